// ubench_host_register.hip -- pageable vs registered-in-place host memory for the 256 MiB fields of an 8192^2 drop-in call
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main()
{
    const size_t bytes = 256u << 20;
    char *h = static_cast<char *>(aligned_alloc(4096, bytes));
    memset(h, 1, bytes);
    void *d;
    CK(hipMalloc(&d, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 3; ++rep) {
        auto t = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double up = ms(t);
        t = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("pageable: H2D %.2f ms  D2H %.2f ms\n", up, ms(t));
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t = std::chrono::steady_clock::now();
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        const double reg = ms(t);
        t = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double up = ms(t);
        t = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const double down = ms(t);
        t = std::chrono::steady_clock::now();
        CK(hipHostUnregister(h));
        printf("registered in place: register %.2f ms  H2D %.2f ms  D2H %.2f ms  unregister %.2f ms\n", reg, up, down, ms(t));
    }
    void *pin;
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    for (int rep = 0; rep < 2; ++rep) {
        auto t = std::chrono::steady_clock::now();
        memcpy(pin, h, bytes);
        const double cp = ms(t);
        t = std::chrono::steady_clock::now();
        CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        printf("pinned staging: memcpy (1 thread) %.2f ms  H2D %.2f ms\n", cp, ms(t));
    }
    return 0;
}
