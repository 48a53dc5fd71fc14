// ubench_xstream_sync.hip -- what does a cross-stream dependency cost?  A chain of N hops between two streams
// (a short kernel on stream 1, then one on stream 2 that must wait for it, and back), ordered by
//   (a) hipEventRecord + hipStreamWaitEvent            (what the slab executor uses around a halo exchange)
//   (b) hipStreamWriteValue32 + hipStreamWaitValue32   (stream memory operations on signal memory)
// Reported: microseconds per hop beyond the kernels' own time, r03.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(unsigned long long cycles, int *sink)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}

int main(int argc, char **argv)
{
    const int hops = argc > 1 ? atoi(argv[1]) : 200;
    const unsigned long long cyc = argc > 2 ? atoll(argv[2]) : 20000;  // ~10 us per kernel
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e12, e21;
    CK(hipEventCreateWithFlags(&e12, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e21, hipEventDisableTiming));
    unsigned *sig = nullptr;
    hipError_t se = hipExtMallocWithFlags(reinterpret_cast<void **>(&sig), 8, hipMallocSignalMemory);
    if (se != hipSuccess) { fprintf(stderr, "signal memory: %s\n", hipGetErrorString(se)); (void)hipGetLastError(); sig = nullptr; }
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !sig) continue;
        if (sig) CK(hipMemset(sig, 0, 8));
        CK(hipDeviceSynchronize());
        unsigned v = 0;
        const auto t0 = now();
        for (int h = 0; h < hops; ++h) {
            if (mode == 0) {  // same stream: the floor
                spin<<<256, 64, 0, s1>>>(cyc, nullptr);
                spin<<<256, 64, 0, s1>>>(cyc, nullptr);
            } else if (mode == 1) {
                spin<<<256, 64, 0, s1>>>(cyc, nullptr);
                CK(hipEventRecord(e12, s1));
                CK(hipStreamWaitEvent(s2, e12, 0));
                spin<<<256, 64, 0, s2>>>(cyc, nullptr);
                CK(hipEventRecord(e21, s2));
                CK(hipStreamWaitEvent(s1, e21, 0));
            } else {
                spin<<<256, 64, 0, s1>>>(cyc, nullptr);
                CK(hipStreamWriteValue32(s1, sig, ++v, 0));
                CK(hipStreamWaitValue32(s2, sig, v, hipStreamWaitValueGte, 0xffffffffu));
                spin<<<256, 64, 0, s2>>>(cyc, nullptr);
                CK(hipStreamWriteValue32(s2, sig, ++v, 0));
                CK(hipStreamWaitValue32(s1, sig, v, hipStreamWaitValueGte, 0xffffffffu));
            }
        }
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        const double us = std::chrono::duration<double, std::micro>(now() - t0).count();
        const char *name[3] = {"same stream (floor)", "events", "stream write/wait value"};
        printf("%-26s %d x 2 kernels: %.2f us per kernel incl. its dependency\n", name[mode], hops, us / (2.0 * hops));
    }
    return 0;
}
