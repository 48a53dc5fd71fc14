// ubench_copy.hip -- what bounds the one-thread-per-cell advection: access width / block shape probes.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o /tmp/uc && /tmp/uc
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 8192;
__global__ void __launch_bounds__(256) copy_f2(float2* o, const float2* a) {
    int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    size_t c = (size_t)j * N + i; o[c] = a[c];
}
__global__ void __launch_bounds__(256) copy_f4(float4* o, const float4* a) {  // 2 cells per thread
    int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    size_t c = (size_t)j * (N / 2) + i; o[c] = a[c];
}
__global__ void __launch_bounds__(256) copy_f2_rows(float2* o, const float2* a) {  // 4 rows per thread
    int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y * 4;
    size_t c = (size_t)j * N + i;
    float2 v0 = a[c], v1 = a[c + N], v2 = a[c + 2 * N], v3 = a[c + 3 * N];
    o[c] = v0; o[c + N] = v1; o[c + 2 * N] = v2; o[c + 3 * N] = v3;
}
__global__ void __launch_bounds__(256) gather_f2(float2* o, const float2* a) {  // own + 4 neighbours
    int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    size_t c = (size_t)j * N + i;
    float2 u = a[c];
    int ii = min(max(i + (int)(u.x * 1e-9f), 0), N - 2), jj = min(max(j + (int)(u.y * 1e-9f), 0), N - 2);
    size_t t = (size_t)jj * N + ii;
    float2 p11 = a[t], p12 = a[t + N], p21 = a[t + 1], p22 = a[t + N + 1];
    o[c] = make_float2(p11.x + p12.x + p21.x + p22.x, p11.y + p12.y + p21.y + p22.y);
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best * 1e3f;
}
int main() {
    float2 *a, *o; size_t bytes = (size_t)N * N * 8;
    hipMalloc(&a, bytes); hipMalloc(&o, bytes); hipMemset(a, 0, bytes);
    printf("copy float2 64x4 blocks      %.1f us\n", timeit([&] { copy_f2<<<dim3(N / 64, N / 4), dim3(64, 4)>>>(o, a); }));
    printf("copy float4 64x4 blocks      %.1f us\n", timeit([&] { copy_f4<<<dim3(N / 128, N / 4), dim3(64, 4)>>>((float4*)o, (const float4*)a); }));
    printf("copy float2 4 rows / thread  %.1f us\n", timeit([&] { copy_f2_rows<<<dim3(N / 256, N / 4), 256>>>(o, a); }));
    printf("own + 4-texel gather float2  %.1f us\n", timeit([&] { gather_f2<<<dim3(N / 64, N / 4), dim3(64, 4)>>>(o, a); }));
    return 0;
}
