// ubench_stream_pitch.hip -- does the fused SOR kernel's streaming pattern (every wave walks a 512-byte wide
// column strip row by row, two arrays read, one written) depend on the ROW PITCH and on the relative placement of
// the three arrays?  At 8192 floats a row is 32 KiB: every row of a strip starts at the same offset within any
// power-of-two interleaving period of the memory system.  Pitches of 8192 + k floats and staggered array bases
// against the library's layout (pitch = dim_x, arrays back to back).
// Build & run: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream_pitch.hip -o /tmp/ubench_stream_pitch && /tmp/ubench_stream_pitch
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 8192;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int AHEAD, bool NT>
__global__ void __launch_bounds__(256) stream3(float *__restrict__ out, const float *__restrict__ p,
                                               const float *__restrict__ d, int rows_per_tile, int strips, size_t pitch)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    const int chunk = tile / strips, strip = tile - chunk * strips;
    const int r0 = chunk * rows_per_tile, r1 = min(r0 + rows_per_tile, N);
    if (r0 >= N) return;
    const size_t col = (size_t)strip * 128 + (size_t)lane * 2;
    v2f a[AHEAD], b[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
        const int r = min(r0 + u, N - 1);
        a[u] = *(const v2f *)(p + (size_t)r * pitch + col);
        b[u] = *(const v2f *)(d + (size_t)r * pitch + col);
    }
    for (int y = r0; y < r1; y += AHEAD) {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            if (y + u < r1) {
                const v2f s = a[u] + b[u];
                const int r = min(y + u + AHEAD, N - 1);
                a[u] = *(const v2f *)(p + (size_t)r * pitch + col);
                b[u] = *(const v2f *)(d + (size_t)r * pitch + col);
                v2f *o = (v2f *)(out + (size_t)(y + u) * pitch + col);
                if (NT)
                    __builtin_nontemporal_store(s, o);
                else
                    *o = s;
            }
        }
    }
}

template <class F>
float timeit(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 8; ++r) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    char *block;
    const size_t max_pitch = N + 4096;
    const size_t arr = max_pitch * (size_t)N * 4 + (64u << 20);
    hipMalloc(&block, 3 * arr);
    hipMemset(block, 0, 3 * arr);
    for (int rep = 0; rep < 30; ++rep)
        stream3<6, false><<<1024, 256>>>((float *)(block + 2 * arr), (float *)block, (float *)(block + arr), 256, 64, N);
    hipDeviceSynchronize();
    const int strips = N / 128;
    for (int rows : {234, 128, 64}) {
        const int chunks = (N + rows - 1) / rows;
        const int tiles = strips * chunks;
        for (int extra : {0, 16, 32, 64, 128, 192, 256, 512, 1024, 2048, 4096}) {
            for (size_t stagger : {(size_t)0, (size_t)(4096 + 256), (size_t)(1u << 20) + 8192 + 512}) {
                const size_t pitch = N + extra;
                const size_t bytes = pitch * (size_t)N * 4;
                // library layout: arrays back to back (stagger 0); staggered: each next array shifted further
                float *p = (float *)block, *d = (float *)(block + bytes + stagger), *o = (float *)(block + 2 * bytes + 2 * stagger);
                const float us = timeit([&] { stream3<6, false><<<(tiles + 3) / 4, 256>>>(o, p, d, rows, strips, pitch); });
                const float usn = timeit([&] { stream3<6, true><<<(tiles + 3) / 4, 256>>>(o, p, d, rows, strips, pitch); });
                printf("rows/wave %3d  pitch 8192+%-4d  stagger %8zu B: %7.1f us %.2f TB/s   nt stores %7.1f us %.2f TB/s\n", rows,
                       extra, stagger, us, 3.0 * N * N * 4 / us / 1e6, usn, 3.0 * N * N * 4 / usn / 1e6);
            }
        }
    }
    return 0;
}
