// ubench_stream_pattern.hip -- the memory ACCESS PATTERN of the fused SOR kernel without its arithmetic:
// every wave streams a column strip bottom-up (one row per step, rows dim_x * 4 bytes apart), reading two
// arrays (p, d) and writing one (p_out), a few rows in flight.  How much of the HBM peak does that pattern
// sustain by itself, and does it depend on the bytes a lane moves per access (8 vs 16), on the rows a wave
// streams, on the rows in flight, on non-temporal stores?
// Build & run: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream_pattern.hip -o tools/ubench_stream_pattern && tools/ubench_stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 8192;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// one wave per tile: strip of 64 * CELLS columns, ROWS rows; AHEAD rows in flight
template <int CELLS, int AHEAD, bool NT>
__global__ void __launch_bounds__(256) stream3(float *__restrict__ out, const float *__restrict__ p,
                                               const float *__restrict__ d, int rows_per_tile, int strips)
{
    typedef float vec __attribute__((ext_vector_type(CELLS)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    const int chunk = tile / strips, strip = tile - chunk * strips;
    const int r0 = chunk * rows_per_tile, r1 = min(r0 + rows_per_tile, N);
    if (r0 >= N) return;
    const size_t col = (size_t)strip * 64 * CELLS + (size_t)lane * CELLS;
    vec a[AHEAD], b[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
        const int r = min(r0 + u, N - 1);
        a[u] = *(const vec *)(p + (size_t)r * N + col);
        b[u] = *(const vec *)(d + (size_t)r * N + col);
    }
    for (int y = r0; y < r1; y += AHEAD) {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            if (y + u < r1) {
                const vec s = a[u] + b[u];
                const int r = min(y + u + AHEAD, N - 1);
                a[u] = *(const vec *)(p + (size_t)r * N + col);
                b[u] = *(const vec *)(d + (size_t)r * N + col);
                vec *o = (vec *)(out + (size_t)(y + u) * N + col);
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
    for (int r = 0; r < 6; ++r) {
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

template <int CELLS, int AHEAD, bool NT>
void run(float *o, const float *p, const float *d, int rows_per_tile)
{
    const int strips = N / (64 * CELLS);
    const int chunks = (N + rows_per_tile - 1) / rows_per_tile;
    const int tiles = strips * chunks;
    const float us = timeit([&] { stream3<CELLS, AHEAD, NT><<<(tiles + 3) / 4, 256>>>(o, p, d, rows_per_tile, strips); });
    printf("%2d B per lane, %2d rows in flight, %4d rows per wave (%5d waves)%s: %7.1f us  %.2f TB/s\n", CELLS * 4, AHEAD,
           rows_per_tile, tiles, NT ? ", nt stores" : "           ", us, 3.0 * N * N * 4 / us / 1e6);
}

int main()
{
    float *p, *d, *o;
    const size_t bytes = (size_t)N * N * 4;
    hipMalloc(&p, bytes);
    hipMalloc(&d, bytes);
    hipMalloc(&o, bytes);
    hipMemset(p, 0, bytes);
    hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 20; ++rep) stream3<2, 6, false><<<1024, 256>>>(o, p, d, 256, 64);  // clocks up
    hipDeviceSynchronize();
    for (int rows : {64, 128, 256, 512}) {
        run<2, 6, false>(o, p, d, rows);
        run<2, 6, true>(o, p, d, rows);
        run<2, 12, false>(o, p, d, rows);
        run<4, 6, false>(o, p, d, rows);
        run<4, 6, true>(o, p, d, rows);
        run<4, 3, false>(o, p, d, rows);
        run<1, 6, false>(o, p, d, rows);
    }
    return 0;
}
