// ubench_pk_chain.hip -- how fast does ONE wave (and 2, 3 per SIMD) get through DEPENDENT packed-fp32
// operations on gfx950?  Decides whether the twin tiles of the SOR kernel (7 packed + 2 DPP
// instructions per cell pair instead of 8.5 per cell) can pay.  Inline asm throughout, so that the
// instruction sequence is exactly what is written (the required wait state between dependent packed
// operations is written by hand: hand-written asm is invisible to the compiler's hazard recognizer).
// Build & run:  hipcc --offload-arch=gfx950 -O2 tools/ubench_pk_chain.hip -o /tmp/ubpk && /tmp/ubpk
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4000
typedef float v2f __attribute__((ext_vector_type(2)));

// shader-clock cycles a wave spent in its loop (s_memtime), written per wave; the host reports the
// mean over waves divided by the instructions of ALL waves resident on the SIMD: a clock-rate
// independent issue interval (the event-timed figure assumes a clock)
__device__ __forceinline__ void stamp(long long *cyc, long long t0)
{
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// 16 dependent scalar adds per iteration
__global__ void k_dep_scalar(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float a = threadIdx.x;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
            : "+v"(a) : "v"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = a;
    stamp(cyc, t0);
}
// 16 dependent packed adds per iteration, one wait state between them (as the compiler emits)
__global__ void k_dep_pk(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    v2f a = {(float)threadIdx.x, 1.0f}, b = {s, s};
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n"
            "v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n"
            "v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n"
            "v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n"
            : "+v"(a) : "v"(b));
    o[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y;
    stamp(cyc, t0);
}
// two interleaved dependent packed chains: 16 packed adds per iteration, no wait states needed
__global__ void k_dep_pk2(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    v2f a = {(float)threadIdx.x, 1.0f}, c = {2.0f, (float)threadIdx.x}, b = {s, s};
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
            "v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
            "v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
            "v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
            : "+v"(a), "+v"(c) : "v"(b));
    o[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + c.x + c.y;
    stamp(cyc, t0);
}
// four interleaved chains
__global__ void k_dep_pk4(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    v2f a = {(float)threadIdx.x, 1.0f}, c = {2.0f, (float)threadIdx.x}, d = {3.0f, 1.0f}, e = {4.0f, 2.0f}, b = {s, s};
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
            "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
            "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
            "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
            : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
    o[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + c.x + c.y + d.x + e.y;
    stamp(cyc, t0);
}
// the twin relaxation as first compiled: 2 DPP adds, (1-w)*own, then the chain of six dependent
// packed operations with a wait state each (9 VALU instructions per cell pair); 2 per iteration
#define RELAX_NAIVE                                                                    \
    "v_add_f32_dpp %[wx], %[ocx], %[ocx] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    "v_add_f32_dpp %[wy], %[ocy], %[ocy] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
    "v_pk_mul_f32 %[co], %[c1], %[own]\n"                                              \
    "v_pk_add_f32 %[t], %[w], %[s]\n s_nop 0\n"                                        \
    "v_pk_add_f32 %[t], %[t], %[n]\n s_nop 0\n"                                        \
    "v_pk_add_f32 %[t], %[d], %[t] neg_lo:[0,1] neg_hi:[0,1]\n s_nop 0\n"              \
    "v_pk_mul_f32 %[t], %[t], %[q]\n s_nop 0\n"                                        \
    "v_pk_mul_f32 %[t], %[om], %[t]\n s_nop 0\n"                                       \
    "v_pk_add_f32 %[n], %[co], %[t]\n s_nop 0\n"
__global__ void k_relax_naive(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    v2f own = {1.0f, 2.0f}, oc = {(float)threadIdx.x, 3.0f}, sd = {0.5f, 0.25f}, n = {0.1f, 0.2f}, d = {0.3f, 0.7f};
    v2f c1 = {-0.96f, -0.96f}, q = {-0.25f, -0.25f}, om = {s, s}, w, co, t;
    for (int i = 0; i < ITER; ++i)
        asm volatile(RELAX_NAIVE RELAX_NAIVE
                     : [n] "+v"(n), [w] "=&v"(w), [co] "=&v"(co), [t] "=&v"(t), [wx] "=&v"(w.x), [wy] "=&v"(w.y)
                     : [own] "v"(own), [ocx] "v"(oc.x), [ocy] "v"(oc.y), [s] "v"(sd), [d] "v"(d), [c1] "v"(c1), [q] "v"(q), [om] "v"(om));
    o[blockIdx.x * blockDim.x + threadIdx.x] = n.x + n.y;
    stamp(cyc, t0);
}

long long *g_cyc = nullptr;

template <class F>
void run(F f, int blocks, const char *name, double valu_per_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f(blocks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f(blocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wps = blocks / 256.0;
    const double ns = ms * 1e6 / (valu_per_iter * ITER * wps);
    const int waves = blocks * 4;
    long long *h = new long long[waves];
    hipMemcpy(h, g_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < waves; ++i) mean += (double)h[i];
    mean /= waves;
    delete[] h;
    printf("%-16s waves/SIMD=%3.0f  %.3f ms  %.2f cycles @2.4GHz per VALU instruction per SIMD;  s_memtime: %.2f counts per "
           "instruction per SIMD\n", name, wps, ms, ns * 2.4, mean / (valu_per_iter * ITER * wps));
}

int main()
{
    float *o;
    hipMalloc(&o, 4096 * 256 * 4);
    hipMalloc(&g_cyc, 4096 * 4 * sizeof(long long));
    for (int i = 0; i < 3; ++i) k_dep_scalar<<<2048, 256>>>(o, 1.0f, g_cyc);  // clocks up
    hipDeviceSynchronize();
    for (int wps : {1, 2, 3, 4}) {
        const int b = 256 * wps;
        run([&](int n) { k_dep_scalar<<<n, 256>>>(o, 1.0001f, g_cyc); }, b, "dep scalar", 16);
        run([&](int n) { k_dep_pk<<<n, 256>>>(o, 1.0001f, g_cyc); }, b, "dep pk +nop", 16);
        run([&](int n) { k_dep_pk2<<<n, 256>>>(o, 1.0001f, g_cyc); }, b, "dep pk x2", 16);
        run([&](int n) { k_dep_pk4<<<n, 256>>>(o, 1.0001f, g_cyc); }, b, "dep pk x4", 16);
        run([&](int n) { k_relax_naive<<<n, 256>>>(o, 1.96f, g_cyc); }, b, "twin relax", 18);
    }
    return 0;
}
