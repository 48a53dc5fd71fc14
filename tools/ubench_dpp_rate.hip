// ubench_dpp_rate.hip -- issue interval of the instructions of ONE relaxation of the fused SOR kernel on gfx950:
// the full-wave DPP shift add (v_add_f32_dpp wave_shl:1), a row-local DPP add, a plain add, and the whole
// 8-instruction relaxation as a dependent chain (pass s + 1 needs the result of pass s), for 1, 2, 3 waves per SIMD.
// Shader-clock cycles per wave loop (s_memtime) / instructions issued on the SIMD.
// Build & run: hipcc --offload-arch=gfx950 -O2 tools/ubench_dpp_rate.hip -o /tmp/ubdpp && /tmp/ubdpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000

__device__ __forceinline__ void stamp(long long *cyc, long long t0)
{
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

#define REP16(x) x x x x x x x x x x x x x x x x

// 16 independent instructions per iteration (4 destinations in rotation; sources never written in the loop)
__global__ void k_plain(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float a = threadIdx.x, b = s, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %5\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %4, %5\n"
            "v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %5\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %4, %5\n"
            "v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %5\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %4, %5\n"
            "v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %5\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %4, %5\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
    o[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    stamp(cyc, t0);
}
#define DPP_KERNEL(NAME, CTRL)                                                                                         \
    __global__ void NAME(float *o, float s, long long *cyc)                                                            \
    {                                                                                                                  \
        const long long t0 = clock64();                                                                                \
        float a = threadIdx.x, b = s, r0 = 0, r1 = 0, r2 = 0, r3 = 0;                                                  \
        for (int i = 0; i < ITER; ++i)                                                                                 \
            asm volatile(                                                                                              \
                "v_add_f32_dpp %0, %4, %5 " CTRL "\n v_add_f32_dpp %1, %4, %5 " CTRL "\n v_add_f32_dpp %2, %4, %5 " CTRL "\n v_add_f32_dpp %3, %4, %5 " CTRL "\n" \
                "v_add_f32_dpp %0, %4, %5 " CTRL "\n v_add_f32_dpp %1, %4, %5 " CTRL "\n v_add_f32_dpp %2, %4, %5 " CTRL "\n v_add_f32_dpp %3, %4, %5 " CTRL "\n" \
                "v_add_f32_dpp %0, %4, %5 " CTRL "\n v_add_f32_dpp %1, %4, %5 " CTRL "\n v_add_f32_dpp %2, %4, %5 " CTRL "\n v_add_f32_dpp %3, %4, %5 " CTRL "\n" \
                "v_add_f32_dpp %0, %4, %5 " CTRL "\n v_add_f32_dpp %1, %4, %5 " CTRL "\n v_add_f32_dpp %2, %4, %5 " CTRL "\n v_add_f32_dpp %3, %4, %5 " CTRL "\n" \
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));                                            \
        o[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                                                  \
        stamp(cyc, t0);                                                                                                \
    }
DPP_KERNEL(k_dpp_wave_shl, "wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_dpp_wave_shr, "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_dpp_row_shl, "row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")

// the relaxation as compiled (DX1, interior): t = shl(oc) + oc ; t += S ; t += N ; t = d - t ; t *= 0.25 ; u = w1 * own ;
// t *= w ; own' = u - t -- and the next relaxation takes own' as its S / N operand.  2 relaxations per asm block,
// 16 instructions; `x` is the value handed from relaxation to relaxation.
__global__ void k_relax_chain(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            REP16("v_add_f32_dpp %1, %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                  "v_add_f32 %1, %1, %0\n"
                  "v_add_f32 %1, %4, %1\n"
                  "v_sub_f32 %1, %5, %1\n"
                  "v_mul_f32 %1, 0x3e800000, %1\n"
                  "v_mul_f32 %2, %7, %6\n"
                  "v_mul_f32 %1, %7, %1\n"
                  "v_sub_f32 %0, %2, %1\n")
            : "+v"(x), "=&v"(t), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}
// the same with a plain add in place of the DPP add
__global__ void k_relax_chain_nodpp(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            REP16("v_add_f32 %1, %3, %3\n"
                  "v_add_f32 %1, %1, %0\n"
                  "v_add_f32 %1, %4, %1\n"
                  "v_sub_f32 %1, %5, %1\n"
                  "v_mul_f32 %1, 0x3e800000, %1\n"
                  "v_mul_f32 %2, %7, %6\n"
                  "v_mul_f32 %1, %7, %1\n"
                  "v_sub_f32 %0, %2, %1\n")
            : "+v"(x), "=&v"(t), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}


// G relaxations whose DPP adds are issued back to back in front of them (the shifted operands of a trip all come from
// the previous trip, so they can be): is the cost of a DPP instruction in a stream of plain ones a per-transition cost?
#define RELAX_TAIL(T)                                                                                  \
    "v_add_f32 " T ", " T ", %0\n"                                                                     \
    "v_add_f32 " T ", %8, " T "\n"                                                                     \
    "v_sub_f32 " T ", %9, " T "\n"                                                                     \
    "v_mul_f32 " T ", 0x3e800000, " T "\n"                                                             \
    "v_mul_f32 %5, %11, %10\n"                                                                         \
    "v_mul_f32 " T ", %11, " T "\n"                                                                    \
    "v_sub_f32 %0, %5, " T "\n"
#define DPPADD(T) "v_add_f32_dpp " T ", %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
__global__ void k_relax_grouped4(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t1, t2, t3, t4, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            DPPADD("%1") DPPADD("%2") DPPADD("%3") DPPADD("%4") RELAX_TAIL("%1") RELAX_TAIL("%2") RELAX_TAIL("%3") RELAX_TAIL("%4")
            DPPADD("%1") DPPADD("%2") DPPADD("%3") DPPADD("%4") RELAX_TAIL("%1") RELAX_TAIL("%2") RELAX_TAIL("%3") RELAX_TAIL("%4")
            DPPADD("%1") DPPADD("%2") DPPADD("%3") DPPADD("%4") RELAX_TAIL("%1") RELAX_TAIL("%2") RELAX_TAIL("%3") RELAX_TAIL("%4")
            DPPADD("%1") DPPADD("%2") DPPADD("%3") DPPADD("%4") RELAX_TAIL("%1") RELAX_TAIL("%2") RELAX_TAIL("%3") RELAX_TAIL("%4")
            : "+v"(x), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(u), "=&v"(u)
            : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}
// the DPP as a separate v_mov_b32_dpp + plain add (9 instructions per relaxation)
__global__ void k_relax_movdpp(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            REP16("v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                  "v_add_f32 %1, %1, %3\n"
                  "v_add_f32 %1, %1, %0\n"
                  "v_add_f32 %1, %4, %1\n"
                  "v_sub_f32 %1, %5, %1\n"
                  "v_mul_f32 %1, 0x3e800000, %1\n"
                  "v_mul_f32 %2, %7, %6\n"
                  "v_mul_f32 %1, %7, %1\n"
                  "v_sub_f32 %0, %2, %1\n")
            : "+v"(x), "=&v"(t), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}

// every second / fourth relaxation without the DPP (what a lane owning 4 / 8 cells of a row would issue)
#define RELAX_DPP                                                                                     \
    "v_add_f32_dpp %1, %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                  \
    "v_add_f32 %1, %1, %0\n v_add_f32 %1, %4, %1\n v_sub_f32 %1, %5, %1\n v_mul_f32 %1, 0x3e800000, %1\n"   \
    "v_mul_f32 %2, %7, %6\n v_mul_f32 %1, %7, %1\n v_sub_f32 %0, %2, %1\n"
#define RELAX_PLAIN                                                                                   \
    "v_add_f32 %1, %3, %3\n"                                                                          \
    "v_add_f32 %1, %1, %0\n v_add_f32 %1, %4, %1\n v_sub_f32 %1, %5, %1\n v_mul_f32 %1, 0x3e800000, %1\n"   \
    "v_mul_f32 %2, %7, %6\n v_mul_f32 %1, %7, %1\n v_sub_f32 %0, %2, %1\n"
__global__ void k_relax_half(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN
                     RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_DPP RELAX_PLAIN
                     : "+v"(x), "=&v"(t), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}
__global__ void k_relax_quarter(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, t, u;
    for (int i = 0; i < ITER; ++i)
        asm volatile(RELAX_DPP RELAX_PLAIN RELAX_PLAIN RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_PLAIN RELAX_PLAIN
                     RELAX_DPP RELAX_PLAIN RELAX_PLAIN RELAX_PLAIN RELAX_DPP RELAX_PLAIN RELAX_PLAIN RELAX_PLAIN
                     : "+v"(x), "=&v"(t), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}
// two INDEPENDENT chains interleaved instruction by instruction, one DPP relaxation + one plain (a lane with 4 cells)
__global__ void k_relax_two_chains(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, y = 3.0f, t, u, t2, u2;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            REP16("v_add_f32_dpp %2, %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                  "v_add_f32 %4, %6, %6\n"
                  "v_add_f32 %2, %2, %0\n v_add_f32 %4, %4, %1\n"
                  "v_add_f32 %2, %7, %2\n v_add_f32 %4, %7, %4\n"
                  "v_sub_f32 %2, %8, %2\n v_sub_f32 %4, %8, %4\n"
                  "v_mul_f32 %2, 0x3e800000, %2\n v_mul_f32 %4, 0x3e800000, %4\n"
                  "v_mul_f32 %3, %10, %9\n v_mul_f32 %5, %10, %9\n"
                  "v_mul_f32 %2, %10, %2\n v_mul_f32 %4, %10, %4\n"
                  "v_sub_f32 %0, %3, %2\n v_sub_f32 %1, %5, %4\n")
            : "+v"(x), "+v"(y), "=&v"(t), "=&v"(u), "=&v"(t2), "=&v"(u2) : "v"(oc), "v"(nn), "v"(d), "v"(own), "s"(s));
    o[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
    stamp(cyc, t0);
}

// the wave shift through the LDS crossbar instead of DPP: ds_bpermute_b32 issued one relaxation ahead of its use
// (9 instructions per relaxation: bpermute + plain add instead of the DPP add), no DPP instruction in the stream
#define RELAX_BP(TNOW, TNEXT, CNT)                                                                    \
    "ds_bpermute_b32 " TNEXT ", %8, %4\n"                                                             \
    "s_waitcnt lgkmcnt(" CNT ")\n"                                                                    \
    "v_add_f32 " TNOW ", " TNOW ", %4\n"                                                              \
    "v_add_f32 " TNOW ", " TNOW ", %0\n v_add_f32 " TNOW ", %5, " TNOW "\n v_sub_f32 " TNOW ", %6, " TNOW "\n" \
    "v_mul_f32 " TNOW ", 0x3e800000, " TNOW "\n v_mul_f32 %3, %9, %7\n v_mul_f32 " TNOW ", %9, " TNOW "\n v_sub_f32 %0, %3, " TNOW "\n"
__global__ void k_relax_bpermute(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float oc = threadIdx.x, nn = s, d = 0.5f, own = 2.0f, x = 1.0f, ta = 0.0f, tb = 0.0f, u;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1") RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1")
            RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1") RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1")
            RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1") RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1")
            RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1") RELAX_BP("%1", "%2", "1") RELAX_BP("%2", "%1", "1")
            : "+v"(x), "+v"(ta), "+v"(tb), "=&v"(u) : "v"(oc), "v"(nn), "v"(d), "v"(own), "v"(addr), "s"(s) : "memory");
    o[blockIdx.x * blockDim.x + threadIdx.x] = x;
    stamp(cyc, t0);
}

// LDS-crossbar throughput: 16 independent ds_bpermute_b32 per iteration, one wait at the end of the block
__global__ void k_bpermute_rate(float *o, float s, long long *cyc)
{
    const long long t0 = clock64();
    float a = threadIdx.x, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "ds_bpermute_b32 %0, %5, %4\n ds_bpermute_b32 %1, %5, %4\n ds_bpermute_b32 %2, %5, %4\n ds_bpermute_b32 %3, %5, %4\n"
            "ds_bpermute_b32 %0, %5, %4\n ds_bpermute_b32 %1, %5, %4\n ds_bpermute_b32 %2, %5, %4\n ds_bpermute_b32 %3, %5, %4\n"
            "ds_bpermute_b32 %0, %5, %4\n ds_bpermute_b32 %1, %5, %4\n ds_bpermute_b32 %2, %5, %4\n ds_bpermute_b32 %3, %5, %4\n"
            "ds_bpermute_b32 %0, %5, %4\n ds_bpermute_b32 %1, %5, %4\n ds_bpermute_b32 %2, %5, %4\n ds_bpermute_b32 %3, %5, %4\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(addr) : "memory");
    o[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    stamp(cyc, t0);
}
// the same with ds_read2st64_b32 (what the rhs ring issues)
__global__ void k_read2_rate(float *o, float s, long long *cyc)
{
    __shared__ float lds[4096];
    const long long t0 = clock64();
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = k;
    __syncthreads();
    float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    const int addr = (threadIdx.x & 63) * 4;
    for (int i = 0; i < ITER; ++i)
        asm volatile(
            "ds_read2st64_b32 v[10:11], %8 offset0:1 offset1:2\n ds_read2st64_b32 v[12:13], %8 offset0:3 offset1:4\n"
            "ds_read2st64_b32 v[14:15], %8 offset0:5 offset1:6\n ds_read2st64_b32 v[16:17], %8 offset0:7 offset1:8\n"
            "ds_read2st64_b32 v[10:11], %8 offset0:1 offset1:2\n ds_read2st64_b32 v[12:13], %8 offset0:3 offset1:4\n"
            "ds_read2st64_b32 v[14:15], %8 offset0:5 offset1:6\n ds_read2st64_b32 v[16:17], %8 offset0:7 offset1:8\n"
            "ds_read2st64_b32 v[10:11], %8 offset0:1 offset1:2\n ds_read2st64_b32 v[12:13], %8 offset0:3 offset1:4\n"
            "ds_read2st64_b32 v[14:15], %8 offset0:5 offset1:6\n ds_read2st64_b32 v[16:17], %8 offset0:7 offset1:8\n"
            "ds_read2st64_b32 v[10:11], %8 offset0:1 offset1:2\n ds_read2st64_b32 v[12:13], %8 offset0:3 offset1:4\n"
            "ds_read2st64_b32 v[14:15], %8 offset0:5 offset1:6\n ds_read2st64_b32 v[16:17], %8 offset0:7 offset1:8\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(addr)
            : "memory", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    o[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + lds[threadIdx.x & 1023];
    stamp(cyc, t0);
}

template <class K>
void run(const char *name, K k, int waves_per_simd, int insts_per_iter, float *o, long long *cyc)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 256 * waves_per_simd;   // 4 SIMDs per CU
    const int blocks = cus;
    for (int rep = 0; rep < 3; ++rep) k<<<blocks, threads>>>(o, 1.0f, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<<<blocks, threads>>>(o, 1.0f, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves = blocks * threads / 64;
    std::vector<long long> h(waves);
    hipMemcpy(h.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= waves;
    const double per_inst_wave = mean / ((double)ITER * insts_per_iter);
    printf("%-22s %d wave(s)/SIMD: %6.2f clock64 ticks per instruction of a wave, %6.2f per instruction issued on the SIMD   (kernel %.1f us, %.0f ticks per us, %.3f ns per instruction issued on the SIMD)\n", name,
           waves_per_simd, per_inst_wave, per_inst_wave / waves_per_simd, ms * 1e3, mean / (ms * 1e3),
           ms * 1e6 / ((double)ITER * insts_per_iter * waves_per_simd));
}

int main()
{
    float *o;
    long long *cyc;
    hipMalloc(&o, 1 << 24);
    hipMalloc(&cyc, 1 << 20);
    for (int w : {1, 2, 3, 4}) {
        run("plain v_add_f32", k_plain, w, 16, o, cyc);
        run("dpp wave_shl:1", k_dpp_wave_shl, w, 16, o, cyc);
        run("dpp wave_shr:1", k_dpp_wave_shr, w, 16, o, cyc);
        run("dpp row_shl:1", k_dpp_row_shl, w, 16, o, cyc);
        run("relaxation chain", k_relax_chain, w, 128, o, cyc);
        run("relaxation, no dpp", k_relax_chain_nodpp, w, 128, o, cyc);
        run("relaxation, dpp x4 up front", k_relax_grouped4, w, 128, o, cyc);
        run("relaxation, mov_dpp + add", k_relax_movdpp, w, 144, o, cyc);
        run("dpp in every 2nd relax", k_relax_half, w, 128, o, cyc);
        run("dpp in every 4th relax", k_relax_quarter, w, 128, o, cyc);
        run("two chains, dpp in one", k_relax_two_chains, w, 256, o, cyc);
        run("relaxation, ds_bpermute", k_relax_bpermute, w, 16 * 8, o, cyc);
        run("ds_bpermute_b32 alone", k_bpermute_rate, w, 16, o, cyc);
        run("ds_read2st64_b32 alone", k_read2_rate, w, 16, o, cyc);   // per 8 "useful" instructions: compare with the chains
    }
    return 0;
}
