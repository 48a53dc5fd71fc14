// ubench_issue.hip -- how often can ONE wave issue, and how many waves does a SIMD need to keep its
// VALU busy?  (r03: the fused SOR kernel is chain-of-dependent-VALU work at 3 waves per SIMD.)
// Every wave runs the same loop of a fixed instruction block and records s_memtime at its start and
// end; waves per SIMD is set by the grid (blocks of 256 threads = one wave per SIMD of a CU) and a
// dynamic-LDS allocation that keeps more blocks from joining.  Reported: cycles per instruction of a
// wave (median / last to finish) and cycles per instruction issued on the SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Mode { DEP = 0, INDEP8, DEP_NOP, DEP_SALU, DPP_INDEP8, RELAX, RELAX_NODPP, RELAX_2CHAIN, DEP_WAITCNT, RELAX_LIT, N_MODES };
static const char *kName[N_MODES] = {"dependent v_add chain", "8 independent v_add", "dependent v_add + s_nop 0 each",
                                     "dependent v_add + s_add_u32 each", "8 independent v_add_dpp wave_shl",
                                     "relaxation (dpp + 7, chain of 5)", "relaxation without dpp",
                                     "two interleaved relaxation chains", "dependent v_add + s_waitcnt each",
                                     "relaxation with 0.25 literal (8-byte mul)"};
static const int kInsts[N_MODES] = {8, 8, 16, 16, 8, 8, 8, 16, 16, 8};   // instructions per block (all kinds)
static const int kValu[N_MODES] = {8, 8, 8, 8, 8, 8, 8, 16, 8, 8};       // VALU instructions per block

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long *out, float *sink, int iters, float a, float b)
{
    extern __shared__ float lds[];
    float x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float s = b, n = a, own = x0, rhs = x1, w = x2, e = x3, r = 0, r2 = 0, n2 = b;
    const float om = a + 1.96f, om1 = 1.0f - om, q = b + 0.25f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == DEP) {
            asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                         "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
        } else if (MODE == INDEP8) {
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s));
        } else if (MODE == DEP_NOP) {
            asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n"
                         "v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0" : "+v"(x0) : "v"(x1));
        } else if (MODE == DEP_SALU) {
            asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n"
                         "v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1"
                         : "+v"(x0) : "v"(x1) : "s20", "scc");
        } else if (MODE == DEP_WAITCNT) {
            asm volatile("v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n"
                         "v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(x0) : "v"(x1));
        } else if (MODE == DPP_INDEP8) {
            asm volatile("v_add_f32_dpp %0, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %2, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %4, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %6, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s));
        } else if (MODE == RELAX || MODE == RELAX_NODPP || MODE == RELAX_LIT) {
            // t = (w + e) + s; t += n(prev result); t = rhs - t; t *= 0.25; t *= omega; u = om1 * own; n = u - t
            if (MODE == RELAX)
                asm volatile("v_add_f32_dpp %1, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %3, %1\n v_add_f32 %1, %1, %0\n v_sub_f32 %1, %4, %1\n"
                             "v_mul_f32 %1, %5, %1\n v_mul_f32 %7, %8, %9\n v_mul_f32 %1, %6, %1\n v_sub_f32 %0, %7, %1"
                             : "+v"(n), "+v"(r) : "v"(w), "v"(s), "v"(rhs), "s"(q), "s"(om), "v"(r2), "s"(om1), "v"(own));
            else if (MODE == RELAX_LIT)
                asm volatile("v_add_f32_dpp %1, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %3, %1\n v_add_f32 %1, %1, %0\n v_sub_f32 %1, %4, %1\n"
                             "v_mul_f32 %1, 0x3e800000, %1\n v_mul_f32 %7, %8, %9\n v_mul_f32 %1, %6, %1\n v_sub_f32 %0, %7, %1"
                             : "+v"(n), "+v"(r) : "v"(w), "v"(s), "v"(rhs), "s"(q), "s"(om), "v"(r2), "s"(om1), "v"(own));
            else
                asm volatile("v_add_f32 %1, %2, %2\n v_add_f32 %1, %3, %1\n v_add_f32 %1, %1, %0\n v_sub_f32 %1, %4, %1\n"
                             "v_mul_f32 %1, %5, %1\n v_mul_f32 %7, %8, %9\n v_mul_f32 %1, %6, %1\n v_sub_f32 %0, %7, %1"
                             : "+v"(n), "+v"(r) : "v"(w), "v"(s), "v"(rhs), "s"(q), "s"(om), "v"(r2), "s"(om1), "v"(own));
        } else if (MODE == RELAX_2CHAIN) {
            asm volatile("v_add_f32_dpp %1, %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32 %1, %6, %1\n v_add_f32 %3, %6, %3\n v_add_f32 %1, %1, %0\n v_add_f32 %3, %3, %2\n v_sub_f32 %1, %7, %1\n v_sub_f32 %3, %7, %3\n"
                         "v_mul_f32 %1, %8, %1\n v_mul_f32 %3, %8, %3\n v_mul_f32 %10, %11, %12\n v_mul_f32 %13, %11, %12\n v_mul_f32 %1, %9, %1\n v_mul_f32 %3, %9, %3\n"
                         "v_sub_f32 %0, %10, %1\n v_sub_f32 %2, %13, %3"
                         : "+v"(n), "+v"(r), "+v"(n2), "+v"(x7) : "v"(w), "v"(e), "v"(s), "v"(rhs), "s"(q), "s"(om), "v"(r2), "s"(om1), "v"(own), "v"(x6));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[3 * wave] = t0;
        out[3 * wave + 1] = t1;
        out[3 * wave + 2] = hw;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + n + r + n2 + lds[threadIdx.x];
}

template <int MODE>
int run(int waves_per_simd, int iters, int prio_unused)
{
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * waves_per_simd;
    // LDS per block so that exactly `waves_per_simd` blocks fit a CU
    const int lds = (160 * 1024 / waves_per_simd) & ~1023;
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    unsigned long long *out;
    float *sink;
    CK(hipMalloc(&out, blocks * 4 * 3 * 8));
    CK(hipMalloc(&sink, blocks * 256 * 4));
    for (int rep = 0; rep < 3; ++rep) k<MODE><<<blocks, 256, lds>>>(out, sink, iters, 1e-9f, 1e-7f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4 * 3);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> life;
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < blocks * 4; ++w) {
        life.push_back((double)(h[3 * w + 1] - h[3 * w]));
        lo = std::min(lo, h[3 * w]);
        hi = std::max(hi, h[3 * w + 1]);
    }
    std::sort(life.begin(), life.end());
    const double per = (double)iters * kInsts[MODE];
    printf("%-44s %d waves/SIMD: cycles per instruction of a wave: first %.2f median %.2f last %.2f | per VALU instruction on the SIMD (last wave): %.2f\n",
           kName[MODE], waves_per_simd, life.front() / per, life[life.size() / 2] / per, life.back() / per,
           life.back() / ((double)iters * kValu[MODE] * waves_per_simd));
    CK(hipFree(out));
    CK(hipFree(sink));
    return 0;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int w = 1; w <= 4; ++w) {
        run<DEP>(w, iters, 0);
        run<INDEP8>(w, iters, 0);
        run<DEP_NOP>(w, iters, 0);
        run<DEP_SALU>(w, iters, 0);
        run<DEP_WAITCNT>(w, iters, 0);
        run<DPP_INDEP8>(w, iters, 0);
        run<RELAX>(w, iters, 0);
        run<RELAX_LIT>(w, iters, 0);
        run<RELAX_NODPP>(w, iters, 0);
        run<RELAX_2CHAIN>(w, iters, 0);
    }
    return 0;
}
