// ubench_stage_pairs.hip -- codegen probe for the next-round SOR design: two pipeline
// stages packed into one v_pk_* instruction, state in register pairs {row r, row r - K}.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -S --cuda-device-only tools/ubench_stage_pairs.hip -o /tmp/pairs.s
// Measured (ROCm 7.2): per pair of relaxations 5 v_pk_add_f32 + 3 v_pk_mul_f32 + 2 v_mov_b32_dpp + 0.4
// v_mov_b32 + 2.2 s_nop = 10.4 VALU instructions against 16 for two scalar relaxations -- PROVIDED the DPP
// moves use bound_ctrl (with bound_ctrl = false the compiler adds "v_mov_b32 dst, 0; s_nop 1" in front of
// every one of them: 12.4 + 1.3 nops).  Not a timing benchmark: the arithmetic is the SOR relaxation, the
// row bookkeeping is only shaped like the pipeline.
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dpp_shr(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ v2f below(v2f x) { return v2f{dpp_shr(x.x), dpp_shr(x.y)}; }
__device__ __forceinline__ v2f above(v2f x) { return v2f{dpp_shl(x.x), dpp_shl(x.y)}; }
__device__ __forceinline__ v2f relax(v2f own, v2f w, v2f e, v2f s, v2f n, v2f d, float om, float om1) {
    v2f sum = ((w + e) + s) + n;
    v2f gs = -0.25f * (d - sum);
    return om1 * own + om * gs;
}
// 4 pair-stages per colour per iteration (NS = 16), rows in a window of 12 pairs
__global__ void k(float* o, const float* in, float om, float om1, int iters) {
    __shared__ float ring[18 * 2 * 64 * 4];
    float* rg = ring + (threadIdx.x >> 6) * 18 * 2 * 64 + (threadIdx.x & 63);
    for (int i = 0; i < 36; ++i) rg[i * 64] = in[i * 64 + threadIdx.x];
    __syncthreads();
    v2f E[12], O[12];
    for (int i = 0; i < 12; ++i) { E[i] = v2f{in[threadIdx.x + i], in[threadIdx.x + 64 + i]}; O[i] = E[i] * 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int U = 0; U < 12; ++U) {
            // entry: new row into lo, transition of row U-8 into hi
            E[U] = v2f{in[it * 12 + U], E[(U + 4) % 12].x};
            O[U].x = in[it * 12 + U + 7];
#pragma unroll
            for (int m = 1; m <= 4; ++m) {
                {
                    const int i0 = (U - (2 * m - 1) + 24) % 12, im = (i0 + 11) % 12, ip = (i0 + 1) % 12;
                    const bool ev = ((U - (2 * m - 1)) & 1) == 0;
                    const v2f oc = O[i0];
                    const v2f w = ev ? below(oc) : oc, e = ev ? oc : above(oc);
                    const v2f d = v2f{rg[((i0 * 2) % 36) * 64], rg[((i0 * 2 + 16) % 36) * 64]};
                    E[i0] = relax(E[i0], w, e, O[im], O[ip], d, om, om1);
                }
                {
                    const int i0 = (U - 2 * m + 24) % 12, im = (i0 + 11) % 12, ip = (i0 + 1) % 12;
                    const bool ev = ((U - 2 * m) & 1) == 0;
                    const v2f oc = E[i0];
                    const v2f w = ev ? oc : below(oc), e = ev ? above(oc) : oc;
                    const v2f d = v2f{rg[((i0 * 2 + 1) % 36) * 64], rg[((i0 * 2 + 17) % 36) * 64]};
                    v2f res = relax(O[i0], w, e, E[im], E[ip], d, om, om1);
                    if (m < 4) O[i0] = res;
                    else { O[i0].x = res.x; O[(i0 + 8) % 12].y = res.x; o[(it * 12 + U) * 64 + (threadIdx.x & 63)] = res.y + E[i0].y; }
                }
            }
        }
    }
}
