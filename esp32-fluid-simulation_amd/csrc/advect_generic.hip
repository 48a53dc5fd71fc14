// advect_generic.hip -- advect<T, float> (advect.h:74-85) for EVERY element type the reference's own
// headers can express: T = C consecutive 32-bit channels of one kind, C = 1..3,
//
//     kind f32   float, Vector2<float>, Vector3<float>          (vector.h:4-126)
//     kind uq32  UQ32,  Vector2<UQ32>,  Vector3<UQ32>           (uq32.h:8-16)
//
// The reference's sample() (advect.h:24-72) is written once against the operators of T; for these types
// every operator acts channel by channel (vector.h:23-61, :83-126), `TPromoted<T>` has float channels
// (advect.h:10-11) and the conversion back to T narrows each channel on its own (uq32.h:13: + 0.5f,
// truncate).  So one kernel over <C, UQ> with the per-channel arithmetic of advect_math.h covers them
// all.  The sketch's own two instantiations (C = 2 / f32 and C = 3 / uq32) have tuned kernels
// (advect_tiled.hip) and are routed there by the launcher; this file serves the others: one thread per
// cell, texels gathered from memory.
//
// Compiled with -ffp-contract=off (bit-exactness contract, see stencil_kernels.hip).
#include "advect_math.h"
#include "kernels.h"

namespace sfl {
namespace {

using namespace advect_math;

constexpr int kTileX = 64, kTileY = 4;

template <bool UQ>
__device__ __forceinline__ float widen(uint32_t bits)
{
    return UQ ? uq_widen(bits) : __builtin_bit_cast(float, bits);
}
// conversion of a promoted (float) channel back to the storage type: identity for float, uq32.h:13 for UQ32
template <bool UQ>
__device__ __forceinline__ uint32_t narrow(float x)
{
    return UQ ? uq_narrow(x) : __builtin_bit_cast(uint32_t, x);
}

template <int C, bool UQ, bool NO_SLIP>
__global__ void __launch_bounds__(kTileX *kTileY)
advect_channels_kernel(uint32_t *__restrict__ next_p, const uint32_t *__restrict__ p, const float2 *__restrict__ vel,
                       int dim_x, int dim_y, float dt)
{
    const int i = blockIdx.x * kTileX + threadIdx.x;
    const int j = blockIdx.y * kTileY + threadIdx.y;
    if (i >= dim_x || j >= dim_y) return;
    const size_t c = (size_t)dim_x * j + i;
    const float2 u = vel[c];
    const float si = (float)i - u.x * dt;  // advect.h:81
    const float sj = (float)j - u.y * dt;
    const SrcPos s = classify(si, sj, dim_x, dim_y);
    const uint32_t *t = p + (size_t)C * ((size_t)dim_x * s.cj + s.ci);
    uint32_t out[C];
    if (!s.x_oob && !s.y_oob) {  // advect.h:37-42: narrowed once, on return
        const uint32_t *n = t + (size_t)C * dim_x;
#pragma unroll
        for (int k = 0; k < C; ++k)
            out[k] = narrow<UQ>(mix1(s.di, mix1(s.dj, widen<UQ>(t[k]), widen<UQ>(n[k])),
                                     mix1(s.dj, widen<UQ>(t[C + k]), widen<UQ>(n[C + k]))));
    } else {
        // "T p_edge" (advect.h:44-55): the corner texel as stored; a wall value narrowed to T once
        if (s.x_oob && s.y_oob) {
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = t[k];
        } else if (s.x_oob) {
            const uint32_t *n = t + (size_t)C * dim_x;
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = narrow<UQ>(mix1(s.dj, widen<UQ>(t[k]), widen<UQ>(n[k])));
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = narrow<UQ>(mix1(s.di, widen<UQ>(t[k]), widen<UQ>(t[C + k])));
        }
        if (NO_SLIP) {  // advect.h:61-71: widened again, scaled, narrowed again
            const float f = wall_discount(s, si, sj, dim_x, dim_y);
#pragma unroll
            for (int k = 0; k < C; ++k) out[k] = narrow<UQ>(f * widen<UQ>(out[k]));
        }
    }
    uint32_t *o = next_p + (size_t)C * c;
#pragma unroll
    for (int k = 0; k < C; ++k) o[k] = out[k];
}

template <int C, bool UQ>
hipError_t launch_c(hipStream_t s, uint32_t *next_p, const uint32_t *p, const float *vel, int dim_x, int dim_y,
                    float dt, bool no_slip)
{
    const dim3 block(kTileX, kTileY, 1);
    const dim3 grid((dim_x + kTileX - 1) / kTileX, (dim_y + kTileY - 1) / kTileY, 1);
    auto *v = reinterpret_cast<const float2 *>(vel);
    if (no_slip)
        advect_channels_kernel<C, UQ, true><<<grid, block, 0, s>>>(next_p, p, v, dim_x, dim_y, dt);
    else
        advect_channels_kernel<C, UQ, false><<<grid, block, 0, s>>>(next_p, p, v, dim_x, dim_y, dt);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_advect_channels(hipStream_t s, void *next_p, const void *p, const float *vel, int dim_x, int dim_y,
                                  float dt, bool no_slip, int channels, int kind)
{
    const Slab g{dim_x, dim_y, 0, dim_y};
    auto *o = static_cast<uint32_t *>(next_p);
    auto *q = static_cast<const uint32_t *>(p);
    // the sketch's two element types: the tuned kernels (LDS-staged tiles from 16 K cells on)
    if (channels == 2 && kind == 0)
        return launch_advect_vec2f(s, static_cast<float *>(next_p), static_cast<const float *>(p), vel, g, 0, dim_y, 0,
                                   dim_y, dt, no_slip, nullptr, nullptr, 0);
    if (channels == 3 && kind == 1)
        return launch_advect_vec3uq32(s, o, q, vel, g, 0, dim_y, 0, dim_y, dt, no_slip, nullptr, nullptr, 0);
    switch (channels * 2 + (kind ? 1 : 0)) {
        case 2: return launch_c<1, false>(s, o, q, vel, dim_x, dim_y, dt, no_slip);
        case 3: return launch_c<1, true>(s, o, q, vel, dim_x, dim_y, dt, no_slip);
        case 5: return launch_c<2, true>(s, o, q, vel, dim_x, dim_y, dt, no_slip);
        case 6: return launch_c<3, false>(s, o, q, vel, dim_x, dim_y, dt, no_slip);
    }
    return hipErrorInvalidValue;
}

}  // namespace sfl
