// advect_math.h -- per-point arithmetic of the semi-Lagrangian advection shared by the one-thread-per-
// cell kernels (stencil_kernels.hip) and the LDS-staged tile kernels (advect_tiled.hip).
//
// Numerics contract: compiled with -ffp-contract=off; every product and sum is
// individually rounded in the order the reference evaluates it.  Reference citations are file:line
// under /root/reference/ESP32-fluid-simulation/.
#pragma once
#include "kernels.h"

namespace sfl {
namespace advect_math {

__device__ __forceinline__ size_t lcell(const Slab &g, int i, int gj)
{
    return (size_t)(gj - g.grow0) * (size_t)g.dim_x + (size_t)i;
}

// lerp(t, a, b) = a*(1-t) + b*t  (advect.h:13-16)
__device__ __forceinline__ float mix1(float t, float a, float b)
{
    const float wa = 1.0f - t;
    const float pa = a * wa;
    const float pb = b * t;
    return pa + pb;
}

// uq32.h:13 / :15
__device__ __forceinline__ uint32_t uq_narrow(float x) { return (uint32_t)(x + 0.5f); }
__device__ __forceinline__ float uq_widen(uint32_t raw) { return (float)raw; }

struct SrcPos {
    bool x_under, y_under, x_oob, y_oob;
    int ci, cj;
    float di, dj;
};

// advect.h:26-35
__device__ __forceinline__ SrcPos classify(float si, float sj, int dim_x, int gdim_y)
{
    SrcPos s;
    const bool x_over = si >= (float)(dim_x - 1);
    const bool y_over = sj >= (float)(gdim_y - 1);
    const float fi = floorf(si), fj = floorf(sj);
    s.x_under = si < 0.0f;
    s.y_under = sj < 0.0f;
    s.x_oob = s.x_under || x_over;
    s.y_oob = s.y_under || y_over;
    s.di = si - fi;
    s.dj = sj - fj;
    s.ci = s.x_oob ? (s.x_under ? 0 : dim_x - 1) : (int)fi;
    s.cj = s.y_oob ? (s.y_under ? 0 : gdim_y - 1) : (int)fj;
    return s;
}

// advect.h:62-70
__device__ __forceinline__ float wall_discount(const SrcPos &s, float si, float sj, int dim_x,
                                               int gdim_y)
{
    float factor = 1.0f;
    if (s.x_oob) {
        const float over = s.x_under ? -si : si - (float)(dim_x - 1);
        factor *= (over < 0.5f) ? (1.0f - 2.0f * over) : 0.0f;
    }
    if (s.y_oob) {
        const float over = s.y_under ? -sj : sj - (float)(gdim_y - 1);
        factor *= (over < 0.5f) ? (1.0f - 2.0f * over) : 0.0f;
    }
    return factor;
}

// rows of p touched by a sample at s: [cj, cj + (y in range ? 1 : 0)]
__device__ __forceinline__ bool rows_available(const SrcPos &s, int valid_begin, int valid_end)
{
    const int last = s.cj + (s.y_oob ? 0 : 1);
    return s.cj >= valid_begin && last < valid_end;
}

struct uq3 {
    uint32_t x, y, z;
};

__device__ __forceinline__ uq3 load_uq3(const uint32_t *p, size_t cell)
{
    const uint32_t *q = p + 3 * cell;
    return {q[0], q[1], q[2]};
}

__device__ __forceinline__ uint32_t uq_mix(float t, uint32_t a, uint32_t b)
{
    return uq_narrow(mix1(t, uq_widen(a), uq_widen(b)));
}

// sample() of a float2 field from its array in memory (advect.h:37-72); gs = geometry of that array
template <bool NO_SLIP>
__device__ __forceinline__ float2 sample_global_vec2f(const float2 *p, const Slab &gs, const SrcPos &s, float si,
                                                      float sj)
{
    const size_t t = lcell(gs, s.ci, s.cj);
    float2 r;
    if (!s.x_oob && !s.y_oob) {
        const float2 p11 = p[t], p12 = p[t + gs.dim_x], p21 = p[t + 1], p22 = p[t + gs.dim_x + 1];
        r.x = mix1(s.di, mix1(s.dj, p11.x, p12.x), mix1(s.dj, p21.x, p22.x));
        r.y = mix1(s.di, mix1(s.dj, p11.y, p12.y), mix1(s.dj, p21.y, p22.y));
    } else {
        if (s.x_oob && s.y_oob) {
            r = p[t];
        } else if (s.x_oob) {
            const float2 a = p[t], b = p[t + gs.dim_x];
            r.x = mix1(s.dj, a.x, b.x);
            r.y = mix1(s.dj, a.y, b.y);
        } else {
            const float2 a = p[t], b = p[t + 1];
            r.x = mix1(s.di, a.x, b.x);
            r.y = mix1(s.di, a.y, b.y);
        }
        if (NO_SLIP) {
            const float f = wall_discount(s, si, sj, gs.dim_x, gs.gdim_y);
            r.x = r.x * f;
            r.y = r.y * f;
        }
    }
    return r;
}

// sample() of a Vector3<UQ32> field from its array in memory (advect.h:37-72 + uq32.h)
template <bool NO_SLIP>
__device__ __forceinline__ uq3 sample_global_uq3(const uint32_t *p, const Slab &gs, const SrcPos &s, float si,
                                                 float sj)
{
    const size_t t = lcell(gs, s.ci, s.cj);
    uq3 r;
    if (!s.x_oob && !s.y_oob) {
        const uq3 p11 = load_uq3(p, t), p12 = load_uq3(p, t + gs.dim_x);
        const uq3 p21 = load_uq3(p, t + 1), p22 = load_uq3(p, t + gs.dim_x + 1);
        r.x = uq_narrow(mix1(s.di, mix1(s.dj, uq_widen(p11.x), uq_widen(p12.x)),
                             mix1(s.dj, uq_widen(p21.x), uq_widen(p22.x))));
        r.y = uq_narrow(mix1(s.di, mix1(s.dj, uq_widen(p11.y), uq_widen(p12.y)),
                             mix1(s.dj, uq_widen(p21.y), uq_widen(p22.y))));
        r.z = uq_narrow(mix1(s.di, mix1(s.dj, uq_widen(p11.z), uq_widen(p12.z)),
                             mix1(s.dj, uq_widen(p21.z), uq_widen(p22.z))));
    } else {
        // "T p_edge" narrows once (advect.h:45-54); returned raw when !no_slip (:57-59)
        if (s.x_oob && s.y_oob) {
            r = load_uq3(p, t);
        } else if (s.x_oob) {
            const uq3 a = load_uq3(p, t), b = load_uq3(p, t + gs.dim_x);
            r = {uq_mix(s.dj, a.x, b.x), uq_mix(s.dj, a.y, b.y), uq_mix(s.dj, a.z, b.z)};
        } else {
            const uq3 a = load_uq3(p, t), b = load_uq3(p, t + 1);
            r = {uq_mix(s.di, a.x, b.x), uq_mix(s.di, a.y, b.y), uq_mix(s.di, a.z, b.z)};
        }
        if (NO_SLIP) {  // widen, scale, narrow again (advect.h:71)
            const float f = wall_discount(s, si, sj, gs.dim_x, gs.gdim_y);
            r.x = uq_narrow(uq_widen(r.x) * f);
            r.y = uq_narrow(uq_widen(r.y) * f);
            r.z = uq_narrow(uq_widen(r.z) * f);
        }
    }
    return r;
}

}  // namespace advect_math
}  // namespace sfl
