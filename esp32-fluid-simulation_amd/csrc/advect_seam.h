// advect_seam.h -- the kernel that joins two sim steps inside sfl_step_n (SFL_OPT_STEP_SEAMS): subtract_gradient + dye advection of
// step k and velocity advection + divergence of step k + 1 in one pass over memory.  Included by advect_tiled.hip (it uses that file's
// tile geometry, windows and sampling helpers); reference: ESP32-fluid-simulation.ino:249-289, finitediff.cpp:9-82, advect.h:24-85.
#pragma once

// ---- the seam between two steps -------------------------------------------------------------------------------
// Inside sfl_step_n the last kernel of step k (subtract_gradient + dye advection, ino:276 + ino:281-287) and the first
// kernel of step k + 1 (velocity advection + divergence, ino:252-256 + ino:274) are ONE kernel: the projected velocity of
// step k is the field step k + 1 advects, and apart from the dye's back-trace nobody else ever reads it -- so it is
// produced in LDS, on the window the advection needs (the pressure window is one cell wider), used by both halves and
// never written to memory: 8 B per cell less to write, the 74 x 42 window per tile less to read back.  Same arithmetic
// in the same order as the two kernels above (a back-trace that leaves a window projects the texels it needs on the fly
// from memory: the same expressions, so the same bits); whole-domain contexts.
constexpr int kPX = kDX + 2, kPY = kDY + 2;   // pressure window

// finitediff.cpp:41-73 for one cell: v - grad p, a missing neighbour's pressure is the cell's own
template <class P>
__device__ __forceinline__ float2 project_cell(float2 u, int i, int gj, int i_max, int j_max, float two_dx_inv, P pressure_at)
{
    const float pc = pressure_at(i, gj);
    const float pw = (i > 0) ? pressure_at(i - 1, gj) : pc;
    const float pe = (i < i_max) ? pressure_at(i + 1, gj) : pc;
    const float ps = (gj > 0) ? pressure_at(i, gj - 1) : pc;
    const float pn = (gj < j_max) ? pressure_at(i, gj + 1) : pc;
    const float gx = (pe - pw) * two_dx_inv;
    const float gy = (pn - ps) * two_dx_inv;
    u.x = u.x - gx;
    u.y = u.y - gy;
    return u;
}

// sample() (advect.h:37-72) of the PROJECTED velocity, texels projected on the fly from v and p in memory
template <bool NO_SLIP>
__device__ __forceinline__ float2 sample_global_projected(const float2 *v, const float *p, const Slab &g, const SrcPos &s,
                                                          float si, float sj, float two_dx_inv)
{
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    auto texel = [&](int i, int gj) {
        return project_cell(v[lcell(g, i, gj)], i, gj, i_max, j_max, two_dx_inv,
                            [&](int a, int b) { return p[lcell(g, a, b)]; });
    };
    float2 r;
    if (!s.x_oob && !s.y_oob) {
        const float2 p11 = texel(s.ci, s.cj), p12 = texel(s.ci, s.cj + 1), p21 = texel(s.ci + 1, s.cj),
                     p22 = texel(s.ci + 1, s.cj + 1);
        r.x = mix1(s.di, mix1(s.dj, p11.x, p12.x), mix1(s.dj, p21.x, p22.x));
        r.y = mix1(s.di, mix1(s.dj, p11.y, p12.y), mix1(s.dj, p21.y, p22.y));
    } else {
        if (s.x_oob && s.y_oob) {
            r = texel(s.ci, s.cj);
        } else if (s.x_oob) {
            const float2 a = texel(s.ci, s.cj), b = texel(s.ci, s.cj + 1);
            r.x = mix1(s.dj, a.x, b.x);
            r.y = mix1(s.dj, a.y, b.y);
        } else {
            const float2 a = texel(s.ci, s.cj), b = texel(s.ci + 1, s.cj);
            r.x = mix1(s.di, a.x, b.x);
            r.y = mix1(s.di, a.y, b.y);
        }
        if (NO_SLIP) {
            const float f = wall_discount(s, si, sj, g.dim_x, g.gdim_y);
            r.x = r.x * f;
            r.y = r.y * f;
        }
    }
    return r;
}

// 512 threads; the register allocator leaves room for 6 waves per SIMD = three blocks per CU (80 VGPRs; two blocks at the 86 it
// would take by itself: 836 against 780 us; four blocks at 64 VGPRs spill: 1390 us)
#ifndef SEAM_DYE_LOADS
#define SEAM_DYE_LOADS 0   // where the dye window's loads are issued: 0 = behind the velocity's advection, 1 = in front of it (A/B)
#endif
#ifndef SEAM_MOCK_NO_P
#define SEAM_MOCK_NO_P 0   // TIMING MOCK (wrong results): the pressure window is not loaded -- what would the seam cost if the last launch of
                           // the solve handed the pressure over without memory? (profiles/r06_pressure_never_stored.txt)
#endif
#if SEAM_MOCK_NO_P && !defined(SFL_ALLOW_TIMING_MOCKS)
#error "SEAM_MOCK_NO_P is a timing mock (wrong results): diagnostic builds only (tools/recipes/build_variant.sh lib ... with -DSFL_ALLOW_TIMING_MOCKS)"
#endif
#ifndef SEAM_THREADS
#define SEAM_THREADS 512
#endif
constexpr int kThreadsSeam = SEAM_THREADS;
template <int THREADS>
__global__ void __launch_bounds__(THREADS, 6)
seam_tiled_kernel(uint32_t *__restrict__ next_col, const uint32_t *col, float2 *__restrict__ next_v,
                  float *__restrict__ div, const float2 *v, const float *pressure, Slab g, TileGrid tg, float dt,
                  float two_dx_inv)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kPlane = kSY * kSX;
    constexpr int kLoadsV = (kDX * kDY + THREADS - 1) / THREADS;
    constexpr int kLoadsC = (kPlane + THREADS - 1) / THREADS;
    static_assert(kRing <= THREADS, "one ring cell per thread");
    constexpr int kLoadsP = (kPX * kPY + THREADS - 1) / THREADS;
    constexpr int kWords = kPX * kPY + 2 * kDX * kDY;   // pressure window + velocity window, in 4-byte words
    static_assert(3 * kPlane <= kWords && kVX * kVY * 2 <= kWords, "one LDS buffer, tenants in turn");
    // ONE buffer (38.2 KB: four blocks per CU), used in turn by the pressure window + the projected velocity window of step
    // k, the dye window, and the advected velocities of step k + 1; what has to survive a change of tenant waits in
    // registers (the dye texels while the velocity is projected and advected, the advected cells while the dye is).
    __shared__ uint32_t lds[kWords];
    float *lds_p = reinterpret_cast<float *>(lds);
    float2 *lds_v = reinterpret_cast<float2 *>(lds + kPX * kPY);
    static_assert((kPX * kPY) % 2 == 0, "the velocity window starts 8-byte aligned");
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    const int x0 = tx * kTX, y0 = ty * kTY;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    const Window wv = window_of<kRD>(x0, y0, g, 0, g.gdim_y);
    const Window wc = window_of<kR>(x0, y0, g, 0, g.gdim_y);
    const int px0 = x0 - kRD - 1, py0 = y0 - kRD - 1;   // the pressure window is one cell wider than the velocity window
    uq3 got_c[kLoadsC];
    {   // every load of the block in flight before the first LDS write
        float2 got_v[kLoadsV];
        float got_p[kLoadsP];
#pragma unroll
        for (int k = 0; k < kLoadsP; ++k) {
            const int e = threadIdx.x + k * THREADS;
            const int r = e / kPX, gi = px0 + (e - r * kPX), gj = py0 + r;
            got_p[k] = (!SEAM_MOCK_NO_P && r < kPY && gi >= 0 && gi <= i_max && gj >= 0 && gj <= j_max) ? pressure[lcell(g, gi, gj)]
                                                                                                       : (SEAM_MOCK_NO_P ? (float)(gi + gj) : 0.0f);
        }
#pragma unroll
        for (int k = 0; k < kLoadsV; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got_v[k] = window_has<kDX>(wv, e) ? v[window_cell<kDX>(wv, g, e)] : float2{0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < kLoadsP; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kPX * kPY) lds_p[e] = got_p[k];
        }
        __syncthreads();
        // ino:276 on the velocity window: every thread projects the cells it loaded, pressure from LDS (five scattered loads
        // per window cell from memory cost 300 of the kernel's 970 us: profiles/r04_step_seam.txt)
#pragma unroll
        for (int k = 0; k < kLoadsV; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e >= kDX * kDY) break;
            float2 u = got_v[k];
            if (window_has<kDX>(wv, e)) {
                const int r = e / kDX, gi = wv.sx0 + (e - r * kDX), gj = wv.sy0 + r;
                u = project_cell(u, gi, gj, i_max, j_max, two_dx_inv,
                                 [&](int a, int b) { return lds_p[(b - py0) * kPX + (a - px0)]; });
            }
            lds_v[e] = u;
        }
    }
    __syncthreads();
#if SEAM_DYE_LOADS == 1
#pragma unroll
    for (int k = 0; k < kLoadsC; ++k) {
        const int e = threadIdx.x + k * THREADS;
        got_c[k] = window_has<kSX>(wc, e) ? load_uq3(col, window_cell<kSX>(wc, g, e)) : uq3{0u, 0u, 0u};
    }
#endif

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    const bool column = i < g.dim_x;
    // ino:252-256 of step k + 1: the projected velocity advects itself (no-slip), tile and the ring around it
    auto advected = [&](int ai, int agj) -> float2 {
        const float2 u = lds_v[(agj - wv.sy0) * kDX + (ai - wv.sx0)];
        const float si = (float)ai - u.x * dt;
        const float sj = (float)agj - u.y * dt;
        const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
        if (in_window(wv, s)) {
            const float2 *q = lds_v + (s.cj - wv.sy0) * kDX + (s.ci - wv.sx0);
            const float2 p11 = q[0], p21 = q[1], p12 = q[kDX], p22 = q[kDX + 1];
            float2 r;
            r.x = mix1(s.di, mix1(s.dj, p11.x, p12.x), mix1(s.dj, p21.x, p22.x));
            r.y = mix1(s.di, mix1(s.dj, p11.y, p12.y), mix1(s.dj, p21.y, p22.y));
            return r;
        }
        return sample_global_projected<true>(v, pressure, g, s, si, sj, two_dx_inv);
    };
    float2 mine[kRows], own[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        mine[r] = own[r] = float2{0.0f, 0.0f};
        if (column && gj < g.gdim_y) {
            own[r] = lds_v[(gj - wv.sy0) * kDX + (i - wv.sx0)];   // the dye's back-trace below needs it
            mine[r] = advected(i, gj);
            next_v[lcell(g, i, gj)] = mine[r];
        }
    }
    const int t = threadIdx.x;
    int ri = -1, rj = -1;
    if (t < kVX) { ri = x0 - 1 + t; rj = y0 - 1; }
    else if (t < 2 * kVX) { ri = x0 - 1 + (t - kVX); rj = y0 + kTY; }
    else if (t < 2 * kVX + kTY) { ri = x0 - 1; rj = y0 + (t - 2 * kVX); }
    else if (t < kRing) { ri = x0 + kTX; rj = y0 + (t - 2 * kVX - kTY); }
    const bool ring = ri >= 0 && ri < g.dim_x && rj >= 0 && rj < g.gdim_y;
    float2 around = float2{0.0f, 0.0f};
    if (ring) around = advected(ri, rj);
    // The dye window is not needed before this point, and its 18 registers per thread are what pushed the kernel over the
    // 80 VGPRs that let three blocks share a CU: with every load of the block up front it spilled 9 registers (16 B of
    // scratch per thread = the 203 MB of writes nobody could explain in profiles/r04_sim_step_summary.txt).  Its loads are
    // issued HERE; the other two blocks of the CU cover their latency.
#if SEAM_DYE_LOADS == 0
#pragma unroll
    for (int k = 0; k < kLoadsC; ++k) {
        const int e = threadIdx.x + k * THREADS;
        got_c[k] = window_has<kSX>(wc, e) ? load_uq3(col, window_cell<kSX>(wc, g, e)) : uq3{0u, 0u, 0u};
    }
#endif
    __syncthreads();   // everybody is done with the velocity window: the dye window moves in
#pragma unroll
    for (int k = 0; k < kLoadsC; ++k) {
        const int e = threadIdx.x + k * THREADS;
        if (e < kPlane) {
            lds[e] = got_c[k].x;
            lds[kPlane + e] = got_c[k].y;
            lds[2 * kPlane + e] = got_c[k].z;
        }
    }
    __syncthreads();
    // ino:281-287: the dye of step k, back-traced with the cell's own projected velocity (free-slip)
    if (column) {
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int gj = y0 + wave + kWaves * r;
            if (gj >= g.gdim_y) break;
            const float2 u = own[r];
            const float si = (float)i - u.x * dt;
            const float sj = (float)gj - u.y * dt;
            const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
            uq3 res;
            if (in_window(wc, s)) {
                const uint32_t *q = lds + (s.cj - wc.sy0) * kSX + (s.ci - wc.sx0);
                uint32_t out[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t *qk = q + k * kPlane;
                    const float p11 = uq_widen(qk[0]), p21 = uq_widen(qk[1]);
                    const float p12 = uq_widen(qk[kSX]), p22 = uq_widen(qk[kSX + 1]);
                    out[k] = uq_narrow(mix1(s.di, mix1(s.dj, p11, p12), mix1(s.dj, p21, p22)));
                }
                res = {out[0], out[1], out[2]};
            } else {
                res = sample_global_uq3<false>(col, g, s, si, sj);
            }
            uint32_t *o = next_col + 3 * lcell(g, i, gj);
            o[0] = res.x;
            o[1] = res.y;
            o[2] = res.z;
        }
    }
    __syncthreads();   // everybody is done with the dye window: the advected velocities move in
    float2 *adv = reinterpret_cast<float2 *>(lds);
#pragma unroll
    for (int r = 0; r < kRows; ++r) adv[(wave + kWaves * r + 1) * kVX + lane + 1] = mine[r];
    if (ring) adv[(rj - (y0 - 1)) * kVX + (ri - (x0 - 1))] = around;
    __syncthreads();

    // ino:274 of step k + 1 (finitediff.cpp:9-39)
    if (!column) return;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int row = wave + kWaves * r, gj = y0 + row;
        if (gj >= g.gdim_y) break;
        const float2 *q = adv + (row + 1) * kVX + lane + 1;
        float sdiv;
        if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // div_expr_fast, finitediff.cpp:29
            const float hx = -q[-1].x + q[1].x;
            const float hy = -q[-kVX].y + q[kVX].y;
            sdiv = hx + hy;
        } else {  // div_expr_safe, :15-20: ghost velocity = -own
            const float2 o2 = q[0];
            sdiv = 0.0f;
            sdiv += (i > 0) ? -q[-1].x : o2.x;
            sdiv += (i < i_max) ? q[1].x : -o2.x;
            sdiv += (gj > 0) ? -q[-kVX].y : o2.y;
            sdiv += (gj < j_max) ? q[kVX].y : -o2.y;
        }
        div[lcell(g, i, gj)] = sdiv * two_dx_inv;
    }
}
