// advect_tiled.hip -- semi-Lagrangian advection with the source footprint of a block staged in LDS
// (gfx950 / MI355X).
//
// A block owns a tile of 64 x 32 output cells and first copies the 72 x 40 window of the advected
// field around it (4 cells of margin: a back-trace of up to 4 cells per step stays inside) from memory
// into LDS with coalesced loads -- 1.4 loads per cell instead of the 4 scattered texel loads + 1 of the
// one-thread-per-cell kernels, whose gather traffic through the vector L1 adds to their HBM time
// (profiles/r01_advect_coherence_probe.txt).  Every cell then back-traces with its own velocity and
// interpolates from the LDS window; a back-trace that leaves the window (or the domain: the wall rules
// of advect.h:37-72) takes the same arithmetic on texels read from memory, so the result does not
// depend on where a texel came from.  Tiles are handed to the XCDs in contiguous ranges so that the
// margins shared by neighbouring tiles are served by one L2.
//
// Numerics contract: -ffp-contract=off, every operation individually rounded in the
// reference's order (advect_math.h).  Reference citations are file:line under
// /root/reference/ESP32-fluid-simulation/.
#include "advect_math.h"
#include "kernels.h"

namespace sfl {
namespace {

using namespace advect_math;

// Measured at 8192^2 (tools/r02_runs/r02_run26.sh): tiles of 16 / 32 / 64 rows 188 / 190 / 235 us (velocity),
// margins of 2 / 4 / 6 cells 209 / 190 / 194 us on per-cell noise of +-3.3 cells -- 32 rows, 4 cells.
#ifndef SFL_TILE_ROWS
#define SFL_TILE_ROWS 32   // (build-time knobs for sweeps: tools/recipes/build_variant.sh)
#endif
constexpr int kTX = 64, kTY = SFL_TILE_ROWS;        // output cells of a tile
constexpr int kR = 4;                    // margin of the staged window (cells)
constexpr int kSX = kTX + 2 * kR, kSY = kTY + 2 * kR;
constexpr int kXcds = 8;

struct TileGrid {
    int nx, ny;   // tiles per row of tiles, rows of tiles
    int per_xcd;  // tiles per XCD (rounded up)
};

// Workgroups are dispatched round-robin over the 8 XCDs: block b runs on XCD b % 8.  Give XCD k the k-th contiguous eighth of the
// tiles, so that neighbouring tiles share an L2 -- in an order in which the sharing happens while the lines are still there:
// STRIP-major.  A tile re-reads its neighbours' margins (window of kR.. cells around 64 x 32 cells: 31 % more rows, 16 % more
// columns); an XCD holds ~96 tiles at a time, and in plain row-major order (rounds 2 - 5) the tile ABOVE came a whole row of tiles
// later -- 128 tiles at 8192 columns, 12 MB of traffic ago, long evicted from the 4 MB L2.  Strips of kStripTiles tile columns are
// walked row by row instead: the tile above is kStripTiles positions away, resident at the same time or just finished (8192^2:
// every XCD walks one strip of 16 x 256 tiles top to bottom).  Speed only: any placement computes the same result.
#ifndef SFL_STRIP_TILES
#define SFL_STRIP_TILES 32   // (a build-time knob for the sweep: tools/recipes/build_variant.sh)
#endif
constexpr int kStripTiles = SFL_STRIP_TILES;
__device__ __forceinline__ bool tile_of_block(const TileGrid &t, int &tx, int &ty)
{
    const int b = blockIdx.x;
    const int q = (b % kXcds) * t.per_xcd + b / kXcds;   // position in the strip-major order
    if (q >= t.nx * t.ny || b / kXcds >= t.per_xcd) return false;
    if (kStripTiles <= 0) {   // (row-major, as before)
        ty = q / t.nx;
        tx = q - ty * t.nx;
        return true;
    }
    const int per_strip = kStripTiles * t.ny;
    const int s = q / per_strip, r = q - s * per_strip;
    const int w = min(kStripTiles, t.nx - s * kStripTiles);   // the last strip may be narrower
    ty = r / w;
    tx = s * kStripTiles + (r - ty * w);
    return true;
}

struct Window {
    int sx0, sy0;            // global cell held by window element (0, 0)
    int lx0, lx1, ly0, ly1;  // loaded part of the window: columns [lx0, lx1), rows [ly0, ly1)
};

// window of MARGIN cells around the tile at (x0, y0), clipped to the domain and to the rows that may be read
template <int MARGIN>
__device__ __forceinline__ Window window_of(int x0, int y0, const Slab &gs, int valid_begin, int valid_end)
{
    Window w;
    w.sx0 = x0 - MARGIN;
    w.sy0 = y0 - MARGIN;
    w.lx0 = max(w.sx0, 0);
    w.lx1 = min(x0 + kTX + MARGIN, gs.dim_x);
    w.ly0 = max(w.sy0, max(valid_begin, 0));
    w.ly1 = min(y0 + kTY + MARGIN, min(valid_end, gs.gdim_y));
    return w;
}

// window element e (row-major, SX elements per row) -> is it part of the loaded window / its cell in the array
template <int SX>
__device__ __forceinline__ bool window_has(const Window &w, int e)
{
    const int r = e / SX, cx = e - r * SX;
    const int gi = w.sx0 + cx, gj = w.sy0 + r;
    return gi >= w.lx0 && gi < w.lx1 && gj >= w.ly0 && gj < w.ly1;  // false beyond the last window row too
}
template <int SX>
__device__ __forceinline__ size_t window_cell(const Window &w, const Slab &gs, int e)
{
    const int r = e / SX, cx = e - r * SX;
    return lcell(gs, w.sx0 + cx, w.sy0 + r);
}

// all four texels of an in-domain sample lie in the loaded window
__device__ __forceinline__ bool in_window(const Window &w, const SrcPos &s)
{
    return !s.x_oob && !s.y_oob && s.ci >= w.lx0 && s.ci + 1 < w.lx1 && s.cj >= w.ly0 && s.cj + 1 < w.ly1;
}

// sample() of a float2 field whose window sits in LDS (SX elements per row); the rare back-trace that
// leaves the window or the domain reads memory instead -- the same arithmetic either way
template <bool NO_SLIP, int SX>
__device__ __forceinline__ float2 sample_window_vec2f(const float2 *tile, const Window &w, const float2 *p,
                                                      const Slab &gs, const SrcPos &s, float si, float sj)
{
    if (in_window(w, s)) {
        const float2 *q = tile + (s.cj - w.sy0) * SX + (s.ci - w.sx0);
        const float2 p11 = q[0], p21 = q[1], p12 = q[SX], p22 = q[SX + 1];
        float2 r;
        r.x = mix1(s.di, mix1(s.dj, p11.x, p12.x), mix1(s.dj, p21.x, p22.x));
        r.y = mix1(s.di, mix1(s.dj, p11.y, p12.y), mix1(s.dj, p21.y, p22.y));
        return r;
    }
    return sample_global_vec2f<NO_SLIP>(p, gs, s, si, sj);
}

// ---- advect<Vector2<float>, float>  (advect.h:24-85) -------------------------------------------
// THREADS / 64 waves; wave k owns rows k, k + waves, ... of the tile, a lane one column.
// SELF: the field advects itself (p == vel with the same geometry, ino:252-253) -- a cell's own
// velocity is then read from the window too.
template <bool NO_SLIP, bool SELF, int THREADS>
__global__ void __launch_bounds__(THREADS)
advect_vec2f_tiled_kernel(float2 *__restrict__ next_p, const float2 *p, const float2 *vel, Slab g, Slab gs,
                          TileGrid tg, int g_begin, int g_end, int valid_begin, int valid_end, float dt,
                          int *halo_flag, int ny1, int g2_begin, int g2_end)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kLoads = (kSX * kSY + THREADS - 1) / THREADS;
    __shared__ float2 tile[kSY * kSX];
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    // a launch covers up to two row ranges (the two bands of a slab next to its cuts in one launch: operators.cpp
    // advect_velocity_planned): the rows of tiles from ny1 on belong to [g2_begin, g2_end)
    if (ty >= ny1) {
        g_begin = g2_begin - ny1 * kTY;
        g_end = g2_end;
    }
    const int x0 = tx * kTX, y0 = g_begin + ty * kTY;
    const Window w = window_of<kR>(x0, y0, gs, valid_begin, valid_end);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    const bool column = i < g.dim_x;
    float2 own[kRows];
    {   // every load of the block in flight before the first LDS write
        float2 got[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got[k] = window_has<kSX>(w, e) ? p[window_cell<kSX>(w, gs, e)] : float2{0.0f, 0.0f};
        }
        if (!SELF) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int gj = y0 + wave + kWaves * r;
                own[r] = (column && gj < g_end) ? vel[lcell(g, i, gj)] : float2{0.0f, 0.0f};
            }
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kSX * kSY) tile[e] = got[k];
        }
    }
    __syncthreads();

    if (!column) return;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        if (gj >= g_end) break;
        const float2 u = SELF ? tile[(gj - w.sy0) * kSX + (i - w.sx0)] : own[r];
        const float si = (float)i - u.x * dt;  // advect.h:81
        const float sj = (float)gj - u.y * dt;
        const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
        if (!rows_available(s, valid_begin, valid_end)) {
            if (halo_flag) atomicOr(halo_flag, 1);
            continue;
        }
        next_p[lcell(g, i, gj)] = sample_window_vec2f<NO_SLIP, kSX>(tile, w, p, gs, s, si, sj);
    }
}

// ---- advect (ino:252-256) + calculate_divergence (ino:274, finitediff.cpp:9-39) in one pass ---------
// Whole-domain contexts, self-advection: the block advects its tile AND the ring of cells around it (from a
// window with one more cell of margin), parks the advected velocities in LDS and differences them there,
// so that the new velocity field is written once and never read back for its divergence.
constexpr int kRD = kR + 1;
constexpr int kDX = kTX + 2 * kRD, kDY = kTY + 2 * kRD;   // window
constexpr int kVX = kTX + 2, kVY = kTY + 2;               // advected cells kept in LDS
constexpr int kRing = 2 * kVX + 2 * kTY;                  // cells around the tile

template <bool NO_SLIP, int THREADS>
__global__ void __launch_bounds__(THREADS)
advect_divergence_tiled_kernel(float2 *__restrict__ next_v, float *__restrict__ div, const float2 *v, Slab g,
                               TileGrid tg, float dt, float two_dx_inv)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kLoads = (kDX * kDY + THREADS - 1) / THREADS;
    static_assert(kRing <= THREADS, "one ring cell per thread");
    static_assert(kVX * kVY <= kDX * kDY, "the advected cells reuse the window's LDS");
    __shared__ float2 lds[kDY * kDX];   // the window, then (from element 0) the kVY x kVX advected cells
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    const int x0 = tx * kTX, y0 = ty * kTY;
    const Window w = window_of<kRD>(x0, y0, g, 0, g.gdim_y);
    {
        float2 got[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got[k] = window_has<kDX>(w, e) ? v[window_cell<kDX>(w, g, e)] : float2{0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kDX * kDY) lds[e] = got[k];
        }
    }
    __syncthreads();

    auto advected = [&](int i, int gj) -> float2 {  // advect.h:78-84 for cell (i, gj) of the domain
        const float2 u = lds[(gj - w.sy0) * kDX + (i - w.sx0)];
        const float si = (float)i - u.x * dt;
        const float sj = (float)gj - u.y * dt;
        const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
        return sample_window_vec2f<NO_SLIP, kDX>(lds, w, v, g, s, si, sj);
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    const bool column = i < g.dim_x;
    float2 mine[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        mine[r] = float2{0.0f, 0.0f};
        if (column && gj < g.gdim_y) {
            mine[r] = advected(i, gj);
            next_v[lcell(g, i, gj)] = mine[r];
        }
    }
    // ring: bottom row, top row, left column, right column of the (kTX + 2) x (kTY + 2) block of cells
    const int t = threadIdx.x;
    int ri = -1, rj = -1;
    if (t < kVX) { ri = x0 - 1 + t; rj = y0 - 1; }
    else if (t < 2 * kVX) { ri = x0 - 1 + (t - kVX); rj = y0 + kTY; }
    else if (t < 2 * kVX + kTY) { ri = x0 - 1; rj = y0 + (t - 2 * kVX); }
    else if (t < kRing) { ri = x0 + kTX; rj = y0 + (t - 2 * kVX - kTY); }
    const bool ring = ri >= 0 && ri < g.dim_x && rj >= 0 && rj < g.gdim_y;
    float2 around = float2{0.0f, 0.0f};
    if (ring) around = advected(ri, rj);
    __syncthreads();   // everybody is done with the window
#pragma unroll
    for (int r = 0; r < kRows; ++r) lds[(wave + kWaves * r + 1) * kVX + lane + 1] = mine[r];
    if (ring) lds[(rj - (y0 - 1)) * kVX + (ri - (x0 - 1))] = around;
    __syncthreads();

    if (!column) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int row = wave + kWaves * r, gj = y0 + row;
        if (gj >= g.gdim_y) break;
        const float2 *q = lds + (row + 1) * kVX + lane + 1;
        float s;
        if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // div_expr_fast, finitediff.cpp:29
            const float hx = -q[-1].x + q[1].x;
            const float hy = -q[-kVX].y + q[kVX].y;
            s = hx + hy;
        } else {  // div_expr_safe, :15-20: ghost velocity = -own
            const float2 own = q[0];
            s = 0.0f;
            s += (i > 0) ? -q[-1].x : own.x;
            s += (i < i_max) ? q[1].x : -own.x;
            s += (gj > 0) ? -q[-kVX].y : own.y;
            s += (gj < j_max) ? q[kVX].y : -own.y;
        }
        div[lcell(g, i, gj)] = s * two_dx_inv;
    }
}

// ---- calculate_divergence (finitediff.cpp:9-39) and subtract_gradient (finitediff.cpp:41-82) as tiles ----------
// Stand-alone operators (inside sfl_step both are fused into the advections above).  Same tile, window margin 1:
// the 66 x 34 window of v (divergence) / of p (gradient) is staged in LDS, all loads of a block in flight at once.
constexpr int kFX = kTX + 2, kFY = kTY + 2;

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
divergence_tiled_kernel(float *__restrict__ div, const float2 *__restrict__ v, Slab g, TileGrid tg, int g_begin,
                        int g_end, float two_dx_inv)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kLoads = (kFX * kFY + THREADS - 1) / THREADS;
    __shared__ float2 win[kFY * kFX];
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    const int x0 = tx * kTX, y0 = g_begin + ty * kTY;
    const Window w = window_of<1>(x0, y0, g, g.grow0, g.grow0 + g.lrows);
    {
        float2 got[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got[k] = window_has<kFX>(w, e) ? v[window_cell<kFX>(w, g, e)] : float2{0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kFX * kFY) win[e] = got[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    if (i >= g.dim_x) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        if (gj >= g_end) break;
        const float2 *q = win + (gj - w.sy0) * kFX + (i - w.sx0);
        float s;
        if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // div_expr_fast, :29
            const float hx = -q[-1].x + q[1].x;
            const float hy = -q[-kFX].y + q[kFX].y;
            s = hx + hy;
        } else {  // div_expr_safe, :15-20: ghost velocity = -own
            const float2 own = q[0];
            s = 0.0f;
            s += (i > 0) ? -q[-1].x : own.x;
            s += (i < i_max) ? q[1].x : -own.x;
            s += (gj > 0) ? -q[-kFX].y : own.y;
            s += (gj < j_max) ? q[kFX].y : -own.y;
        }
        div[lcell(g, i, gj)] = s * two_dx_inv;
    }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
gradient_tiled_kernel(float2 *v, const float *__restrict__ p, Slab g, TileGrid tg, int g_begin, int g_end,
                      float two_dx_inv)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kLoads = (kFX * kFY + THREADS - 1) / THREADS;
    __shared__ float win[kFY * kFX];
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    const int x0 = tx * kTX, y0 = g_begin + ty * kTY;
    const Window w = window_of<1>(x0, y0, g, g.grow0, g.grow0 + g.lrows);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    const bool column = i < g.dim_x;
    float2 own[kRows];
    {
        float got[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got[k] = window_has<kFX>(w, e) ? p[window_cell<kFX>(w, g, e)] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int gj = y0 + wave + kWaves * r;
            own[r] = (column && gj < g_end) ? v[lcell(g, i, gj)] : float2{0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kFX * kFY) win[e] = got[k];
        }
    }
    __syncthreads();
    if (!column) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        if (gj >= g_end) break;
        const float *q = win + (gj - w.sy0) * kFX + (i - w.sx0);
        const float pc = q[0];
        const float pw = (i > 0) ? q[-1] : pc;      // finitediff.cpp:47-69: a missing neighbour is the cell itself
        const float pe = (i < i_max) ? q[1] : pc;
        const float ps = (gj > 0) ? q[-kFX] : pc;
        const float pn = (gj < j_max) ? q[kFX] : pc;
        const float gx = (pe - pw) * two_dx_inv;
        const float gy = (pn - ps) * two_dx_inv;
        float2 u = own[r];
        u.x = u.x - gx;
        u.y = u.y - gy;
        v[lcell(g, i, gj)] = u;
    }
}

// ---- advect<Vector3<UQ32>, float>  (advect.h:24-85 + uq32.h) -------------------------------------
// The window holds the three channels in three planes (4-byte LDS accesses, no 12-byte alignment
// question).  FUSE_GRAD as in stencil_kernels.hip: the projection of the cell's own velocity
// (finitediff.cpp:41-82) happens here, in place, before the back-trace -- and before the barrier, so
// that the loads of v and p travel together with those of the window.
// REACH (slabs, FUSE_GRAD): the back-traces of this kernel are those of the NEXT step's velocity advection -- same projected
// velocity, same dt, same rows -- so the kernel also leaves what backtrace_reach_kernel would measure afterwards (three launches,
// 31 us on a 1024-row slab, on the critical path between two steps): `halo_flag` then points at word [2] of a reach report
// (context.h kReachWords) and words [0], [1], [3] .. [7] are raised with atomicMax, one per wave and word.
template <bool NO_SLIP, bool FUSE_GRAD, int THREADS, bool REACH = false>
__global__ void __launch_bounds__(THREADS)
advect_vec3uq32_tiled_kernel(uint32_t *__restrict__ next_p, const uint32_t *p, float2 *vel, Slab g, Slab gs,
                             TileGrid tg, int g_begin, int g_end, int valid_begin, int valid_end, float dt,
                             int *halo_flag, const float *__restrict__ pressure, float two_dx_inv)
{
    constexpr int kWaves = THREADS / 64, kRows = kTY / kWaves;
    constexpr int kPlane = kSY * kSX;
    constexpr int kLoads = (kPlane + THREADS - 1) / THREADS;
    __shared__ uint32_t tile[3 * kPlane];
    int tx, ty;
    if (!tile_of_block(tg, tx, ty)) return;
    const int x0 = tx * kTX, y0 = g_begin + ty * kTY;
    const Window w = window_of<kR>(x0, y0, gs, valid_begin, valid_end);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = x0 + lane;
    const bool column = i < g.dim_x;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    float2 own[kRows];
    {
        uq3 got[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            got[k] = window_has<kSX>(w, e) ? load_uq3(p, window_cell<kSX>(w, gs, e)) : uq3{0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int gj = y0 + wave + kWaves * r;
            own[r] = float2{0.0f, 0.0f};
            if (column && gj < g_end) {
                const size_t c = lcell(g, i, gj);
                float2 u = vel[c];
                if (FUSE_GRAD) {
                    const float pc = pressure[c];
                    const float pw = (i > 0) ? pressure[c - 1] : pc;
                    const float pe = (i < i_max) ? pressure[c + 1] : pc;
                    const float ps = (gj > 0) ? pressure[c - g.dim_x] : pc;
                    const float pn = (gj < j_max) ? pressure[c + g.dim_x] : pc;
                    const float gx = (pe - pw) * two_dx_inv;
                    const float gy = (pn - ps) * two_dx_inv;
                    u.x = u.x - gx;
                    u.y = u.y - gy;
                    vel[c] = u;
                }
                own[r] = u;
            }
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int e = threadIdx.x + k * THREADS;
            if (e < kPlane) {
                tile[e] = got[k].x;
                tile[kPlane + e] = got[k].y;
                tile[2 * kPlane + e] = got[k].z;
            }
        }
    }
    __syncthreads();

    if (!REACH && !column) return;
    int need[7] = {0, 0, 0, 0, 0, 0, 0};   // REACH: report words [0], [1], [4], [5], [6], [7], [3]
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int gj = y0 + wave + kWaves * r;
        if (gj >= g_end || !column) break;
        const float2 u = own[r];
        const float si = (float)i - u.x * dt;
        const float sj = (float)gj - u.y * dt;
        const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
        if (REACH) {   // exactly backtrace_reach_kernel's expressions for rows [g_begin, g_end), [g_begin, g_begin + 1), [g_end - 1, g_end)
            const int top = s.cj + (s.y_oob ? 0 : 1);
            need[0] = max(need[0], g_begin - s.cj);
            need[1] = max(need[1], top - (g_end - 1));
            if (gj == g_begin) {   // (wave-uniform)
                need[2] = max(need[2], g_begin - s.cj);
                need[3] = max(need[3], top - g_begin);
            }
            if (gj == g_end - 1) {
                need[4] = max(need[4], g_end - 1 - s.cj);
                need[5] = max(need[5], top - (g_end - 1));
            }
            // [3]: how many rows from its own row a cell's sources lie at most -- rows further than that from both ends of the
            // slab never read beyond it (slab_step.cpp advect_interior_early)
            need[6] = max(need[6], max(gj - s.cj, top - gj));
        }
        if (!rows_available(s, valid_begin, valid_end)) {
            if (halo_flag) atomicOr(halo_flag, 1);
            continue;
        }
        uq3 res;
        if (in_window(w, s)) {
            const uint32_t *q = tile + (s.cj - w.sy0) * kSX + (s.ci - w.sx0);
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
            res = sample_global_uq3<NO_SLIP>(p, gs, s, si, sj);
        }
        uint32_t *o = next_p + 3 * lcell(g, i, gj);
        o[0] = res.x;
        o[1] = res.y;
        o[2] = res.z;
    }
    if (REACH) {   // every lane of the block is here again: one look at each word per BLOCK (word [3] is positive in every wave:
                   // an atomic per wave on it took 330 us, a look per wave still 50), an atomic only where there is something to add
        __syncthreads();   // (the window is no longer read: its LDS takes the waves' maxima)
        int *scratch = reinterpret_cast<int *>(tile);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            int v = need[k];
            for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
            if (lane == 0) scratch[wave * 8 + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int k = threadIdx.x;
            int v = 0;
            for (int w2 = 0; w2 < kWaves; ++w2) v = max(v, scratch[w2 * 8 + k]);
            int *word = halo_flag - 2 + (k < 2 ? k : k < 6 ? k + 2 : 3);
            if (v > 0 && v > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, v);
        }
    }
}

#include "advect_seam.h"

// threads per block: the dye kernel's window takes 34.5 KB of LDS (4 blocks per CU), so it needs 8 waves
// per block to fill a CU (392 us at 8192^2 against 404 with 4); the velocity kernel's 23 KB leave room
// for 6 blocks of 4 waves (193 us either way)
#ifndef SFL_DYE_THREADS
#define SFL_DYE_THREADS 512
#endif
constexpr int kThreadsVec2 = 256, kThreadsDye = SFL_DYE_THREADS;

TileGrid tile_grid(int dim_x, int rows)
{
    TileGrid t;
    t.nx = (dim_x + kTX - 1) / kTX;
    t.ny = (rows + kTY - 1) / kTY;
    t.per_xcd = (t.nx * t.ny + kXcds - 1) / kXcds;
    return t;
}

}  // namespace

hipError_t launch_advect_vec2f_tiled(hipStream_t s, float *next_p, const float *p, const float *vel, Slab g,
                                     int g_begin, int g_end, int valid_begin, int valid_end, float dt,
                                     bool no_slip, int *halo_flag, const Slab *src, int g2_begin, int g2_end)
{
    if (g_end <= g_begin) return hipSuccess;
    const Slab gs = src ? *src : g;
    const int ny1 = (g_end - g_begin + kTY - 1) / kTY, ny2 = g2_end > g2_begin ? (g2_end - g2_begin + kTY - 1) / kTY : 0;
    const TileGrid tg = tile_grid(g.dim_x, (ny1 + ny2) * kTY);
    const dim3 grid(tg.per_xcd * kXcds), block(kThreadsVec2);
    auto *o = reinterpret_cast<float2 *>(next_p);
    auto *pi = reinterpret_cast<const float2 *>(p);
    auto *vi = reinterpret_cast<const float2 *>(vel);
    const bool self = p == vel && !src;
#define SFL_GO(NS_, SELF_)                                                                 \
    advect_vec2f_tiled_kernel<NS_, SELF_, kThreadsVec2><<<grid, block, 0, s>>>(            \
        o, pi, vi, g, gs, tg, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, ny1, g2_begin, g2_end)
    if (no_slip) {
        if (self) { SFL_GO(true, true); } else { SFL_GO(true, false); }
    } else {
        if (self) { SFL_GO(false, true); } else { SFL_GO(false, false); }
    }
#undef SFL_GO
    return hipGetLastError();
}

hipError_t launch_advect_vec3uq32_tiled(hipStream_t s, uint32_t *next_p, const uint32_t *p, float *vel,
                                        const float *pressure, Slab g, int g_begin, int g_end, int valid_begin,
                                        int valid_end, float dt, bool no_slip, int *halo_flag,
                                        float two_dx_inv, const Slab *src, bool reach)
{
    if (g_end <= g_begin) return hipSuccess;
    const Slab gs = src ? *src : g;
    const TileGrid tg = tile_grid(g.dim_x, g_end - g_begin);
    const dim3 grid(tg.per_xcd * kXcds), block(kThreadsDye);
    auto *vi = reinterpret_cast<float2 *>(vel);
    if (reach) {   // (see the kernel: halo_flag = word [2] of a reach report, the projection fused in)
        if (!pressure || !halo_flag || src) return hipErrorInvalidValue;
        if (no_slip)
            advect_vec3uq32_tiled_kernel<true, true, kThreadsDye, true><<<grid, block, 0, s>>>(
                next_p, p, vi, g, gs, tg, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, pressure, two_dx_inv);
        else
            advect_vec3uq32_tiled_kernel<false, true, kThreadsDye, true><<<grid, block, 0, s>>>(
                next_p, p, vi, g, gs, tg, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, pressure, two_dx_inv);
        return hipGetLastError();
    }
#define SFL_GO(NS_, FG_)                                                                                    \
    advect_vec3uq32_tiled_kernel<NS_, FG_, kThreadsDye><<<grid, block, 0, s>>>(                             \
        next_p, p, vi, g, gs, tg, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, pressure, two_dx_inv)
    if (no_slip) {
        if (pressure) { SFL_GO(true, true); } else { SFL_GO(true, false); }
    } else {
        if (pressure) { SFL_GO(false, true); } else { SFL_GO(false, false); }
    }
#undef SFL_GO
    return hipGetLastError();
}

hipError_t launch_advect_divergence_tiled(hipStream_t s, float *next_v, float *div, const float *v, Slab g,
                                          float dt, bool no_slip, float two_dx_inv)
{
    const TileGrid tg = tile_grid(g.dim_x, g.gdim_y);
    const dim3 grid(tg.per_xcd * kXcds), block(512);
    auto *o = reinterpret_cast<float2 *>(next_v);
    auto *vi = reinterpret_cast<const float2 *>(v);
    if (no_slip)
        advect_divergence_tiled_kernel<true, 512><<<grid, block, 0, s>>>(o, div, vi, g, tg, dt, two_dx_inv);
    else
        advect_divergence_tiled_kernel<false, 512><<<grid, block, 0, s>>>(o, div, vi, g, tg, dt, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_step_seam_tiled(hipStream_t s, uint32_t *next_col, const uint32_t *col, float *next_v, float *div,
                                  const float *v, const float *pressure, Slab g, float dt, float two_dx_inv)
{
    const TileGrid tg = tile_grid(g.dim_x, g.gdim_y);
    seam_tiled_kernel<kThreadsSeam><<<dim3(tg.per_xcd * kXcds), dim3(kThreadsSeam), 0, s>>>(
        next_col, col, reinterpret_cast<float2 *>(next_v), div, reinterpret_cast<const float2 *>(v), pressure, g, tg, dt,
        two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_divergence_tiled(hipStream_t s, float *div, const float *v, Slab g, int g_begin, int g_end,
                                   float two_dx_inv)
{
    if (g_end <= g_begin) return hipSuccess;
    const TileGrid tg = tile_grid(g.dim_x, g_end - g_begin);
    divergence_tiled_kernel<256><<<dim3(tg.per_xcd * kXcds), dim3(256), 0, s>>>(
        div, reinterpret_cast<const float2 *>(v), g, tg, g_begin, g_end, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_gradient_tiled(hipStream_t s, float *v, const float *p, Slab g, int g_begin, int g_end,
                                 float two_dx_inv)
{
    if (g_end <= g_begin) return hipSuccess;
    const TileGrid tg = tile_grid(g.dim_x, g_end - g_begin);
    gradient_tiled_kernel<256><<<dim3(tg.per_xcd * kXcds), dim3(256), 0, s>>>(reinterpret_cast<float2 *>(v), p, g, tg,
                                                                            g_begin, g_end, two_dx_inv);
    return hipGetLastError();
}

}  // namespace sfl
