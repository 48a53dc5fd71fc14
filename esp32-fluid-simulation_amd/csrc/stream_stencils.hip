// stream_stencils.hip -- row-streaming divergence and pressure-gradient kernels for gfx950.
//
// The one-thread-per-cell kernels of stencil_kernels.hip issue 4-5 eight-byte loads per cell and
// use half of each (velocity is AoS {x, y}: a vertical neighbour is fetched for its .y only), so
// they are bound by the load path (~3 TB/s) although their HBM traffic is near ideal.  Here one
// 64-lane wave streams a strip of rows bottom-up: a lane owns two adjacent cells, every row is
// loaded exactly once with one 16-byte (velocity) / 8-byte (pressure) access per lane, vertical
// neighbours wait in a 4-row register ring (3-row window + 1 row in flight), horizontal
// neighbours come from DPP wave shifts.  Lanes 0 and 63 are halo lanes (their W / E neighbour
// would live in another wave), so a wave produces 124 columns.
//
// Same numerics contract as stencil_kernels.hip (-ffp-contract=off, reference evaluation order):
//   calculate_divergence  finitediff.cpp:9-39    interior ((-W.x + E.x) + (-S.y + N.y)) * k,
//                                                perimeter running sum from 0 with ghost = -own
//   subtract_gradient     finitediff.cpp:41-82   v - ((pE - pW) * k, (pN - pS) * k), a missing
//                                                neighbour's pressure is the cell's own
// Used when dim_x is even and the arrays are 16-byte aligned; otherwise the launchers in
// stencil_kernels.hip fall back to the one-thread-per-cell kernels (still HIP).
#include "kernels.h"

namespace sfl {
namespace {

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;
constexpr int kValidCols = 124;  // lanes 1..62 x 2 cells

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lane_below(float x)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_above(float x)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

struct TileGeom {
    int n_strips, n_chunks, rows_per_chunk;
};

// wave-uniform tile coordinates + per-lane columns shared by both kernels
struct Tile {
    int lane, x0, xa, r0, r1;
    int col_off;      // clamped even column of cell a (elements)
    bool out;         // this lane's two cells are produced by this wave
    bool edge;        // tile touches the domain perimeter (wave-uniform)
    bool valid;

    __device__ __forceinline__ Tile(const Slab &g, const TileGeom &t, int g_begin, int g_end)
    {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        const int tile = blockIdx.x * kWaves + wave;
        valid = tile < t.n_strips * t.n_chunks;
        const int chunk = tile / t.n_strips, strip = tile - chunk * t.n_strips;
        x0 = strip * kValidCols - 2;
        xa = x0 + 2 * lane;
        r0 = g_begin + chunk * t.rows_per_chunk;
        r1 = min(r0 + t.rows_per_chunk, g_end);
        col_off = min(max(xa, 0), g.dim_x - 2);
        out = lane >= 1 && lane <= 62 && xa >= 0 && xa < g.dim_x;
        edge = x0 <= 0 || x0 + 128 >= g.dim_x || r0 <= 1 || r1 >= g.gdim_y - 1;
    }
};

// ---- divergence ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
divergence_stream_kernel(float *div, const float *v, Slab g, TileGeom t, int g_begin, int g_end,
                         float k)
{
    const Tile tl(g, t, g_begin, g_end);
    if (!tl.valid) return;
    const size_t bytes_v = (size_t)g.lrows * g.dim_x * 8, bytes_d = (size_t)g.lrows * g.dim_x * 4;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(v), 0, bytes_v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes_v, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        div, 0, bytes_d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes_d, 0x00020000);
    const int row_lo = max(g.grow0, 0), row_hi = min(g.grow0 + g.lrows, g.gdim_y);
    const int voff = tl.col_off * 8, doff = tl.xa * 4;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;

    auto load = [&](int r) -> v4f {  // {a.x, a.y, b.x, b.y} of row r (row clamped: always valid)
        const int rc = min(max(r, row_lo), row_hi - 1);
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(
                                           rv, voff, (rc - g.grow0) * g.dim_x * 8, 0));
    };

    v4f ring[4];
    ring[0] = load(tl.r0 - 1);
    ring[1] = load(tl.r0);
    ring[2] = load(tl.r0 + 1);

    auto row = [&](int y, const v4f &prev, const v4f &cur, const v4f &next) {
        // W / E neighbours: cell a <- (lane-1).b | own b ; cell b <- own a | (lane+1).a
        const float wx_a = lane_below(cur.z), ex_a = cur.z;
        const float wx_b = cur.x, ex_b = lane_above(cur.x);
        float da = ((-wx_a + ex_a) + (-prev.y + next.y)) * k;   // div_expr_fast, :29-30
        float db = ((-wx_b + ex_b) + (-prev.w + next.w)) * k;
        if (tl.edge) {  // div_expr_safe, :15-22 on perimeter cells
            const bool bottom = y == 0, top = y == j_max;
            auto safe = [&](int i, float wx, float ex, float own_x, float own_y, float sy, float ny) {
                float s = 0.0f;
                s += (i > 0) ? -wx : own_x;
                s += (i < i_max) ? ex : -own_x;
                s += bottom ? own_y : -sy;
                s += top ? -own_y : ny;
                return s * k;
            };
            const bool pa = bottom || top || tl.xa == 0 || tl.xa == i_max;
            const bool pb = bottom || top || tl.xa + 1 == 0 || tl.xa + 1 == i_max;
            const float sa = safe(tl.xa, wx_a, ex_a, cur.x, cur.y, prev.y, next.y);
            const float sb = safe(tl.xa + 1, wx_b, ex_b, cur.z, cur.w, prev.w, next.w);
            da = pa ? sa : da;
            db = pb ? sb : db;
        }
        if (tl.out) {
            v2f o;
            o.x = da;
            o.y = db;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, o), rd, doff,
                                                  (y - g.grow0) * g.dim_x * 4, 0);
        }
    };

    for (int y = tl.r0; y < tl.r1; y += 4) {
#define SFL_DIV_STEP(U)                                                        \
    if (y + U < tl.r1) {                                                       \
        ring[(U + 3) & 3] = load(y + U + 2);                                   \
        row(y + U, ring[U & 3], ring[(U + 1) & 3], ring[(U + 2) & 3]);         \
    }
        SFL_DIV_STEP(0) SFL_DIV_STEP(1) SFL_DIV_STEP(2) SFL_DIV_STEP(3)
#undef SFL_DIV_STEP
    }
}

// ---- subtract_gradient (in place on v) --------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
gradient_stream_kernel(float *v, const float *p, Slab g, TileGeom t, int g_begin, int g_end, float k)
{
    const Tile tl(g, t, g_begin, g_end);
    if (!tl.valid) return;
    const size_t bytes_v = (size_t)g.lrows * g.dim_x * 8, bytes_p = (size_t)g.lrows * g.dim_x * 4;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        v, 0, bytes_v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes_v, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p), 0, bytes_p > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes_p, 0x00020000);
    const int row_lo = max(g.grow0, 0), row_hi = min(g.grow0 + g.lrows, g.gdim_y);
    const int poff = tl.col_off * 4, voff = tl.xa * 8;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;

    auto load_p = [&](int r) -> v2f {
        const int rc = min(max(r, row_lo), row_hi - 1);
        return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(
                                           rp, poff, (rc - g.grow0) * g.dim_x * 4, 0));
    };
    auto load_v = [&](int r) -> v4f {  // only lanes that will store use the value; clamp anyway
        const int rc = min(max(r, row_lo), row_hi - 1);
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(
                                           rv, tl.col_off * 8, (rc - g.grow0) * g.dim_x * 8, 0));
    };

    v2f ring[4];
    v4f vel[2];
    ring[0] = load_p(tl.r0 - 1);
    ring[1] = load_p(tl.r0);
    ring[2] = load_p(tl.r0 + 1);
    vel[0] = load_v(tl.r0);

    auto row = [&](int y, const v2f &prev, const v2f &cur, const v2f &next, const v4f &u) {
        float pw_a = lane_below(cur.y), pe_a = cur.y;   // grad_sub_expr_fast, :67-72
        float pw_b = cur.x, pe_b = lane_above(cur.x);
        float ps_a = prev.x, pn_a = next.x, ps_b = prev.y, pn_b = next.y;
        if (tl.edge) {  // grad_sub_expr_safe, :51-54: a missing neighbour is the cell itself
            if (tl.xa == 0) pw_a = cur.x;
            if (tl.xa == i_max) pe_a = cur.x;
            if (tl.xa + 1 == i_max) pe_b = cur.y;
            if (y == 0) { ps_a = cur.x; ps_b = cur.y; }
            if (y == j_max) { pn_a = cur.x; pn_b = cur.y; }
        }
        const float gx_a = (pe_a - pw_a) * k, gy_a = (pn_a - ps_a) * k;
        const float gx_b = (pe_b - pw_b) * k, gy_b = (pn_b - ps_b) * k;
        if (tl.out) {
            const v4f o = {u.x - gx_a, u.y - gy_a, u.z - gx_b, u.w - gy_b};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, o), rv, voff,
                                                   (y - g.grow0) * g.dim_x * 8, 0);
        }
    };

    for (int y = tl.r0; y < tl.r1; y += 4) {
#define SFL_GRAD_STEP(U)                                                             \
    if (y + U < tl.r1) {                                                             \
        ring[(U + 3) & 3] = load_p(y + U + 2);                                       \
        vel[(U + 1) & 1] = load_v(y + U + 1);                                        \
        row(y + U, ring[U & 3], ring[(U + 1) & 3], ring[(U + 2) & 3], vel[U & 1]);   \
    }
        SFL_GRAD_STEP(0) SFL_GRAD_STEP(1) SFL_GRAD_STEP(2) SFL_GRAD_STEP(3)
#undef SFL_GRAD_STEP
    }
}

inline TileGeom make_geom(int dim_x, int rows)
{
    TileGeom t;
    t.n_strips = (dim_x + kValidCols - 1) / kValidCols;
    // ~8 waves per SIMD on 256 CUs, but chunks of at least 32 rows (2 halo rows each)
    int chunks = (8192 + t.n_strips - 1) / t.n_strips;
    int rpc = (rows + chunks - 1) / chunks;
    if (rpc < 32) rpc = 32;
    if (rpc > rows) rpc = rows;
    t.rows_per_chunk = rpc;
    t.n_chunks = (rows + rpc - 1) / rpc;
    return t;
}

}  // namespace

bool stream_stencils_applicable(const Slab &g, const void *a, const void *b)
{
    return g.dim_x % 2 == 0 && g.dim_x >= 4 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 &&
           (size_t)g.lrows * g.dim_x * 8 <= 0xFFFFFFFFull;
}

hipError_t launch_divergence_stream(hipStream_t s, float *div, const float *v, Slab g, int g_begin,
                                    int g_end, float two_dx_inv)
{
    const TileGeom t = make_geom(g.dim_x, g_end - g_begin);
    const int blocks = (t.n_strips * t.n_chunks + kWaves - 1) / kWaves;
    divergence_stream_kernel<<<blocks, kThreads, 0, s>>>(div, v, g, t, g_begin, g_end, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_gradient_stream(hipStream_t s, float *v, const float *p, Slab g, int g_begin,
                                  int g_end, float two_dx_inv)
{
    const TileGeom t = make_geom(g.dim_x, g_end - g_begin);
    const int blocks = (t.n_strips * t.n_chunks + kWaves - 1) / kWaves;
    gradient_stream_kernel<<<blocks, kThreads, 0, s>>>(v, p, g, t, g_begin, g_end, two_dx_inv);
    return hipGetLastError();
}

}  // namespace sfl
