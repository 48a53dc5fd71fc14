// sor_stream_emu.cpp -- CPU emulation of one gfx950 wavefront running the fused SOR pipeline.
//
// TEST HARNESS ONLY.  It instantiates the PRODUCT's pipeline (csrc/sor_stream_core.h -- the
// exact header the GPU kernel is built from) with a backend whose value type is "64 lanes of
// float", executes tiles one after the other on the CPU and lets pytest compare the result
// with the oracle bit for bit.  Nothing here is reachable from the product library.
//
// What this proves without a GPU: slot rotation, row parity, colour order, the E/O dependency
// schedule, the -0.0f boundary algebra, tile validity margins and the tiling arithmetic.
// What it cannot prove: the DPP lane shifts, LDS addressing and load/store guards of the real
// backend (sor_fused.hip) -- those are covered by the `-m gpu` parity tests.
//
// Build: g++ -std=c++17 -O1 -ffp-contract=off -shared -fPIC   (tests/cpp/Makefile)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "sor_stream_core.h"

namespace {

struct V64 {
    float l[64];
};
struct M64 {
    bool l[64];
};

#define EMU_BINOP(OP)                                      \
    inline V64 operator OP(const V64 &a, const V64 &b)     \
    {                                                      \
        V64 r;                                             \
        for (int i = 0; i < 64; ++i) r.l[i] = a.l[i] OP b.l[i]; \
        return r;                                          \
    }
EMU_BINOP(+)
EMU_BINOP(-)
EMU_BINOP(*)
#undef EMU_BINOP

template <bool FOLD>
struct EmuBackendT {
    using V = V64;
    using M = M64;
    static constexpr int kPrefetch = 3;
    static constexpr bool kFoldQuarter = FOLD;   // SFL_OPT_SOR_FOLD (sor_stream_core.h relax)

    const float *p_in;
    const float *d;
    float *p_out;
    int dim_x, gdim_y, grow0, row_lo, row_hi;
    int x0;       // column of lane 0's cell a
    int out_lo, out_hi;
    int row_sign = 1;  // pipeline row index t -> domain row row_sign * t (-1: tile streamed top-down)
    bool vec2;    // emulate the 8-byte access variant (pair handled as a whole)
    bool poison_on;
    int tile_r0, tile_r1;  // output rows of the tile being streamed
    int *stray_stores;
    std::vector<float> ring;  // [slot][plane][lane]

    V splat(float x) const
    {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = x;
        return r;
    }
    V select(const M &m, const V &a, const V &b) const
    {
        V r;
        for (int i = 0; i < 64; ++i) r.l[i] = m.l[i] ? a.l[i] : b.l[i];
        return r;
    }
    M mask_and(const M &m, bool row) const
    {
        M r;
        for (int i = 0; i < 64; ++i) r.l[i] = m.l[i] && row;
        return r;
    }
    // DPP wave_shr:1 / wave_shl:1 with bound_ctrl: the missing lane reads 0
    V from_lower_lane(const V &x) const
    {
        V r;
        r.l[0] = 0.0f;
        for (int i = 1; i < 64; ++i) r.l[i] = x.l[i - 1];
        return r;
    }
    V from_upper_lane(const V &x) const
    {
        V r;
        r.l[63] = 0.0f;
        for (int i = 0; i < 63; ++i) r.l[i] = x.l[i + 1];
        return r;
    }
    sfl::sor::RowFacts row_facts(int t) const
    {
        const int r = row_sign * t;
        return {r >= 0 && r < gdim_y, r > 0 && r < gdim_y - 1};
    }
    template <class P>
    void poison(P &pp) const
    {
        if (!poison_on) return;
        // every pipeline register starts as NaN: a stale read would surface in the output
        float *f = reinterpret_cast<float *>(&pp);
        for (size_t k = 0; k < sizeof(P) / sizeof(float); ++k)
            f[k] = std::numeric_limits<float>::quiet_NaN();
    }
    // Mirrors the GPU backend: loads are unconditional with clamped row / column.  To be
    // stricter than the hardware, whatever a CLAMPED access would return is replaced by NaN:
    // the pipeline must mask it (domain boundary) or keep it out of the exact interior.
    void load_row(int t, V &pa, V &pb, V &da, V &db) const
    {
        const int r = row_sign * t;
        const bool row_ok = r >= row_lo && r < row_hi;
        const float nan = std::numeric_limits<float>::quiet_NaN();
        for (int i = 0; i < 64; ++i) {
            const int xa = x0 + 2 * i;
            const bool a_in = xa >= 0 && xa < dim_x, b_in = xa + 1 >= 0 && xa + 1 < dim_x;
            const size_t c = row_ok ? (size_t)(r - grow0) * dim_x + xa : 0;
            const bool a_ok = row_ok && (vec2 ? (a_in && b_in) : a_in);
            const bool b_ok = row_ok && (vec2 ? (a_in && b_in) : b_in);
            da.l[i] = a_ok ? d[c] : nan;
            db.l[i] = b_ok ? d[c + 1] : nan;
            if (p_in) {
                pa.l[i] = a_ok ? p_in[c] : nan;
                pb.l[i] = b_ok ? p_in[c + 1] : nan;
            }
        }
    }
    void store_row(int t, const V &a, const V &b) const
    {
        const int r = row_sign * t;
        if (r < tile_r0 || r >= tile_r1) {  // a store outside the tile's output rows is a bug
            ++*stray_stores;
            return;
        }
        for (int i = 0; i < 64; ++i) {
            const int xa = x0 + 2 * i;
            const bool a_in = xa >= 0 && xa < dim_x, b_in = xa + 1 >= 0 && xa + 1 < dim_x;
            const bool a_out = a_in && xa >= out_lo && xa < out_hi;
            const bool b_out = b_in && xa + 1 >= out_lo && xa + 1 < out_hi;
            const size_t c = (size_t)(r - grow0) * dim_x + xa;
            if (vec2 ? a_out : a_out) p_out[c] = a.l[i];
            if (vec2 ? a_out : b_out) p_out[c + 1] = b.l[i];
        }
    }
    void ring_store(int slot, int plane, const V &x)
    {
        std::memcpy(&ring[(size_t)(slot * 2 + plane) * 64], x.l, sizeof(x.l));
    }
    V ring_load(int slot, int plane) const
    {
        V r;
        std::memcpy(r.l, &ring[(size_t)(slot * 2 + plane) * 64], sizeof(r.l));
        return r;
    }
    V detach(const V &x) const { return x; }
    void pin() const {}
    static constexpr int kTurnRows = 6;
    void next_turn() {}
};

template <class EmuBackend>
sfl::sor::EdgeCell<EmuBackend> edge_cells(int x0, int which, int dim_x)
{
    sfl::sor::EdgeCell<EmuBackend> ec;
    const float k2 = (float)(-1.0 / 2.0), k3 = (float)(-1.0 / 3.0), k4 = -0.25f;
    for (int i = 0; i < 64; ++i) {
        const int x = x0 + 2 * i + which;
        const int nh = (x > 0 ? 1 : 0) + (x < dim_x - 1 ? 1 : 0);
        ec.in.l[i] = x >= 0 && x < dim_x;
        ec.k_full.l[i] = (nh == 2) ? k4 : (nh == 1) ? k3 : k2;
        ec.k_part.l[i] = (nh == 2) ? k3 : k2;
        ec.z_full.l[i] = (nh == 2) ? -0.0f : 0.0f;
    }
    return ec;
}

template <int NS, bool FOLD>
int run_tiles_as(float *p_out, const float *p_in, const float *d, int dim_x, int gdim_y, int grow0,
               int lrows, int g_begin, int g_end, float dx, float omega, int rows_per_chunk,
               bool vec2, bool poison, bool force_edge, bool balance, int flip, int *flipped_tiles)
{
    using namespace sfl::sor;
    using EmuBackend = EmuBackendT<FOLD>;
    const Tiling t = make_tiling(NS, 128, 2, dim_x, gdim_y, g_begin, g_end, rows_per_chunk,
                                 balance ? kEdgeRowCost16 : 0, flip);
    int stray = 0;
    {
        for (int tile = 0; tile < t.n_tiles; ++tile) {
            const TileRect rect = tile_rect(t, tile);
            const int strip = rect.strip;
            EmuBackend bk;
            bk.p_in = p_in;
            bk.d = d;
            bk.p_out = p_out;
            bk.dim_x = dim_x;
            bk.gdim_y = gdim_y;
            bk.grow0 = grow0;
            bk.row_lo = grow0 > 0 ? grow0 : 0;
            bk.row_hi = (grow0 + lrows < gdim_y) ? grow0 + lrows : gdim_y;
            bk.x0 = strip_x0(t, strip);
            bk.out_lo = bk.x0 + t.halo_cols;
            bk.out_hi = bk.x0 + t.tile_cols - t.halo_cols;
            bk.vec2 = vec2;
            bk.poison_on = poison;
            bk.ring.assign((size_t)ring_rows(NS) * 2 * 64, poison ? std::numeric_limits<float>::quiet_NaN() : 0.0f);
            const int r0 = rect.r0, r1 = rect.r1;
            bk.tile_r0 = r0;
            bk.tile_r1 = r1;
            bk.stray_stores = &stray;
            Consts<EmuBackend> c{bk.splat(dx), bk.splat(omega), bk.splat(1.0f - omega), bk.splat(-0.25f * omega)};
            const bool edge = force_edge || tile_touches_boundary(t, rect, gdim_y);
            const bool flipped = !edge && tile_may_flip(t, rect);
            const bool dx1 = dx == 1.0f;
            const auto eca = edge_cells<EmuBackend>(bk.x0, 0, dim_x), ecb = edge_cells<EmuBackend>(bk.x0, 1, dim_x);
            const bool zero_in = p_in == nullptr;
#define EMU_RUN(EDGE, DX1, ZERO) stream_tile<EmuBackend, NS, EDGE, DX1, ZERO>(bk, c, eca, ecb, r0, r1)
            if (edge) {
                if (dx1) { if (zero_in) EMU_RUN(true, true, true); else EMU_RUN(true, true, false); }
                else     { if (zero_in) EMU_RUN(true, false, true); else EMU_RUN(true, false, false); }
            } else if (flipped) {  // streamed top-down: pipeline index = -row (what the product does)
                bk.row_sign = -1;
#define EMU_FLIP(DX1, ZERO) stream_tile<EmuBackend, NS, false, DX1, ZERO, true>(bk, c, eca, ecb, 1 - r1, 1 - r0)
                if (dx1) { if (zero_in) EMU_FLIP(true, true); else EMU_FLIP(true, false); }
                else     { if (zero_in) EMU_FLIP(false, true); else EMU_FLIP(false, false); }
#undef EMU_FLIP
                ++*flipped_tiles;
            } else {
                if (dx1) { if (zero_in) EMU_RUN(false, true, true); else EMU_RUN(false, true, false); }
                else     { if (zero_in) EMU_RUN(false, false, true); else EMU_RUN(false, false, false); }
            }
#undef EMU_RUN
        }
    }
    return stray;
}

}  // namespace

// One non-template entry per fuse depth and arithmetic (exact / folded quarter); the depths and arithmetics are spread over
// several objects (-DEMU_NS_GROUP=0..3 -DEMU_FOLD_PART=0/1, see Makefile) because a single translation unit takes eight minutes.
#ifndef EMU_NS_GROUP
#define EMU_NS_GROUP (-1)  // everything in one translation unit
#endif
#ifndef EMU_FOLD_PART
#define EMU_FOLD_PART (-1)  // both arithmetics in this translation unit
#endif
#define EMU_ARGS float *p_out, const float *p_in, const float *d, int dim_x, int gdim_y, int grow0, \
                 int lrows, int g_begin, int g_end, float dx, float omega, int rows_per_chunk,      \
                 bool vec2, bool poison, bool force_edge, bool balance, int flip, int *flipped_tiles
#define EMU_DECLARE(N) int emu_run_ns##N##_f0(EMU_ARGS); int emu_run_ns##N##_f1(EMU_ARGS);
#define EMU_DEFINE_F(N, F)                                                                        \
    int emu_run_ns##N##_f##F(EMU_ARGS)                                                            \
    {                                                                                             \
        return run_tiles_as<N, F != 0>(p_out, p_in, d, dim_x, gdim_y, grow0, lrows, g_begin, g_end, dx, omega, \
                                       rows_per_chunk, vec2, poison, force_edge, balance, flip, flipped_tiles); \
    }
#if EMU_FOLD_PART == 0
#define EMU_DEFINE(N) EMU_DEFINE_F(N, 0)
#elif EMU_FOLD_PART == 1
#define EMU_DEFINE(N) EMU_DEFINE_F(N, 1)
#else
#define EMU_DEFINE(N) EMU_DEFINE_F(N, 0) EMU_DEFINE_F(N, 1)
#endif
EMU_DECLARE(2) EMU_DECLARE(4) EMU_DECLARE(6) EMU_DECLARE(8)
EMU_DECLARE(10) EMU_DECLARE(12) EMU_DECLARE(14) EMU_DECLARE(16)
#if EMU_NS_GROUP == 0 || EMU_NS_GROUP == -1
EMU_DEFINE(2) EMU_DEFINE(4) EMU_DEFINE(6) EMU_DEFINE(8)
#endif
#if EMU_NS_GROUP == 1 || EMU_NS_GROUP == -1
EMU_DEFINE(10) EMU_DEFINE(12)
#endif
#if EMU_NS_GROUP == 2 || EMU_NS_GROUP == -1
EMU_DEFINE(14)
#endif
#if EMU_NS_GROUP == 3 || EMU_NS_GROUP == -1
EMU_DEFINE(16)
#endif

#if (EMU_NS_GROUP == 0 || EMU_NS_GROUP == -1) && EMU_FOLD_PART != 1
// flags: bit0 = emulate the VEC2 access variant, bit1 = NaN-poison pipeline state,
//        bit2 = force the EDGE path for every tile, bit3 = uniform tiling (no short boundary tiles),
//        bit4 = every second chunk of an inner strip is streamed top-down (the product's default): the odd chunks,
//        bit5 = ... the even chunks instead (the product's odd launches of a solve on big slabs)
//        bit6 = SFL_OPT_SOR_FOLD: the interior relaxation's single product by -0.25f * omega (default: the reference's two products)
// returns the number of tiles streamed top-down (>= 0), or a negative error
extern "C" __attribute__((visibility("default"))) int
emu_sor_fused(float *p_out, const float *p_in, const float *d, int dim_x, int gdim_y, int grow0,
              int lrows, int g_begin, int g_end, int ns, float dx, float omega,
              int rows_per_chunk, int flags)
{
    const bool vec2 = flags & 1, poison = flags & 2, force_edge = flags & 4, balance = !(flags & 8);
    const int flip = (flags & 32) ? 2 : (flags & 16) ? 1 : 0;
    const bool fold = flags & 64;
    int flipped_tiles = 0;
    if (vec2 && (dim_x & 1)) return -1;
#define EMU_CASE(N)                                                                          \
    case N:                                                                                  \
        return (fold ? emu_run_ns##N##_f1 : emu_run_ns##N##_f0)(p_out, p_in, d, dim_x, gdim_y, grow0, lrows, g_begin, g_end, dx, \
                             omega, rows_per_chunk, vec2, poison, force_edge, balance, flip, &flipped_tiles) ? -3 : flipped_tiles;
    switch (ns) {
        EMU_CASE(2) EMU_CASE(4) EMU_CASE(6) EMU_CASE(8) EMU_CASE(10) EMU_CASE(12) EMU_CASE(14)
        EMU_CASE(16)
    }
#undef EMU_CASE
    return -2;
}

// Tiling property check: adds 1 to cover[row * dim_x + col] for every cell a tile of the launch
// would store (its exact columns x its output rows; balance = 0 or the boundary row cost in
// sixteenths) and returns the number of tiles; *n_edge
// receives how many of them take the EDGE path.  A correct tiling leaves exactly 1 on every cell
// of rows [g_begin, g_end) and 0 elsewhere.
extern "C" __attribute__((visibility("default"))) int
emu_tiling_cover(int ns, int tile_cols, int col_align, int dim_x, int gdim_y, int g_begin, int g_end,
                 int rows_per_chunk, int balance, int *cover, int *n_edge)
{
    using namespace sfl::sor;
    const Tiling t = make_tiling(ns, tile_cols, col_align, dim_x, gdim_y, g_begin, g_end, rows_per_chunk,
                                 balance);
    *n_edge = 0;
    for (int tile = 0; tile < t.n_tiles; ++tile) {
        const TileRect r = tile_rect(t, tile);
        if (r.strip < 0 || r.strip >= t.n_strips || r.r0 >= r.r1) return -1;
        if (tile_touches_boundary(t, r, gdim_y)) ++*n_edge;
        const int x0 = strip_x0(t, r.strip);
        const int lo = x0 + t.halo_cols, hi = x0 + t.tile_cols - t.halo_cols;
        for (int y = r.r0; y < r.r1; ++y)
            for (int x = (lo < 0 ? 0 : lo); x < (hi < dim_x ? hi : dim_x); ++x)
                if (y >= 0 && y < gdim_y) ++cover[(size_t)y * dim_x + x];
    }
    return t.n_tiles;
}

// Dispatch order (sor::tile_rect's rotation): for any rotation of the chunk numbers of the inner strips (rot_c) and of the
// boundary strips (rot_e) the map position -> (strip, output rows) must visit every tile of the tiling exactly once, and the
// first fc x n_inner positions must be the chunks [rot_c, rot_c + fc) of the inner strips.  Returns n_tiles, or -1.
extern "C" __attribute__((visibility("default"))) int
emu_tile_order(int ns, int tile_cols, int col_align, int dim_x, int gdim_y, int g_begin, int g_end, int rows_per_chunk,
               int balance, int c0, int c1, int e0, int e1)
{
    using namespace sfl::sor;
    const Tiling t = make_tiling(ns, tile_cols, col_align, dim_x, gdim_y, g_begin, g_end, rows_per_chunk, balance);
    auto clip = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    const int rot_c = t.n_chunks ? clip(c0, t.n_chunks - 1) : 0, rot_e = t.n_chunks_edge ? clip(e0, t.n_chunks_edge - 1) : 0;
    const int fc = clip(c1, t.n_chunks - rot_c);   // free chunks of the inner strips: [rot_c, rot_c + fc)
    (void)e1;
    std::vector<char> seen(t.n_tiles, 0);
    for (int q = 0; q < t.n_tiles; ++q) {
        const TileRect r = tile_rect(t, q, rot_c, rot_e);
        if (r.strip < 0 || r.strip >= t.n_strips || r.r0 >= r.r1 || r.r0 < t.g_begin || r.r1 > t.g_end) return -1;
        const int chunk = chunk_of_row(t, r.strip, r.r0);
        const int k = tile_index(t, r.strip, chunk);
        const TileRect plain = tile_rect(t, k);
        if (plain.strip != r.strip || plain.r0 != r.r0 || plain.r1 != r.r1) return -1;   // a tile of the tiling, as the tiling cuts it
        if (k < 0 || k >= t.n_tiles || seen[k]) return -1;
        seen[k] = 1;
        if (q < fc * t.n_inner && !(strip_is_inner(t, r.strip) && chunk >= rot_c && chunk < rot_c + fc)) return -1;
    }
    return t.n_tiles;
}
#endif  // EMU_NS_GROUP 0
