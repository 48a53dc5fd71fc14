// sor_stream_core.h -- the fused red-black SOR pipeline, written once against a small
// "wave backend" so that the SAME code runs (a) on gfx950, one 64-lane wavefront per tile
// with V = float in VGPRs (sor_fused.hip), and (b) lane-by-lane on a CPU in the test
// harness with V = 64 floats (tests/cpp/sor_stream_emu.cpp), where it is checked against
// the oracle without a GPU.
//
// What it computes: NS consecutive colour passes (NS even: NS/2 full iterations of
// poisson.cpp:121-124, even (i+j) first) of the in-place red-black SOR of
// poisson.cpp:14-112 on one tile, in ONE sweep over the tile's rows, reading p and d once
// and writing p once.  Bit-identical to running the passes one after the other.
//
// Tile = 64 lanes x 2 columns (128 columns) x a run of rows streamed bottom-up (or top-down: FLIP).  A lane's
// value type V holds its cell of ONE colour in a row: V = float, cell `a` at the even column x_a, cell `b` at
// x_a + 1.  (Rounds 1 and 2 also carried 4-cells-per-lane flavours on packed and on scalar fp32; a plain fp32
// instruction of a wave64 occupies the SIMD for 2 cycles, a packed one for 4, and neither flavour was ever
// faster: DESIGN.md 4.1.)  from_lower_lane / from_upper_lane shift a colour vector by one position of that
// colour towards higher / lower columns.  Colours: a cell is "E" when (column + row) is even (updated by the
// first pass of an iteration, poisson.cpp:22) and "O" otherwise.  In a row of even parity the E cell of a
// lane is `a`, in an odd row it is `b`.
//
// Dependencies (5-point stencil, neighbours always have the other colour):
//   E_m[r] = relax(E_{m-1}[r]; W/E from O_{m-1}[r]; S = O_{m-1}[r-1]; N = O_{m-1}[r+1]; d_E[r])
//   O_m[r] = relax(O_{m-1}[r]; W/E from E_m[r];     S = E_m[r-1];     N = E_m[r+1];     d_O[r])
// for version m = 1 .. NS/2 (version 0 = the loaded data).  Software pipeline: in the
// iteration that receives input row y it computes E_m[y-(2m-1)] and O_m[y-2m] for every m, so
// the finished row y-NS leaves the pipeline NS iterations after it entered.  The last reader
// of a version is the relaxation that produces the next one, so a row keeps ONE register per
// colour for its whole stay and is updated in place (as the reference updates memory in place):
// 2 * RING registers, addressed by (row mod RING).  The right-hand side d of a row is needed
// NS iterations long; it waits in a per-lane ring of RING >= NS + 1 rows in LDS (nobody else
// reads it), same index.  The loop body is unrolled RING times, RING a multiple of 6 (row parity
// 2, prefetch depths 2 / 3 / 6), so that every register, row parity and ring slot is a
// compile-time constant: no address arithmetic, no register moves.
//
// Validity: a tile is loaded with NS extra columns / rows on every side; pass s of the NS
// spoils one more ring of cells, the tile's interior [NS from each loaded edge] is exact.
// Cells outside the domain hold -0.0f, the additive identity (x + -0.0f == x bitwise for
// every x), so perimeter cells can add "absent" neighbours unconditionally:
//   interior  sum = ((W + E) + S) + N                    poisson.cpp:107  (pois_sor_fast)
//   perimeter sum = ((((+0 + W') + E') + S') + N'        poisson.cpp:69-86 (pois_gs_safe)
// and both are evaluated as (((z + W) + E) + S) + N with z = -0.0f / +0.0f.
#pragma once
#include <utility>

#ifndef SFL_EDGE_PROLOGUE_MAX_NS
#define SFL_EDGE_PROLOGUE_MAX_NS 12
#endif

#if defined(__HIPCC__)
#define SFL_HD __host__ __device__ __forceinline__
#else
#define SFL_HD inline
#endif

namespace sfl {
namespace sor {


constexpr int wrapn(int x, int n) { return ((x % n) + n) % n; }
constexpr bool is_even(int x) { return ((x % 2) + 2) % 2 == 0; }
// rows of d alive at once (NS + 1), rounded up to a multiple of 6; also the unroll factor
constexpr int ring_rows(int ns) { return ((ns + 1 + 5) / 6) * 6; }

// Per-lane facts used only by tiles that touch the domain boundary.
template <class B>
struct EdgeCell {
    typename B::M in;       // column inside [0, dim_x)
    typename B::V k_full;   // -1/n for a row with both vertical neighbours (n = 2 + #horizontal)
    typename B::V k_part;   // -1/n for the bottom / top row            (n = 1 + #horizontal)
    typename B::V z_full;   // +0 when the column is a wall column, else -0 (rows with both
                            // vertical neighbours); bottom / top rows always use +0
};

struct RowFacts {  // wave-uniform
    bool in_dom;   // 0 <= r < gdim_y
    bool full;     // 0 < r < gdim_y - 1
};

template <class B>
struct Consts {
    typename B::V dx, omega, one_minus_omega, neg_quarter_omega;
};

// B::kPrefetch = rows in flight ahead of the pipeline (must divide 6, hence every RING)
// B::kFoldQuarter = the interior relaxation multiplies once by -0.25f * omega instead of by -0.25f and by omega (relax)
// E[k] / O[k] hold the two colours of the row whose index is k modulo RING -- the NEWEST version
// of that row computed so far: a relaxation overwrites its own input (the previous version's last
// reader is the relaxation that replaces it, see iterate()), exactly as SOR does in memory.
template <class B, int NS>
struct Pipe {
    using V = typename B::V;
    V E[ring_rows(NS)];
    V O[ring_rows(NS)];
    V pa[B::kPrefetch], pb[B::kPrefetch];  // prefetched p rows (cell a / cell b)
    V da[B::kPrefetch], db[B::kPrefetch];  // prefetched d rows
};

// One relaxation (poisson.cpp:63-112).
template <class B, bool EDGE>
SFL_HD typename B::V relax(const B &bk, const Consts<B> &c, typename B::V own, typename B::V w,
                           typename B::V e, typename B::V s, typename B::V n, typename B::V d,
                           const EdgeCell<B> &ec, RowFacts rf)
{
    using V = typename B::V;
    const V rhs = d;   // fl(dx * d), formed when the row entered the ring (iterate)
    if (!EDGE) {
        const V sum = ((w + e) + s) + n;
        // poisson.cpp:107-111: p_gs = -0.25f * (dx * d - sum); p = (1 - omega) * p + omega * p_gs -- every product rounded on its
        // own, 8 vector instructions, the reference's bits on EVERY input.  This is what the library runs unless asked otherwise.
        if (!B::kFoldQuarter) return c.one_minus_omega * own + c.omega * (bk.splat(-0.25f) * (rhs - sum));
        // SFL_OPT_SOR_FOLD = 1 (opt-in): fl(omega * fl(-0.25f * t)), t = dx * d - sum, as ONE rounded product fl((-0.25f * omega) * t) --
        // 7 instructions, the relaxation's dependency chain one shorter (+2.4 .. 3 % on 8192^2).  Scaling by -0.25 is exact unless
        // the product underflows inexactly, so the two are the same bits whenever t is a multiple of 2^-147, i.e. unless an operand
        // (dx * d or a neighbour's p) is a nonzero number below 2^-124 = 4.7e-38.  That is no edge of float: the front of a solution
        // that decays into a quiescent region (zero right-hand side -- the sketch's own start, ino:199,264-276) passes through that
        // range from ~63 iterations on, and there the reference rounds -0.25f * t to a denormal first and the two products can differ
        // by one unit of 2^-149, which later passes carry along (tests/test_gpu_parity.py test_quiescent_*; an omega whose
        // own quarter would underflow is solved unfolded whatever the option says, sor_executor.cpp).
        return c.one_minus_omega * own + c.neg_quarter_omega * (rhs - sum);
    }
    const V z = rf.full ? ec.z_full : bk.splat(0.0f);
    const V k = rf.full ? ec.k_full : ec.k_part;
    const V sum = (((z + w) + e) + s) + n;
    const V gs = k * (rhs - sum);
    const V out = c.one_minus_omega * own + c.omega * gs;
    return bk.select(bk.mask_and(ec.in, rf.in_dom), out, bk.splat(-0.0f));
}

// Trips of a tile: the first ones are the pipeline's prologue.  Pass s (1 .. NS) only has to be
// exact on rows [out_begin - (NS - s), ...), and it works on row y - s while row y enters, so in
// the tile's J-th iteration (J = 0 for the first input row, out_begin - NS) pass s is needed
// only if J >= 2 s: the prologue trips leave the others out (a quarter of the work of a 26-row
// tile at NS = 12).  Leaving a pass out keeps an older version in the row's register; by the
// same inequality nothing that is needed ever reads it.  The tile may start one row earlier
// (even alignment), which only makes J an over-estimate.
constexpr int kRhsAhead = 1;    // stages between the LDS read of a right-hand-side pair and its use
constexpr int kSteadyTrip = 2;  // TRIP = 0, 1: prologue trips; kSteadyTrip: every pass runs
constexpr bool pass_runs(int trip, int ring, int u, int s)
{
    return trip >= kSteadyTrip || trip * ring + u >= 2 * s;
}
// The boundary path gets prologue trips only at the shallower fuse depths: at NS >= 14 they push
// the kernel past 168 VGPRs (3 -> 2 waves per SIMD).
constexpr bool edge_prologue(int ns) { return ns <= SFL_EDGE_PROLOGUE_MAX_NS; }
constexpr int prologue_trips(int ns) { return (2 * ns + ring_rows(ns) - 1) / ring_rows(ns); }
// After the prologue trips the leaving row y - NS is at or above out_begin (the tile starts at
// out_begin - NS or one row lower), and a full trip never runs past out_end + NS: steady full
// trips can store without looking.  (A conditional store costs a branch per row and, as
// compiled, eight SGPR spill moves for its buffer descriptor.)
constexpr bool steady_rows_are_output(int ns) { return prologue_trips(ns) * ring_rows(ns) >= 2 * ns + 1; }

// The work of ONE pipeline iteration: input row y (already waiting in prefetch slot U mod kPrefetch)
// enters, row y - NS leaves.  U = (y - y_start) mod RING is a compile-time constant.
// FLIP: the tile is streamed TOP-DOWN (the backend maps pipeline row index t to domain row -t, same
// parity): the pipeline's "previous row" is then the N neighbour and its "next row" the S neighbour,
// and the reference's sum ((W + E) + S) + N keeps its order by swapping the two operands here.
template <class B, int NS, bool EDGE, bool DX1, bool ZERO_IN, int TRIP, bool GUARD_STORE, bool FLIP, int U>
SFL_HD void iterate(B &bk, Pipe<B, NS> &pp, const Consts<B> &c, const EdgeCell<B> &eca,
                    const EdgeCell<B> &ecb, int y, int out_begin, int out_end)
{
    using V = typename B::V;
    constexpr int RING = ring_rows(NS);
    constexpr int kPrefetch = B::kPrefetch;
    constexpr int Q = U % kPrefetch;

    // the waves that share a SIMD take turns at the top issue priority (Lane2::next_turn; a no-op in the emulator)
    if (U % B::kTurnRows == 0) bk.next_turn();
#ifdef SFL_PROBE_COOP
    // TIMING MOCK (diagnostic builds only, wrong results): what sharing column halos between the waves of a block
    // would add to every row -- see Lane2::coop_mock
    bk.template coop_mock<NS, U>(pp);
#endif

    // ---- row y enters: hand it to version 0, park its d in the ring, refill the slot ----
    {
        // detach(): an explicit register copy, so that the prefetch registers are free to
        // receive the next load at once (otherwise the compiler keeps the old value alive in
        // them and has to drain all loads in flight at the loop back-edge to rotate registers)
        V a = bk.detach(pp.pa[Q]), b = bk.detach(pp.pb[Q]);
        // poisson.cpp:94 / :109 multiply dx * d anew in every relaxation -- the same two operands, the same rounded product every
        // time: it is formed ONCE, here, and the ring holds fl(dx * d) (dx == 1: d itself, kernels without the multiplication)
        const V fa = DX1 ? pp.da[Q] : c.dx * pp.da[Q], fb = DX1 ? pp.db[Q] : c.dx * pp.db[Q];
        bk.ring_store(U, 0, is_even(U) ? fa : fb);  // plane 0: dx * d of the E cell
        bk.ring_store(U, 1, is_even(U) ? fb : fa);  // plane 1: dx * d of the O cell
        bk.load_row(y + kPrefetch, pp.pa[Q], pp.pb[Q], pp.da[Q], pp.db[Q]);
        if (ZERO_IN) a = b = bk.splat(0.0f);  // poisson.cpp:117-119, fused
        if (EDGE) {  // cells outside the domain hold the additive identity
            const RowFacts rf = bk.row_facts(y);
            a = bk.select(bk.mask_and(eca.in, rf.in_dom), a, bk.splat(-0.0f));
            b = bk.select(bk.mask_and(ecb.in, rf.in_dom), b, bk.splat(-0.0f));
        }
        constexpr bool ev = is_even(U);
        pp.E[U] = ev ? a : b;  // replaces row y - RING, which left the pipeline RING - NS rows ago
        pp.O[U] = ev ? b : a;
    }

    // Versions at this point, for stage m (E part: row r = y - 2m + 1, O part: row r = y - 2m):
    //   O[r + 1] was brought to version m - 1 by stage m - 1 of THIS iteration, O[r] by the
    //   previous iteration, O[r - 1] two iterations ago; none has reached version m yet (O_m of
    //   row y - 2m is computed below, after E_m).  Likewise E[r + 1] = E_m (just computed),
    //   E[r] = E_m (previous iteration), E[r - 1] = E_m (stage m + 1 comes later).  So every
    //   relaxation reads exactly the versions the reference's in-place sweep reads
    //   (poisson.cpp:14-61), and may overwrite its own input register.
    // The right-hand sides of a stage are read from the LDS ring ONE STAGE AHEAD of their use: the
    // relaxations of an iteration form one dependency chain (every pass needs the previous pass's result of
    // the same iteration as its S / N operand), all waves of a launch are resident at once, so a launch lasts
    // as long as that chain -- and an LDS read issued right in front of its use (what the compiler does to
    // save a register pair) puts the LDS latency on the chain once per stage.  d_e / d_o[m + 1] are
    // requested while stage m computes; bk.pin() keeps the compiler from sinking the reads back down.
    // Measured (profiles/r02_rhs_read_ahead.txt): 8192^2 x 80 1.94 -> 1.89 ms, 8192 x 2048 0.597 -> 0.556 ms.
    constexpr int kAhead = kRhsAhead;
    V d_e[NS / 2 + 1], d_o[NS / 2 + 1];            // (only kAhead + 1 of each alive at a time)
#pragma unroll
    for (int m = 1; m <= kAhead && m <= NS / 2; ++m) {
        d_e[m] = bk.ring_load(wrapn(U - (2 * m - 1), RING), 0);
        d_o[m] = bk.ring_load(wrapn(U - 2 * m, RING), 1);
    }
#pragma unroll
    for (int m = 1; m <= NS / 2; ++m) {
        if (m + kAhead <= NS / 2) {
            d_e[m + kAhead] = bk.ring_load(wrapn(U - (2 * (m + kAhead) - 1), RING), 0);
            d_o[m + kAhead] = bk.ring_load(wrapn(U - 2 * (m + kAhead), RING), 1);
            bk.pin();
        }
        // ---- E_m of row y - (2m - 1) ----
        if (pass_runs(TRIP, RING, U, 2 * m - 1)) {
            const int lag = 2 * m - 1;
            const int r = y - lag;
            const int rel = U - lag;                 // compile time after unrolling
            const int i0 = wrapn(rel, RING), im = wrapn(rel - 1, RING), ip = wrapn(rel + 1, RING);
            const bool ev = is_even(rel);            // E cell is `a` in even rows
            const V own = pp.E[i0];
            const V oc = pp.O[i0];
            const V w = ev ? bk.from_lower_lane(oc) : oc;
            const V e = ev ? oc : bk.from_upper_lane(oc);
            const V d = d_e[m];
            const RowFacts rf = bk.row_facts(r);
            pp.E[i0] = relax<B, EDGE>(bk, c, own, w, e, pp.O[FLIP ? ip : im], pp.O[FLIP ? im : ip], d,
                                           ev ? eca : ecb, rf);
        }
        // ---- O_m of row y - 2m ----
        if (pass_runs(TRIP, RING, U, 2 * m)) {
            const int lag = 2 * m;
            const int r = y - lag;
            const int rel = U - lag;
            const int i0 = wrapn(rel, RING), im = wrapn(rel - 1, RING), ip = wrapn(rel + 1, RING);
            const bool ev = is_even(rel);            // O cell is `b` in even rows
            const V own = pp.O[i0];
            const V oc = pp.E[i0];
            const V w = ev ? oc : bk.from_lower_lane(oc);
            const V e = ev ? bk.from_upper_lane(oc) : oc;
            const V d = d_o[m];
            const RowFacts rf = bk.row_facts(r);
            const V res = relax<B, EDGE>(bk, c, own, w, e, pp.E[FLIP ? ip : im], pp.E[FLIP ? im : ip], d,
                                              ev ? ecb : eca, rf);
            if (m < NS / 2) {
                pp.O[i0] = res;
            } else if (!GUARD_STORE || (r >= out_begin && r < out_end)) {  // finished row leaves
                if (ev)
                    bk.store_row(r, oc, res);
                else
                    bk.store_row(r, res, oc);
            }
        }
    }
}

// One trip = RING consecutive iterations, fully unrolled.  FULL trips are straight-line code (no
// branch between iterations: with a branch the compiler's wait-count pass loses track of the
// loads in flight and drains them all -- s_waitcnt vmcnt(0) -- once per trip); only the last,
// partial trip of a tile checks after every iteration whether the remaining rows still need to
// enter.
template <class B, int NS, bool EDGE, bool DX1, bool ZERO_IN, int TRIP, bool PARTIAL, bool FLIP, int... Us>
SFL_HD void run_unrolled(B &bk, Pipe<B, NS> &pp, const Consts<B> &c, const EdgeCell<B> &eca,
                         const EdgeCell<B> &ecb, int y, int out_begin, int out_end,
                         std::integer_sequence<int, Us...>)
{
    const int y_stop = out_end + NS;
    // steady full trips run only after the prologue trips on the paths that have them
    constexpr bool guard = PARTIAL || TRIP < kSteadyTrip || !steady_rows_are_output(NS) ||
                           (EDGE && !edge_prologue(NS));
    (void)(((!PARTIAL || y + Us < y_stop) &&
            (iterate<B, NS, EDGE, DX1, ZERO_IN, TRIP, guard, FLIP, Us>(bk, pp, c, eca, ecb, y + Us, out_begin,
                                                                       out_end),
             true)) && ...);
}

// Stream one tile: output rows [out_begin, out_end) IN PIPELINE ROW INDICES (= domain rows, or their
// negatives when FLIP), all NS passes.
template <class B, int NS, bool EDGE, bool DX1, bool ZERO_IN, bool FLIP = false>
SFL_HD void stream_tile(B &bk, const Consts<B> &c, const EdgeCell<B> &eca,
                        const EdgeCell<B> &ecb, int out_begin, int out_end)
{
    static_assert(NS >= 2 && NS % 2 == 0, "fuse an even number of colour passes");
    constexpr int RING = ring_rows(NS);
    constexpr int kPrefetch = B::kPrefetch;
    static_assert(RING % 6 == 0 && 6 % kPrefetch == 0 && RING >= NS + 1, "ring geometry");
    Pipe<B, NS> pp;
    bk.poison(pp);  // no-op on the GPU; NaN-fills in the emulator to prove nothing stale leaks

    // first input row: NS below the first output row, rounded down to an even row so that
    // row parity == iteration parity
    int y = out_begin - NS;
    y -= (y & 1);
    const int y_stop = out_end + NS;  // first row that need not enter

#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) bk.load_row(y + u, pp.pa[u], pp.pb[u], pp.da[u], pp.db[u]);

    constexpr auto us = std::make_integer_sequence<int, RING>{};
    if ((!EDGE || edge_prologue(NS)) && y + RING <= y_stop) {  // prologue trips: passes join one by one
        run_unrolled<B, NS, EDGE, DX1, ZERO_IN, 0, false, FLIP>(bk, pp, c, eca, ecb, y, out_begin, out_end, us);
        y += RING;
        if (prologue_trips(NS) >= 2 && y + RING <= y_stop) {
                run_unrolled<B, NS, EDGE, DX1, ZERO_IN, 1, false, FLIP>(bk, pp, c, eca, ecb, y, out_begin, out_end, us);
            y += RING;
        }
    }
    static_assert(prologue_trips(NS) <= kSteadyTrip, "prologue trips");
    for (; y + RING <= y_stop; y += RING)
        run_unrolled<B, NS, EDGE, DX1, ZERO_IN, kSteadyTrip, false, FLIP>(bk, pp, c, eca, ecb, y, out_begin,
                                                                    out_end, us);
    if (y < y_stop)
        run_unrolled<B, NS, EDGE, DX1, ZERO_IN, kSteadyTrip, true, FLIP>(bk, pp, c, eca, ecb, y, out_begin,
                                                                   out_end, us);
}

// ---- tiling arithmetic shared by the launcher, the kernel and the emulator -----------------
// A wave tile is `tile_cols` columns wide (64 lanes x 2 or 4 cells); its outer `halo_cols`
// columns on each side (NS rounded up to the lane's access granularity) are spoiled by the NS
// passes, the rest is exact.  Strips are laid out so that strip 0's exact interior starts at
// column 0.
//
// Tiles that touch the domain boundary run the EDGE path, which costs about 1.6x the
// instructions of the interior path per row.  All tiles of a launch are resident at once, so the
// launch lasts as long as its slowest wave: boundary tiles are therefore given fewer rows
// (`rows_edge`, chosen so that (rows_edge + 2 NS) * 1.6 ~ rows_per_chunk + NS: the boundary path has
// no prologue trips, they would cost it a wave of occupancy).  Boundary
// tiles are: every tile of a boundary strip (strip 0 and the strips whose columns reach
// dim_x), and the first / last chunk of the other ("inner") strips when the row range reaches
// the bottom / top of the domain.  Measured on 8192^2, NS = 16: 261 -> 2xx us per launch.
struct Tiling {
    int ns;               // passes fused
    int dim_x;
    int g_begin, g_end;   // output rows
    int rows_per_chunk;   // rows of an interior tile
    int rows_edge;        // rows of a tile in a boundary strip
    int rows_first;       // rows of the first chunk of an inner strip (0: no short first chunk)
    int rows_last;        // rows of the last chunk of an inner strip (0: no short last chunk)
    int n_strips;         // all strips
    int n_inner;          // inner strips are 1 .. n_inner
    int n_chunks;         // chunks of an inner strip
    int n_chunks_edge;    // chunks of a boundary strip
    int n_tiles;
    int tile_cols, halo_cols;
    int rotate;           // GPU backend, set by the launcher: 1 = the waves of a SIMD take turns at the top issue priority,
                          // (2 = this wave at the top priority throughout: set per tile by the kernel for sender tiles)
    int flip;             // every second chunk of an inner strip is streamed top-down (see tile_rect):
                          // 1 = the odd chunks, 2 = the even ones, 0 = none
};

struct TileRect {
    int strip;
    int r0, r1;  // output rows [r0, r1)
    int flip;    // streamed top-down (interior tiles only)
};

SFL_HD int strip_step(const Tiling &t) { return t.tile_cols - 2 * t.halo_cols; }

// column of lane 0's first cell for a strip (may be negative: columns left of the domain)
SFL_HD int strip_x0(const Tiling &t, int strip) { return strip * strip_step(t) - t.halo_cols; }

// rows given to a boundary tile when interior tiles get `rows_per_chunk`; `sixteenths` / 16 is
// the cost of an interior row relative to a boundary row.  A tile costs about rows + NS
// row-steps with prologue trips and rows + 2 NS without.  Boundary tiles are never shorter than
// kMinEdgeRows; when even such a tile would outlast the interior ones by more than a fifth,
// shortening cannot equalise anything (small grids: the warm-up rows dominate and the extra tiles
// only cost parallelism, measured 1024^2: -30 %) and 0 is returned: keep the uniform tiling.
#ifndef SFL_EDGE_ROW_COST16
#define SFL_EDGE_ROW_COST16 10   // (a build-time knob for the sweep only: tools/recipes/build_variant.sh)
#endif
constexpr int kEdgeRowCost16 = SFL_EDGE_ROW_COST16;
constexpr int kEdgeRowCostQueued16 = 8;   // launches with more tiles than wave slots (sor_fused.hip launch_variant)
constexpr int kMinEdgeRows = 8;
SFL_HD int balanced_edge_rows(int rows_per_chunk, int ns, int sixteenths)
{
    const int warm = edge_prologue(ns) ? ns : 2 * ns;
    const int r = (rows_per_chunk + ns) * sixteenths / 16 - warm;
    if (r >= kMinEdgeRows) return r;
    // cost of a minimum-height boundary tile against an interior tile, in interior row-steps
    return (kMinEdgeRows + warm) * 16 * 5 <= (rows_per_chunk + ns) * sixteenths * 6 ? kMinEdgeRows : 0;
}

// `balance16` = 0: every tile gets rows_per_chunk rows; otherwise pass kEdgeRowCost16.
SFL_HD Tiling make_tiling(int ns, int tile_cols, int col_align, int dim_x, int gdim_y, int g_begin,
                          int g_end, int rows_per_chunk, int balance16, int flip = 0)
{
    Tiling t;
    t.rotate = 0;
    t.flip = flip;
    const int rows = g_end - g_begin;
    t.ns = ns;
    t.dim_x = dim_x;
    t.g_begin = g_begin;
    t.g_end = g_end;
    t.rows_per_chunk = rows_per_chunk;
    t.tile_cols = tile_cols;
    t.halo_cols = (ns + col_align - 1) / col_align * col_align;
    t.n_strips = (dim_x + strip_step(t) - 1) / strip_step(t);

    int n_right = 0;  // strips whose columns reach the right wall
    while (n_right < t.n_strips && strip_x0(t, t.n_strips - 1 - n_right) + tile_cols >= dim_x) ++n_right;
    t.n_inner = t.n_strips - 1 - n_right;
    if (t.n_inner < 0) t.n_inner = 0;

    t.rows_edge = rows_per_chunk;
    t.rows_first = t.rows_last = 0;
    if (balance16 > 0) {
        const int re = balanced_edge_rows(rows_per_chunk, ns, balance16);
        if (re > 0 && re < rows_per_chunk) {
            t.rows_edge = re;
            // short first / last chunk of the inner strips, long enough that the next chunk is
            // clear of the boundary (see tile_touches_boundary)
            const bool bottom = g_begin - ns - 1 <= 0;
            const bool top = g_end + ns + ring_rows(ns) >= gdim_y;
            const int first = bottom ? (re > ns + 2 ? re : ns + 2) : 0;
            const int last = top ? (re > ns + ring_rows(ns) + 1 ? re : ns + ring_rows(ns) + 1) : 0;
            if (first + last + rows_per_chunk <= rows) {
                t.rows_first = first;
                t.rows_last = last;
            }
        }
    }
    const int mid = rows - t.rows_first - t.rows_last;
    t.n_chunks = (t.rows_first > 0) + (mid + rows_per_chunk - 1) / rows_per_chunk + (t.rows_last > 0);
    t.n_chunks_edge = (rows + t.rows_edge - 1) / t.rows_edge;
    t.n_tiles = t.n_inner * t.n_chunks + (t.n_strips - t.n_inner) * t.n_chunks_edge;
    return t;
}

// Tile index -> strip and output rows.  Inner strips come first, chunk-major (the waves of a
// block are neighbouring strips of one chunk), then the boundary strips.
// Alternating stream direction (t.flip): the even chunks of an inner strip are streamed bottom-up,
// the odd ones top-down.  Vertically adjacent tiles re-read 2 NS rows of each other; all tiles of a
// launch start together and take the same time, so with one direction a tile reads the shared rows
// at the END of its life and its upper neighbour at the BEGINNING of its own -- a tile's life apart,
// long evicted from L2.  With alternating directions both sharers reach a shared band at the same
// moment (both at the start, or both at the end): the second read hits the XCD's L2.
//
// DISPATCH ORDER (rot_c, rot_e): a launch whose cut-adjacent tiles wait INSIDE the kernel for a halo message (kernels.h HaloWait)
// must not hand those tiles out first.  The dispatcher deals workgroups to the XCDs in strict rotation and an XCD's share of the
// tiles in index order; with the bottom rows first the first XCD fills up with waiting tiles, none of which ever retires -- and
// the workgroups of the kernel that DELIVERS the message (RCCL's, dealt to the XCDs in the same rotation) find no room there: the
// launch waits for the message and the message for the launch, until the wait gives up (seen with sfl_comm_emulate_rccl on
// 16384 x 2048 slabs at a 160-row halo, profiles/r05_emulate_rccl.txt).  Chunks are numbered bottom-up and a tile waits when its
// rows come within reach of a cut, so the tiles that do NOT wait are one range of chunks [c0, c0 + fc) per kind of strip: the
// index `tile` is taken as a POSITION whose chunk number is rotated by rot = c0 -- the free chunks first, then the waiting ones
// above them, then (wrapped round) the waiting ones below.  A rotation, not a general permutation, on purpose: one add and one
// compare; a position -> tile map with quotients cost the NS = 16 kernel 1400 extra v_readlane / v_writelane in its streaming
// loop (SGPR pressure at the top of the kernel decides what is spilled for the whole loop) and 5 % of its time.
SFL_HD TileRect tile_rect(const Tiling &t, int tile, int rot_c = 0, int rot_e = 0)
{
    TileRect r;
    r.flip = 0;
    const int inner_tiles = t.n_inner * t.n_chunks;
    if (tile < inner_tiles) {
        int chunk = tile / t.n_inner;
        const int strip_in_chunk = tile - chunk * t.n_inner;
        chunk += rot_c;
        if (chunk >= t.n_chunks) chunk -= t.n_chunks;
        r.flip = t.flip && ((chunk & 1) == (t.flip & 1));  // flip = 1: the odd chunks, 2: the even ones
        r.strip = 1 + strip_in_chunk;
        const int has_first = t.rows_first > 0;
        if (has_first && chunk == 0) {
            r.r0 = t.g_begin;
            r.r1 = t.g_begin + t.rows_first;
        } else if (t.rows_last > 0 && chunk == t.n_chunks - 1) {
            r.r0 = t.g_end - t.rows_last;
            r.r1 = t.g_end;
        } else {
            const int mid_end = t.g_end - t.rows_last;
            r.r0 = t.g_begin + t.rows_first + (chunk - has_first) * t.rows_per_chunk;
            r.r1 = r.r0 + t.rows_per_chunk < mid_end ? r.r0 + t.rows_per_chunk : mid_end;
        }
    } else {
        const int u = tile - inner_tiles;
        const int e = u / t.n_chunks_edge;
        int chunk = u - e * t.n_chunks_edge + rot_e;
        if (chunk >= t.n_chunks_edge) chunk -= t.n_chunks_edge;
        r.strip = e == 0 ? 0 : t.n_inner + e;
        r.r0 = t.g_begin + chunk * t.rows_edge;
        r.r1 = r.r0 + t.rows_edge < t.g_end ? r.r0 + t.rows_edge : t.g_end;
    }
    return r;
}

// ---- strip_is_inner / chunk_of_row / tile_index invert tile_rect (the launcher's dispatch order: sor_fused.hip launch_variant) ----
SFL_HD bool strip_is_inner(const Tiling &t, int strip) { return strip >= 1 && strip <= t.n_inner; }

// chunk of `strip` whose output rows hold row r (g_begin <= r < g_end)
SFL_HD int chunk_of_row(const Tiling &t, int strip, int r)
{
    if (!strip_is_inner(t, strip)) return (r - t.g_begin) / t.rows_edge;
    const int has_first = t.rows_first > 0;
    if (has_first && r < t.g_begin + t.rows_first) return 0;
    if (t.rows_last > 0 && r >= t.g_end - t.rows_last) return t.n_chunks - 1;
    return has_first + (r - t.g_begin - t.rows_first) / t.rows_per_chunk;
}

SFL_HD int tile_index(const Tiling &t, int strip, int chunk)
{
    if (strip_is_inner(t, strip)) return chunk * t.n_inner + (strip - 1);
    return t.n_inner * t.n_chunks + (strip == 0 ? 0 : strip - t.n_inner) * t.n_chunks_edge + chunk;
}

// may the tile be streamed top-down?  (the RING rows of slack the pipeline needs beyond its last
// output row then lie BELOW the tile)
SFL_HD bool tile_may_flip(const Tiling &t, const TileRect &r)
{
    return r.flip && r.r0 - t.ns - ring_rows(t.ns) > 0;
}

// does the tile touch the domain boundary (=> EDGE path)?
SFL_HD bool tile_touches_boundary(const Tiling &t, const TileRect &r, int gdim_y)
{
    const int x0 = strip_x0(t, r.strip);
    // rows entering the pipeline: [r0 - ns - 1, r1 + ns + ring); columns [x0, x0 + tile_cols)
    return x0 <= 0 || x0 + t.tile_cols >= t.dim_x || r.r0 - t.ns - 1 <= 0 ||
           r.r1 + t.ns + ring_rows(t.ns) >= gdim_y;
}

}  // namespace sor
}  // namespace sfl
