// sor_stream_pairs.h -- the interior path of the fused red-black SOR pipeline with TWO pipeline
// stages per packed instruction (experimental; sor_stream_core.h describes the pipeline itself).
//
// In sor_stream_core.h stage m (colour passes 2m-1 and 2m) works on rows y-(2m-1) / y-2m while row
// y enters.  Stage m + Q (Q = NS/4) does the same work on rows that entered earlier.  Here the two
// are carried out together: a register PAIR holds {row r, row r - KP} of one colour, KP = NS/2 + 2,
// and every relaxation is done on pairs (v_pk_add_f32 / v_pk_mul_f32 on gfx950: two relaxations
// in 8 packed operations + 2 DPP moves instead of 16 instructions).
//
//   lower half of pair P[r]: row r,      versions 0 .. Q   (stages 1 .. Q)
//   upper half of pair P[r]: row r - KP, versions Q .. 2Q  (stages Q+1 .. 2Q = NS/2)
//
// When row y enters, the upper halves of its pairs are loaded from the lower halves of P[y - KP],
// which reached version Q two (O colour) / three (E colour) iterations earlier.  KP is NS/2 + 2 and
// not NS/2: with NS/2 the first packed stage's upper half would need, as its N neighbour, the value
// the LAST packed stage's lower half produces later in the same iteration; + 2 (not + 1) keeps the
// row parity of both halves equal, so both take the same DPP shift.  Consequence: a row leaves
// NS + 2 iterations after it entered (two more warm-up rows than the scalar pipeline).
//
// The right-hand side waits in the same kind of LDS ring as in the scalar pipeline (one 4-byte
// word per row, colour plane and lane), NS + 3 rows deep; a packed stage fetches its operand pair
// with two reads (slot of row r, slot of row r - KP).  Measured alternatives: a ring of 8-byte
// {d[r], d[r - KP]} elements (one ds_read_b64 per operand pair, d[y - KP] read back and rewritten
// when row y enters) is much slower (LDS bound: 364 against 210 us per launch at NS = 16).
//
// Every neighbour a half reads is the same half of the neighbouring row's pair, at exactly the
// version the scalar pipeline reads (the argument of sor_stream_core.h applies to each half), so
// the results are bit-identical.  Interior tiles only: boundary tiles keep the scalar EDGE path.
#pragma once
#include "sor_stream_core.h"

namespace sfl {
namespace sor {

// rows of d alive at once in the paired pipeline (NS + 3), rounded up to a multiple of 6; also the
// unroll factor and the modulus of the register arrays
constexpr int pair_ring_rows(int ns) { return ((ns + 3 + 5) / 6) * 6; }
// built for the depths where it pays (NS % 4 == 0 is required by the design)
constexpr bool pairs_supported(int ns) { return ns == 12 || ns == 16; }

template <class B, int NS>
struct PairPipe {
    using V = typename B::V;
    using V2 = typename B::V2;
    V2 E[pair_ring_rows(NS)];
    V2 O[pair_ring_rows(NS)];
    V pa[B::kPrefetch], pb[B::kPrefetch];
    V da[B::kPrefetch], db[B::kPrefetch];
};

// One relaxation of a pair, interior formula (poisson.cpp:101-112), each half rounded as the
// scalar relax<> rounds it.
template <class B, bool DX1>
SFL_HD typename B::V2 relax_pair(const B &bk, float dx, float omega, float one_minus_omega,
                                 typename B::V2 own, typename B::V2 w, typename B::V2 e,
                                 typename B::V2 s, typename B::V2 n, typename B::V2 d)
{
    using V2 = typename B::V2;
    const V2 sum = ((w + e) + s) + n;
    const V2 rhs = DX1 ? d : bk.scale2(dx, d);
    const V2 gs = bk.scale2(-0.25f, rhs - sum);
    return bk.scale2(one_minus_omega, own) + bk.scale2(omega, gs);
}

// Packed stage M of the iteration that receives row y (U = y mod RING): every index below is a
// constant expression (a runtime `for m` loop that is unrolled later leaves the arrays dynamically
// indexed for the first optimisation passes, and the small ones are then not split into registers).
template <class B, int NS, bool DX1, bool GUARD_STORE, int U, int M>
SFL_HD void pair_stage(B &bk, PairPipe<B, NS> &pp, float dx, float omega, float one_minus_omega, int y,
                       int out_begin, int out_end)
{
    using V2 = typename B::V2;
    constexpr int RING = pair_ring_rows(NS);
    constexpr int Q = NS / 4;
    constexpr int KP = NS / 2 + 2;
    // ---- E: lower half = E_M of row y - (2M - 1), upper half = E_{M+Q} of that row - KP ----
    {
        constexpr int rel = U - (2 * M - 1);
        constexpr int i0 = wrapn(rel, RING), im = wrapn(rel - 1, RING), ip = wrapn(rel + 1, RING);
        constexpr bool ev = is_even(rel);
        const V2 oc = pp.O[i0];
        const V2 w = ev ? bk.from_lower_lane2(oc) : oc;
        const V2 e = ev ? oc : bk.from_upper_lane2(oc);
        const V2 d = bk.ring_load2(i0, wrapn(rel - KP, RING), 0);
        pp.E[i0] = relax_pair<B, DX1>(bk, dx, omega, one_minus_omega, pp.E[i0], w, e, pp.O[im], pp.O[ip], d);
    }
    // ---- O: lower half = O_M of row y - 2M, upper half = O_{M+Q} of that row - KP ----
    {
        constexpr int rel = U - 2 * M;
        const int r_hi = y - 2 * M - KP;  // M == Q: y - NS - 2, the row that leaves
        constexpr int i0 = wrapn(rel, RING), im = wrapn(rel - 1, RING), ip = wrapn(rel + 1, RING);
        constexpr bool ev = is_even(rel);
        const V2 oc = pp.E[i0];
        const V2 w = ev ? oc : bk.from_lower_lane2(oc);
        const V2 e = ev ? bk.from_upper_lane2(oc) : oc;
        const V2 d = bk.ring_load2(i0, wrapn(rel - KP, RING), 1);
        const V2 res = relax_pair<B, DX1>(bk, dx, omega, one_minus_omega, pp.O[i0], w, e, pp.E[im], pp.E[ip], d);
        pp.O[i0] = res;
        if (M == Q && (!GUARD_STORE || (r_hi >= out_begin && r_hi < out_end))) {
            if (ev)
                bk.store_row(r_hi, bk.hi(oc), bk.hi(res));
            else
                bk.store_row(r_hi, bk.hi(res), bk.hi(oc));
        }
    }
}

template <class B, int NS, bool DX1, bool GUARD_STORE, int U, int... Ms>
SFL_HD void pair_stages(B &bk, PairPipe<B, NS> &pp, float dx, float omega, float one_minus_omega, int y,
                        int out_begin, int out_end, std::integer_sequence<int, Ms...>)
{
    (pair_stage<B, NS, DX1, GUARD_STORE, U, Ms + 1>(bk, pp, dx, omega, one_minus_omega, y, out_begin, out_end), ...);
}

template <class B, int NS, bool DX1, bool ZERO_IN, bool GUARD_STORE, int U>
SFL_HD void iterate_pairs(B &bk, PairPipe<B, NS> &pp, float dx, float omega, float one_minus_omega,
                          int y, int out_begin, int out_end)
{
    using V = typename B::V;
    constexpr int RING = pair_ring_rows(NS);
    constexpr int KP = NS / 2 + 2;
    constexpr int kPrefetch = B::kPrefetch;
    constexpr int P = U % kPrefetch;

    // ---- row y enters the lower halves; row y - KP moves to the upper halves ----
    {
        V a = bk.detach(pp.pa[P]), b = bk.detach(pp.pb[P]);
        const V fa = pp.da[P], fb = pp.db[P];
        constexpr bool ev = is_even(U);
        constexpr int from = wrapn(U - KP, RING);
        bk.ring_store(U, 0, ev ? fa : fb);  // plane 0: d of the E cell
        bk.ring_store(U, 1, ev ? fb : fa);  // plane 1: d of the O cell
        bk.load_row(y + kPrefetch, pp.pa[P], pp.pb[P], pp.da[P], pp.db[P]);
        if (ZERO_IN) a = b = bk.splat(0.0f);
        pp.E[U] = bk.make2(ev ? a : b, bk.lo(pp.E[from]));
        pp.O[U] = bk.make2(ev ? b : a, bk.lo(pp.O[from]));
    }
    pair_stages<B, NS, DX1, GUARD_STORE, U>(bk, pp, dx, omega, one_minus_omega, y, out_begin, out_end,
                                             std::make_integer_sequence<int, NS / 4>{});
}

template <class B, int NS, bool DX1, bool ZERO_IN, bool PARTIAL, bool GUARD_STORE, int... Us>
SFL_HD void run_unrolled_pairs(B &bk, PairPipe<B, NS> &pp, float dx, float omega,
                               float one_minus_omega, int y, int out_begin, int out_end, int y_stop,
                               std::integer_sequence<int, Us...>)
{
    (void)(((!PARTIAL || y + Us < y_stop) &&
            (iterate_pairs<B, NS, DX1, ZERO_IN, GUARD_STORE, Us>(bk, pp, dx, omega, one_minus_omega,
                                                                 y + Us, out_begin, out_end),
             true)) && ...);
}

// Stream one interior tile: output rows [out_begin, out_end), all NS passes.
template <class B, int NS, bool DX1, bool ZERO_IN>
SFL_HD void stream_tile_pairs(B &bk, float dx, float omega, float one_minus_omega, int out_begin,
                              int out_end)
{
    static_assert(pairs_supported(NS) && NS % 4 == 0, "the paired pipeline needs NS % 4 == 0");
    constexpr int RING = pair_ring_rows(NS);
    constexpr int kPrefetch = B::kPrefetch;
    static_assert(RING % 6 == 0 && 6 % kPrefetch == 0 && RING >= NS + 3, "ring geometry");
    PairPipe<B, NS> pp;
    bk.poison(pp);

    int y = out_begin - NS;
    y -= (y & 1);
    const int y_stop = out_end + NS + 2;  // the row that leaves is y - NS - 2

#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) bk.load_row(y + u, pp.pa[u], pp.pb[u], pp.da[u], pp.db[u]);

    constexpr auto us = std::make_integer_sequence<int, RING>{};
    for (; y + RING <= y_stop; y += RING)
        run_unrolled_pairs<B, NS, DX1, ZERO_IN, false, true>(bk, pp, dx, omega, one_minus_omega, y,
                                                             out_begin, out_end, y_stop, us);
    if (y < y_stop)
        run_unrolled_pairs<B, NS, DX1, ZERO_IN, true, true>(bk, pp, dx, omega, one_minus_omega, y,
                                                            out_begin, out_end, y_stop, us);
}

}  // namespace sor
}  // namespace sfl
