// slab_plan.cpp -- see slab_plan.h.  Pure host code.
#include "slab_plan.h"

namespace sfl {

std::vector<int> sor_pass_plan(int iters, int fuse)
{
    std::vector<int> passes;
    int left = 2 * iters;
    while (left > 0) {
        const int n = left < fuse ? left : fuse;
        passes.push_back(n);
        left -= n;
    }
    return passes;
}

static sfl_plan_step exchange(int field, int rows, int skip = 0)
{
    sfl_plan_step s{};
    s.kind = SFL_STEP_EXCHANGE;
    s.field = field;
    s.rows = rows;
    s.g_begin = skip;
    return s;
}

std::vector<sfl_plan_step> plan_poisson(int dim_y, int nranks, int rank, int iters, int fuse,
                                        int kernel, int halo, int tail)
{
    std::vector<sfl_plan_step> prog;
    int g0, g1;
    slab_rows(dim_y, nranks, rank, &g0, &g1);
    const bool multi = nranks > 1;

    if (kernel == 1) {
        // baseline: zero fill, then one launch per colour pass; a pass reads the other
        // colour one row beyond the slab, so one ghost row per side is refreshed before it
        // (not before the very first: p is zero everywhere)
        sfl_plan_step z{};
        z.kind = SFL_STEP_ZERO;
        z.field = SFL_FIELD_PRESSURE;
        prog.push_back(z);
        for (int s = 0; s < 2 * iters; ++s) {
            if (multi && s > 0) prog.push_back(exchange(SFL_FIELD_PRESSURE, 1));
            sfl_plan_step c{};
            c.kind = SFL_STEP_SOR;
            c.g_begin = g0;
            c.g_end = g1;
            c.nsweeps = 1;
            c.first_colour = s & 1;
            prog.push_back(c);
        }
        return prog;
    }

    // Fused kernel.  Launches are grouped into SUPERSTEPS whose colour passes add up to at most
    // `halo` rows: one exchange of that many rows of p per superstep, after which launch i of the
    // superstep still owns exact input on own +- (rows left) and therefore produces output rows
    // own +- (rows left after it) -- the ghost rows are recomputed redundantly instead of being
    // exchanged after every launch (messages are latency-bound, so the lever is the
    // exchange COUNT).  The first superstep needs no exchange: p is zero everywhere.
    const std::vector<int> passes = sor_pass_plan(iters, fuse);
    if (passes.empty()) return prog;   // iters == 0: nothing to launch, nothing to exchange (the early-exchange branch below
                                       // indexed its empty tables at n - 1: found by UBSan, tests/cpp/host_san_driver.cpp)
    if (halo < fuse) halo = fuse;

    // kernel 3 = kernel 2's launches with IN-TIME exchanges at every halo depth: the halo of a superstep is sent after the
    // launch that produces it (the executor lets it leave as soon as that launch's cut-adjacent tiles are done and lets
    // only the next launch's cut-adjacent tiles wait for it: sor_executor.cpp run_poisson_in_time)
    const bool in_time = kernel == 3;
    if (tail < 0 || !multi || (!in_time && halo < 2 * fuse + tail) || (in_time && halo < fuse + tail)) tail = 0;
    if (multi && halo >= 2 * fuse && !in_time) {
        // EARLY exchanges (halo of at least two launches).  A launch needs `nsweeps` valid ghost rows to produce
        // its own rows; everything deeper only feeds later launches.  So the halo for the next group of launches
        // is sent ONE LAUNCH EARLY -- before the last launch e of the running group, while the ghost rows are
        // still valid `n_e` deep: only the rows beyond that depth travel (skip = n_e, rows = halo - n_e), launch
        // e produces the owned rows from what is already there, and the receiver repeats launch e's passes on
        // the received rows (output rows own +- left, left <= halo - n_e).  The executor can therefore run the
        // owned rows of launch e WHILE the message is in flight and relax the ghost rows behind it on the
        // exchange stream: no launch is split, nothing on the compute stream waits for the wire except the
        // launch after e.  Groups: the first may hold `halo` passes (p starts at zero: nothing to send), the
        // following ones halo - n_e (what is left of the received rows after launch e's passes).
        const size_t n = passes.size();
        // extra[j] = ghost rows that must still be valid after launch j = passes of the launches up to and
        // including the next exchange launch (whose owned rows are produced from what is there)
        std::vector<int> xchg_before(n, 0), extra(n, 0);
        auto place_exchanges = [&](int t, std::vector<int> *where) {
            int count = 0, budget = halo - t;
            for (size_t j = 0; j < n; ++j) {
                if (passes[j] > budget) {  // launch j does not fit what is left: the exchange goes before launch j - 1
                    if (where) (*where)[j - 1] = 1;
                    ++count;
                    budget = halo - t - passes[j - 1];
                }
                budget -= passes[j];
            }
            return count;
        };
        // The tail saves ONE 1-row exchange after the solve; it must not cost exchanges inside it (fuse 16 / halo 64
        // is one row short of four launches per superstep with a tail: 11 exchanges instead of 7 over 200
        // iterations).  Dropped then -- subtract_gradient exchanges its row as it always could.
        if (tail > 0 && place_exchanges(tail, nullptr) > place_exchanges(0, nullptr)) tail = 0;
        place_exchanges(tail, &xchg_before);
        extra[n - 1] = tail;
        for (size_t j = n; j-- > 0;) {
            // rows needed after launch j: the following launches up to and including the next exchange launch
            // (whose owned rows + tail come from what is there: the exchange skips passes + tail rows)
            if (j + 1 < n) extra[j] = xchg_before[j + 1] ? passes[j + 1] + tail : extra[j + 1] + passes[j + 1];
        }
        int deepest = 0;  // pass 1 of launch j relaxes its output rows +- (n_j - 1): rows of the right-hand side
        for (size_t j = 0; j < n; ++j) deepest = extra[j] + passes[j] > deepest ? extra[j] + passes[j] : deepest;
        if (deepest > 1) prog.push_back(exchange(SFL_FIELD_DIVERGENCE, deepest - 1));
        for (size_t j = 0; j < n; ++j) {
            if (xchg_before[j])
                prog.push_back(exchange(SFL_FIELD_PRESSURE, halo - passes[j] - tail, passes[j] + tail));
            sfl_plan_step c{};
            c.kind = SFL_STEP_SOR;
            c.g_begin = g0 - extra[j] < 0 ? 0 : g0 - extra[j];
            c.g_end = g1 + extra[j] > dim_y ? dim_y : g1 + extra[j];
            c.nsweeps = passes[j];
            c.first_colour = 0;
            c.from_zero = j == 0;
            prog.push_back(c);
        }
        return prog;
    }
    // (a tail -- in-time plans only get here with one -- is carried by every superstep: `tail` ghost rows stay exact behind
    // each launch, so the last launch of the solve leaves them as well; a superstep then holds halo - tail passes)
    std::vector<std::vector<int>> groups;
    for (int n : passes) {
        int sum = 0;
        if (!groups.empty())
            for (int m : groups.back()) sum += m;
        if (groups.empty() || !multi || sum + n > halo - tail) groups.emplace_back();
        groups.back().push_back(n);
    }
    if (multi) {
        int deepest = 0;
        for (const auto &g : groups) {
            int sum = 0;
            for (int m : g) sum += m;
            deepest = sum > deepest ? sum : deepest;
        }
        // pass j of a launch relaxes its output rows +- (n - j): the right-hand side is needed
        // one row less than p, once per solve
        if (deepest + tail > 1) prog.push_back(exchange(SFL_FIELD_DIVERGENCE, deepest + tail - 1));
    }
    bool first = true;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        int left = 0;
        for (int m : groups[gi]) left += m;
        // (the `tail` ghost rows next to the cut are still exact: the last launch of the superstep before left them)
        if (multi && gi > 0) prog.push_back(exchange(SFL_FIELD_PRESSURE, left, tail));
        for (int n : groups[gi]) {
            left -= n;
            const int extra = multi ? left + tail : 0;
            sfl_plan_step c{};
            c.kind = SFL_STEP_SOR;
            c.g_begin = g0 - extra < 0 ? 0 : g0 - extra;
            c.g_end = g1 + extra > dim_y ? dim_y : g1 + extra;
            c.nsweeps = n;
            c.first_colour = 0;
            c.from_zero = first;
            first = false;
            prog.push_back(c);
        }
    }
    return prog;
}

}  // namespace sfl
