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

static sfl_plan_step exchange(int field, int rows)
{
    sfl_plan_step s{};
    s.kind = SFL_STEP_EXCHANGE;
    s.field = field;
    s.rows = rows;
    return s;
}

std::vector<sfl_plan_step> plan_poisson(int dim_y, int nranks, int rank, int iters, int fuse,
                                        int kernel, int halo)
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
    if (halo < fuse) halo = fuse;
    std::vector<std::vector<int>> groups;
    for (int n : passes) {
        int sum = 0;
        if (!groups.empty())
            for (int m : groups.back()) sum += m;
        if (groups.empty() || !multi || sum + n > halo) groups.emplace_back();
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
        if (deepest > 1) prog.push_back(exchange(SFL_FIELD_DIVERGENCE, deepest - 1));
    }
    bool first = true;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        int left = 0;
        for (int m : groups[gi]) left += m;
        if (multi && gi > 0) prog.push_back(exchange(SFL_FIELD_PRESSURE, left));
        for (int n : groups[gi]) {
            left -= n;
            const int extra = multi ? left : 0;
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
