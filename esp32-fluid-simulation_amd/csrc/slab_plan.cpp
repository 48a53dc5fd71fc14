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
                                        int kernel)
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

    const std::vector<int> passes = sor_pass_plan(iters, fuse);
    if (multi && !passes.empty()) {
        // pass j of a launch relaxes rows own +- (n - j): the right-hand side is needed n - 1
        // rows beyond the slab, once per solve
        int widest = 0;
        for (int n : passes) widest = n > widest ? n : widest;
        if (widest > 1) prog.push_back(exchange(SFL_FIELD_DIVERGENCE, widest - 1));
    }
    for (size_t k = 0; k < passes.size(); ++k) {
        if (multi && k > 0) prog.push_back(exchange(SFL_FIELD_PRESSURE, passes[k]));
        sfl_plan_step c{};
        c.kind = SFL_STEP_SOR;
        c.g_begin = g0;
        c.g_end = g1;
        c.nsweeps = passes[k];
        c.first_colour = 0;
        c.from_zero = (k == 0);
        prog.push_back(c);
    }
    return prog;
}

}  // namespace sfl
