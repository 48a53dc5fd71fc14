// slab_plan.h -- pure host arithmetic of the row-slab decomposition and of the launch /
// halo-exchange schedule of one poisson_solve.  No HIP, no RCCL: this translation unit is what
// the CPU (gloo, world_size 2) tests exercise through the C ABI (sfl_slab_rows,
// sfl_sor_pass_plan, sfl_plan_poisson), and what the GPU executor in sor_executor.cpp walks.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/sfl.h"

namespace sfl {

// rank g owns global rows [dim_y*g/n, dim_y*(g+1)/n)   
inline void slab_rows(int dim_y, int nranks, int rank, int *begin, int *end)
{
    *begin = (int)((int64_t)dim_y * rank / nranks);
    *end = (int)((int64_t)dim_y * (rank + 1) / nranks);
}

// 2*iters colour passes cut into launches of at most `fuse` (even) passes.
std::vector<int> sor_pass_plan(int iters, int fuse);

// Program of one poisson_solve for one rank; every rank's program has the same length and the
// same kinds at the same positions (exchanges are matched pairs).
// halo = rows of p exchanged per superstep (>= fuse; kernel 2 only)
// tail = ghost rows that must still be exact when the solve ends (early exchanges only: every launch then extends
// `tail` rows further into the ghost rows; sfl_step on slabs asks for 1 -- the row subtract_gradient reads --
// and saves the 1-row exchange of p after the solve)
std::vector<sfl_plan_step> plan_poisson(int dim_y, int nranks, int rank, int iters, int fuse,
                                        int kernel, int halo, int tail = 0);

}  // namespace sfl
