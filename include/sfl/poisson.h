// sfl/poisson.h -- red-black SOR pressure solve with the reference's exact signature
// (ESP32-fluid-simulation/poisson.h:4-5).  p is overwritten (zero initial guess, poisson.cpp:117).
// Host pointers; runs the fused HIP kernel through sfl_host_poisson_solve.  Link with
// libsfl_dropin.so.
#ifndef SFL_POISSON_H
#define SFL_POISSON_H

void poisson_solve(float *p, float *div, int dim_x, int dim_y, float dx, int iters, float omega);

#endif  // SFL_POISSON_H
