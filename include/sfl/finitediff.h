// sfl/finitediff.h -- divergence and pressure-gradient subtraction with the reference's exact
// signatures (ESP32-fluid-simulation/finitediff.h:6-10).  Host pointers in, host pointers out;
// the work is done by the HIP kernels behind the C ABI (sfl_host_calculate_divergence,
// sfl_host_subtract_gradient).  Like the reference they return void; a failure (no GPU, bad
// dimensions) prints the library's message and aborts.  Link with libsfl_dropin.so.
#ifndef SFL_FINITEDIFF_H
#define SFL_FINITEDIFF_H

#include "vector.h"

void calculate_divergence(float *div, Vector2<float> *v, int dim_x, int dim_y, float dx);

void subtract_gradient(Vector2<float> *v, float *p, int dim_x, int dim_y, float dx);

#endif  // SFL_FINITEDIFF_H
