// dropin.cpp -- C++ host layer: the reference's free functions (finitediff.h:6-10, poisson.h:4-5,
// advect.h:74-76) implemented on top of the C ABI (include/sfl.h).  Built as libsfl_dropin.so,
// which depends on libsfl_hip.so.  The reference's functions return void and never fail; here a
// failure of the GPU path is fatal and loud (message + abort) -- never a silent CPU substitute.
#include <cstdio>
#include <cstdlib>

#include "sfl.h"
#include "sfl/advect.h"
#include "sfl/finitediff.h"
#include "sfl/poisson.h"

namespace {
void must(int rc, const char *what)
{
    if (rc == SFL_OK) return;
    std::fprintf(stderr, "sfl: %s failed (%d): %s\n", what, rc, sfl_last_error());
    std::abort();
}
float *flat(Vector2<float> *v) { return reinterpret_cast<float *>(v); }
}  // namespace

void calculate_divergence(float *div, Vector2<float> *v, int dim_x, int dim_y, float dx)
{
    must(sfl_host_calculate_divergence(div, flat(v), dim_x, dim_y, dx), "calculate_divergence");
}

void subtract_gradient(Vector2<float> *v, float *p, int dim_x, int dim_y, float dx)
{
    must(sfl_host_subtract_gradient(flat(v), p, dim_x, dim_y, dx), "subtract_gradient");
}

void poisson_solve(float *p, float *div, int dim_x, int dim_y, float dx, int iters, float omega)
{
    must(sfl_host_poisson_solve(p, div, dim_x, dim_y, dx, iters, omega), "poisson_solve");
}

namespace sfl_dropin {
void advect_vec2f(Vector2<float> *next_p, Vector2<float> *p, Vector2<float> *vel, int dim_x,
                  int dim_y, float dt, bool no_slip)
{
    must(sfl_host_advect_vec2f(flat(next_p), flat(p), flat(vel), dim_x, dim_y, dt, no_slip),
         "advect<Vector2<float>>");
}

void advect_vec3uq32(Vector3<UQ32> *next_p, Vector3<UQ32> *p, Vector2<float> *vel, int dim_x,
                     int dim_y, float dt, bool no_slip)
{
    must(sfl_host_advect_vec3uq32(reinterpret_cast<uint32_t *>(next_p),
                                  reinterpret_cast<uint32_t *>(p), flat(vel), dim_x, dim_y, dt,
                                  no_slip),
         "advect<Vector3<UQ32>>");
}

void advect_channels(void *next_p, void *p, Vector2<float> *vel, int dim_x, int dim_y, float dt, bool no_slip,
                     int channels, bool uq32)
{
    must(sfl_host_advect_channels(next_p, p, flat(vel), dim_x, dim_y, dt, no_slip, channels,
                                  uq32 ? SFL_CHANNEL_UQ32 : SFL_CHANNEL_F32),
         "advect<T>");
}
}  // namespace sfl_dropin
