// ref_shim.cpp -- C entry points onto the UNMODIFIED reference sources.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no
// solver arithmetic: it includes the reference's own headers from where they
// lie (-I/root/reference/ESP32-fluid-simulation, given by oracle/Makefile) and
// forwards flat arrays to the reference's functions so that Python (ctypes)
// can call them.  It is compiled together with the reference's finitediff.cpp
// and poisson.cpp into oracle/_ref/libsf_ref.so, which is git-ignored.
//
// Reference entry points wrapped (file:line under ESP32-fluid-simulation/):
//   advect<T,U>            advect.h:74-85
//   calculate_divergence   finitediff.h:6-7   (finitediff.cpp:33-39)
//   subtract_gradient      finitediff.h:9-10  (finitediff.cpp:75-82)
//   poisson_solve          poisson.h:4-5      (poisson.cpp:114-125)
#include <cstdint>
#include <cstring>

#include "vector.h"
#include "uq32.h"
#include "advect.h"
#include "finitediff.h"
#include "poisson.h"

static_assert(sizeof(Vector2<float>) == 8, "Vector2<float> must be two packed floats");
static_assert(sizeof(Vector3<UQ32>) == 12, "Vector3<UQ32> must be three packed uint32");

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_advect_vec2f(float *next_p, float *p, float *vel, int dim_x, int dim_y,
                              float dt, int no_slip)
{
    advect(reinterpret_cast<Vector2<float> *>(next_p), reinterpret_cast<Vector2<float> *>(p),
           reinterpret_cast<Vector2<float> *>(vel), dim_x, dim_y, dt, no_slip != 0);
}

REF_API void ref_advect_vec3uq32(uint32_t *next_p, uint32_t *p, float *vel, int dim_x,
                                 int dim_y, float dt, int no_slip)
{
    advect(reinterpret_cast<Vector3<UQ32> *>(next_p), reinterpret_cast<Vector3<UQ32> *>(p),
           reinterpret_cast<Vector2<float> *>(vel), dim_x, dim_y, dt, no_slip != 0);
}

// advect<T, float> for every element type the reference's headers can express (channels x kind as in
// include/sfl.h: kind 0 = float channels, 1 = UQ32 channels); returns 0, or -1 for a combination that is none
template <class T>
static void advect_as(void *next_p, void *p, float *vel, int dim_x, int dim_y, float dt, int no_slip)
{
    advect(static_cast<T *>(next_p), static_cast<T *>(p), reinterpret_cast<Vector2<float> *>(vel), dim_x, dim_y, dt,
           no_slip != 0);
}
REF_API int ref_advect_channels(void *next_p, void *p, float *vel, int dim_x, int dim_y, float dt, int no_slip,
                                int channels, int kind)
{
    switch (channels * 2 + (kind ? 1 : 0)) {
        case 2: advect_as<float>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
        case 3: advect_as<UQ32>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
        case 4: advect_as<Vector2<float>>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
        case 5: advect_as<Vector2<UQ32>>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
        case 6: advect_as<Vector3<float>>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
        case 7: advect_as<Vector3<UQ32>>(next_p, p, vel, dim_x, dim_y, dt, no_slip); return 0;
    }
    return -1;
}

REF_API void ref_divergence(float *div, float *v, int dim_x, int dim_y, float dx)
{
    calculate_divergence(div, reinterpret_cast<Vector2<float> *>(v), dim_x, dim_y, dx);
}

REF_API void ref_subtract_gradient(float *v, float *p, int dim_x, int dim_y, float dx)
{
    subtract_gradient(reinterpret_cast<Vector2<float> *>(v), p, dim_x, dim_y, dx);
}

REF_API void ref_poisson_solve(float *p, float *div, int dim_x, int dim_y, float dx, int iters,
                               float omega)
{
    poisson_solve(p, div, dim_x, dim_y, dx, iters, omega);
}

// One sim step in the call order of ESP32-fluid-simulation.ino:252-287
// (no touch-force injection, no RTOS semaphores).  v / colour updated in place.
REF_API int ref_step(float *v, uint32_t *colour, float *div, float *p, int dim_x, int dim_y,
                     float dt, float dx, int iters, float omega)
{
    const size_t n = static_cast<size_t>(dim_x) * dim_y;
    Vector2<float> *vf = reinterpret_cast<Vector2<float> *>(v);
    Vector3<UQ32> *cf = reinterpret_cast<Vector3<UQ32> *>(colour);
    Vector2<float> *v_temp = new Vector2<float>[n];
    advect(v_temp, vf, vf, dim_x, dim_y, dt, true);
    std::memcpy(v, v_temp, n * sizeof(Vector2<float>));
    delete[] v_temp;
    calculate_divergence(div, vf, dim_x, dim_y, dx);
    poisson_solve(p, div, dim_x, dim_y, dx, iters, omega);
    subtract_gradient(vf, p, dim_x, dim_y, dx);
    Vector3<UQ32> *c_temp = new Vector3<UQ32>[n];
    advect(c_temp, cf, vf, dim_x, dim_y, dt, false);
    std::memcpy(colour, c_temp, n * sizeof(Vector3<UQ32>));
    delete[] c_temp;
    return 0;
}
