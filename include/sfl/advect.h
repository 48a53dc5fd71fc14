// sfl/advect.h -- semi-Lagrangian advection with the reference's template signature
// (ESP32-fluid-simulation/advect.h:74-76):
//
//     template <class T, class U>
//     void advect(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip);
//
// The two instantiations the sketch uses are GPU paths (ino:253 and ino:282):
//     T = Vector2<float>, U = float   ->  sfl_host_advect_vec2f
//     T = Vector3<UQ32>,  U = float   ->  sfl_host_advect_vec3uq32
// Any other element type is rejected at compile time: there is no CPU fallback in this library.
// next_p must not alias p; p may alias vel (self-advection).  Link with libsfl_dropin.so.
#ifndef SFL_ADVECT_H
#define SFL_ADVECT_H

#include <type_traits>

#include "operations.h"
#include "uq32.h"
#include "vector.h"

namespace sfl_dropin {
void advect_vec2f(Vector2<float> *next_p, Vector2<float> *p, Vector2<float> *vel, int dim_x,
                  int dim_y, float dt, bool no_slip);
void advect_vec3uq32(Vector3<UQ32> *next_p, Vector3<UQ32> *p, Vector2<float> *vel, int dim_x,
                     int dim_y, float dt, bool no_slip);
}  // namespace sfl_dropin

template <class T, class U>
void advect(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip)
{
    constexpr bool velocity_field = std::is_same<T, Vector2<float>>::value;
    constexpr bool dye_field = std::is_same<T, Vector3<UQ32>>::value;
    static_assert(std::is_same<U, float>::value, "advect: the velocity field must be Vector2<float>");
    static_assert(velocity_field || dye_field,
                  "advect: GPU kernels exist for T = Vector2<float> and T = Vector3<UQ32> "
                  "(the instantiations of the sketch); other element types are not supported");
    if constexpr (velocity_field)
        sfl_dropin::advect_vec2f(next_p, p, vel, dim_x, dim_y, dt, no_slip);
    else
        sfl_dropin::advect_vec3uq32(next_p, p, vel, dim_x, dim_y, dt, no_slip);
}

#endif  // SFL_ADVECT_H
