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
//
// The reference's header also exposes its per-POINT helpers to every includer (advect.h:10-72):
// TPromoted, lerp, billinear_interpolate and sample.  They are kept here as header templates with
// the same names, argument order and arithmetic (each product and sum rounded on its own, result
// narrowed to T on return), usable from host and device code: a caller that samples a field at a
// few points of its own (a probe, a particle) recompiles unchanged.  They evaluate ONE point; the
// data-parallel path -- advect() over a whole field -- always runs the HIP kernels.
#ifndef SFL_ADVECT_H
#define SFL_ADVECT_H

#include <cmath>
#include <type_traits>
#include <utility>

#include "operations.h"
#include "uq32.h"
#include "vector.h"

namespace sfl_dropin {
void advect_vec2f(Vector2<float> *next_p, Vector2<float> *p, Vector2<float> *vel, int dim_x,
                  int dim_y, float dt, bool no_slip);
void advect_vec3uq32(Vector3<UQ32> *next_p, Vector3<UQ32> *p, Vector2<float> *vel, int dim_x,
                     int dim_y, float dt, bool no_slip);
}  // namespace sfl_dropin

// ---- per-point helpers (advect.h:10-72) -------------------------------------------------------
// what T becomes when scaled by a float: float stays float, Vector<anything> -> Vector<float>
template <typename T>
using TPromoted = decltype(std::declval<T>() * std::declval<float>());

// a weighted by (1 - t) plus b weighted by t: two products, then one sum (advect.h:13-16)
template <class T>
SFL_XPU static TPromoted<T> lerp(float t, T a, T b)
{
    const TPromoted<T> left = a * (1 - t), right = b * t;
    return left + right;
}

// first along j on both columns, then along i (advect.h:18-22); p11 = (i, j), p12 = (i, j + 1),
// p21 = (i + 1, j), p22 = (i + 1, j + 1)
template <class T>
SFL_XPU static TPromoted<T> billinear_interpolate(float di, float dj, T p11, T p12, T p21, T p22)
{
    const TPromoted<T> column_i = lerp(dj, p11, p12), column_i1 = lerp(dj, p21, p22);
    return lerp(di, column_i, column_i1);
}

namespace sfl_detail {
// how far a coordinate lies outside [0, last]: <0 below, >0 at or above `last`, 0 inside
SFL_XPU inline int side_of(float c, int last) { return c < 0 ? -1 : (c >= last ? 1 : 0); }
// no-slip weight of one axis: falls linearly to zero half a cell outside the wall (advect.h:64-70)
SFL_XPU inline float wall_weight(float c, int last, int side)
{
    const float beyond = side < 0 ? -c : c - last;
    return beyond < 0.5 ? (1 - 2 * beyond) : 0;
}
}  // namespace sfl_detail

// Field value at the real-valued position (i, j) (advect.h:24-72): bilinear inside; along the wall
// when one coordinate is outside; the corner texel when both are; with no_slip the wall value is
// scaled towards zero.  Every branch narrows to T exactly where the reference does.
template <class T>
SFL_XPU static T sample(T *p, float i, float j, int dim_x, int dim_y, bool no_slip)
{
    const int last_x = dim_x - 1, last_y = dim_y - 1;
    const int sx = sfl_detail::side_of(i, last_x), sy = sfl_detail::side_of(j, last_y);
    const float fi = floorf(i), fj = floorf(j);
    const float di = i - fi, dj = j - fj;
    if (sx == 0 && sy == 0) {
        T *c = p + index(fi, fj, dim_x);
        return billinear_interpolate(di, dj, c[0], c[dim_x], c[1], c[dim_x + 1]);
    }
    const int wall_i = sx < 0 ? 0 : last_x, wall_j = sy < 0 ? 0 : last_y;
    T on_wall;
    if (sx != 0 && sy != 0) {
        on_wall = p[index(wall_i, wall_j, dim_x)];
    } else if (sx != 0) {
        T *c = p + index(wall_i, fj, dim_x);
        on_wall = lerp(dj, c[0], c[dim_x]);
    } else {
        T *c = p + index(fi, wall_j, dim_x);
        on_wall = lerp(di, c[0], c[1]);
    }
    if (!no_slip) return on_wall;
    float weight = 1.0f;
    if (sx != 0) weight *= sfl_detail::wall_weight(i, last_x, sx);
    if (sy != 0) weight *= sfl_detail::wall_weight(j, last_y, sy);
    return weight * on_wall;
}

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
