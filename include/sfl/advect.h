// sfl/advect.h -- semi-Lagrangian advection with the reference's template signature
// (ESP32-fluid-simulation/advect.h:74-76):
//
//     template <class T, class U>
//     void advect(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip);
//
// advect() over a whole field ALWAYS runs on the GPU; there is no CPU fallback in this library.
//   * Every element type the reference's headers can express -- float, UQ32, Vector2 / Vector3 of
//     either (vector.h:4-126, uq32.h:8-16) -- with a Vector2<float> velocity: HIP kernels inside
//     libsfl_hip.so, reached from any C++ compiler through libsfl_dropin.so.  The sketch's two
//     instantiations (Vector2<float>, ino:253; Vector3<UQ32>, ino:282) take the tuned tile kernels.
//   * Any OTHER element or velocity component type (a caller's own struct with the operators sample()
//     needs): when this header is compiled by hipcc, a kernel is instantiated from the templates below
//     (advect_device for device arrays; advect() itself stages host arrays).  A plain C++ compiler cannot
//     build device code for a type it has never seen: there the call is rejected at compile time.
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
#include <cstdio>
#include <cstdlib>
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
// element = `channels` consecutive 32-bit channels, uq32 ? UQ32 raw : float (sfl_host_advect_channels)
void advect_channels(void *next_p, void *p, Vector2<float> *vel, int dim_x, int dim_y, float dt, bool no_slip,
                     int channels, bool uq32);
}  // namespace sfl_dropin

// Element types with kernels inside the library: channels (0 = none) and channel kind.
template <class T> struct sfl_element { static constexpr int channels = 0; static constexpr bool uq32 = false; };
template <> struct sfl_element<float> { static constexpr int channels = 1; static constexpr bool uq32 = false; };
template <> struct sfl_element<UQ32> { static constexpr int channels = 1; static constexpr bool uq32 = true; };
template <> struct sfl_element<Vector2<float>> { static constexpr int channels = 2; static constexpr bool uq32 = false; };
template <> struct sfl_element<Vector2<UQ32>> { static constexpr int channels = 2; static constexpr bool uq32 = true; };
template <> struct sfl_element<Vector3<float>> { static constexpr int channels = 3; static constexpr bool uq32 = false; };
template <> struct sfl_element<Vector3<UQ32>> { static constexpr int channels = 3; static constexpr bool uq32 = true; };

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

#if defined(__HIPCC__)
// advect() for ANY element / velocity type, instantiated from the caller's own T by hipcc: one thread per
// cell, the header's own sample() (so exactly the arithmetic a host loop over sample() would do).
template <class T, class U>
__global__ void sfl_advect_any_kernel(T *next_p, T *p, const Vector2<U> *vel, int dim_x, int dim_y, float dt,
                                      bool no_slip)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= dim_x || j >= dim_y) return;
    const int ij = index(i, j, dim_x);
    const Vector2<float> source = Vector2<float>(i, j) - vel[ij] * dt;   // advect.h:81
    next_p[ij] = sample(p, source.x, source.y, dim_x, dim_y, no_slip);
}

// DEVICE arrays; asynchronous on `stream`.
template <class T, class U>
inline hipError_t advect_device(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip,
                                hipStream_t stream = nullptr)
{
    const dim3 block(64, 4), grid((dim_x + 63) / 64, (dim_y + 3) / 4);
    sfl_advect_any_kernel<T, U><<<grid, block, 0, stream>>>(next_p, p, vel, dim_x, dim_y, dt, no_slip);
    return hipGetLastError();
}

namespace sfl_detail {
// host arrays of a type without a kernel in the library: stage, run the instantiated kernel, fetch; loud on failure
template <class T, class U>
inline void advect_staged(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip)
{
    const size_t n = (size_t)dim_x * dim_y;
    T *d_next = nullptr, *d_p = nullptr;
    Vector2<U> *d_vel = nullptr;
    auto must = [](hipError_t e, const char *what) {
        if (e == hipSuccess) return;
        fprintf(stderr, "sfl: advect<T> (%s) failed: %s\n", what, hipGetErrorString(e));
        abort();
    };
    must(hipMalloc(reinterpret_cast<void **>(&d_next), n * sizeof(T)), "hipMalloc");
    must(hipMalloc(reinterpret_cast<void **>(&d_p), n * sizeof(T)), "hipMalloc");
    must(hipMalloc(reinterpret_cast<void **>(&d_vel), n * sizeof(Vector2<U>)), "hipMalloc");
    must(hipMemcpy(d_p, p, n * sizeof(T), hipMemcpyHostToDevice), "upload");
    must(hipMemcpy(d_vel, vel, n * sizeof(Vector2<U>), hipMemcpyHostToDevice), "upload");
    must(advect_device(d_next, d_p, d_vel, dim_x, dim_y, dt, no_slip), "launch");
    must(hipMemcpy(next_p, d_next, n * sizeof(T), hipMemcpyDeviceToHost), "download");
    (void)hipFree(d_next);
    (void)hipFree(d_p);
    (void)hipFree(d_vel);
}
}  // namespace sfl_detail
#endif  // __HIPCC__

template <class T, class U>
void advect(T *next_p, T *p, Vector2<U> *vel, int dim_x, int dim_y, float dt, bool no_slip)
{
    constexpr bool in_library = sfl_element<T>::channels > 0 && std::is_same<U, float>::value;
    if constexpr (std::is_same<T, Vector2<float>>::value && in_library)
        sfl_dropin::advect_vec2f(next_p, p, vel, dim_x, dim_y, dt, no_slip);
    else if constexpr (std::is_same<T, Vector3<UQ32>>::value && in_library)
        sfl_dropin::advect_vec3uq32(next_p, p, vel, dim_x, dim_y, dt, no_slip);
    else if constexpr (in_library)
        sfl_dropin::advect_channels(next_p, p, vel, dim_x, dim_y, dt, no_slip, sfl_element<T>::channels,
                                    sfl_element<T>::uq32);
    else {
#if defined(__HIPCC__)
        sfl_detail::advect_staged(next_p, p, vel, dim_x, dim_y, dt, no_slip);
#else
        static_assert(in_library,
                      "advect: the library holds GPU kernels for float, UQ32 and Vector2 / Vector3 of either, "
                      "advected by a Vector2<float> velocity; for any other element or velocity type compile the "
                      "caller with hipcc (the kernel is then instantiated from this header) -- there is no CPU "
                      "fallback");
#endif
    }
}

#endif  // SFL_ADVECT_H
