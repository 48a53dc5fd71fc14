// sfl/operations.h -- field indexing and the generic per-cell driver, source compatible with the
// reference (ESP32-fluid-simulation/operations.h:4-38).
//
//   index(i, j, dim_x)      dim_x * j + i, i fastest                         (operations.h:7-9)
//   kernel_func_t<T, U>     U (*)(T *cell, int i, int j, int dim_x, int dim_y, void *ctx); `cell`
//                           points AT the centre element                     (operations.h:4-5)
//   domain_iter(...)        interior cells through expr_fast, perimeter cells through expr_safe;
//                           `wrt` may alias `rd`                             (operations.h:11-38)
//
// domain_iter takes HOST function pointers, so it necessarily runs on the host: it is kept for
// source compatibility with callers that bring their own per-cell expressions.  The four
// expression pairs the sketch itself uses (divergence, gradient subtraction, the two SOR
// flavours) are NOT routed through it here -- they are the HIP kernels behind finitediff.h and
// poisson.h.  domain_for_each() below is the device-side generalisation for user expressions
// written as functors (HIP builds only).
#ifndef SFL_OPERATIONS_H
#define SFL_OPERATIONS_H

#include "vector.h"

template <class T, class U>
using kernel_func_t = U (*)(T *, int i, int j, int dim_x, int dim_y, void *ctx);

SFL_XPU static inline int index(int i, int j, int dim_x) { return j * dim_x + i; }

template <class T, class U>
void domain_iter(kernel_func_t<T, U> expr_safe, kernel_func_t<T, U> expr_fast, U *wrt, T *rd,
                 int dim_x, int dim_y, void *ctx)
{
    const int last_i = dim_x - 1, last_j = dim_y - 1;
    auto visit = [&](kernel_func_t<T, U> expr, int i, int j) {
        const int c = index(i, j, dim_x);
        wrt[c] = expr(rd + c, i, j, dim_x, dim_y, ctx);
    };
    // same visiting order as the reference: interior, then bottom / top rows (corners included,
    // alternating per column), then left / right columns -- it matters when wrt aliases rd
    for (int j = 1; j < last_j; ++j)
        for (int i = 1; i < last_i; ++i) visit(expr_fast, i, j);
    for (int i = 0; i <= last_i; ++i) {
        visit(expr_safe, i, 0);
        visit(expr_safe, i, last_j);
    }
    for (int j = 1; j < last_j; ++j) {
        visit(expr_safe, 0, j);
        visit(expr_safe, last_i, j);
    }
}

#if defined(__HIPCC__)
// Device-side counterpart: `safe` / `fast` are functors
//   U operator()(const T *cell, int i, int j, int dim_x, int dim_y) const   (device callable)
// applied to every cell of DEVICE arrays rd -> wrt (wrt must NOT alias rd: cells are visited
// concurrently).  One thread per cell, rows of 256 threads.
template <class T, class U, class Safe, class Fast>
__global__ void sfl_domain_for_each_kernel(Safe safe, Fast fast, U *wrt, const T *rd, int dim_x, int dim_y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= dim_x) return;
    const int c = index(i, j, dim_x);
    const bool inner = i > 0 && i < dim_x - 1 && j > 0 && j < dim_y - 1;
    wrt[c] = inner ? fast(rd + c, i, j, dim_x, dim_y) : safe(rd + c, i, j, dim_x, dim_y);
}

template <class T, class U, class Safe, class Fast>
inline hipError_t domain_for_each(Safe safe, Fast fast, U *wrt, const T *rd, int dim_x, int dim_y,
                                  hipStream_t stream = nullptr)
{
    const dim3 grid((dim_x + 255) / 256, dim_y);
    sfl_domain_for_each_kernel<T, U><<<grid, 256, 0, stream>>>(safe, fast, wrt, rd, dim_x, dim_y);
    return hipGetLastError();
}
#endif

#endif  // SFL_OPERATIONS_H
