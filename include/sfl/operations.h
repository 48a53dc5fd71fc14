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
// poisson.h.  domain_for_each() / domain_for_each_red_black() below are the device-side generalisation for
// user expressions written as functors (HIP builds only), in place included.
#ifndef SFL_OPERATIONS_H
#define SFL_OPERATIONS_H

#include <type_traits>

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
// ---- device-side counterparts (HIP builds only) -------------------------------------------
// `safe` / `fast` are functors  U operator()(const T *cell, int i, int j, int dim_x, int dim_y) const  (device
// callable; `cell` points AT the centre element as in kernel_func_t, state travels in the functor instead of
// `void *ctx`), applied to DEVICE arrays.
//
//   domain_for_each(safe, fast, wrt, rd, ...)            the generic driver, operations.h:11-38.
//       wrt != rd   one thread per cell, all at once.
//       wrt == rd   (the reference allows it and uses it, finitediff.cpp:80) -- the bits of the reference's
//                   visiting order, for expressions that read, of rd, the centre and cells of its own row and
//                   column (the five-point neighbourhood and beyond): in the reference's order a cell sees its W / S
//                   side updated and its E / N side not yet; anti-diagonals of the interior keep exactly that and
//                   are swept one after the other by ONE workgroup, the perimeter then by one thread in the
//                   reference's sequence (bottom / top per column, left / right per row).  Correct, and bound by
//                   the dependency chain: ~2 us per diagonal.  Expressions that read only the centre element of rd
//                   (subtract_gradient's do: the neighbours they read belong to another field) are order-free:
//                   pass sfl_in_place::pointwise and every cell runs at once.
//   domain_for_each_red_black(safe, fast, field, ...)     the colour-split in-place driver of poisson.cpp:14-61
//       (domain_iter_red_black): all cells of even i + j, then all cells of odd i + j; an expression may read the
//       four neighbours (the other colour) and the centre.  Two launches, every cell of a colour at once.
enum class sfl_in_place { reference_order, pointwise };

template <class T, class U, class Safe, class Fast>
__global__ void sfl_domain_for_each_kernel(Safe safe, Fast fast, U *wrt, const T *rd, int dim_x, int dim_y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= dim_x) return;
    const int c = index(i, j, dim_x);
    const bool inner = i > 0 && i < dim_x - 1 && j > 0 && j < dim_y - 1;
    wrt[c] = inner ? fast(rd + c, i, j, dim_x, dim_y) : safe(rd + c, i, j, dim_x, dim_y);
}

// wrt == rd in the reference's order (see above): launched as ONE workgroup
template <class T, class Safe, class Fast>
__global__ void sfl_domain_for_each_in_order_kernel(Safe safe, Fast fast, T *field, int dim_x, int dim_y)
{
    const int last_i = dim_x - 1, last_j = dim_y - 1;
    const int ni = dim_x - 2, nj = dim_y - 2;   // interior extent
    // interior (operations.h:17-23), by anti-diagonals: cell (i, j) after (i - 1, j) and (i, j - 1), before
    // (i + 1, j) and (i, j + 1), as in the row-major loop
    for (int d = 0; d < ni + nj - 1; ++d) {
        const int lo = d - (ni - 1) > 0 ? d - (ni - 1) : 0, hi = d < nj - 1 ? d : nj - 1;
        for (int jj = lo + (int)threadIdx.x; jj <= hi; jj += (int)blockDim.x) {
            const int i = 1 + d - jj, j = 1 + jj, c = index(i, j, dim_x);
            field[c] = fast(field + c, i, j, dim_x, dim_y);
        }
        __syncthreads();   // (with its workgroup-scope fence: the next diagonal reads this one's cells)
    }
    if (threadIdx.x != 0) return;
    auto visit = [&](int i, int j) {
        const int c = index(i, j, dim_x);
        field[c] = safe(field + c, i, j, dim_x, dim_y);
    };
    for (int i = 0; i <= last_i; ++i) {   // operations.h:26-30
        visit(i, 0);
        visit(i, last_j);
    }
    for (int j = 1; j < last_j; ++j) {    // operations.h:33-37
        visit(0, j);
        visit(last_i, j);
    }
}

template <class T, class U, class Safe, class Fast>
inline hipError_t domain_for_each(Safe safe, Fast fast, U *wrt, const T *rd, int dim_x, int dim_y,
                                  hipStream_t stream = nullptr, sfl_in_place mode = sfl_in_place::reference_order)
{
    if (static_cast<const void *>(wrt) == static_cast<const void *>(rd) && mode == sfl_in_place::reference_order) {
        if constexpr (std::is_same<T, U>::value) {
            sfl_domain_for_each_in_order_kernel<T><<<1, 1024, 0, stream>>>(safe, fast, wrt, dim_x, dim_y);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;   // one array cannot hold two element types
        }
    }
    const dim3 grid((dim_x + 255) / 256, dim_y);
    sfl_domain_for_each_kernel<T, U><<<grid, 256, 0, stream>>>(safe, fast, wrt, rd, dim_x, dim_y);
    return hipGetLastError();
}

// one colour pass of poisson.cpp:14-61: the cells with (i + j) % 2 == odd
template <class T, class Safe, class Fast>
__global__ void sfl_domain_colour_pass_kernel(Safe safe, Fast fast, T *field, int dim_x, int dim_y, int odd)
{
    const int j = blockIdx.y;
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + ((j + odd) & 1);
    if (i >= dim_x) return;
    const int c = index(i, j, dim_x);
    const bool inner = i > 0 && i < dim_x - 1 && j > 0 && j < dim_y - 1;
    field[c] = inner ? fast(field + c, i, j, dim_x, dim_y) : safe(field + c, i, j, dim_x, dim_y);
}

template <class T, class Safe, class Fast>
inline hipError_t domain_for_each_red_black(Safe safe, Fast fast, T *field, int dim_x, int dim_y,
                                            hipStream_t stream = nullptr)
{
    const dim3 grid(((dim_x + 1) / 2 + 255) / 256, dim_y);
    for (int odd = 0; odd < 2; ++odd)   // even i + j first (poisson.cpp:22, :57-60)
        sfl_domain_colour_pass_kernel<T><<<grid, 256, 0, stream>>>(safe, fast, field, dim_x, dim_y, odd);
    return hipGetLastError();
}
#endif

#endif  // SFL_OPERATIONS_H
