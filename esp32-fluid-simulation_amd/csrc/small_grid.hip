// small_grid.hip -- the whole sim step, or the whole pressure solve, of a SMALL grid in one launch of
// one workgroup (gfx950 / MI355X).
//
// The sketch's own grid is 61 x 81 cells (ino:12-14): 20 KB of velocity + pressure + divergence.  With
// one kernel per operator and one per few colour passes a step of that size is six to ten dependent
// launches of ~5 us each and the GPU idles in between (profiles/r02_experiments_without_gain.txt: a
// hipGraph does not help, the latency is on the GPU side).  Here ONE workgroup of 1024 threads keeps
// the advected velocity, the divergence and the pressure of the whole grid in its CU's LDS (16 B per
// cell: up to 6144 cells = 96 KB) and runs advection -> forces -> divergence -> the red-black SOR
// iterations (a workgroup barrier between colour passes) -> projection -> dye advection back to back.
// The arithmetic is that of the one-thread-per-cell kernels (stencil_kernels.hip, advect_math.h),
// expression by expression: same bits.
//
// Numerics contract: -ffp-contract=off, every operation individually rounded in the
// reference's order.  Reference citations are file:line under /root/reference/ESP32-fluid-simulation/.
#include "advect_math.h"
#include "kernels.h"

namespace sfl {
namespace {

using namespace advect_math;

constexpr int kThreads = 1024;
// cells of one colour a thread may own (registers): a grid qualifies when its rows x ceil(dim_x / 2) positions of
// one colour fit (small_grid_fits; only widths of 3 .. 7 cells with thousands of rows do not)
constexpr int kCellsPerColour = kSmallGridMaxCells / 2 / kThreads;

struct Lds {
    float2 *v;   // advected (then projected) velocity
    float *d;    // divergence
    float *p;    // pressure
};

__device__ __forceinline__ Lds carve(char *base, int cells)
{
    Lds l;
    l.v = reinterpret_cast<float2 *>(base);
    l.d = reinterpret_cast<float *>(base + (size_t)cells * 8);
    l.p = l.d + cells;
    return l;
}

// iters x two colour passes of poisson.cpp:14-112 on p (zero-filled here, :117-119) in LDS.
// A thread owns the same cells in every pass: their pressure, dx * d, -1/n and the boundary facts stay in
// registers; LDS holds p for the neighbours.  A pass is branch-free: all neighbour reads of the thread's cells go
// out together, and both reference formulas are evaluated as (((z + W) + E) + S) + N -- an absent neighbour
// contributes -0.0f, the additive identity, z = -0.0f inside and +0.0f on the perimeter (the fused kernel's
// formulation, sor_stream_core.h): interior ((W + E) + S) + N  (pois_sor_fast, :107-109), perimeter the running
// sum from 0 over the neighbours present (pois_gs_safe, :67-89).
__device__ __forceinline__ void sor_in_lds(float *p, const float *d, int dim_x, int dim_y, int iters, SorParams prm)
{
    const int cells = dim_x * dim_y, half = (dim_x + 1) / 2;
    const int i_max = dim_x - 1, j_max = dim_y - 1;
    for (int c = threadIdx.x; c < cells; c += kThreads) p[c] = 0.0f;
    __syncthreads();   // (also: d is complete)
    const int per_colour = dim_y * half;                              // positions of one colour, row-major
    const int kmax = (per_colour + kThreads - 1) / kThreads;          // block-uniform, <= kCellsPerColour
    int cm[2][kCellsPerColour];      // cell index | neighbour mask << 16 (bit 0 W, 1 E, 2 S, 3 N present; bit 4: cell exists)
    float own[2][kCellsPerColour], rhs[2][kCellsPerColour], kf[2][kCellsPerColour], z[2][kCellsPerColour];
#pragma unroll
    for (int colour = 0; colour < 2; ++colour)
#pragma unroll
        for (int k = 0; k < kCellsPerColour; ++k) {
            const int q = threadIdx.x + k * kThreads;
            const int gj = q / half, ii = q - gj * half;
            const int i = 2 * ii + ((gj + colour) & 1);
            const bool have = k < kmax && gj < dim_y && i < dim_x;
            const int c = have ? gj * dim_x + i : 0;
            const int m = (i > 0 ? 1 : 0) | (i < i_max ? 2 : 0) | (gj > 0 ? 4 : 0) | (gj < j_max ? 8 : 0);
            const int present = __builtin_popcount(m);
            cm[colour][k] = have ? (c | ((m | 16) << 16)) : 0;
            own[colour][k] = 0.0f;
            rhs[colour][k] = have ? prm.dx * d[c] : 0.0f;   // dx * d, :108 / :88 (the same product every pass)
            kf[colour][k] = (present == 2) ? (float)(-1.0 / 2.0) : (present == 3) ? (float)(-1.0 / 3.0) : -0.25f;  // :67
            z[colour][k] = (present == 4) ? -0.0f : 0.0f;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int colour = 0; colour < 2; ++colour) {  // colour 0 = even (i + j) first, poisson.cpp:22,57-60
            float w[kCellsPerColour], e[kCellsPerColour], s[kCellsPerColour], n[kCellsPerColour];
#pragma unroll
            for (int k = 0; k < kCellsPerColour; ++k) {
                if (k >= kmax) break;
                const int c = cm[colour][k] & 0xffff, m = cm[colour][k] >> 16;
                w[k] = (m & 1) ? p[c - 1] : -0.0f;
                e[k] = (m & 2) ? p[c + 1] : -0.0f;
                s[k] = (m & 4) ? p[c - dim_x] : -0.0f;
                n[k] = (m & 8) ? p[c + dim_x] : -0.0f;
            }
#pragma unroll
            for (int k = 0; k < kCellsPerColour; ++k) {
                if (k >= kmax) break;
                const float sum = (((z[colour][k] + w[k]) + e[k]) + s[k]) + n[k];
                const float p_gs = kf[colour][k] * (rhs[colour][k] - sum);
                const float fresh = prm.one_minus_omega * own[colour][k] + prm.omega * p_gs;  // :98, :111
                own[colour][k] = fresh;
                if (cm[colour][k] >> 20) p[cm[colour][k] & 0xffff] = fresh;
            }
            __syncthreads();
        }
    }
}

// ---- poisson_solve (poisson.cpp:114-125) alone ---------------------------------------------------
__global__ void __launch_bounds__(kThreads)
small_solve_kernel(float *__restrict__ p_out, const float *__restrict__ d_in, int dim_x, int dim_y, int iters,
                   SorParams prm)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int cells = dim_x * dim_y;
    const Lds l = carve(lds_raw, cells);
    for (int c = threadIdx.x; c < cells; c += kThreads) l.d[c] = d_in[c];
    sor_in_lds(l.p, l.d, dim_x, dim_y, iters, prm);   // (its first barrier also covers the copy of d)
    for (int c = threadIdx.x; c < cells; c += kThreads) p_out[c] = l.p[c];
}

// ---- one whole step, ino:252-287 -------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
small_step_kernel(SmallStep a)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int dim_x = a.dim_x, dim_y = a.dim_y, cells = dim_x * dim_y;
    const Lds l = carve(lds_raw, cells);
    const Slab g{dim_x, dim_y, 0, dim_y};
    const float2 *v_in = reinterpret_cast<const float2 *>(a.v_in);
    float2 *v_out = reinterpret_cast<float2 *>(a.v_out);

    // advect(v_next, v, v, dt, no_slip): ino:252-256, advect.h:78-84
    for (int c = threadIdx.x; c < cells; c += kThreads) {
        const int gj = c / dim_x, i = c - gj * dim_x;
        const float2 u = v_in[c];
        const float si = (float)i - u.x * a.dt;
        const float sj = (float)gj - u.y * a.dt;
        const SrcPos s = classify(si, sj, dim_x, dim_y);
        l.v[c] = sample_global_vec2f<true>(v_in, g, s, si, sj);
    }
    __syncthreads();
    // drag forces, in queue order: later entries win (ino:264-269)
    if (a.n_forces > 0) {
        if (threadIdx.x == 0)
            for (int k = 0; k < a.n_forces; ++k) {
                const int i = a.force_cells[2 * k], gj = a.force_cells[2 * k + 1];
                if (i < 0 || i >= dim_x || gj < 0 || gj >= dim_y) continue;
                l.v[gj * dim_x + i] = make_float2(a.force_vel[2 * k], a.force_vel[2 * k + 1]);
            }
        __syncthreads();
    }
    // calculate_divergence: ino:274, finitediff.cpp:9-39
    const int i_max = dim_x - 1, j_max = dim_y - 1;
    for (int c = threadIdx.x; c < cells; c += kThreads) {
        const int gj = c / dim_x, i = c - gj * dim_x;
        const float2 *q = l.v + c;
        float s;
        if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // div_expr_fast, :29
            const float hx = -q[-1].x + q[1].x;
            const float hy = -q[-dim_x].y + q[dim_x].y;
            s = hx + hy;
        } else {  // div_expr_safe, :15-20: ghost velocity = -own
            const float2 own = q[0];
            s = 0.0f;
            s += (i > 0) ? -q[-1].x : own.x;
            s += (i < i_max) ? q[1].x : -own.x;
            s += (gj > 0) ? -q[-dim_x].y : own.y;
            s += (gj < j_max) ? q[dim_x].y : -own.y;
        }
        const float dv = s * a.two_dx_inv;
        l.d[c] = dv;
        a.div[c] = dv;
    }
    // poisson_solve: ino:275 (the barrier inside also orders the divergence writes above)
    sor_in_lds(l.p, l.d, dim_x, dim_y, a.iters, a.prm);
    // subtract_gradient (ino:276, finitediff.cpp:41-82), then the dye back-trace with the projected velocity of
    // the cell itself (ino:281-287, advect.h:81) -- per cell, no barrier needed in between
    const uint32_t *col_in = a.col_in;
    for (int c = threadIdx.x; c < cells; c += kThreads) {
        const int gj = c / dim_x, i = c - gj * dim_x;
        const float pc = l.p[c];
        const float pw = (i > 0) ? l.p[c - 1] : pc;
        const float pe = (i < i_max) ? l.p[c + 1] : pc;
        const float ps = (gj > 0) ? l.p[c - dim_x] : pc;
        const float pn = (gj < j_max) ? l.p[c + dim_x] : pc;
        const float gx = (pe - pw) * a.two_dx_inv;
        const float gy = (pn - ps) * a.two_dx_inv;
        float2 u = l.v[c];
        u.x = u.x - gx;
        u.y = u.y - gy;
        v_out[c] = u;
        a.p[c] = pc;
        const float si = (float)i - u.x * a.dt;
        const float sj = (float)gj - u.y * a.dt;
        const SrcPos s = classify(si, sj, dim_x, dim_y);
        const uq3 r = sample_global_uq3<false>(col_in, g, s, si, sj);
        uint32_t *o = a.col_out + 3 * (size_t)c;
        o[0] = r.x;
        o[1] = r.y;
        o[2] = r.z;
    }
}

// more than 64 KB of dynamic LDS has to be granted once per kernel and device
hipError_t allow_lds(const void *kernel, bool *granted)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (granted[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmallGridMaxCells * 16);
    if (e == hipSuccess) granted[dev] = true;
    return e;
}

}  // namespace

bool small_grid_fits(int dim_x, int dim_y)
{
    return (int64_t)dim_x * dim_y <= kSmallGridMaxCells && (int64_t)dim_y * ((dim_x + 1) / 2) <= kCellsPerColour * kThreads;
}

hipError_t launch_small_solve(hipStream_t s, float *p, const float *d, int dim_x, int dim_y, int iters, SorParams prm)
{
    static bool granted[64];
    const size_t lds = (size_t)dim_x * dim_y * 16;
    hipError_t e = allow_lds(reinterpret_cast<const void *>(small_solve_kernel), granted);
    if (e != hipSuccess) return e;
    small_solve_kernel<<<1, kThreads, lds, s>>>(p, d, dim_x, dim_y, iters, prm);
    return hipGetLastError();
}

hipError_t launch_small_step(hipStream_t s, const SmallStep &a)
{
    static bool granted[64];
    const size_t lds = (size_t)a.dim_x * a.dim_y * 16;
    hipError_t e = allow_lds(reinterpret_cast<const void *>(small_step_kernel), granted);
    if (e != hipSuccess) return e;
    small_step_kernel<<<1, kThreads, lds, s>>>(a);
    return hipGetLastError();
}

}  // namespace sfl
