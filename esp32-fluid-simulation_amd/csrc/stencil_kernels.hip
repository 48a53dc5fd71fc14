// stencil_kernels.hip -- advection, divergence, pressure-gradient, baseline SOR colour pass,
// zero fill and force injection for gfx950 (MI355X).
//
// Numerics contract: this file is compiled with -ffp-contract=off; every
// product and sum below is individually rounded in the order the reference evaluates it, so
// results are bit-identical to the reference built without contraction.  Do not "simplify"
// expressions such as `0.0f + w` (it maps -0 to +0) or reassociate sums.
//
// Reference citations are file:line under /root/reference/ESP32-fluid-simulation/.
#include "advect_math.h"
#include "kernels.h"

namespace sfl {
namespace {

using namespace advect_math;

constexpr int kBlock = 256;
// advection blocks are 64 x 4 cells (one wave per row segment, four rows per block): a sample reads
// rows cj and cj + 1, so vertically adjacent waves share cache lines through the CU's L1 -- 20 %
// faster on incoherent velocity fields, neutral on smooth ones (profiles/r01_advect_coherence_probe.txt)
constexpr int kAdvTileX = 64, kAdvTileY = 4;

// ---- advect<Vector2<float>, float>  (advect.h:24-85) ---------------------------------
template <bool NO_SLIP>
__global__ void __launch_bounds__(kBlock)
advect_vec2f_kernel(float2 *__restrict__ next_p, const float2 *p, const float2 *vel, Slab g, Slab gs,
                    int g_begin, int g_end, int valid_begin, int valid_end, float dt, int *halo_flag)
{   // g: geometry of next_p and vel (the slab); gs: geometry of the advected field p (the slab's own
    // array, or the gathered whole domain when the back-trace outruns the ghost rows)
    const int i = blockIdx.x * kAdvTileX + threadIdx.x;
    const int gj = g_begin + blockIdx.y * kAdvTileY + threadIdx.y;
    if (gj >= g_end) return;
    if (i >= g.dim_x) return;
    const size_t c = lcell(g, i, gj);
    const float2 u = vel[c];
    const float si = (float)i - u.x * dt;  // advect.h:81
    const float sj = (float)gj - u.y * dt;
    const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
    if (!rows_available(s, valid_begin, valid_end)) {
        if (halo_flag) atomicOr(halo_flag, 1);
        return;
    }
    const float2 r = sample_global_vec2f<NO_SLIP>(p, gs, s, si, sj);
    next_p[c] = r;
}

// ---- advect<Vector3<UQ32>, float>  (advect.h:24-85 + uq32.h) ---------------------------
// FUSE_GRAD: the projection step subtract_gradient (finitediff.cpp:41-82) is applied to the
// cell's own velocity first -- the dye back-trace reads ONLY vel[ij] (advect.h:81), so the
// projected velocity can be produced here, written back in place and used at once; this saves
// the separate pass over v of ino:276 followed by ino:282 (same arithmetic, same results).
template <bool NO_SLIP, bool FUSE_GRAD>
__global__ void __launch_bounds__(kBlock)
advect_vec3uq32_kernel(uint32_t *__restrict__ next_p, const uint32_t *p, float2 *vel, Slab g, Slab gs,
                       int g_begin, int g_end, int valid_begin, int valid_end, float dt, int *halo_flag,
                       const float *__restrict__ pressure, float two_dx_inv)
{
    const int i = blockIdx.x * kAdvTileX + threadIdx.x;
    const int gj = g_begin + blockIdx.y * kAdvTileY + threadIdx.y;
    if (gj >= g_end) return;
    if (i >= g.dim_x) return;
    const size_t c = lcell(g, i, gj);
    float2 u = vel[c];
    if (FUSE_GRAD) {
        const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
        const float pc = pressure[c];
        const float pw = (i > 0) ? pressure[c - 1] : pc;
        const float pe = (i < i_max) ? pressure[c + 1] : pc;
        const float ps = (gj > 0) ? pressure[c - g.dim_x] : pc;
        const float pn = (gj < j_max) ? pressure[c + g.dim_x] : pc;
        const float gx = (pe - pw) * two_dx_inv;
        const float gy = (pn - ps) * two_dx_inv;
        u.x = u.x - gx;
        u.y = u.y - gy;
        vel[c] = u;
    }
    const float si = (float)i - u.x * dt;
    const float sj = (float)gj - u.y * dt;
    const SrcPos s = classify(si, sj, g.dim_x, g.gdim_y);
    if (!rows_available(s, valid_begin, valid_end)) {
        if (halo_flag) atomicOr(halo_flag, 1);
        return;
    }
    const uq3 r = sample_global_uq3<NO_SLIP>(p, gs, s, si, sj);
    uint32_t *o = next_p + 3 * c;
    o[0] = r.x;
    o[1] = r.y;
    o[2] = r.z;
}

// ---- how far do the back-traces of a slab reach beyond its owned rows? ------------------------
// reach[0] = max over owned cells of (g_begin - first source row), reach[1] = max of (last source
// row - (g_end - 1)), both >= 0: the halo rows an advection of this slab needs below / above
// (exactly the rows rows_available() will ask for).  One atomicMax per wave.
__global__ void __launch_bounds__(kBlock)
backtrace_reach_kernel(int *reach, const float2 *__restrict__ vel, Slab g, int g_begin, int g_end, float dt)
{
    const int i = blockIdx.x * kAdvTileX + threadIdx.x;
    const int gj = g_begin + blockIdx.y * kAdvTileY + threadIdx.y;
    int below = 0, above = 0;
    if (gj < g_end && i < g.dim_x) {
        const float2 u = vel[lcell(g, i, gj)];
        const SrcPos s = classify((float)i - u.x * dt, (float)gj - u.y * dt, g.dim_x, g.gdim_y);
        below = max(g_begin - s.cj, 0);
        above = max(s.cj + (s.y_oob ? 0 : 1) - (g_end - 1), 0);
    }
    for (int off = 32; off > 0; off >>= 1) {
        below = max(below, __shfl_xor(below, off));
        above = max(above, __shfl_xor(above, off));
    }
    if ((threadIdx.x & 63) == 0) {
        if (below > 0) atomicMax(reach, below);
        if (above > 0) atomicMax(reach + 1, above);
    }
}

// ---- calculate_divergence (finitediff.cpp:9-39) -------------------------------------------
__global__ void __launch_bounds__(kBlock)
divergence_kernel(float *__restrict__ div, const float2 *__restrict__ v, Slab g, int g_begin,
                  float two_dx_inv)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int gj = g_begin + blockIdx.y;
    if (i >= g.dim_x) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    const size_t c = lcell(g, i, gj);
    float s;
    if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // div_expr_fast, :29
        const float hx = -v[c - 1].x + v[c + 1].x;
        const float hy = -v[c - g.dim_x].y + v[c + g.dim_x].y;
        s = hx + hy;
    } else {  // div_expr_safe, :15-20: ghost velocity = -own
        const float2 own = v[c];
        s = 0.0f;
        s += (i > 0) ? -v[c - 1].x : own.x;
        s += (i < i_max) ? v[c + 1].x : -own.x;
        s += (gj > 0) ? -v[c - g.dim_x].y : own.y;
        s += (gj < j_max) ? v[c + g.dim_x].y : -own.y;
    }
    div[c] = s * two_dx_inv;
}

// ---- subtract_gradient (finitediff.cpp:41-82), in place on v ------------------------------
__global__ void __launch_bounds__(kBlock)
subtract_gradient_kernel(float2 *v, const float *__restrict__ p, Slab g, int g_begin,
                         float two_dx_inv)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int gj = g_begin + blockIdx.y;
    if (i >= g.dim_x) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    const size_t c = lcell(g, i, gj);
    const float pc = p[c];
    const float pw = (i > 0) ? p[c - 1] : pc;
    const float pe = (i < i_max) ? p[c + 1] : pc;
    const float ps = (gj > 0) ? p[c - g.dim_x] : pc;
    const float pn = (gj < j_max) ? p[c + g.dim_x] : pc;
    const float gx = (pe - pw) * two_dx_inv;
    const float gy = (pn - ps) * two_dx_inv;
    float2 u = v[c];
    u.x = u.x - gx;
    u.y = u.y - gy;
    v[c] = u;
}

// ---- baseline SOR colour pass (poisson.cpp:14-112), in place --------------------------------
__global__ void __launch_bounds__(kBlock)
sor_half_sweep_kernel(float *p, const float *__restrict__ d, Slab g, int g_begin, int colour,
                      SorParams prm)
{
    const int gj = g_begin + blockIdx.y;
    const int i = 2 * (blockIdx.x * kBlock + threadIdx.x) + ((gj + colour) & 1);
    if (i >= g.dim_x) return;
    const int i_max = g.dim_x - 1, j_max = g.gdim_y - 1;
    const size_t c = lcell(g, i, gj);
    float p_gs;
    if (i > 0 && i < i_max && gj > 0 && gj < j_max) {  // pois_sor_fast, :107-109
        const float sum = p[c - 1] + p[c + 1] + p[c - g.dim_x] + p[c + g.dim_x];
        p_gs = -0.25f * (prm.dx * d[c] - sum);
    } else {  // pois_gs_safe, :67-89
        float sum = 0.0f;
        int n = 0;
        if (i > 0) { sum += p[c - 1]; ++n; }
        if (i < i_max) { sum += p[c + 1]; ++n; }
        if (gj > 0) { sum += p[c - g.dim_x]; ++n; }
        if (gj < j_max) { sum += p[c + g.dim_x]; ++n; }
        const float k = (n == 2) ? (float)(-1.0 / 2.0) : (n == 3) ? (float)(-1.0 / 3.0) : -0.25f;
        p_gs = k * (prm.dx * d[c] - sum);
    }
    p[c] = prm.one_minus_omega * p[c] + prm.omega * p_gs;  // :98, :111
}

// ---- dye visualiser (draw task, ino:116-176) ---------------------------------
// One thread per output pixel.  Pixel (sy, sx) belongs to cell block (i, j) = (sy / S, sx / S)
// at offset (ii, jj); it replays the sketch's strength-reduced lerps for its own offsets only:
// left / right edge values after ii increments (:134-153), then jj increments across (:156-161),
// narrowing to UQ32 (:168), RGB565 pack (:170-172), optional byte swap (:173).
__global__ void __launch_bounds__(kBlock)
render_rgb565_kernel(uint16_t *__restrict__ image, const uint32_t *__restrict__ colour, int dim_x,
                     int dim_y, int scaling, int byteswap)
{
    const int width = scaling * (dim_y - 1);
    const int sx = blockIdx.x * kBlock + threadIdx.x;
    const int sy = blockIdx.y;
    if (sx >= width) return;
    const int i = sy / scaling, ii = sy - i * scaling;
    const int j = sx / scaling, jj = sx - j * scaling;
    const float inv = 1.0f / (float)scaling;
    const uint32_t *t1 = colour + 3 * ((size_t)dim_x * j + i);
    const uint32_t *t2 = colour + 3 * ((size_t)dim_x * (j + 1) + i);
    uint32_t raw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = uq_widen(t1[k]), r = uq_widen(t2[k]);
        const float dl = (uq_widen(t1[3 + k]) - l) * inv;
        const float dr = (uq_widen(t2[3 + k]) - r) * inv;
        for (int n = 0; n < ii; ++n) {
            l += dl;
            r += dr;
        }
        float c = l;
        const float dc = (r - l) * inv;
        for (int n = 0; n < jj; ++n) c += dc;
        raw[k] = uq_narrow(c);
    }
    uint16_t px = (uint16_t)(((raw[0] & 0xF8000000u) >> 16) | ((raw[1] & 0xFC000000u) >> 21) |
                             ((raw[2] & 0xF8000000u) >> 27));
    if (byteswap) px = (uint16_t)((px >> 8) | (px << 8));
    image[(size_t)sy * width + sx] = px;
}

// ---- initial condition of the sketch's setup() (ino:196-241) ------------------
// float -> UQ32 with SATURATION for values >= 2^32: the sketch converts UINT32_MAX (2^32 as a
// float) and blurred sums that reach 2^32, which is undefined behaviour in C++; the ESP32's and
// this GPU's conversion instructions saturate, and so does the oracle's restatement.
__device__ __forceinline__ uint32_t uq_narrow_sat(float x)
{
    const float y = x + 0.5f;
    return y >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)y;
}

// three 120-degree dye sectors (ino:204-218) + zero velocity (ino:197-201), one thread per cell
__global__ void __launch_bounds__(kBlock)
setup_sectors_kernel(float2 *__restrict__ v, uint32_t *__restrict__ colour, int dim_x, int dim_y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x, j = blockIdx.y;
    if (i >= dim_x) return;
    const size_t c = (size_t)dim_x * j + i;
    v[c] = make_float2(0.0f, 0.0f);
    const double third_pi = 3.1415926535897932384626433832795 / 3;  // Arduino's PI / 3
    const float angle = atan2f((float)(-(i - dim_x / 2)), (float)(j - dim_y / 2));
    const int sector = ((double)angle < -third_pi) ? 0 : ((double)angle < third_pi) ? 1 : 2;
    for (int k = 0; k < 3; ++k)
        colour[3 * c + k] = uq_narrow_sat(k == sector ? (float)4294967295u : 0.0f);
}

// The sketch's blurs run IN PLACE and sequentially (ino:219-241): along the blurred axis the
// previous cell is already blurred, the next one is not -- a recurrence.  Lines are independent:
// one thread per line.  ALONG_J: line = fixed i, walks j (stride dim_x); else fixed j, walks i.
template <bool ALONG_J>
__global__ void __launch_bounds__(kBlock)
setup_blur_kernel(uint32_t *colour, int dim_x, int dim_y)
{
    const int line = blockIdx.x * kBlock + threadIdx.x;
    const int n_lines = ALONG_J ? dim_x : dim_y, len = ALONG_J ? dim_y : dim_x;
    if (line >= n_lines) return;
    const size_t stride = ALONG_J ? (size_t)dim_x * 3 : 3;
    uint32_t *c = colour + (ALONG_J ? (size_t)line * 3 : (size_t)line * dim_x * 3);
    for (int t = 0; t < len; ++t, c += stride) {
        const uint32_t *prev = (t == 0) ? c : c - stride;
        const uint32_t *next = (t == len - 1) ? c : c + stride;
        for (int k = 0; k < 3; ++k) {
            const float s = (0.25f * uq_widen(prev[k]) + 0.5f * uq_widen(c[k])) + 0.25f * uq_widen(next[k]);
            c[k] = uq_narrow_sat(s);
        }
    }
}

__global__ void __launch_bounds__(kBlock)
zero_rows_kernel(float *f, size_t first, size_t count)
{
    size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (; k < count; k += stride) f[first + k] = 0.0f;
}

// Sequential on purpose: later entries overwrite earlier ones like the sketch's queue drain
// (ino:264-269); n is a handful of touch events.
__global__ void apply_forces_kernel(float2 *v, Slab g, int g_begin, int g_end,
                                    const int *cells_ij, const float *vel_xy, int n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int k = 0; k < n; ++k) {
        const int i = cells_ij[2 * k], gj = cells_ij[2 * k + 1];
        if (i < 0 || i >= g.dim_x || gj < g_begin || gj >= g_end) continue;
        v[lcell(g, i, gj)] = make_float2(vel_xy[2 * k], vel_xy[2 * k + 1]);
    }
}

inline dim3 grid_cells(int cells_per_row, int rows) { return dim3((cells_per_row + kBlock - 1) / kBlock, rows, 1); }
#define SFL_ADV_GRID(dim_x, rows)                                                         \
    const dim3 ablock(kAdvTileX, kAdvTileY, 1);                                           \
    const dim3 agrid(((dim_x) + kAdvTileX - 1) / kAdvTileX, ((rows) + kAdvTileY - 1) / kAdvTileY, 1)

}  // namespace

// the tiled kernels (advect_tiled.hip) or the one-thread-per-cell ones of this file?
static bool use_tiled_advect(int kernel, const Slab &g, int g_begin, int g_end)
{
    return kernel == 2 || (kernel == 0 && (int64_t)g.dim_x * (g_end - g_begin) >= kAdvectTiledMinCells);
}

hipError_t launch_backtrace_reach(hipStream_t s, int *reach, const float *vel, Slab g, int g_begin, int g_end,
                                  float dt)
{
    if (g_end <= g_begin) return hipSuccess;
    SFL_ADV_GRID(g.dim_x, g_end - g_begin);
    backtrace_reach_kernel<<<agrid, ablock, 0, s>>>(reach, reinterpret_cast<const float2 *>(vel), g, g_begin,
                                                    g_end, dt);
    return hipGetLastError();
}

hipError_t launch_advect_vec2f(hipStream_t s, float *next_p, const float *p, const float *vel,
                               Slab g, int g_begin, int g_end, int valid_begin, int valid_end,
                               float dt, bool no_slip, int *halo_flag, const Slab *src, int kernel, int g2_begin, int g2_end)
{
    if (g_end <= g_begin) return hipSuccess;
    if (use_tiled_advect(kernel, g, g_begin, g_end))
        return launch_advect_vec2f_tiled(s, next_p, p, vel, g, g_begin, g_end, valid_begin, valid_end, dt,
                                         no_slip, halo_flag, src, g2_begin, g2_end);
    if (g2_end > g2_begin) {   // the one-thread-per-cell kernel: one launch per range
        const hipError_t e = launch_advect_vec2f(s, next_p, p, vel, g, g2_begin, g2_end, valid_begin, valid_end, dt, no_slip,
                                                 halo_flag, src, kernel);
        if (e != hipSuccess) return e;
    }
    const Slab gs = src ? *src : g;
    SFL_ADV_GRID(g.dim_x, g_end - g_begin);
    auto *o = reinterpret_cast<float2 *>(next_p);
    auto *pi = reinterpret_cast<const float2 *>(p);
    auto *vi = reinterpret_cast<const float2 *>(vel);
    if (no_slip)
        advect_vec2f_kernel<true><<<agrid, ablock, 0, s>>>(o, pi, vi, g, gs, g_begin, g_end, valid_begin,
                                                          valid_end, dt, halo_flag);
    else
        advect_vec2f_kernel<false><<<agrid, ablock, 0, s>>>(o, pi, vi, g, gs, g_begin, g_end, valid_begin,
                                                           valid_end, dt, halo_flag);
    return hipGetLastError();
}

hipError_t launch_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p,
                                  const float *vel, Slab g, int g_begin, int g_end,
                                  int valid_begin, int valid_end, float dt, bool no_slip,
                                  int *halo_flag, const Slab *src, int kernel)
{
    if (g_end <= g_begin) return hipSuccess;
    if (use_tiled_advect(kernel, g, g_begin, g_end))
        return launch_advect_vec3uq32_tiled(s, next_p, p, const_cast<float *>(vel), nullptr, g, g_begin, g_end,
                                            valid_begin, valid_end, dt, no_slip, halo_flag, 0.0f, src);
    const Slab gs = src ? *src : g;
    SFL_ADV_GRID(g.dim_x, g_end - g_begin);
    auto *vi = reinterpret_cast<float2 *>(const_cast<float *>(vel));  // read-only without FUSE_GRAD
    if (no_slip)
        advect_vec3uq32_kernel<true, false><<<agrid, ablock, 0, s>>>(
            next_p, p, vi, g, gs, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, nullptr, 0.0f);
    else
        advect_vec3uq32_kernel<false, false><<<agrid, ablock, 0, s>>>(
            next_p, p, vi, g, gs, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, nullptr, 0.0f);
    return hipGetLastError();
}

hipError_t launch_project_advect_vec3uq32(hipStream_t s, uint32_t *next_p, const uint32_t *p,
                                          float *vel, const float *pressure, Slab g, int g_begin,
                                          int g_end, int valid_begin, int valid_end, float dt,
                                          bool no_slip, int *halo_flag, float two_dx_inv, int kernel, bool *reach_measured)
{
    if (reach_measured) *reach_measured = false;
    if (g_end <= g_begin) return hipSuccess;
    if (use_tiled_advect(kernel, g, g_begin, g_end)) {
        const bool reach = reach_measured != nullptr && halo_flag != nullptr;
        if (reach) *reach_measured = true;
        return launch_advect_vec3uq32_tiled(s, next_p, p, vel, pressure, g, g_begin, g_end, valid_begin,
                                            valid_end, dt, no_slip, halo_flag, two_dx_inv, nullptr, reach);
    }
    SFL_ADV_GRID(g.dim_x, g_end - g_begin);
    auto *vi = reinterpret_cast<float2 *>(vel);
    if (no_slip)
        advect_vec3uq32_kernel<true, true><<<agrid, ablock, 0, s>>>(
            next_p, p, vi, g, g, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, pressure, two_dx_inv);
    else
        advect_vec3uq32_kernel<false, true><<<agrid, ablock, 0, s>>>(
            next_p, p, vi, g, g, g_begin, g_end, valid_begin, valid_end, dt, halo_flag, pressure, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_divergence(hipStream_t s, float *div, const float *v, Slab g, int g_begin,
                             int g_end, float two_dx_inv, int kernel)
{
    if (g_end <= g_begin) return hipSuccess;
    if (use_tiled_advect(kernel, g, g_begin, g_end))
        return launch_divergence_tiled(s, div, v, g, g_begin, g_end, two_dx_inv);
    divergence_kernel<<<grid_cells(g.dim_x, g_end - g_begin), kBlock, 0, s>>>(
        div, reinterpret_cast<const float2 *>(v), g, g_begin, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_subtract_gradient(hipStream_t s, float *v, const float *p, Slab g, int g_begin,
                                    int g_end, float two_dx_inv, int kernel)
{
    if (g_end <= g_begin) return hipSuccess;
    if (use_tiled_advect(kernel, g, g_begin, g_end))
        return launch_gradient_tiled(s, v, p, g, g_begin, g_end, two_dx_inv);
    subtract_gradient_kernel<<<grid_cells(g.dim_x, g_end - g_begin), kBlock, 0, s>>>(
        reinterpret_cast<float2 *>(v), p, g, g_begin, two_dx_inv);
    return hipGetLastError();
}

hipError_t launch_sor_half_sweep(hipStream_t s, float *p, const float *d, Slab g, int g_begin,
                                 int g_end, int colour, SorParams prm)
{
    if (g_end <= g_begin) return hipSuccess;
    sor_half_sweep_kernel<<<grid_cells((g.dim_x + 1) / 2, g_end - g_begin), kBlock, 0, s>>>(
        p, d, g, g_begin, colour, prm);
    return hipGetLastError();
}

hipError_t launch_zero_rows(hipStream_t s, float *f, Slab g, int g_begin, int g_end)
{
    if (g_end <= g_begin) return hipSuccess;
    const size_t first = (size_t)(g_begin - g.grow0) * g.dim_x;
    const size_t count = (size_t)(g_end - g_begin) * g.dim_x;
    const size_t want = (count + kBlock - 1) / kBlock;
    const int blocks = (int)(want < 4096 ? want : 4096);
    zero_rows_kernel<<<blocks, kBlock, 0, s>>>(f, first, count);
    return hipGetLastError();
}

hipError_t launch_setup_sketch_fields(hipStream_t s, float *v, uint32_t *colour, int dim_x, int dim_y)
{
    setup_sectors_kernel<<<dim3((dim_x + kBlock - 1) / kBlock, dim_y), kBlock, 0, s>>>(
        reinterpret_cast<float2 *>(v), colour, dim_x, dim_y);
    setup_blur_kernel<true><<<(dim_x + kBlock - 1) / kBlock, kBlock, 0, s>>>(colour, dim_x, dim_y);
    setup_blur_kernel<false><<<(dim_y + kBlock - 1) / kBlock, kBlock, 0, s>>>(colour, dim_x, dim_y);
    return hipGetLastError();
}

hipError_t launch_render_rgb565(hipStream_t s, uint16_t *image, const uint32_t *colour, int dim_x,
                                int dim_y, int scaling, bool byteswap)
{
    const int width = scaling * (dim_y - 1), height = scaling * (dim_x - 1);
    if (width <= 0 || height <= 0) return hipSuccess;
    render_rgb565_kernel<<<dim3((width + kBlock - 1) / kBlock, height), kBlock, 0, s>>>(
        image, colour, dim_x, dim_y, scaling, byteswap ? 1 : 0);
    return hipGetLastError();
}

// One wave that does nothing for `us` microseconds of the constant 100 MHz real-time clock: the wire delay of an
// EMULATED halo message (sfl_comm_emulate + SFL_OPT_EMULATE_WIRE_US; measurement aid, never on a product path).
__global__ void spin_us_kernel(int us)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ticks = (unsigned long long)us * 100ull;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

hipError_t launch_spin_us(hipStream_t s, int us)
{
    if (us <= 0) return hipSuccess;
    spin_us_kernel<<<1, 64, 0, s>>>(us);
    return hipGetLastError();
}

// dst_a <- src_a, dst_b <- src_b (see kernels.h): 16-byte accesses when everything is 16-byte aligned, words otherwise
template <class W>
__global__ void __launch_bounds__(kBlock)
copy_bands_kernel(W *dst_a, const W *src_a, W *dst_b, const W *src_b, size_t words)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < words; k += stride) {
        if (dst_a) dst_a[k] = src_a[k];
        if (dst_b) dst_b[k] = src_b[k];
    }
}

hipError_t launch_copy_bands(hipStream_t s, void *dst_a, const void *src_a, void *dst_b, const void *src_b, size_t bytes)
{
    if (bytes == 0 || (!dst_a && !dst_b)) return hipSuccess;
    const uintptr_t all = reinterpret_cast<uintptr_t>(dst_a) | reinterpret_cast<uintptr_t>(src_a) |
                          reinterpret_cast<uintptr_t>(dst_b) | reinterpret_cast<uintptr_t>(src_b) | bytes;
    if ((all & 15) == 0) {
        const size_t words = bytes / 16, want = (words + kBlock - 1) / kBlock;
        copy_bands_kernel<uint4><<<(int)(want < 2048 ? want : 2048), kBlock, 0, s>>>(
            static_cast<uint4 *>(dst_a), static_cast<const uint4 *>(src_a), static_cast<uint4 *>(dst_b),
            static_cast<const uint4 *>(src_b), words);
    } else {   // (every field is made of 4-byte words)
        const size_t words = bytes / 4, want = (words + kBlock - 1) / kBlock;
        copy_bands_kernel<uint32_t><<<(int)(want < 2048 ? want : 2048), kBlock, 0, s>>>(
            static_cast<uint32_t *>(dst_a), static_cast<const uint32_t *>(src_a), static_cast<uint32_t *>(dst_b),
            static_cast<const uint32_t *>(src_b), words);
    }
    return hipGetLastError();
}

// The word cut-adjacent tiles poll inside a launch (kernels.h HaloWait).  This kernel starts when everything queued
// before it on its stream -- the halo message, the kernels that relaxed the arrived rows -- has completed and been
// written back; a relaxed agent-scope store is then all the flag needs.
__global__ void signal_arrival_kernel(int *flag, int value)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_signal_arrival(hipStream_t s, int *flag, int value)
{
    signal_arrival_kernel<<<1, 64, 0, s>>>(flag, value);
    return hipGetLastError();
}

__global__ void wait_count_kernel(const int *count, int target, int *timed_out, int timeout_us)
{
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
    while ((int)((unsigned)__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)target) < 0) {
        __builtin_amdgcn_s_sleep(20);
        if (__builtin_amdgcn_s_memrealtime() - t_begin > 100ull * (unsigned long long)timeout_us) {
            if (threadIdx.x == 0) atomicOr(timed_out, 8);
            break;
        }
    }
}

hipError_t launch_wait_count(hipStream_t s, const int *count, int target, int *timed_out, int timeout_us)
{
    wait_count_kernel<<<1, 64, 0, s>>>(count, target, timed_out, timeout_us > 0 ? timeout_us : kHaloWaitDefaultTimeoutUs);
    return hipGetLastError();
}

hipError_t launch_apply_forces(hipStream_t s, float *v, Slab g, int g_begin, int g_end,
                               const int *cells_ij, const float *vel_xy, int n)
{
    if (n <= 0 || g_end <= g_begin) return hipSuccess;
    apply_forces_kernel<<<1, 64, 0, s>>>(reinterpret_cast<float2 *>(v), g, g_begin, g_end,
                                         cells_ij, vel_xy, n);
    return hipGetLastError();
}

}  // namespace sfl
