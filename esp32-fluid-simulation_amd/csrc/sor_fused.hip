// sor_fused.hip -- gfx950 backend + launcher of the fused red-black SOR pipeline
// (sor_stream_core.h).  One 64-lane wavefront streams one tile; lanes exchange their W / E
// neighbours with DPP wave shifts (folded into v_add_f32_dpp), S / N neighbours stay in
// VGPRs, the right-hand side waits in a per-lane LDS ring.  No barriers, no atomics, no MFMA:
// this is a bandwidth / VALU-issue bound stencil.
//
// Compiled with -ffp-contract=off (bit-exactness contract, see stencil_kernels.hip).
#include "kernels.h"
#include "sor_stream_core.h"

namespace sfl {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = 64 * kWavesPerBlock;

// DPP full-wave shifts (GFX9 wave_shr:1 / wave_shl:1).  Lane 0 / lane 63 receive 0, which only
// ever feeds cells of the tile's invalid rim.
__device__ __forceinline__ float lane_below(float x)  // value of lane - 1
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_above(float x)  // value of lane + 1
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

// VEC2: dim_x even and all base pointers 8-byte aligned -> one 8-byte access per lane and row.
template <int NS, bool VEC2>
struct WaveBackend {
    using V = float;
    using M = bool;

    const float *p_in;   // nullptr: p is implicitly +0 everywhere in the domain
    const float *d;
    float *p_out;
    float *ring;         // this lane's slot 0 / plane 0 word in LDS
    int dim_x, gdim_y;
    int grow0;           // global row of local row 0
    int row_lo, row_hi;  // global rows present in the local arrays AND inside the domain
    int xa;              // column of cell a (even; may be < 0 or >= dim_x)
    bool a_in, b_in;     // columns inside the domain
    bool a_out, b_out;   // columns this tile is responsible for (valid interior)

    __device__ __forceinline__ V splat(float x) const { return x; }
    __device__ __forceinline__ V select(M m, V a, V b) const { return m ? a : b; }
    __device__ __forceinline__ M mask_and(M m, bool row) const { return m && row; }
    __device__ __forceinline__ V from_lower_lane(V x) const { return lane_below(x); }
    __device__ __forceinline__ V from_upper_lane(V x) const { return lane_above(x); }
    __device__ __forceinline__ sor::RowFacts row_facts(int r) const
    {
        return {r >= 0 && r < gdim_y, r > 0 && r < gdim_y - 1};
    }
    template <class P>
    __device__ __forceinline__ void poison(P &) const {}

    __device__ __forceinline__ size_t cell(int r) const
    {
        return (size_t)(r - grow0) * (size_t)dim_x + (size_t)(long)xa;
    }

    __device__ __forceinline__ void load_row(int r, V &pa, V &pb, V &da, V &db) const
    {
        const bool row_ok = r >= row_lo && r < row_hi;
        const bool in_dom = r >= 0 && r < gdim_y;
        // outside the domain: -0 (additive identity); inside but not loadable: don't care
        const float p_default = (in_dom && p_in == nullptr) ? 0.0f : -0.0f;
        float va = a_in ? p_default : -0.0f, vb = b_in ? p_default : -0.0f;
        float fa = 0.0f, fb = 0.0f;
        if (row_ok) {
            const size_t c = cell(r);
            if (VEC2) {
                if (a_in) {  // pair is inside as a whole (dim_x even, xa even)
                    const float2 f = *reinterpret_cast<const float2 *>(d + c);
                    fa = f.x;
                    fb = f.y;
                    if (p_in) {
                        const float2 q = *reinterpret_cast<const float2 *>(p_in + c);
                        va = q.x;
                        vb = q.y;
                    }
                }
            } else {
                if (a_in) {
                    fa = d[c];
                    if (p_in) va = p_in[c];
                }
                if (b_in) {
                    fb = d[c + 1];
                    if (p_in) vb = p_in[c + 1];
                }
            }
        }
        pa = va;
        pb = vb;
        da = fa;
        db = fb;
    }

    __device__ __forceinline__ void store_row(int r, V a, V b) const
    {
        const size_t c = cell(r);
        if (VEC2) {
            if (a_out) *reinterpret_cast<float2 *>(p_out + c) = make_float2(a, b);
        } else {
            if (a_out) p_out[c] = a;
            if (b_out) p_out[c + 1] = b;
        }
    }

    // ring word (row slot, plane) of this lane: [slot][plane][64 lanes]
    __device__ __forceinline__ void ring_store(int slot, int plane, V x) const
    {
        ring[(slot * 2 + plane) * 64] = x;
    }
    __device__ __forceinline__ V ring_load(int slot, int plane) const
    {
        return ring[(slot * 2 + plane) * 64];
    }
};

template <int NS, bool VEC2>
__device__ __forceinline__ sor::EdgeCell<WaveBackend<NS, VEC2>> edge_cell(int x, int dim_x)
{
    sor::EdgeCell<WaveBackend<NS, VEC2>> ec;
    const bool in = x >= 0 && x < dim_x;
    const int nh = (x > 0 ? 1 : 0) + (x < dim_x - 1 ? 1 : 0);  // horizontal neighbours present
    // -1/n evaluated in double and narrowed, poisson.cpp:67
    const float k2 = (float)(-1.0 / 2.0), k3 = (float)(-1.0 / 3.0), k4 = -0.25f;
    ec.in = in;
    ec.k_full = (nh == 2) ? k4 : (nh == 1) ? k3 : k2;
    ec.k_part = (nh == 2) ? k3 : k2;  // nh == 0 only when dim_x == 1 (rejected by the API)
    ec.z_full = (nh == 2) ? -0.0f : 0.0f;
    return ec;
}

template <int NS, bool VEC2, bool DX1>
__global__ void __launch_bounds__(kThreads)
sor_fused_kernel(float *p_out, const float *p_in, const float *d, Slab g, sor::Tiling t,
                 SorParams prm)
{
    using B = WaveBackend<NS, VEC2>;
    __shared__ float ring_mem[kWavesPerBlock][sor::ring_rows(NS) * 2 * 64];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWavesPerBlock + wave;
    if (tile >= t.n_strips * t.n_chunks) return;  // wave-uniform
    const int chunk = tile / t.n_strips;
    const int strip = tile - chunk * t.n_strips;

    const int x0 = sor::strip_x0(t, strip);
    const int r0 = t.g_begin + chunk * t.rows_per_chunk;
    const int r1 = min(r0 + t.rows_per_chunk, t.g_end);

    B bk;
    bk.p_in = p_in;
    bk.d = d;
    bk.p_out = p_out;
    bk.ring = &ring_mem[wave][lane];
    bk.dim_x = g.dim_x;
    bk.gdim_y = g.gdim_y;
    bk.grow0 = g.grow0;
    bk.row_lo = max(g.grow0, 0);
    bk.row_hi = min(g.grow0 + g.lrows, g.gdim_y);
    bk.xa = x0 + 2 * lane;
    bk.a_in = bk.xa >= 0 && bk.xa < g.dim_x;
    bk.b_in = bk.xa + 1 >= 0 && bk.xa + 1 < g.dim_x;
    const int out_lo = x0 + NS, out_hi = x0 + sor::kTileCols - NS;
    bk.a_out = bk.a_in && bk.xa >= out_lo && bk.xa < out_hi;
    bk.b_out = bk.b_in && bk.xa + 1 >= out_lo && bk.xa + 1 < out_hi;

    sor::Consts<B> c{prm.dx, prm.omega, prm.one_minus_omega};

    if (sor::tile_touches_boundary(t, strip, chunk, g.gdim_y)) {  // wave-uniform
        const auto eca = edge_cell<NS, VEC2>(bk.xa, g.dim_x);
        const auto ecb = edge_cell<NS, VEC2>(bk.xa + 1, g.dim_x);
        sor::stream_tile<B, NS, true, DX1>(bk, c, eca, ecb, r0, r1);
    } else {
        const sor::EdgeCell<B> none{};
        sor::stream_tile<B, NS, false, DX1>(bk, c, none, none, r0, r1);
    }
}

template <int NS>
hipError_t launch_ns(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                     const sor::Tiling &t, SorParams prm)
{
    const int tiles = t.n_strips * t.n_chunks;
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const bool aligned = (g.dim_x % 2 == 0) && ((reinterpret_cast<uintptr_t>(p_out) & 7) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p_in) & 7) == 0) &&
                         ((reinterpret_cast<uintptr_t>(d) & 7) == 0);
    const bool dx1 = prm.dx == 1.0f;
#define SFL_LAUNCH(VEC2, DX1) \
    sor_fused_kernel<NS, VEC2, DX1><<<blocks, kThreads, 0, s>>>(p_out, p_in, d, g, t, prm)
    if (aligned) {
        if (dx1) SFL_LAUNCH(true, true); else SFL_LAUNCH(true, false);
    } else {
        if (dx1) SFL_LAUNCH(false, true); else SFL_LAUNCH(false, false);
    }
#undef SFL_LAUNCH
    return hipGetLastError();
}

}  // namespace

hipError_t launch_sor_fused(hipStream_t s, float *p_out, const float *p_in, const float *d,
                            Slab g, int g_begin, int g_end, int nsweeps, int first_colour,
                            SorParams prm, int rows_per_chunk)
{
    if (g_end <= g_begin) return hipSuccess;
    if (first_colour != 0 || nsweeps < 2 || nsweeps > SFL_MAX_FUSE || (nsweeps & 1) ||
        p_out == p_in || p_out == nullptr || d == nullptr)
        return hipErrorInvalidValue;
    if (rows_per_chunk <= 0) {
        // enough tiles to fill 256 CUs x 4 SIMDs a few times over, but chunks long enough
        // that the 2*NS warm-up rows stay a small fraction
        const int strips = (g.dim_x + sor::strip_step(nsweeps) - 1) / sor::strip_step(nsweeps);
        const int rows = g_end - g_begin;
        int want_chunks = (256 * 4 * 4 + strips - 1) / strips;
        if (want_chunks < 1) want_chunks = 1;
        rows_per_chunk = (rows + want_chunks - 1) / want_chunks;
        const int min_rows = 8 * nsweeps;
        if (rows_per_chunk < min_rows) rows_per_chunk = min_rows;
        if (rows_per_chunk > rows) rows_per_chunk = rows;
    }
    const sor::Tiling t = sor::make_tiling(nsweeps, g.dim_x, g_begin, g_end, rows_per_chunk);
    switch (nsweeps) {
        case 2: return launch_ns<2>(s, p_out, p_in, d, g, t, prm);
        case 4: return launch_ns<4>(s, p_out, p_in, d, g, t, prm);
        case 6: return launch_ns<6>(s, p_out, p_in, d, g, t, prm);
        case 8: return launch_ns<8>(s, p_out, p_in, d, g, t, prm);
        case 10: return launch_ns<10>(s, p_out, p_in, d, g, t, prm);
        case 12: return launch_ns<12>(s, p_out, p_in, d, g, t, prm);
        case 14: return launch_ns<14>(s, p_out, p_in, d, g, t, prm);
        case 16: return launch_ns<16>(s, p_out, p_in, d, g, t, prm);
    }
    return hipErrorInvalidValue;
}

}  // namespace sfl
