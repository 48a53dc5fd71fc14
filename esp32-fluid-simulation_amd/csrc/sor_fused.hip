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

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

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
// ZERO_IN: p is implicitly zero on entry (first launch of a solve): p is never read.
//
// Loads are UNCONDITIONAL and branch-free: the row index is clamped into the rows the local
// array holds and the lane's column into the domain, so every address is valid; what a
// clamped access returns is irrelevant -- cells outside the domain are overwritten with -0.0f
// when they enter the pipeline (EDGE tiles), and clamped rows only ever feed rows outside the
// tile's exact interior.  This keeps the prefetched rows in flight across iterations (a guarded
// load would have to be waited for inside its branch).
template <int NS, bool VEC2, bool ZERO_IN>
struct WaveBackend {
    using V = float;
    using M = bool;
    static constexpr int RING = sor::ring_rows(NS);

    // buffer resources (base = local row 0 of each array): loads / stores take a per-lane byte
    // offset (loop invariant) plus a wave-uniform row offset in an SGPR -- no address VALU
    __amdgpu_buffer_rsrc_t rs_p, rs_d, rs_out;
    float *ring;            // this lane's word of ring slot 0 / plane 0 in LDS
    int dim_x, gdim_y;
    int grow0;              // global row of local row 0
    int row_lo, row_hi;     // global rows present in the local arrays AND inside the domain
    int off_a, off_b;       // byte offsets of the clamped load columns of cell a / b
    int off_out;            // byte offset of the true column of cell a
    bool a_out, b_out;      // columns this tile is responsible for (valid interior, in the domain)

    __device__ __forceinline__ V splat(float x) const { return x; }
    __device__ __forceinline__ V select(M m, V a, V b) const { return m ? a : b; }
    __device__ __forceinline__ M mask_and(M m, bool row) const { return m && row; }
    __device__ __forceinline__ V from_lower_lane(V x) const { return lane_below(x); }
    __device__ __forceinline__ V from_upper_lane(V x) const { return lane_above(x); }
    __device__ __forceinline__ V detach(V x) const
    {
        V r;
        asm("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
        return r;
    }
    __device__ __forceinline__ sor::RowFacts row_facts(int r) const
    {
        return {r >= 0 && r < gdim_y, r > 0 && r < gdim_y - 1};
    }
    template <class P>
    __device__ __forceinline__ void poison(P &) const {}

    __device__ __forceinline__ int row_bytes(int r) const  // wave-uniform
    {
        return (r - grow0) * dim_x * 4;
    }

    __device__ __forceinline__ void load_row(int r, V &pa, V &pb, V &da, V &db) const
    {
        const int soff = row_bytes(min(max(r, row_lo), row_hi - 1));
        if (VEC2) {
            const v2f f = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs_d, off_a, soff, 0));
            da = f.x;
            db = f.y;
        } else {
            da = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off_a, soff, 0));
            db = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off_b, soff, 0));
        }
        if (!ZERO_IN) {
            if (VEC2) {
                const v2f q = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs_p, off_a, soff, 0));
                pa = q.x;
                pb = q.y;
            } else {
                pa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_p, off_a, soff, 0));
                pb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_p, off_b, soff, 0));
            }
        }
    }

    __device__ __forceinline__ void store_row(int r, V a, V b) const
    {
        const int soff = row_bytes(r);
        if (VEC2) {
            if (a_out) {
                v2f o;
                o.x = a;
                o.y = b;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, o), rs_out, off_out, soff, 0);
            }
        } else {
            if (a_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), rs_out, off_out, soff, 0);
            if (b_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, b), rs_out, off_out + 4, soff, 0);
        }
    }

    // ring: [RING slots][2 planes][64 lanes]; slot and plane are compile-time constants at
    // every call site, so each access is one DS instruction with an immediate offset
    __device__ __forceinline__ void ring_store(int slot, int plane, V x) const
    {
        ring[(slot * 2 + plane) * 64] = x;
    }
    __device__ __forceinline__ V ring_load(int slot, int plane) const
    {
        return ring[(slot * 2 + plane) * 64];
    }
};

template <class B>
__device__ __forceinline__ sor::EdgeCell<B> edge_cell(int x, int dim_x)
{
    sor::EdgeCell<B> ec;
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

template <int NS, bool VEC2, bool DX1, bool ZERO_IN>
__global__ void __launch_bounds__(kThreads)
sor_fused_kernel(float *p_out, const float *p_in, const float *d, Slab g, sor::Tiling t,
                 SorParams prm)
{
    using B = WaveBackend<NS, VEC2, ZERO_IN>;
    __shared__ float ring_mem[kWavesPerBlock][B::RING * 2 * 64];

    // everything derived from the wave index is wave-uniform: tell the compiler (SGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWavesPerBlock + wave;
    if (tile >= t.n_strips * t.n_chunks) return;
    const int chunk = tile / t.n_strips;
    const int strip = tile - chunk * t.n_strips;

    const int x0 = sor::strip_x0(t, strip);
    const int r0 = t.g_begin + chunk * t.rows_per_chunk;
    const int r1 = min(r0 + t.rows_per_chunk, t.g_end);

    B bk;
    const size_t bytes = (size_t)g.lrows * (size_t)g.dim_x * 4;
    const unsigned records = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    bk.rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ZERO_IN ? d : p_in), 0, records, 0x00020000);
    bk.rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d), 0, records, 0x00020000);
    bk.rs_out = __builtin_amdgcn_make_buffer_rsrc(p_out, 0, records, 0x00020000);
    bk.ring = &ring_mem[wave][lane];
    bk.dim_x = g.dim_x;
    bk.gdim_y = g.gdim_y;
    bk.grow0 = g.grow0;
    bk.row_lo = max(g.grow0, 0);
    bk.row_hi = min(g.grow0 + g.lrows, g.gdim_y);
    const int xa = x0 + 2 * lane;  // true column of cell a (even; may be < 0 or >= dim_x)
    const bool a_in = xa >= 0 && xa < g.dim_x;
    const bool b_in = xa + 1 >= 0 && xa + 1 < g.dim_x;
    if (VEC2) {  // dim_x even: the pair is inside or outside as a whole
        bk.off_a = 4 * min(max(xa, 0), g.dim_x - 2);
        bk.off_b = bk.off_a + 4;
    } else {
        bk.off_a = 4 * min(max(xa, 0), g.dim_x - 1);
        bk.off_b = 4 * min(max(xa + 1, 0), g.dim_x - 1);
    }
    bk.off_out = 4 * xa;
    const int out_lo = x0 + NS, out_hi = x0 + sor::kTileCols - NS;
    bk.a_out = a_in && xa >= out_lo && xa < out_hi;
    bk.b_out = b_in && xa + 1 >= out_lo && xa + 1 < out_hi;

    sor::Consts<B> c{prm.dx, prm.omega, prm.one_minus_omega};

    if (sor::tile_touches_boundary(t, strip, chunk, g.gdim_y)) {  // wave-uniform
        const auto eca = edge_cell<B>(xa, g.dim_x);
        const auto ecb = edge_cell<B>(xa + 1, g.dim_x);
        sor::stream_tile<B, NS, true, DX1, ZERO_IN>(bk, c, eca, ecb, r0, r1);
    } else {
        const sor::EdgeCell<B> none{};
        sor::stream_tile<B, NS, false, DX1, ZERO_IN>(bk, c, none, none, r0, r1);
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
#define SFL_LAUNCH(VEC2, DX1)                                                                  \
    do {                                                                                       \
        if (p_in)                                                                              \
            sor_fused_kernel<NS, VEC2, DX1, false><<<blocks, kThreads, 0, s>>>(p_out, p_in, d, g, t, prm); \
        else                                                                                   \
            sor_fused_kernel<NS, VEC2, DX1, true><<<blocks, kThreads, 0, s>>>(p_out, p_in, d, g, t, prm);  \
    } while (0)
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
