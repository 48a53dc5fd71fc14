// sor_fused.hip -- gfx950 backends + launcher of the fused red-black SOR pipeline
// (sor_stream_core.h).  One 64-lane wavefront streams one tile; lanes exchange their W / E
// neighbours with DPP wave shifts, S / N neighbours stay in VGPRs, the right-hand side waits in
// a per-lane LDS ring.  No barriers, no atomics, no MFMA: a bandwidth / VALU-issue bound stencil.
//
// One lane flavour: 2 cells per lane, V = float, 128-column tiles, 8-byte accesses where dim_x is even
// and the arrays are 8-byte aligned, 4-byte ones otherwise (any dim_x).  (Round 1 also carried a
// 4-cells-per-lane flavour on packed fp32; packed fp32 issues at half the rate of plain fp32 on
// gfx950 -- profiles/r02_experiments_without_gain.txt -- it was never faster and is gone.)
//
// Compiled with -ffp-contract=off (bit-exactness contract, see stencil_kernels.hip).
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "sor_stream_core.h"

#ifndef SFL_DX_PART
#define SFL_DX_PART (-1)  // both halves in this translation unit
#endif

namespace sfl {
namespace {

constexpr int kWavesPerBlock = 4;
#ifndef SFL_NT_STORE_CELLS
#define SFL_NT_STORE_CELLS (24u << 20)
#endif
constexpr size_t kNtStoreCells = SFL_NT_STORE_CELLS;  // local cells from which p is stored non-temporally
constexpr int kFlipTiles = 1;                // alternate the stream direction of vertically adjacent tiles
constexpr int kThreads = 64 * kWavesPerBlock;

#ifdef SFL_SOR_TRACE
// Diagnostic builds only (tools/sor_clock_probe.hip; never defined for the product library): every wave
// records when it started and ended on the shader clock (s_memtime) AND on the constant 100 MHz
// real-time clock (s_memrealtime), plus where it ran -- 6 words per tile.
__device__ unsigned long long *g_sor_trace;
struct WaveTrace {
    unsigned long long t0, w0;
    unsigned hwid, xcc;
    __device__ __forceinline__ void begin()
    {
        t0 = __builtin_readcyclecounter();
        w0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    }
    __device__ __forceinline__ void end(int tile, int kind) const
    {
        const unsigned long long t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && g_sor_trace) {
            unsigned long long *o = g_sor_trace + 6 * (size_t)tile;
            o[0] = t0; o[1] = t1; o[2] = w0; o[3] = w1; o[4] = hwid; o[5] = ((unsigned long long)kind << 32) | xcc;
        }
    }
};
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

// DPP full-wave shifts (GFX9 wave_shr:1 / wave_shl:1).  Lane 0 / lane 63 receive 0, which only
// ever feeds cells of the tile's invalid rim.
#ifndef SFL_PRIO_LEVELS
#define SFL_PRIO_LEVELS 4  // priority levels the waves of a SIMD rotate through (1 = leave the priority alone)
#endif
#ifndef SFL_PRIO_FORCE
#define SFL_PRIO_FORCE (-1)  // diagnostic builds: 0 / 1 = rotation off / on whatever the launch; -1 = the launcher decides
#endif
#ifndef SFL_PRIO_ROWS
#define SFL_PRIO_ROWS 2    // rows (pipeline iterations) a wave spends on one level; must divide 6
#endif
#ifndef SFL_PROBE_NO_LDS
#define SFL_PROBE_NO_LDS 0   // diagnostic builds only: no rhs ring traffic (wrong results)
#endif
#ifndef SFL_PROBE_NO_LOAD
#define SFL_PROBE_NO_LOAD 0  // diagnostic builds only: no global loads (wrong results)
#endif
#ifndef SFL_PROBE_P_LOAD_AUX
#define SFL_PROBE_P_LOAD_AUX 0   // diagnostic builds only: cache-policy bits of the p loads (16 = sc1: agent scope, bypasses L1)
#endif
#ifndef SFL_PROBE_P_STORE_AUX
#define SFL_PROBE_P_STORE_AUX 0  // diagnostic builds only: ... of the p stores (16 = sc1: written through the XCD's L2)
#endif
#ifndef SFL_PROBE_SHIFT
#define SFL_PROBE_SHIFT 0  // diagnostic builds only (tools/sor_clock_probe.hip): 1 = no lane shift at all, 2 = row_shr / row_shl
#endif
__device__ __forceinline__ float lane_below(float x)  // value of lane - 1
{
    if (SFL_PROBE_SHIFT == 1) return x;
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), SFL_PROBE_SHIFT == 2 ? 0x111 : 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_above(float x)  // value of lane + 1
{
    if (SFL_PROBE_SHIFT == 1) return x;
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), SFL_PROBE_SHIFT == 2 ? 0x101 : 0x130, 0xf, 0xf, false));
}

// State shared by both flavours.  Loads are UNCONDITIONAL and branch-free: the row index is
// clamped into the rows the local array holds and the lane's column into the domain, so every
// address is valid; what a clamped access returns is irrelevant -- cells outside the domain are
// overwritten with -0.0f when they enter the pipeline (EDGE tiles), and clamped rows only ever
// feed rows outside the tile's exact interior.  This keeps the prefetched rows in flight across
// iterations (a guarded load would have to be waited for inside its branch).  Addresses are a
// per-lane byte offset (loop invariant VGPR) plus a wave-uniform row offset (SGPR) into a buffer
// resource: no address arithmetic on the vector ALU.
struct WaveCommon {
    __amdgpu_buffer_rsrc_t rs_p, rs_d, rs_out;
    int dim_x, gdim_y;
    int grow0;           // global row of local row 0
    int row_lo, row_hi;  // global rows present in the local arrays AND inside the domain
    int row_sign;        // +1: pipeline row index = domain row; -1: its negative (tile streamed top-down)
    int prio_turn;       // rotating issue priority: this wave's turn counter (see next_turn)
    int prio_on;         // ... enabled for this launch (wave-uniform)

    // the pipeline speaks in row INDICES t; domain row = row_sign * t (same parity either way)
    __device__ __forceinline__ sor::RowFacts row_facts(int t) const
    {
        const int r = row_sign * t;
        return {r >= 0 && r < gdim_y, r > 0 && r < gdim_y - 1};
    }
    template <class P>
    __device__ __forceinline__ void poison(P &) const {}

    // Rotating issue priority.  The SIMD's arbiter serves the waves it holds by priority, then AGE: with
    // equal priorities the oldest wave issues whenever it can (one dependent VALU instruction per ~4.3
    // cycles, 2 of them busy), the second fills the gaps and the third starves -- measured with
    // tools/sor_clock_probe.hip at 8192^2, NS = 16: the three waves of a SIMD finish after 228 k, 262 k and
    // 362 k cycles, the last one running alone (35 % VALU use) for the final quarter of the launch.  Every
    // wave therefore moves to the next priority level at each trip (its start level comes from its hardware
    // wave slot, so the waves of a SIMD start on different levels): over its life each wave spends the same
    // share of trips at each level, all advance at the same pace and the SIMD stays full to the end.
    // Only for launches whose tiles are all resident at once (Tiling::rotate, set by the launcher): when tiles
    // queue up behind the resident ones, a finished wave is replaced at once, the SIMDs stay full by themselves
    // and the rotation only costs (16384^2, 2.3 rounds: 784 -> 804 us per launch with it; 8192^2, one round:
    // 228 -> 217 us; profiles/r03_priority_rotation.txt).
    __device__ __forceinline__ void start_turns()
    {
        if (prio_on == 2) {   // a sender tile (HaloWait::done): top priority from the first instruction, no turns
            __builtin_amdgcn_s_setprio(3);
            return;
        }
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));  // wave slot on the SIMD
        prio_turn = (int)(hw % SFL_PRIO_LEVELS);
    }
    __device__ __forceinline__ void next_turn()
    {
        if (SFL_PRIO_LEVELS <= 1) return;
        prio_turn = prio_turn + 1 >= SFL_PRIO_LEVELS ? prio_turn + 1 - SFL_PRIO_LEVELS : prio_turn + 1;
        // s_setprio takes an immediate: select it with scalar branches INSIDE one asm statement, so that the
        // straight-line trip stays straight-line for the compiler (a visible branch makes its wait-count pass
        // drain the loads in flight)
        asm volatile("s_cmp_lg_u32 %1, 1\n\t"
                     "s_cbranch_scc1 .Lsfl_pe_%=\n\t"
                     "s_cmp_lg_u32 %0, 0\n\t"
                     "s_cbranch_scc1 .Lsfl_p1_%=\n\t"
                     "s_setprio 0\n\t"
                     "s_branch .Lsfl_pe_%=\n"
                     ".Lsfl_p1_%=:\n\t"
                     "s_cmp_lg_u32 %0, 1\n\t"
                     "s_cbranch_scc1 .Lsfl_p2_%=\n\t"
                     "s_setprio 1\n\t"
                     "s_branch .Lsfl_pe_%=\n"
                     ".Lsfl_p2_%=:\n\t"
                     "s_cmp_lg_u32 %0, 2\n\t"
                     "s_cbranch_scc1 .Lsfl_p3_%=\n\t"
                     "s_setprio 2\n\t"
                     "s_branch .Lsfl_pe_%=\n"
                     ".Lsfl_p3_%=:\n\t"
                     "s_setprio 3\n"
                     ".Lsfl_pe_%=:"
                     :
                     : "s"(prio_turn), "s"(prio_on)
                     : "scc");
    }
    __device__ __forceinline__ int row_bytes(int t) const { return (row_sign * t - grow0) * dim_x * 4; }
    __device__ __forceinline__ int load_row_bytes(int t) const
    {
        return (min(max(row_sign * t, row_lo), row_hi - 1) - grow0) * dim_x * 4;
    }
};

// ---- 2 cells per lane --------------------------------------------------------------------
// VEC: dim_x even and 8-byte aligned arrays -> one 8-byte access per lane and row.
// NT (VEC only): the finished rows are stored non-temporally.  A launch writes every p row once and
// reads it back a whole launch later: on slabs whose arrays exceed the caches the nt hint keeps the
// write stream from displacing the halo rows neighbouring tiles are about to re-read (8192^2:
// -1.1 %, 8192 x 4096: -2.3 .. -5 %); on cache-resident slabs the next launch WANTS those rows in
// cache (8192 x 1024: +9 %), so the launcher sets it from the slab size.
// ST = cache policy of the p stores (VEC only): 0 plain; 2 non-temporal (NT, above); 16 = sc1, WRITTEN THROUGH to memory -- the launch
// in front of an in-time halo exchange, whose sender tiles publish their rows to a copy / send kernel on another stream (or GPU)
// while the launch is still running: with plain stores every sender would have to write back its XCD's whole L2 first
// (buffer_wbl2: the launch took 48 instead of 24 us), written-through rows only have to be waited for (8192 x 1024: +0.5 %
// for the launch, profiles/r04_experiments_without_gain.txt 3).
// LD = cache policy of the p loads (VEC only): 0 plain; 16 = sc1, past the CU's L1 -- the chained launch (sor_chain_kernel), whose
// tiles read rows that other CUs stored (written-through) while the launch is running.
template <int NS, bool VEC, bool ZERO_IN, int ST = 0, int LD = 0>
struct Lane2 : WaveCommon {
    using V = float;
    using M = bool;
        // three rows in flight ahead of the pipeline: six cost 12 more VGPRs (and, with the rhs read-ahead, spills
    // at NS = 16) without being faster (profiles/r02_rhs_read_ahead.txt)
    static constexpr int kTileCols = 128, kColAlign = 2, kCells = 2, kPrefetch = 3, kTurnRows = SFL_PRIO_ROWS;
    static constexpr int kStoreAux = VEC ? ST : 0;
    // LDS per wave: the rhs ring (RING rows x 2 planes x 64 lanes x 4 B)
    static constexpr int kRingFloats = sor::ring_rows(NS) * 2 * 64;

    float *ring;         // this lane's word of ring slot 0 / plane 0 in LDS
    int off_a, off_b;    // byte offsets of the clamped load columns of cell a / b
    int off_out;         // byte offset of the true column of cell a
    bool a_out, b_out;   // columns this tile is responsible for (exact interior, in the domain)

    __device__ __forceinline__ void setup(float *ring_base, int lane, int x0, int halo)
    {
        ring = ring_base + lane;
        const int xa = x0 + 2 * lane;
        if (VEC) {  // dim_x even: the pair is inside or outside as a whole
            off_a = 4 * min(max(xa, 0), dim_x - 2);
            off_b = off_a + 4;
        } else {
            off_a = 4 * min(max(xa, 0), dim_x - 1);
            off_b = 4 * min(max(xa + 1, 0), dim_x - 1);
        }
        off_out = 4 * xa;
        const int out_lo = x0 + halo, out_hi = x0 + kTileCols - halo;
        a_out = xa >= 0 && xa < dim_x && xa >= out_lo && xa < out_hi;
        b_out = xa + 1 >= 0 && xa + 1 < dim_x && xa + 1 >= out_lo && xa + 1 < out_hi;
    }
    __device__ __forceinline__ sor::EdgeCell<Lane2> edge_cell(int lane, int x0, int which) const
    {
        const int x = x0 + 2 * lane + which;
        sor::EdgeCell<Lane2> ec;
        const int nh = (x > 0 ? 1 : 0) + (x < dim_x - 1 ? 1 : 0);  // horizontal neighbours present
        // -1/n evaluated in double and narrowed, poisson.cpp:67
        const float k2 = (float)(-1.0 / 2.0), k3 = (float)(-1.0 / 3.0), k4 = -0.25f;
        ec.in = x >= 0 && x < dim_x;
        ec.k_full = (nh == 2) ? k4 : (nh == 1) ? k3 : k2;
        ec.k_part = (nh == 2) ? k3 : k2;  // nh == 0 only when dim_x == 1 (rejected by the API)
        ec.z_full = (nh == 2) ? -0.0f : 0.0f;
        return ec;
    }

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
    __device__ __forceinline__ void load_row(int r, V &pa, V &pb, V &da, V &db) const
    {
        if (SFL_PROBE_NO_LOAD) {
            asm volatile("" : "+v"(pa), "+v"(pb), "+v"(da), "+v"(db));
            return;
        }
        const int soff = load_row_bytes(r);
        if (VEC) {
            const v2f f = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs_d, off_a, soff, 0));
            da = f.x;
            db = f.y;
        } else {
            da = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off_a, soff, 0));
            db = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off_b, soff, 0));
        }
        if (!ZERO_IN) {
            if (VEC) {
                const v2f q = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs_p, off_a, soff, SFL_PROBE_P_LOAD_AUX ? SFL_PROBE_P_LOAD_AUX : LD));
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
        if (VEC) {
            if (a_out) {
                v2f o;
                o.x = a;
                o.y = b;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, o), rs_out, off_out, soff, SFL_PROBE_P_STORE_AUX ? SFL_PROBE_P_STORE_AUX : ST);
            }
        } else {
            if (a_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), rs_out, off_out, soff, 0);
            if (b_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, b), rs_out, off_out + 4, soff, 0);
        }
    }

#ifdef SFL_PROBE_COOP
    // TIMING MOCK of cooperative tiles (tools/sor_clock_probe.hip -DSFL_PROBE_COOP; never in the product): the waves of a
    // block would be neighbouring strips that overlap by one lane on each side and keep each other's edge lanes exact
    // instead of letting NS columns per side go stale.  Per row every wave then (1) publishes the newest state of its two
    // outermost exact lanes -- NS values each: one per row in flight -- to LDS, (2) meets its neighbours at a barrier (all
    // cross-wave operands of row y were produced in row y - 1), (3) overwrites the state of its two ghost lanes with the
    // neighbours' values: exec-masked LDS accesses, no vector-ALU instruction.  Here the values go round in the same way
    // but between arbitrary lanes of the block, so the results are garbage; instruction mix, LDS traffic and the
    // barrier are the real thing.  The tiling is NOT changed: compare the time per launch with the shipped kernel's and
    // multiply by the tiles the scheme would save.
    float *coop_pub, *coop_get;   // this lane's publish / pick-up word in the block's exchange area
    bool coop_is_pub, coop_is_ghost;
#if SFL_PROBE_COOP == 3
    // TIMING MOCK of a two-wave VERTICAL pass pipeline (VERDICT r04 item 5 b): wave A of a pair would run passes 1 .. NS/2 and
    // hand every finished row to wave B (passes NS/2 + 1 .. NS) through an LDS ring with a row-granular flag -- no
    // s_barrier.  Here the waves of a pair are neighbouring tiles of the shipped tiling that do exactly that traffic, with
    // the real dependency: the even wave writes a row (ds_write_b64 per lane) into an 8-row ring, drains, publishes its row
    // count and stays at most 6 rows ahead of its partner (one ds_read_b32 per row); the odd wave waits for the count, takes
    // the row from the ring INTO ITS PIPELINE (the row that enters is what it read: garbage results) and publishes its own
    // progress.  Compare the time per launch with the shipped kernel's: the difference is what the hand-over costs a row.
    float *vp_ring;                    // this lane's two words in row slot 0 of the pair's ring
    int *vp_mine, *vp_other;           // rows this wave / its partner has gone through
    int vp_rows;
    bool vp_producer;
    template <int NSW, int U, class P>
    __device__ __forceinline__ void coop_mock(P &pp)
    {
        constexpr int RING = sor::ring_rows(NSW);
        constexpr int Q = U % kPrefetch;
        ++vp_rows;
        float2 *slot = reinterpret_cast<float2 *>(vp_ring + (vp_rows & 7) * 128);
        if (vp_producer) {
            for (int spin = 0; spin < 4000 && vp_rows - __hip_atomic_load(vp_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > 6; ++spin)
                __builtin_amdgcn_s_sleep(1);
            *slot = float2{pp.E[sor::wrapn(U - 1, RING)], pp.O[sor::wrapn(U - 2, RING)]};
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the row is in LDS before the count says so
            __hip_atomic_store(vp_mine, vp_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            for (int spin = 0; spin < 4000 && __hip_atomic_load(vp_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < vp_rows; ++spin)
                __builtin_amdgcn_s_sleep(1);
            const float2 got = *slot;
            pp.pa[Q] = got.x;
            pp.pb[Q] = got.y;
            __hip_atomic_store(vp_mine, vp_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
#elif SFL_PROBE_COOP == 2
    static __device__ __forceinline__ unsigned lds_addr(float *p)
    {
        return (unsigned)(size_t)(__attribute__((address_space(3))) float *)p;
    }
    // hand-scheduled flavour: ONE exec toggle around the NS publishing writes, ONE around the NS ghost-lane reads (which
    // land directly in the state registers), no vector-ALU instruction at all
    template <int NSW, int U, int K, class P>
    __device__ __forceinline__ void coop_pair(P &pp, unsigned long long m_pub, unsigned long long m_get)
    {
        constexpr int RING = sor::ring_rows(NSW);
        constexpr int par = U & 1;
        constexpr int e0 = sor::wrapn(U - 1 - 4 * K, RING), e1 = sor::wrapn(U - 3 - 4 * K, RING);
        constexpr int o0 = sor::wrapn(U - 2 - 4 * K, RING), o1 = sor::wrapn(U - 4 - 4 * K, RING);
        constexpr int b = (par * NSW + 4 * K) * 32;
        asm volatile("s_mov_b64 exec, %[m]\n\t"
                     "ds_write_b32 %[a], %[v0] offset:%[f0]\n\t"
                     "ds_write_b32 %[a], %[v1] offset:%[f1]\n\t"
                     "ds_write_b32 %[a], %[v2] offset:%[f2]\n\t"
                     "ds_write_b32 %[a], %[v3] offset:%[f3]\n\t"
                     "s_mov_b64 exec, -1"
                     :
                     : [m] "s"(m_pub), [a] "v"(lds_addr(coop_pub)), [v0] "v"(pp.E[e0]), [v1] "v"(pp.E[e1]), [v2] "v"(pp.O[o0]),
                       [v3] "v"(pp.O[o1]), [f0] "n"(b), [f1] "n"(b + 32), [f2] "n"(b + 64), [f3] "n"(b + 96)
                     : "memory");
    }
    template <int NSW, int U, int K, class P>
    __device__ __forceinline__ void coop_pick(P &pp, unsigned long long m_get)
    {
        constexpr int RING = sor::ring_rows(NSW);
        constexpr int par = U & 1;
        constexpr int e0 = sor::wrapn(U - 1 - 4 * K, RING), e1 = sor::wrapn(U - 3 - 4 * K, RING);
        constexpr int o0 = sor::wrapn(U - 2 - 4 * K, RING), o1 = sor::wrapn(U - 4 - 4 * K, RING);
        constexpr int b = (par * NSW + 4 * K) * 32;
        asm volatile("s_mov_b64 exec, %[m]\n\t"
                     "ds_read_b32 %[v0], %[a] offset:%[f0]\n\t"
                     "ds_read_b32 %[v1], %[a] offset:%[f1]\n\t"
                     "ds_read_b32 %[v2], %[a] offset:%[f2]\n\t"
                     "ds_read_b32 %[v3], %[a] offset:%[f3]\n\t"
                     "s_mov_b64 exec, -1"
                     : [v0] "+v"(pp.E[e0]), [v1] "+v"(pp.E[e1]), [v2] "+v"(pp.O[o0]), [v3] "+v"(pp.O[o1])
                     : [m] "s"(m_get), [a] "v"(lds_addr(coop_get)), [f0] "n"(b), [f1] "n"(b + 32), [f2] "n"(b + 64), [f3] "n"(b + 96)
                     : "memory");
    }
    template <int NSW, int U, class P>
    __device__ __forceinline__ void coop_mock(P &pp)
    {
        const unsigned long long m_pub = (1ull << 1) | (1ull << 62), m_get = 1ull | (1ull << 63);
        coop_pair<NSW, U, 0>(pp, m_pub, m_get);
        if (NSW >= 8) coop_pair<NSW, U, 1>(pp, m_pub, m_get);
        if (NSW >= 12) coop_pair<NSW, U, 2>(pp, m_pub, m_get);
        if (NSW >= 16) coop_pair<NSW, U, 3>(pp, m_pub, m_get);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        coop_pick<NSW, U, 0>(pp, m_get);
        if (NSW >= 8) coop_pick<NSW, U, 1>(pp, m_get);
        if (NSW >= 12) coop_pick<NSW, U, 2>(pp, m_get);
        if (NSW >= 16) coop_pick<NSW, U, 3>(pp, m_get);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#else
    template <int NSW, int U, class P>
    __device__ __forceinline__ void coop_mock(P &pp)
    {
        constexpr int RING = sor::ring_rows(NSW);
        constexpr int par = U & 1;
        if (coop_is_pub) {
#pragma unroll
            for (int k = 0; k < NSW / 2; ++k) {
                coop_pub[(par * NSW + k) * 8] = pp.E[sor::wrapn(U - 1 - 2 * k, RING)];
                coop_pub[(par * NSW + NSW / 2 + k) * 8] = pp.O[sor::wrapn(U - 2 - 2 * k, RING)];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (coop_is_ghost) {
#pragma unroll
            for (int k = 0; k < NSW / 2; ++k) {
                pp.E[sor::wrapn(U - 1 - 2 * k, RING)] = coop_get[(par * NSW + k) * 8];
                pp.O[sor::wrapn(U - 2 - 2 * k, RING)] = coop_get[(par * NSW + NSW / 2 + k) * 8];
            }
        }
    }
#endif
#endif

    // ring: [RING slots][2 planes][64 lanes]; slot and plane are compile-time constants at
    // every call site, so each access is one DS instruction with an immediate offset
    __device__ __forceinline__ void ring_store(int slot, int plane, V x) const
    {
        if (SFL_PROBE_NO_LDS) return;
        ring[(slot * 2 + plane) * 64] = x;
    }
    __device__ __forceinline__ void pin() const { __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ V ring_load(int slot, int plane) const
    {
        if (SFL_PROBE_NO_LDS) {
            V r = __builtin_bit_cast(float, off_out);
            asm volatile("" : "+v"(r));
            return r;
        }
        return ring[(slot * 2 + plane) * 64];
    }
};

// Occupancy the register allocator must keep: a launch lasts as long as one wave's chain of iterations, and
// that chain is served best with >= 3 waves on the SIMD (NS >= 12: 168 VGPRs) / 4 (NS <= 10: 128).
// NS = 14 does not fit 168 registers on its boundary path (2 - 7 registers spilled to scratch in round 3's builds);
// it is never the automatic choice, so it is simply given the registers it asks for (2 waves per SIMD).
#ifndef SFL_MIN_WAVES_DEEP
#define SFL_MIN_WAVES_DEEP 3
#endif
#ifndef SFL_PROBE_NO_EDGE
#define SFL_PROBE_NO_EDGE 0  // diagnostic builds only: every tile takes the interior path (wrong results at the walls)
#endif
#ifndef SFL_SOR_TRACE
// The SFL_PROBE_* / SFL_CHAIN_* switches compute WRONG results (or change cache policies the protocol relies on); they exist for
// tools/sor_clock_probe.hip, which includes this file with SFL_SOR_TRACE defined.  A product library never sees them set:
// `make EXTRA_FLAGS=-DSFL_PROBE_NO_LOAD=1` stops here.
#ifdef SFL_PROBE_COOP
#error "SFL_PROBE_COOP is a timing mock (wrong results): only with SFL_SOR_TRACE (tools/sor_clock_probe.hip)"
#endif
static_assert(SFL_PROBE_NO_LDS == 0 && SFL_PROBE_NO_LOAD == 0 && SFL_PROBE_SHIFT == 0 && SFL_PROBE_NO_EDGE == 0 &&
                  SFL_PROBE_P_LOAD_AUX == 0 && SFL_PROBE_P_STORE_AUX == 0,
              "SFL_PROBE_* switches give wrong results: diagnostic builds only (define SFL_SOR_TRACE, tools/sor_clock_probe.hip)");
#endif
constexpr int min_waves_per_simd(int ns) { return ns == 14 ? 2 : ns >= 12 ? SFL_MIN_WAVES_DEEP : 4; }

// One tile: its NS passes over output rows rect.[r0, r1) of strip rect.strip, from p_in (ZERO_IN: from zero) to p_out.
// Returns the path taken (0 interior bottom-up, 1 boundary, 2 interior top-down).
template <class B, int NS, bool DX1, bool ZERO_IN>
__device__ __forceinline__ int relax_tile(float *p_out, const float *p_in, const float *d, const Slab &g, const sor::Tiling &t,
                                          const sor::TileRect &rect, const SorParams &prm, bool sender, float *ring_base, int lane
#ifdef SFL_PROBE_COOP
                                          , float *coop_mem, int wave
#endif
                                          )
{
    const int x0 = sor::strip_x0(t, rect.strip);
    const int r0 = rect.r0, r1 = rect.r1;
    const size_t bytes = (size_t)g.lrows * (size_t)g.dim_x * 4;
    const unsigned records = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    // The backend (buffer resources, lane offsets) is built inside each branch: built once in
    // front of the branch, its SGPRs live through the register-hungry boundary path too and the
    // allocator parks the store's descriptor in spill lanes, reloading it for every row of the
    // interior path as well (8 v_readlane / v_writelane per row, 6 % of its VALU-class instructions).
    auto backend = [&]() {
        B bk;
        bk.rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ZERO_IN ? d : p_in), 0, records, 0x00020000);
        bk.rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d), 0, records, 0x00020000);
        bk.rs_out = __builtin_amdgcn_make_buffer_rsrc(p_out, 0, records, 0x00020000);
        bk.dim_x = g.dim_x;
        bk.gdim_y = g.gdim_y;
        bk.grow0 = g.grow0;
        bk.row_lo = max(g.grow0, 0);
        bk.row_hi = min(g.grow0 + g.lrows, g.gdim_y);
        bk.row_sign = 1;
        bk.prio_on = sender ? 2 : t.rotate;   // senders first: the message is waiting for them
        bk.start_turns();
        bk.setup(ring_base, lane, x0, t.halo_cols);
#if defined(SFL_PROBE_COOP) && SFL_PROBE_COOP == 3
        {
            float *pair = coop_mem + (wave >> 1) * (8 * 128 + 16);
            bk.vp_ring = pair + lane * 2;
            bk.vp_mine = reinterpret_cast<int *>(pair + 8 * 128) + (wave & 1);
            bk.vp_other = reinterpret_cast<int *>(pair + 8 * 128) + 1 - (wave & 1);
            bk.vp_rows = 0;
            bk.vp_producer = (wave & 1) == 0;
        }
#endif
#ifdef SFL_PROBE_COOP
        bk.coop_is_pub = lane == 1 || lane == 62;
        bk.coop_is_ghost = lane == 0 || lane == 63;
        bk.coop_pub = coop_mem + wave * 2 + (lane == 62);
        // lane 0 picks up what the wave on its left published from lane 62, lane 63 what the wave on its right did from lane 1
        bk.coop_get = coop_mem + ((wave + (lane == 0 ? kWavesPerBlock - 1 : 1)) % kWavesPerBlock) * 2 + (lane == 0);
#endif
        return bk;
    };
    if (!SFL_PROBE_NO_EDGE && sor::tile_touches_boundary(t, rect, g.gdim_y)) {  // wave-uniform
        B bk = backend();
        sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega)};
        const auto eca = bk.edge_cell(lane, x0, 0);
        const auto ecb = bk.edge_cell(lane, x0, 1);
        sor::stream_tile<B, NS, true, DX1, ZERO_IN>(bk, c, eca, ecb, r0, r1);
        return 1;
    }
    if (sor::tile_may_flip(t, rect)) {  // streamed top-down: pipeline index = -row
        B bk = backend();
        bk.row_sign = -1;
        sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega)};
        const sor::EdgeCell<B> none{};
        sor::stream_tile<B, NS, false, DX1, ZERO_IN, true>(bk, c, none, none, 1 - r1, 1 - r0);
        return 2;
    }
    B bk = backend();
    sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega)};
    const sor::EdgeCell<B> none{};
    sor::stream_tile<B, NS, false, DX1, ZERO_IN>(bk, c, none, none, r0, r1);
    return 0;
}

template <class B, int NS, bool DX1, bool ZERO_IN>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd(NS))))
sor_fused_kernel(float *p_out, const float *p_in, const float *d, Slab g, sor::Tiling t1,
                 sor::Tiling t2, SorParams prm, HaloWait hw)
{
    __shared__ __attribute__((aligned(16))) float ring_mem[kWavesPerBlock][B::kRingFloats];
#if defined(SFL_PROBE_COOP) && SFL_PROBE_COOP == 3
    __shared__ float coop_mem[2 * (8 * 128 + 16)];   // per pair of waves: an 8-row ring of 64 x 2 words + the two row counts
    for (int k = threadIdx.x; k < 2 * (8 * 128 + 16); k += kThreads) coop_mem[k] = 0.0f;
    __syncthreads();
#elif defined(SFL_PROBE_COOP)
    __shared__ float coop_mem[2 * NS * 8];   // [row parity][value][wave x {left edge, right edge}]
#endif

    // everything derived from the wave index is wave-uniform: tell the compiler (SGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // XCD-aware block order: the dispatcher deals consecutive blocks round-robin to the 8 XCDs
    // (each with a private L2); give every XCD a CONTIGUOUS range of tiles instead, so that
    // horizontally / vertically adjacent tiles -- which re-read each other's halo columns and
    // rows -- hit in the same L2.  Speed only: any placement computes the same result.
    // The blocks that hold tiles which wait for a halo message inside the launch keep the dispatcher's own rotation over the
    // XCDs and come LAST (sor::tile_of_position has the reason); the contiguous ranges are dealt over the blocks in front.
    const int nblocks = gridDim.x;
    int block = blockIdx.x;
    const int free_blocks = t2.n_tiles ? nblocks : (sor::free_tiles(t1) / kWavesPerBlock);
    if (block < free_blocks) {
        const int per = free_blocks >> 3, rem = free_blocks & 7;
        const int xcd = block & 7, idx = block >> 3;
        block = xcd * per + min(xcd, rem) + idx;  // bijective on [0, free_blocks) for every count
    }
    // a launch covers up to two row ranges (the two cut-adjacent bands of a slab in one launch):
    // the tiles of the second tiling follow those of the first
    int tile = block * kWavesPerBlock + wave;
    if (tile >= t1.n_tiles + t2.n_tiles) return;
    if (tile < t1.n_tiles) tile = sor::tile_of_position(t1, tile);
#ifdef SFL_SOR_TRACE
    WaveTrace trace;
    trace.begin();
    const int trace_tile = tile;
    int trace_kind = 0;
#endif
    const bool second = tile >= t1.n_tiles;  // wave-uniform
    const sor::Tiling t = second ? t2 : t1;
    if (second) tile -= t1.n_tiles;
    const sor::TileRect rect = sor::tile_rect(t, tile);
    const int r0 = rect.r0, r1 = rect.r1;

    // Halo arrival inside the launch (kernels.h HaloWait): a tile that reads a row a halo message writes -- in either
    // stream direction it reads at most NS + RING rows beyond its output rows -- waits for the message's epoch; the
    // other tiles of the launch are already running.  One relaxed poll per turn (all lanes read the one word: one
    // request), then ONE agent-scope acquire, so that the rows another CU / queue / GPU wrote while this launch was
    // resident are read from memory, not from this CU's L1.
    // "Reads a row" is meant in cache lines: when the row pitch is not a multiple of the line, the line that holds the first
    // bytes of an owned row also holds the last bytes of the ghost row below it.  A tile that does NOT wait must not touch such
    // a line either: its fill, requested before the message landed, can be installed in the CU's L1 after a waiting tile's
    // acquire has invalidated it, and the waiting tile then reads the ghost row's old bytes from it (seen once in 26 k solves
    // on 3000- and 2999-column slabs, never on pitches of whole lines: tools/unaligned_stress.py).  Hence `line_rows`.
    const int line_rows = (g.dim_x & 63) ? 1 + 63 / g.dim_x : 0;   // rows a 256-byte span reaches across a row boundary
    const int reach = NS + sor::ring_rows(NS) + line_rows;
    if (hw.flag != nullptr && (r0 - reach < hw.own_lo || r1 + reach > hw.own_hi)) {  // wave-uniform
        const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
        bool arrived = true;
        while ((int)((unsigned)__hip_atomic_load(hw.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)hw.epoch) < 0) {
            __builtin_amdgcn_s_sleep(20);
            if (__builtin_amdgcn_s_memrealtime() - t_begin > 100ull * (unsigned long long)hw.timeout_us) {
                arrived = false;
                break;
            }
        }
        if (!arrived && lane == 0) atomicOr(hw.timed_out, 1);
        if (hw.system_scope)   // (wave-uniform) rows written by a peer GPU over xGMI
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        else
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }

    // a SENDER tile (kernels.h HaloWait::done): its output rows are part of the next halo message
    const bool sender = hw.done != nullptr && (r0 < hw.send_lo_end || r1 > hw.send_hi_begin);   // wave-uniform

#ifdef SFL_PROBE_COOP
    const int kind = relax_tile<B, NS, DX1, ZERO_IN>(p_out, p_in, d, g, t, rect, prm, sender, ring_mem[wave], lane, coop_mem, wave);
#else
    const int kind = relax_tile<B, NS, DX1, ZERO_IN>(p_out, p_in, d, g, t, rect, prm, sender, ring_mem[wave], lane);
#endif
#ifdef SFL_SOR_TRACE
    trace_kind = kind;
#else
    (void)kind;
#endif
    if (sender) {
        // This wave's rows must be in memory before it counts itself: the copy / send kernel that picks them up runs on any
        // XCD, or on another GPU.  Written-through stores (B::kStoreAux == 16) only have to be waited for -- every storing
        // wave drains its own (CDNA guide, inter-workgroup rules R1); plain stores (odd widths: the 4-byte path) need the
        // XCD's L2 written back, with the wait restated where the compiler cannot drop it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (B::kStoreAux != 16) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_fetch_add(hw.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef SFL_SOR_TRACE
    __builtin_amdgcn_s_waitcnt(0);  // the wave's stores have left
    trace.end(trace_tile, trace_kind);
#endif
}

// ---- chained supersteps (kernels.h launch_sor_chain) ---------------------------------------------------------------------
// What a launch boundary costs a thin slab: 2.4-3.9 us of dispatch / drain per launch of 20 us, and a SIMD whose older wave
// has finished runs its younger one alone, at 60 % of the pair's rate, for the last seventh of every launch
// (profiles/r04_thin_share_lower_bound.txt).  Here a wave goes straight on to its tile of the next superstep; what it needs
// from the previous superstep are the tiles within NS + 3 rows and one strip of its own.
#ifndef SFL_PROBE_CHAIN_NO_DEPS
#define SFL_PROBE_CHAIN_NO_DEPS 0   // diagnostic builds only: nobody waits for anybody (wrong results): the cost of the waits
#endif
#ifndef SFL_CHAIN_ST
#define SFL_CHAIN_ST 16             // cache policy of the chain's p stores / loads (diagnostic builds: 0 = plain, wrong results)
#endif
#ifndef SFL_CHAIN_LD
#define SFL_CHAIN_LD 16
#endif
#ifndef SFL_CHAIN_FLAG_STRIDE
#define SFL_CHAIN_FLAG_STRIDE 32    // ints between the words of two tiles: a 128-byte line each
#endif
#ifndef SFL_CHAIN_SLEEP
#define SFL_CHAIN_SLEEP 1
#endif
#ifndef SFL_SOR_TRACE
static_assert(SFL_PROBE_CHAIN_NO_DEPS == 0 && SFL_CHAIN_ST == 16 && SFL_CHAIN_LD == 16,
              "the chained launch's hand-off needs written-through stores, L1-bypassing loads and its waits: diagnostic builds only");
#endif
struct ChainLink {
    sor::Tiling t;
    HaloWait hw;
    const int *guard_flag;
    int guard_epoch, guard_lo_end, guard_hi_begin;
};
struct ChainArgs {
    int n_steps, waves, epoch;
    int timeout_us;
    int *flags;
    int *timed_out;
    ChainLink link[kMaxChain];
};

// rows beyond its output rows that a tile touches, in either stream direction: NS rows of input, the row that makes the first
// input row even, kPrefetch rows in flight past the last one -- and, on pitches that are not whole cache lines, the rows that
// share a line with them (see the arrival wait of sor_fused_kernel)
template <class B, int NS>
__device__ __forceinline__ int chain_reach(const Slab &g)
{
    return NS + B::kPrefetch + ((g.dim_x & 63) ? 1 + 63 / g.dim_x : 0);
}

// Wait until every tile of tiling `prev` whose output rows intersect [lo, hi) in strips strip - 1 .. strip + 1 has published
// `want` (or a later value).  Lane k polls the k-th such tile; one relaxed agent-scope load per lane and turn.
__device__ __forceinline__ __attribute__((unused)) bool chain_wait(const sor::Tiling &prev, int strip, int lo, int hi, const int *flags, int want, int lane, int timeout_us)
{
    int c0[3], n[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int c1;
        if (sor::chunks_touching(prev, strip - 1 + k, lo, hi, &c0[k], &c1)) n[k] = c1 - c0[k] + 1;
    }
    const int total = n[0] + n[1] + n[2];   // wave-uniform
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
    for (int base = 0; base < total; base += 64) {
        const int k = base + lane;
        int idx = -1;
        if (k < n[0]) idx = sor::tile_index(prev, strip - 1, c0[0] + k);
        else if (k < n[0] + n[1]) idx = sor::tile_index(prev, strip, c0[1] + k - n[0]);
        else if (k < total) idx = sor::tile_index(prev, strip + 1, c0[2] + k - n[0] - n[1]);
        for (;;) {
            const bool behind = idx >= 0 && (int)((unsigned)__hip_atomic_load(flags + (size_t)idx * SFL_CHAIN_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)want) < 0;
            if (!__builtin_amdgcn_ballot_w64(behind)) break;
            __builtin_amdgcn_s_sleep(SFL_CHAIN_SLEEP);
            if (__builtin_amdgcn_s_memrealtime() - t_begin > 100ull * (unsigned long long)timeout_us) return false;
        }
    }
    return true;
}

template <class B, int NS, bool DX1>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd(NS))))
sor_chain_kernel(float *pa, float *pb, const float *d, Slab g, SorParams prm, ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) float ring_mem[kWavesPerBlock][B::kRingFloats];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // XCD-contiguous slots, as in sor_fused_kernel: slot k of every superstep is (nearly) the same rectangle, so a wave's
    // neighbours in one superstep are its neighbours in the next, on the same XCD
    const int nblocks = gridDim.x;
    int block = blockIdx.x;
    {
        const int per = nblocks >> 3, rem = nblocks & 7;
        const int xcd = block & 7, idx = block >> 3;
        block = xcd * per + min(xcd, rem) + idx;
    }
    const int slot = block * kWavesPerBlock + wave;
    if (slot >= a.waves) return;
    const int reach = chain_reach<B, NS>(g);

    for (int s = 0; s < a.n_steps; ++s) {
        const sor::Tiling t = a.link[s].t;
        const HaloWait hw = a.link[s].hw;
        const float *p_in = (s & 1) ? pb : pa;
        float *p_out = (s & 1) ? pa : pb;
        for (int tile = slot; tile < t.n_tiles; tile += a.waves) {
            const sor::TileRect rect = sor::tile_rect(t, tile);
            const int r0 = rect.r0, r1 = rect.r1;
            int late = 0;   // which wait gave up (bits of *timed_out: 2 = for the tiles around, 4 = for a halo message)
            // the previous superstep's tiles around this one: their output is this tile's input, and this tile's output
            // replaces their input (the two arrays take turns)
            if (s > 0 && !SFL_PROBE_CHAIN_NO_DEPS)
                late = chain_wait(a.link[s - 1].t, rect.strip, r0 - reach, r1 + reach, a.flags, a.epoch + s, lane, a.timeout_us) ? 0 : 2;
            // the halo message of the exchange in front of this superstep (see sor_fused_kernel; no acquire: sc1 loads), and
            // the message two supersteps back whose source this tile overwrites (kernels.h ChainStep::guard_flag)
            const bool incoming = hw.flag != nullptr && (r0 - reach < hw.own_lo || r1 + reach > hw.own_hi);
            const bool outgoing = a.link[s].guard_flag != nullptr && (r0 < a.link[s].guard_lo_end || r1 > a.link[s].guard_hi_begin);
            if (incoming || outgoing) {
                const int *word = incoming ? hw.flag : a.link[s].guard_flag;
                const int want = incoming ? hw.epoch : a.link[s].guard_epoch;   // the later of the two when both apply
                const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
                while ((int)((unsigned)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)want) < 0) {
                    __builtin_amdgcn_s_sleep(20);
                    if (__builtin_amdgcn_s_memrealtime() - t_begin > 100ull * (unsigned long long)a.timeout_us) {
                        late |= 4;
                        break;
                    }
                }
            }
            if (late && lane == 0) atomicOr(a.timed_out, late);
            const bool sender = hw.done != nullptr && (r0 < hw.send_lo_end || r1 > hw.send_hi_begin);
            relax_tile<B, NS, DX1, false>(p_out, p_in, d, g, t, rect, prm, sender, ring_mem[wave], lane);
            // publish: the rows are written through; once this wave's stores have left, the word may say so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(a.flags + (size_t)tile * SFL_CHAIN_FLAG_STRIDE, a.epoch + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sender) __hip_atomic_fetch_add(hw.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sender) __builtin_amdgcn_s_setprio(0);
        }
    }
}

// Resident waves of one instantiation on the whole device (occupancy query, cached).
template <class B, int NS, bool DX1, bool ZERO_IN>
int resident_waves()
{
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, cus = 0, blocks = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, sor_fused_kernel<B, NS, DX1, ZERO_IN>,
                                                     kThreads, 0) != hipSuccess ||
        blocks < 1 || cus < 1) {
        (void)hipGetLastError();
        return 256 * 4 * 3;  // not cached: try again next time
    }
    cached = cus * blocks * kWavesPerBlock;
    return cached;
}

inline int device_simds()
{
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        return 1024;
    }
    cached = cus * 4;
    return cached;
}

// Output rows per interior wave tile (boundary tiles get fewer: sor::make_tiling).  Every tile
// costs about the same (rpc + 2 NS rows streamed, of which the prologue trips skip ~NS rows' worth
// of passes), a SIMD works through its tiles essentially one VALU stream at a time, so a launch
// takes about ceil(tiles / SIMDs) * (rpc + NS) row-steps -- provided each SIMD holds ~2+ waves to cover DS / memory latency (measured on
// 8192 x {1024, 8192}, profiles/r01_rows_per_chunk.txt: fewer than ~2 waves per SIMD costs 1.4x,
// 2..3 waves ~1.08x).  Pick the chunk count that minimises that, never exceeding the
// resident-wave capacity by less than a full round.
template <class B>
int auto_rows_per_chunk(const Slab &g, int g_begin, int g_end, int ns, int waves, int simds)
{
    const int rows = g_end - g_begin;
    int best_rows = rows;
    double best_cost = 1e300;
    const int max_chunks = (rows + 7) / 8;
    for (int chunks = 1; chunks <= max_chunks; ++chunks) {
        const int rpc = (rows + chunks - 1) / chunks;
        const sor::Tiling t = sor::make_tiling(ns, B::kTileCols, B::kColAlign, g.dim_x, g.gdim_y, g_begin,
                                               g_end, rpc, sor::kEdgeRowCost16, kFlipTiles);
        const long tiles = t.n_tiles;
        const double per_simd = (double)tiles / simds;
        const long serial = (tiles + simds - 1) / simds;          // tiles one SIMD works through
        const long rounds = (tiles + waves - 1) / waves;          // residency rounds
        const double penalty = per_simd >= 2.8 ? 1.0 : per_simd >= 1.9 ? 1.08 : 1.45;
        double cost = (double)(serial > rounds ? serial : rounds) * (rpc + ns + 2) * penalty;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best_rows = rpc;
        }
        if (tiles > 12L * simds) break;
    }
    return best_rows;
}

template <class B, int NS, bool DX1, bool ZERO_IN>
hipError_t launch_variant(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                          SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    auto tiling = [&](int g_begin, int g_end) {
        if (g_end <= g_begin) {
            sor::Tiling none{};
            none.n_tiles = 0;
            return none;
        }
        int rpc = rows_per_chunk > g_end - g_begin ? g_end - g_begin : rows_per_chunk;
        if (rpc <= 0)
            rpc = auto_rows_per_chunk<B>(g, g_begin, g_end, NS, resident_waves<B, NS, DX1, ZERO_IN>(), device_simds());
        return sor::make_tiling(NS, B::kTileCols, B::kColAlign, g.dim_x, g.gdim_y, g_begin, g_end, rpc,
                                sor::kEdgeRowCost16, kFlipTiles ? 1 + (sweep & 1) : 0);
    };
    sor::Tiling t1 = tiling(rows.g_begin, rows.g_end), t2 = tiling(rows.g2_begin, rows.g2_end);
    const int tiles = t1.n_tiles + t2.n_tiles;
    if (tiles == 0) return hipSuccess;
    // rotating issue priority (WaveCommon::next_turn) only when every tile is resident from the start AND the SIMDs
    // hold three waves: with two, the second wave fills the first one's gaps anyway (8192 x 1024, NS = 10: 24.6 us per
    // launch with and without), and a thin slab's launches run next to the halo exchange's kernels, which should
    // not have to compete with raised priorities
    t1.rotate = t2.rotate = SFL_PRIO_FORCE >= 0 ? SFL_PRIO_FORCE
                                          : tiles <= resident_waves<B, NS, DX1, ZERO_IN>() && 2 * tiles > 5 * device_simds();
    const int blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    HaloWait hw = wait ? *wait : HaloWait{nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, 0, 0};
    if (hw.timeout_us <= 0) hw.timeout_us = kHaloWaitDefaultTimeoutUs;
    if (hw.flag != nullptr && t2.n_tiles == 0) {
        // the tiles that will wait inside the launch go last (sor::tile_of_position): the same test as the kernel's, on the
        // chunks of an inner strip and of a boundary strip (chunks are numbered bottom-up: the free ones are one range)
        const int line_rows = (g.dim_x & 63) ? 1 + 63 / g.dim_x : 0;
        const int reach = NS + sor::ring_rows(NS) + line_rows;
        auto free_range = [&](int strip, int n_chunks, int *c0, int *c1) {
            *c0 = *c1 = 0;
            bool any = false;
            for (int c = 0; c < n_chunks; ++c) {
                const sor::TileRect r = sor::tile_rect(t1, sor::tile_index(t1, strip, c));
                if (r.r0 - reach < hw.own_lo || r.r1 + reach > hw.own_hi) continue;
                if (!any) *c0 = c;
                any = true;
                *c1 = c + 1;
            }
        };
        if (t1.n_inner > 0) free_range(1, t1.n_chunks, &t1.free_c0, &t1.free_c1);
        free_range(0, t1.n_chunks_edge, &t1.free_e0, &t1.free_e1);
    }
    if (senders) {   // the same test as the kernel's, on the same tilings
        int n = 0;
        if (hw.done)
            for (const sor::Tiling *t : {&t1, &t2})
                for (int k = 0; k < t->n_tiles; ++k) {
                    const sor::TileRect r = sor::tile_rect(*t, k);
                    n += r.r0 < hw.send_lo_end || r.r1 > hw.send_hi_begin;
                }
        *senders = n;
    }
    sor_fused_kernel<B, NS, DX1, ZERO_IN><<<blocks, kThreads, 0, s>>>(p_out, p_in, d, g, t1, t2, prm, hw);
    return hipGetLastError();
}

template <class B, int NS, bool DX1>
hipError_t launch_chain_variant(hipStream_t s, float *pa, float *pb, const float *d, Slab g, const ChainStep *steps, int n_steps,
                                SorParams prm, int rows_per_chunk, int *flags, int flag_words, int epoch, int *timed_out,
                                int max_waves, int *senders, int tiles_at_most, bool *launched)
{
    if (launched) *launched = false;
    static int resident = 0;   // waves of this kernel the device holds at once
    if (!resident) {
        int dev = 0, cus = 0, blocks = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, sor_chain_kernel<B, NS, DX1>, kThreads, 0) != hipSuccess ||
            blocks < 1 || cus < 1) {
            (void)hipGetLastError();
            return hipErrorInvalidValue;
        }
        resident = cus * blocks * kWavesPerBlock;
    }
    ChainArgs a;
    a.n_steps = n_steps;
    a.epoch = epoch;
    a.flags = flags;
    a.timed_out = timed_out;
    a.timeout_us = steps[0].hw.timeout_us > 0 ? steps[0].hw.timeout_us : kHaloWaitDefaultTimeoutUs;
    int most = 0;
    for (int i = 0; i < n_steps; ++i) {
        const ChainStep &st = steps[i];
        int rpc = rows_per_chunk > st.g_end - st.g_begin ? st.g_end - st.g_begin : rows_per_chunk;
        if (rpc <= 0) rpc = auto_rows_per_chunk<B>(g, st.g_begin, st.g_end, NS, resident, device_simds());
        sor::Tiling t = sor::make_tiling(NS, B::kTileCols, B::kColAlign, g.dim_x, g.gdim_y, st.g_begin, st.g_end, rpc,
                                         sor::kEdgeRowCost16, kFlipTiles ? 1 + (st.sweep & 1) : 0);
        t.rotate = 0;
        a.link[i].t = t;
        a.link[i].hw = st.hw;
        a.link[i].guard_flag = st.guard_flag;
        a.link[i].guard_epoch = st.guard_epoch;
        a.link[i].guard_lo_end = st.guard_lo_end;
        a.link[i].guard_hi_begin = st.guard_hi_begin;
        if (t.n_tiles > most) most = t.n_tiles;
        if (senders) {
            int n = 0;
            if (st.hw.done)
                for (int k = 0; k < t.n_tiles; ++k) {
                    const sor::TileRect r = sor::tile_rect(t, k);
                    n += r.r0 < st.hw.send_lo_end || r.r1 > st.hw.send_hi_begin;
                }
            senders[i] = n;
        }
    }
    if (getenv("SFL_DEBUG_CHAIN"))
        fprintf(stderr, "sor chain: %d supersteps, rows [%d, %d) .. [%d, %d), most tiles %d (limit %d), resident %d, max waves %d\n", n_steps,
                steps[0].g_begin, steps[0].g_end, steps[n_steps - 1].g_begin, steps[n_steps - 1].g_end, most, tiles_at_most, resident, max_waves);
    if (most == 0 || (tiles_at_most > 0 && most > tiles_at_most)) return hipSuccess;
    if ((long)most * SFL_CHAIN_FLAG_STRIDE > (long)flag_words) return hipErrorInvalidValue;
    if (launched) *launched = true;
    int waves = most;
    if (waves > resident) waves = resident;
    if (max_waves > 0 && waves > max_waves) waves = max_waves;
    a.waves = waves;
    const int blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    sor_chain_kernel<B, NS, DX1><<<blocks, kThreads, 0, s>>>(pa, pb, d, g, prm, a);
    return hipGetLastError();
}

template <class B, int NS, bool ZERO_IN>
hipError_t launch_dx(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                     SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    // (a translation unit may hold only the dx == 1 kernels or only the general ones: SFL_DX_PART)
#if SFL_DX_PART != 1
    if (prm.dx == 1.0f)
        return launch_variant<B, NS, true, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
#endif
#if SFL_DX_PART != 0
    if (prm.dx != 1.0f)
        return launch_variant<B, NS, false, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
#endif
    return hipErrorInvalidValue;
}

template <int NS, bool ZERO_IN>
hipError_t launch_lane(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                       SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    const uintptr_t all = reinterpret_cast<uintptr_t>(p_out) | reinterpret_cast<uintptr_t>(p_in) |
                          reinterpret_cast<uintptr_t>(d);
    const bool can2v = (g.dim_x % 2 == 0) && (all & 7) == 0;
    if (can2v) {
        // written through when sender tiles publish rows of this launch while it runs (see Lane2)
        if (wait && wait->done)
            return launch_dx<Lane2<NS, true, ZERO_IN, 16>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
        // non-temporal stores once the slab's arrays no longer fit the caches
        if ((size_t)g.lrows * (size_t)g.dim_x >= kNtStoreCells)
            return launch_dx<Lane2<NS, true, ZERO_IN, 2>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
        return launch_dx<Lane2<NS, true, ZERO_IN>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    }
    return launch_dx<Lane2<NS, false, ZERO_IN>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
}

template <int NS>
hipError_t launch_ns(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                     SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    if (p_in == nullptr)
        return launch_lane<NS, true>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    return launch_lane<NS, false>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
}

}  // namespace

// One non-template entry per fuse depth and per half (dx == 1 / any dx); the depths and halves are spread over
// translation units (SFL_NS_GROUP = 0..5, SFL_DX_PART = 0 / 1, see csrc/Makefile) so that they compile in parallel.
#ifndef SFL_NS_GROUP
#define SFL_NS_GROUP (-1)  // single translation unit: every depth
#endif
#define SFL_ENTRY_ARGS                                                                             \
    hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g, SorRows rows, SorParams prm, \
        int rows_per_chunk, int sweep, const HaloWait *wait, int *senders
#define SFL_DEFINE_PART(N, P)                                                                      \
    hipError_t launch_sor_fused_ns##N##_p##P(SFL_ENTRY_ARGS)                                       \
    {                                                                                             \
        return launch_ns<N>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);   \
    }
#if SFL_DX_PART == 0
#define SFL_DEFINE_NS(N) SFL_DEFINE_PART(N, 0)
#elif SFL_DX_PART == 1
#define SFL_DEFINE_NS(N) SFL_DEFINE_PART(N, 1)
#else
#define SFL_DEFINE_NS(N) SFL_DEFINE_PART(N, 0) SFL_DEFINE_PART(N, 1)
#endif
#define SFL_DECLARE_NS(N)                                                                         \
    hipError_t launch_sor_fused_ns##N##_p0(SFL_ENTRY_ARGS);                                       \
    hipError_t launch_sor_fused_ns##N##_p1(SFL_ENTRY_ARGS);
SFL_DECLARE_NS(2) SFL_DECLARE_NS(4) SFL_DECLARE_NS(6) SFL_DECLARE_NS(8)
SFL_DECLARE_NS(10) SFL_DECLARE_NS(12) SFL_DECLARE_NS(14) SFL_DECLARE_NS(16)
#if SFL_NS_GROUP == 0 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(2) SFL_DEFINE_NS(4) SFL_DEFINE_NS(6)
#endif
#if SFL_NS_GROUP == 1 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(8)
#endif
#if SFL_NS_GROUP == 2 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(10)
#endif
#if SFL_NS_GROUP == 3 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(12)
#endif
#if SFL_NS_GROUP == 4 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(14)
#endif
#if SFL_NS_GROUP == 5 || SFL_NS_GROUP == -1
SFL_DEFINE_NS(16)
#endif

// The chained launch: its kernels live in translation units of their own (SFL_NS_GROUP 6: fuse 8 and 10, 7: fuse 12 and 16).
#define SFL_CHAIN_ARGS                                                                                                  \
    hipStream_t s, float *pa, float *pb, const float *d, Slab g, const ChainStep *steps, int n_steps, SorParams prm,   \
        int rows_per_chunk, int *flags, int flag_words, int epoch, int *timed_out, int max_waves, int *senders,        \
        int tiles_at_most, bool *launched
#define SFL_DEFINE_CHAIN_PART(N, P, DX1)                                                                                \
    hipError_t launch_sor_chain_ns##N##_p##P(SFL_CHAIN_ARGS)                                                            \
    {                                                                                                                  \
        return launch_chain_variant<Lane2<N, true, false, SFL_CHAIN_ST, SFL_CHAIN_LD>, N, DX1>(s, pa, pb, d, g, steps, n_steps, prm,        \
                                                                           rows_per_chunk, flags, flag_words, epoch,   \
                                                                           timed_out, max_waves, senders,              \
                                                                           tiles_at_most, launched);                   \
    }
#if SFL_DX_PART == 0
#define SFL_DEFINE_CHAIN(N) SFL_DEFINE_CHAIN_PART(N, 0, true)
#elif SFL_DX_PART == 1
#define SFL_DEFINE_CHAIN(N) SFL_DEFINE_CHAIN_PART(N, 1, false)
#else
#define SFL_DEFINE_CHAIN(N) SFL_DEFINE_CHAIN_PART(N, 0, true) SFL_DEFINE_CHAIN_PART(N, 1, false)
#endif
#define SFL_DECLARE_CHAIN(N)                                  \
    hipError_t launch_sor_chain_ns##N##_p0(SFL_CHAIN_ARGS);   \
    hipError_t launch_sor_chain_ns##N##_p1(SFL_CHAIN_ARGS);
SFL_DECLARE_CHAIN(8) SFL_DECLARE_CHAIN(10) SFL_DECLARE_CHAIN(12) SFL_DECLARE_CHAIN(16)
#if SFL_NS_GROUP == 6 || SFL_NS_GROUP == -1
SFL_DEFINE_CHAIN(8) SFL_DEFINE_CHAIN(10)
#endif
#if SFL_NS_GROUP == 7 || SFL_NS_GROUP == -1
SFL_DEFINE_CHAIN(12) SFL_DEFINE_CHAIN(16)
#endif

#if (SFL_NS_GROUP == 0 || SFL_NS_GROUP == -1) && SFL_DX_PART != 1
bool sor_chain_supported(const float *pa, const float *pb, const float *d, Slab g, int nsweeps)
{
    const uintptr_t all = reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb) | reinterpret_cast<uintptr_t>(d);
    return (nsweeps == 8 || nsweeps == 10 || nsweeps == 12 || nsweeps == 16) && g.dim_x % 2 == 0 && (all & 7) == 0 &&
           pa != nullptr && pb != nullptr && pa != pb && d != nullptr;
}

hipError_t launch_sor_chain(hipStream_t s, float *pa, float *pb, const float *d, Slab g, const ChainStep *steps, int n_steps,
                            int nsweeps, SorParams prm, int rows_per_chunk, int *flags, int flag_words, int epoch,
                            int *timed_out, int max_waves, int *senders, int tiles_at_most, bool *launched)
{
    if (launched) *launched = false;
    if (n_steps < 1 || n_steps > kMaxChain || !sor_chain_supported(pa, pb, d, g, nsweeps) || flags == nullptr || timed_out == nullptr)
        return hipErrorInvalidValue;
    for (int i = 0; i < n_steps; ++i)
        if (steps[i].g_end <= steps[i].g_begin) return hipErrorInvalidValue;
    const bool dx1 = prm.dx == 1.0f;
#define SFL_CASE(N)                                                                                                        \
    case N:                                                                                                                \
        return dx1 ? launch_sor_chain_ns##N##_p0(s, pa, pb, d, g, steps, n_steps, prm, rows_per_chunk, flags, flag_words,   \
                                                 epoch, timed_out, max_waves, senders, tiles_at_most, launched)            \
                   : launch_sor_chain_ns##N##_p1(s, pa, pb, d, g, steps, n_steps, prm, rows_per_chunk, flags, flag_words,   \
                                                 epoch, timed_out, max_waves, senders, tiles_at_most, launched);
    switch (nsweeps) {
        SFL_CASE(8) SFL_CASE(10) SFL_CASE(12) SFL_CASE(16)
    }
#undef SFL_CASE
    return hipErrorInvalidValue;
}
#endif

#if (SFL_NS_GROUP == 0 || SFL_NS_GROUP == -1) && SFL_DX_PART != 1
hipError_t launch_sor_fused(hipStream_t s, float *p_out, const float *p_in, const float *d,
                            Slab g, SorRows rows, int nsweeps, int first_colour,
                            SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    if (senders) *senders = 0;
    if (rows.g_end <= rows.g_begin && rows.g2_end <= rows.g2_begin) return hipSuccess;
    if (first_colour != 0 || nsweeps < 2 || nsweeps > SFL_MAX_FUSE || (nsweeps & 1) ||
        p_out == p_in || p_out == nullptr || d == nullptr)
        return hipErrorInvalidValue;
    const bool dx1 = prm.dx == 1.0f;
#define SFL_CASE(N)                                                                                              \
    case N:                                                                                                      \
        return dx1 ? launch_sor_fused_ns##N##_p0(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders) \
                   : launch_sor_fused_ns##N##_p1(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    switch (nsweeps) {
        SFL_CASE(2) SFL_CASE(4) SFL_CASE(6) SFL_CASE(8) SFL_CASE(10) SFL_CASE(12) SFL_CASE(14) SFL_CASE(16)
    }
#undef SFL_CASE
    return hipErrorInvalidValue;
}
#endif

}  // namespace sfl
