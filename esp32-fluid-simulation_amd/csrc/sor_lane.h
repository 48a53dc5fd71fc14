// sor_lane.h -- the gfx950 backend of the fused red-black SOR pipeline (sor_stream_core.h): what one 64-lane wavefront keeps and does
// per row.  W / E neighbours by DPP wave shifts, S / N neighbours in VGPRs, loads and stores through buffer resources with a
// loop-invariant lane offset and an SGPR row offset, the right-hand side in a per-lane LDS ring, rotating issue priority.
// Included by sor_fused.hip (inside namespace sfl, anonymous namespace); reference: poisson.cpp:63-112.
#pragma once

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
#ifndef SFL_PROBE_NO_STORE
#define SFL_PROBE_NO_STORE 0  // diagnostic builds only: the finished rows are not stored (what would a last launch
                              // of a solve cost whose pressure nobody reads from memory? profiles/r06_pressure_never_stored.txt)
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
// FOLD = the interior relaxation's one product by -0.25f * omega (SFL_OPT_SOR_FOLD = 1; sor_stream_core.h relax); false: the
// reference's two products, its bits on every input.
template <int NS, bool VEC, bool ZERO_IN, int ST = 0, bool FOLD = false>
struct Lane2 : WaveCommon {
    using V = float;
    using M = bool;
    static constexpr bool kFoldQuarter = FOLD;
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
                const v2f q = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs_p, off_a, soff, SFL_PROBE_P_LOAD_AUX));
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
        if (SFL_PROBE_NO_STORE) {   // (the values stay "used": the relaxations are not optimised away)
            asm volatile("" ::"v"(a), "v"(b));
            return;
        }
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
#include "sor_probe_mocks.inc"
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
