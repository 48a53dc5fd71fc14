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

#include "sor_lane.h"


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
// The SFL_PROBE_* switches compute WRONG results; they exist for
// tools/sor_clock_probe.hip, which includes this file with SFL_SOR_TRACE defined.  A product library never sees them set:
// `make EXTRA_FLAGS=-DSFL_PROBE_NO_LOAD=1` stops here.
#ifdef SFL_PROBE_COOP
#error "SFL_PROBE_COOP is a timing mock (wrong results): only with SFL_SOR_TRACE (tools/sor_clock_probe.hip)"
#endif
static_assert(SFL_PROBE_NO_LDS == 0 && SFL_PROBE_NO_LOAD == 0 && SFL_PROBE_SHIFT == 0 && SFL_PROBE_NO_EDGE == 0 &&
                  SFL_PROBE_P_LOAD_AUX == 0 && SFL_PROBE_P_STORE_AUX == 0 && SFL_PROBE_NO_STORE == 0,
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
        sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega), bk.splat(prm.neg_quarter_omega)};
        const auto eca = bk.edge_cell(lane, x0, 0);
        const auto ecb = bk.edge_cell(lane, x0, 1);
        sor::stream_tile<B, NS, true, DX1, ZERO_IN>(bk, c, eca, ecb, r0, r1);
        return 1;
    }
    if (sor::tile_may_flip(t, rect)) {  // streamed top-down: pipeline index = -row
        B bk = backend();
        bk.row_sign = -1;
        sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega), bk.splat(prm.neg_quarter_omega)};
        const sor::EdgeCell<B> none{};
        sor::stream_tile<B, NS, false, DX1, ZERO_IN, true>(bk, c, none, none, 1 - r1, 1 - r0);
        return 2;
    }
    B bk = backend();
    sor::Consts<B> c{bk.splat(prm.dx), bk.splat(prm.omega), bk.splat(prm.one_minus_omega), bk.splat(prm.neg_quarter_omega)};
    const sor::EdgeCell<B> none{};
    sor::stream_tile<B, NS, false, DX1, ZERO_IN>(bk, c, none, none, r0, r1);
    return 0;
}

template <class B, int NS, bool DX1, bool ZERO_IN>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd(NS))))
sor_fused_kernel(float *p_out, const float *p_in, const float *d, Slab g, sor::Tiling t1,
                 sor::Tiling t2, SorParams prm, HaloWait hw, int rot_c, int rot_e, int free_blocks)
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
    // XCDs and come LAST (sor::tile_rect has the reason); the contiguous ranges are dealt over the blocks in front.
    int block = blockIdx.x;
    if (block < free_blocks) {
        const int per = free_blocks >> 3, rem = free_blocks & 7;
        const int xcd = block & 7, idx = block >> 3;
        block = xcd * per + min(xcd, rem) + idx;  // bijective on [0, free_blocks) for every count
    }
    // a launch covers up to two row ranges (the two cut-adjacent bands of a slab in one launch):
    // the tiles of the second tiling follow those of the first
    int tile = block * kWavesPerBlock + wave;
    if (tile >= t1.n_tiles + t2.n_tiles) return;
#ifdef SFL_SOR_TRACE
    WaveTrace trace;
    trace.begin();
    const int trace_tile = tile;
    int trace_kind = 0;
#endif
    const bool second = tile >= t1.n_tiles;  // wave-uniform
    const sor::Tiling t = second ? t2 : t1;
    if (second) tile -= t1.n_tiles;
    const sor::TileRect rect = sor::tile_rect(t, tile, second ? 0 : rot_c, second ? 0 : rot_e);
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
int auto_rows_per_chunk(const Slab &g, int g_begin, int g_end, int ns, int waves, int simds, int edge_cost16)
{
    const int rows = g_end - g_begin;
    int best_rows = rows;
    double best_cost = 1e300;
    const int max_chunks = (rows + 7) / 8;
    for (int chunks = 1; chunks <= max_chunks; ++chunks) {
        const int rpc = (rows + chunks - 1) / chunks;
        const sor::Tiling t = sor::make_tiling(ns, B::kTileCols, B::kColAlign, g.dim_x, g.gdim_y, g_begin,
                                               g_end, rpc, edge_cost16, kFlipTiles);
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
        const int resident = resident_waves<B, NS, DX1, ZERO_IN>();
        auto with_edge_cost = [&](int cost16) {
            int rpc = rows_per_chunk > g_end - g_begin ? g_end - g_begin : rows_per_chunk;
            if (rpc <= 0) rpc = auto_rows_per_chunk<B>(g, g_begin, g_end, NS, resident, device_simds(), cost16);
            return sor::make_tiling(NS, B::kTileCols, B::kColAlign, g.dim_x, g.gdim_y, g_begin, g_end, rpc, cost16,
                                    kFlipTiles ? 1 + (sweep & 1) : 0);
        };
        const sor::Tiling t = with_edge_cost(sor::kEdgeRowCost16);
        // More tiles than wave slots (one GPU from ~12288^2 up): a boundary tile that ends early hands its slot to a waiting
        // tile instead of idling beside its SIMD's other waves, so boundary tiles are cut shorter still (16384^2 x 200:
        // 18.16 -> 17.7 ms per solve; with every tile resident the same cut costs 0.5 - 2 %: profiles/r05_fold_quarter_omega.txt)
        return t.n_tiles > resident ? with_edge_cost(sor::kEdgeRowCostQueued16) : t;
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
    // dispatch order (sor::tile_rect): nothing special unless tiles of this launch will wait inside it -- then the chunks that do
    // not wait first (the same test as the kernel's, on the chunks of an inner strip and of a boundary strip), and only their
    // blocks in the XCD-contiguous deal
    int rot_c = 0, rot_e = 0, free_blocks = blocks;
    if (hw.flag != nullptr && t2.n_tiles == 0) {
        const int line_rows = (g.dim_x & 63) ? 1 + 63 / g.dim_x : 0;
        const int reach = NS + sor::ring_rows(NS) + line_rows;
        auto free_range = [&](int strip, int n_chunks, int *c0, int *count) {
            *c0 = *count = 0;
            for (int c = 0; c < n_chunks; ++c) {
                const sor::TileRect r = sor::tile_rect(t1, sor::tile_index(t1, strip, c));
                if (r.r0 - reach < hw.own_lo || r.r1 + reach > hw.own_hi) continue;
                if (*count == 0) *c0 = c;
                ++*count;
            }
        };
        int fc = 0, fe = 0;
        if (t1.n_inner > 0) free_range(1, t1.n_chunks, &rot_c, &fc);
        free_range(0, t1.n_chunks_edge, &rot_e, &fe);
        free_blocks = fc * t1.n_inner / kWavesPerBlock;
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
    sor_fused_kernel<B, NS, DX1, ZERO_IN><<<blocks, kThreads, 0, s>>>(p_out, p_in, d, g, t1, t2, prm, hw, rot_c, rot_e, free_blocks);
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

template <int NS, bool ZERO_IN, bool FOLD>
hipError_t launch_lane(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                       SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    const uintptr_t all = reinterpret_cast<uintptr_t>(p_out) | reinterpret_cast<uintptr_t>(p_in) |
                          reinterpret_cast<uintptr_t>(d);
    const bool can2v = (g.dim_x % 2 == 0) && (all & 7) == 0;
    if (can2v) {
        // written through when sender tiles publish rows of this launch while it runs (see Lane2)
        if (wait && wait->done)
            return launch_dx<Lane2<NS, true, ZERO_IN, 16, FOLD>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
        // non-temporal stores once the slab's arrays no longer fit the caches
        if ((size_t)g.lrows * (size_t)g.dim_x >= kNtStoreCells)
            return launch_dx<Lane2<NS, true, ZERO_IN, 2, FOLD>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
        return launch_dx<Lane2<NS, true, ZERO_IN, 0, FOLD>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    }
    return launch_dx<Lane2<NS, false, ZERO_IN, 0, FOLD>, NS, ZERO_IN>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
}

template <int NS, bool FOLD>
hipError_t launch_ns(hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g,
                     SorRows rows, SorParams prm, int rows_per_chunk, int sweep, const HaloWait *wait, int *senders)
{
    if (p_in == nullptr)
        return launch_lane<NS, true, FOLD>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    return launch_lane<NS, false, FOLD>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
}

}  // namespace

// One non-template entry per fuse depth, per half (dx == 1 / any dx) and per arithmetic (exact / folded quarter); the depths, halves
// and arithmetics are spread over translation units (SFL_NS_GROUP = 0..5, SFL_DX_PART = 0 / 1, SFL_FOLD_PART = 0 / 1, see
// csrc/Makefile) so that they compile in parallel.
#ifndef SFL_NS_GROUP
#define SFL_NS_GROUP (-1)  // single translation unit: every depth
#endif
#ifndef SFL_FOLD_PART
#define SFL_FOLD_PART (-1)  // both arithmetics in this translation unit
#endif
#define SFL_ENTRY_ARGS                                                                             \
    hipStream_t s, float *p_out, const float *p_in, const float *d, Slab g, SorRows rows, SorParams prm, \
        int rows_per_chunk, int sweep, const HaloWait *wait, int *senders
#define SFL_DEFINE_PART(N, P, F)                                                                   \
    hipError_t launch_sor_fused_ns##N##_p##P##_f##F(SFL_ENTRY_ARGS)                                \
    {                                                                                             \
        return launch_ns<N, F != 0>(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders); \
    }
#if SFL_DX_PART == 0
#define SFL_DEFINE_DX(N, F) SFL_DEFINE_PART(N, 0, F)
#elif SFL_DX_PART == 1
#define SFL_DEFINE_DX(N, F) SFL_DEFINE_PART(N, 1, F)
#else
#define SFL_DEFINE_DX(N, F) SFL_DEFINE_PART(N, 0, F) SFL_DEFINE_PART(N, 1, F)
#endif
#if SFL_FOLD_PART == 0
#define SFL_DEFINE_NS(N) SFL_DEFINE_DX(N, 0)
#elif SFL_FOLD_PART == 1
#define SFL_DEFINE_NS(N) SFL_DEFINE_DX(N, 1)
#else
#define SFL_DEFINE_NS(N) SFL_DEFINE_DX(N, 0) SFL_DEFINE_DX(N, 1)
#endif
#define SFL_DECLARE_NS(N)                                                                         \
    hipError_t launch_sor_fused_ns##N##_p0_f0(SFL_ENTRY_ARGS);                                    \
    hipError_t launch_sor_fused_ns##N##_p1_f0(SFL_ENTRY_ARGS);                                    \
    hipError_t launch_sor_fused_ns##N##_p0_f1(SFL_ENTRY_ARGS);                                    \
    hipError_t launch_sor_fused_ns##N##_p1_f1(SFL_ENTRY_ARGS);
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

#if (SFL_NS_GROUP == 0 || SFL_NS_GROUP == -1) && SFL_DX_PART != 1 && SFL_FOLD_PART != 1
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
        if (prm.fold)                                                                                            \
            return dx1 ? launch_sor_fused_ns##N##_p0_f1(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders) \
                       : launch_sor_fused_ns##N##_p1_f1(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders); \
        return dx1 ? launch_sor_fused_ns##N##_p0_f0(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders) \
                   : launch_sor_fused_ns##N##_p1_f0(s, p_out, p_in, d, g, rows, prm, rows_per_chunk, sweep, wait, senders);
    switch (nsweeps) {
        SFL_CASE(2) SFL_CASE(4) SFL_CASE(6) SFL_CASE(8) SFL_CASE(10) SFL_CASE(12) SFL_CASE(14) SFL_CASE(16)
    }
#undef SFL_CASE
    return hipErrorInvalidValue;
}
#endif

}  // namespace sfl
