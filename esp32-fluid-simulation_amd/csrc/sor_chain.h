// sor_chain.h -- CHAINED supersteps (kernels.h launch_sor_chain; SFL_OPT_SOR_CHAIN): up to 16 consecutive launches of the fused SOR
// kernel as one launch whose waves walk from superstep to superstep, each tile waiting only for the tiles around it.  Included by
// sor_fused.hip behind launch_variant (it uses relax_tile, auto_rows_per_chunk and device_simds of that file); an option, not the
// default (profiles/r04_chained_launch.txt).
#pragma once

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
        if (rpc <= 0) rpc = auto_rows_per_chunk<B>(g, st.g_begin, st.g_end, NS, resident, device_simds(), sor::kEdgeRowCost16);
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
