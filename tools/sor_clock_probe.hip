// sor_clock_probe.hip -- what clock does sor_fused_kernel really run at, and do its waves all live
// for the whole launch?  ("what binds the kernel": a clock64()-only reading said 1.22 GHz,
// GRBM_GUI_ACTIVE / wall said 2.1 GHz.)
//
// The kernel source is compiled here with SFL_SOR_TRACE: every wave records s_memtime (shader clock)
// and s_memrealtime (constant 100 MHz) at its start and end, plus HW_ID / XCC_ID.  From one launch:
//   * shader clock of each wave = d(s_memtime) / d(s_memrealtime) x 100 MHz  -- no assumption about
//     how long the wave lived;
//   * when each wave started and ended relative to the first wave of the launch: residency rounds,
//     tails, idle SIMDs.
// Build (one fuse depth per binary):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DSFL_NS_GROUP=5 -DPROBE_NS=16 \
//       tools/sor_clock_probe.hip -o tools/sor_clock_probe_ns16
// Run:  sor_clock_probe_ns16 <dim_x> <dim_y> [launches] [rows_per_chunk] [csv path] [loop seconds]
#define SFL_SOR_TRACE 1
#define SFL_DX_PART 0
#define SFL_FOLD_PART 0
#include "../esp32-fluid-simulation_amd/csrc/sor_fused.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#ifndef PROBE_NS
#define PROBE_NS 16
#endif
#define PASTE2(a, b, c) a##b##c
#define PASTE(a, b, c) PASTE2(a, b, c)
#define PROBE_LAUNCH PASTE(launch_sor_fused_ns, PROBE_NS, _p0_f0)

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static double pct(std::vector<double> v, double q)
{
    std::sort(v.begin(), v.end());
    return v[(size_t)(q * (v.size() - 1))];
}

int main(int argc, char **argv)
{
    const int dim_x = argc > 1 ? atoi(argv[1]) : 8192, dim_y = argc > 2 ? atoi(argv[2]) : 8192;
    const int launches = argc > 3 ? atoi(argv[3]) : 40;
    const int rpc = argc > 4 ? atoi(argv[4]) : 0;
    const char *csv = argc > 5 ? argv[5] : nullptr;
    const double loop_s = argc > 6 ? atof(argv[6]) : 0.0;
    const size_t cells = (size_t)dim_x * dim_y;
    float *d = nullptr, *p0 = nullptr, *p1 = nullptr;
    CK(hipMalloc(&d, cells * 4));
    CK(hipMalloc(&p0, cells * 4));
    CK(hipMalloc(&p1, cells * 4));
    {
        std::vector<float> h(cells);
        unsigned s = 12345;
        for (size_t k = 0; k < cells; ++k) {
            s = s * 1664525u + 1013904223u;
            h[k] = (float)((int)((s >> 8) % 2001) - 1000) / 1000.0f;
        }
        CK(hipMemcpy(d, h.data(), cells * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(p0, 0, cells * 4));
    CK(hipMemset(p1, 0, cells * 4));
    const size_t max_tiles = 1 << 16;
    unsigned long long *trace = nullptr;
    CK(hipMalloc(&trace, max_tiles * 6 * 8));
    CK(hipMemset(trace, 0, max_tiles * 6 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sfl::g_sor_trace), &trace, sizeof trace));

    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const sfl::Slab g{dim_x, dim_y, 0, dim_y};
    const sfl::SorRows rows{0, dim_y, 0, 0};
    sfl::SorParams prm{1.0f, 1.96f, 1.0f - 1.96f, -0.25f * 1.96f};
    const int alternate = getenv("PROBE_ALTERNATE") ? atoi(getenv("PROBE_ALTERNATE")) : 0;
    int n_launch = 0;
    auto launch = [&](float *out, const float *in) {
        return sfl::PROBE_LAUNCH(st, out, in, d, g, rows, prm, rpc, alternate ? n_launch++ : 0, nullptr, nullptr);
    };

    if (loop_s > 0) {  // a steady load for clock polling from outside (rocm-smi / amd-smi)
        const auto t0 = std::chrono::steady_clock::now();
        long n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < loop_s) {
            for (int k = 0; k < 20; ++k) {
                CK(launch(p1, p0));
                std::swap(p0, p1);
            }
            CK(hipStreamSynchronize(st));
            n += 20;
        }
        printf("loop: %ld launches in %.2f s\n", n, loop_s);
    }
    for (int k = 0; k < launches; ++k) {  // priming
        CK(launch(p1, p0));
        std::swap(p0, p1);
    }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < launches; ++k) {
        CK(launch(p1, p0));
        std::swap(p0, p1);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double launch_us = ms * 1e3 / launches;

    std::vector<unsigned long long> h(max_tiles * 6);
    CK(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
    size_t n = 0;
    while (n < max_tiles && h[6 * n + 1] != 0) ++n;
    if (n == 0) {
        fprintf(stderr, "no trace records\n");
        return 1;
    }
    unsigned long long wbase = ~0ull, wlast = 0;
    for (size_t k = 0; k < n; ++k) {
        wbase = std::min(wbase, h[6 * k + 2]);
        wlast = std::max(wlast, h[6 * k + 3]);
    }
    std::vector<double> ghz, life_us, start_us, end_us, cyc;
    std::map<unsigned long long, int> per_simd;
    std::vector<double> kind_life[3], kind_cyc[3];
    for (size_t k = 0; k < n; ++k) {
        const double dt = (double)(h[6 * k + 1] - h[6 * k + 0]), dw = (double)(h[6 * k + 3] - h[6 * k + 2]);
        const int kind = (int)(h[6 * k + 5] >> 32);
        ghz.push_back(dw > 0 ? dt / dw * 0.1 : 0);
        life_us.push_back(dw * 0.01);
        cyc.push_back(dt);
        start_us.push_back((double)(h[6 * k + 2] - wbase) * 0.01);
        end_us.push_back((double)(h[6 * k + 3] - wbase) * 0.01);
        kind_life[kind].push_back(dw * 0.01);
        kind_cyc[kind].push_back(dt);
        // SIMD identity: xcc, se, sh, cu, simd bits of HW_ID (wave slot bits [3:0] dropped)
        const unsigned long long key = ((h[6 * k + 5] & 0xffffffffull) << 32) | (h[6 * k + 4] & 0xfff0ull & ~0xc0ull) |
                                       (h[6 * k + 4] & 0xe000ull);
        ++per_simd[key];
    }
    printf("grid %d x %d, NS %d, rows_per_chunk %d: %zu waves traced; launch by HIP events %.2f us\n", dim_x, dim_y,
           PROBE_NS, rpc, n, launch_us);
    printf("launch span on the 100 MHz clock (first wave start -> last wave end): %.2f us\n",
           (double)(wlast - wbase) * 0.01);
    printf("shader clock per wave, GHz  (d s_memtime / d s_memrealtime): min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f\n",
           pct(ghz, 0), pct(ghz, 0.1), pct(ghz, 0.5), pct(ghz, 0.9), pct(ghz, 1));
    printf("wave lifetime, us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", pct(life_us, 0), pct(life_us, 0.1),
           pct(life_us, 0.5), pct(life_us, 0.9), pct(life_us, 1));
    printf("wave lifetime, shader cycles: min %.0f  median %.0f  max %.0f\n", pct(cyc, 0), pct(cyc, 0.5), pct(cyc, 1));
    printf("wave start after launch start, us: p10 %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f\n", pct(start_us, 0.1),
           pct(start_us, 0.5), pct(start_us, 0.9), pct(start_us, 0.99), pct(start_us, 1));
    printf("wave end after launch start, us:   min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", pct(end_us, 0),
           pct(end_us, 0.1), pct(end_us, 0.5), pct(end_us, 0.9), pct(end_us, 1));
    const char *kn[3] = {"interior bottom-up", "boundary (EDGE)", "interior top-down"};
    for (int k = 0; k < 3; ++k)
        if (!kind_life[k].empty())
            printf("  %-20s %5zu waves: lifetime median %.1f us (max %.1f), %.0f cycles\n", kn[k], kind_life[k].size(),
                   pct(kind_life[k], 0.5), pct(kind_life[k], 1), pct(kind_cyc[k], 0.5));
    {
        std::map<int, int> hist;
        for (auto &kv : per_simd) ++hist[kv.second];
        printf("distinct SIMDs seen %zu; waves per SIMD histogram:", per_simd.size());
        for (auto &kv : hist) printf("  %d waves: %d SIMDs", kv.first, kv.second);
        printf("\n");
    }
    // how much of the launch is a SIMD busy with at least one wave: sum of lifetimes / (SIMDs x span)
    {
        double sum = 0;
        for (double x : life_us) sum += x;
        printf("mean waves alive over the span: %.1f (of %zu)\n", sum / ((double)(wlast - wbase) * 0.01), n);
    }
    {   // the slowest waves and what they were streaming (the tiling the launcher chose, recomputed here)
        using B = sfl::Lane2<PROBE_NS, true, false, false>;
        const int waves = sfl::resident_waves<B, PROBE_NS, true, false>();
        const int use_rpc = rpc > 0 ? rpc : sfl::auto_rows_per_chunk<B>(g, 0, dim_y, PROBE_NS, waves, sfl::device_simds(), sfl::sor::kEdgeRowCost16);
        const sfl::sor::Tiling t = sfl::sor::make_tiling(PROBE_NS, B::kTileCols, B::kColAlign, dim_x, dim_y, 0, dim_y, use_rpc,
                                                         sfl::sor::kEdgeRowCost16, 1);
        printf("tiling: rows per tile %d (boundary strips %d, first / last chunk %d / %d), %d strips (%d inner), %d tiles\n",
               t.rows_per_chunk, t.rows_edge, t.rows_first, t.rows_last, t.n_strips, t.n_inner, t.n_tiles);
        std::vector<std::pair<double, size_t>> order;
        for (size_t k = 0; k < n; ++k) order.push_back({end_us[k], k});
        std::sort(order.rbegin(), order.rend());
        for (int k = 0; k < 12 && k < (int)order.size(); ++k) {
            const size_t tile = order[k].second;
            const sfl::sor::TileRect r = sfl::sor::tile_rect(t, (int)tile);
            printf("  end %.1f us  life %.1f us  tile %zu kind %llu  strip %d rows [%d, %d) = %d\n", order[k].first, life_us[tile], tile,
                   h[6 * tile + 5] >> 32, r.strip, r.r0, r.r1, r.r1 - r.r0);
        }
        // by class: inner strips' first / last chunk, boundary strips, the rest
        double worst[4] = {0, 0, 0, 0};
        const char *cls[4] = {"interior", "inner strip, first chunk", "inner strip, last chunk", "boundary strip"};
        for (size_t k = 0; k < n; ++k) {
            const sfl::sor::TileRect r = sfl::sor::tile_rect(t, (int)k);
            const bool inner = (int)k < t.n_inner * t.n_chunks;
            const int c = !inner ? 3 : (r.r0 == 0 ? 1 : (r.r1 == dim_y ? 2 : 0));
            worst[c] = std::max(worst[c], end_us[k]);
        }
        for (int c = 0; c < 4; ++c) printf("  last wave of class '%s' ends at %.1f us\n", cls[c], worst[c]);
    }
    if (csv) {
        FILE *f = fopen(csv, "w");
        if (f) {
            fprintf(f, "tile,t0,t1,w0,w1,hwid,xcc,kind\n");
            for (size_t k = 0; k < n; ++k)
                fprintf(f, "%zu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", k, h[6 * k], h[6 * k + 1], h[6 * k + 2] - wbase,
                        h[6 * k + 3] - wbase, h[6 * k + 4], h[6 * k + 5] & 0xffffffffull, h[6 * k + 5] >> 32);
            fclose(f);
        }
    }
    return 0;
}
