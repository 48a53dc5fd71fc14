// host_san_driver.cpp -- TEST HARNESS (tests/cpp, `make san_host`; tests/test_host_sanitizers.py): the HOST side of the library
// -- csrc/slab_plan.cpp, the plan / slab / option / argument-check entry points of the C ABI -- built with
// -fsanitize=address,undefined and driven without a GPU (the kernels are stubs: launch_stubs.cpp).
//   1. sfl_slab_rows / sfl_sor_pass_plan / sfl_plan_poisson / sfl_plan_poisson_tail over a seeded fuzz of
//      (dim_y, nranks, iters, fuse, kernel, halo, tail), every rank: the invariants an executor relies on --
//      slabs partition the rows; every rank's program has the same shape; launches cover the owned rows; no launch and no
//      exchange reaches beyond the ghost rows a context allocates (160) or the thinnest slab; passes add up to 2 x iters;
//      a launch never needs rows of p that no exchange or earlier launch left exact;
//   2. the argument checks: every bad call comes back as SFL_ERR_INVALID / SFL_ERR_STATE with a message, NULL outputs
//      are refused, capacity-limited writes stay inside the caller's array;
//   3. what exists without a device: contexts are refused with SFL_ERR_HIP / SFL_ERR_INVALID and leave *out NULL.
// Exit status 0 = no finding; the sanitizers abort the process on theirs.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/sfl.h"

static int failures = 0;
static long validity_checks = 0;
#define CHECK(cond, ...)                                     \
    do {                                                     \
        if (!(cond)) {                                       \
            if (failures++ < 20) {                           \
                fprintf(stderr, "CHECK failed: %s -- ", #cond); \
                fprintf(stderr, __VA_ARGS__);                \
                fprintf(stderr, "\n");                       \
            }                                                \
        }                                                    \
    } while (0)

static unsigned long long state = 88172645463325252ull;
static unsigned rnd()
{
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return (unsigned)(state >> 11);
}
static int between(int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); }

constexpr int kGhost = 160;   // csrc/context.h kGhostRows
#define SFL_MAX_FUSE 16        // csrc/kernels.h (the deepest fusion the ABI accepts: include/sfl.h SFL_OPT_SOR_FUSE)

static std::vector<sfl_plan_step> plan(int dim_y, int nranks, int rank, int iters, int fuse, int kernel, int halo, int tail)
{
    int n = -1;
    const int rc = sfl_plan_poisson_tail(dim_y, nranks, rank, iters, fuse, kernel, halo, tail, nullptr, 0, &n);
    CHECK(rc == SFL_OK && n >= 0, "plan query failed: %s", sfl_last_error());
    std::vector<sfl_plan_step> v((size_t)std::max(n, 1) + 2);
    const sfl_plan_step canary{-77, -77, -77, -77, -77, -77, -77, -77};
    v[(size_t)n] = v[(size_t)n + 1] = canary;
    int m = -1;
    CHECK(sfl_plan_poisson_tail(dim_y, nranks, rank, iters, fuse, kernel, halo, tail, v.data(), n, &m) == SFL_OK && m == n, "second query");
    CHECK(v[(size_t)n].kind == -77 && v[(size_t)n + 1].kind == -77, "plan wrote beyond the capacity it was given");
    if (n > 1) {   // a smaller capacity: only that many steps are written
        std::vector<sfl_plan_step> w((size_t)n, canary);
        CHECK(sfl_plan_poisson_tail(dim_y, nranks, rank, iters, fuse, kernel, halo, tail, w.data(), n / 2, &m) == SFL_OK && m == n, "capped query");
        CHECK(w[(size_t)n / 2].kind == -77, "capped plan wrote step %d", n / 2);
    }
    v.resize((size_t)n);
    return v;
}

static void fuzz_plans(int cases)
{
    for (int c = 0; c < cases; ++c) {
        const int nranks = between(1, 9);
        const int dim_y = between(std::max(2, nranks), c % 7 == 0 ? 20000 : 3000);
        const int iters = c % 11 == 0 ? 0 : between(1, 220);
        const int kernel = between(1, 3);
        const int fuse = 2 * between(1, SFL_MAX_FUSE / 2);
        const int halo = c % 5 == 0 ? 0 : between(0, 200);
        const int tail = c % 3 == 0 ? between(0, 3) : 0;
        int thinnest = dim_y, prev_end = 0;
        for (int r = 0; r < nranks; ++r) {
            int b = -1, e = -1;
            CHECK(sfl_slab_rows(dim_y, nranks, r, &b, &e) == SFL_OK, "slab rows");
            CHECK(b == prev_end && e >= b && e <= dim_y, "slabs must partition the rows: rank %d of %d owns [%d, %d) after %d", r, nranks, b, e, prev_end);
            prev_end = e;
            thinnest = std::min(thinnest, e - b);
        }
        CHECK(prev_end == dim_y, "the last slab ends at %d of %d", prev_end, dim_y);
        std::vector<sfl_plan_step> first;
        for (int r = 0; r < nranks; ++r) {
            const std::vector<sfl_plan_step> p = plan(dim_y, nranks, r, iters, fuse, kernel, halo, tail);
            int g0, g1;
            sfl_slab_rows(dim_y, nranks, r, &g0, &g1);
            if (r == 0) first = p;
            CHECK(p.size() == first.size(), "rank %d: %zu steps, rank 0 has %zu", r, p.size(), first.size());
            int passes = 0;
            // how deep the ghost rows of p / of the right-hand side are EXACT on a side that has a neighbour (p starts at zero
            // everywhere: exact to any depth until the first launch has run)
            long valid_p = 1 << 30, valid_d = 0;
            for (size_t k = 0; k < p.size() && k < first.size(); ++k) {
                const sfl_plan_step &s = p[k];
                // (depths the product itself would use: effective_halo clamps to the ghost rows and to the thinnest slab, and the
                // executor refuses an exchange beyond either)
                if (nranks > 1 && kernel >= 2 && std::max(halo, fuse) + tail <= std::min(kGhost, thinnest)) {
                    if (s.kind == SFL_STEP_EXCHANGE && s.field == SFL_FIELD_DIVERGENCE) valid_d = s.g_begin + s.rows;
                    if (s.kind == SFL_STEP_EXCHANGE && s.field == SFL_FIELD_PRESSURE) {
                        CHECK(valid_p >= s.g_begin, "rank %d step %zu: an exchange skips %d ghost rows of which only %ld are exact", r, k, s.g_begin, valid_p);
                        valid_p = s.g_begin + s.rows;
                    }
                    if (s.kind == SFL_STEP_SOR) {
                        for (int side = 0; side < 2; ++side) {
                            if ((side == 0 && r == 0) || (side == 1 && r == nranks - 1)) continue;   // the domain's own boundary
                            const long ext = side == 0 ? g0 - s.g_begin : s.g_end - g1;
                            ++validity_checks;
                            CHECK(s.from_zero || ext + s.nsweeps <= valid_p,
                                  "rank %d step %zu: a launch of %d passes that keeps %ld ghost rows exact reads p %ld deep, %ld are exact (dim_y %d ranks %d iters %d fuse %d kernel %d halo %d tail %d)",
                                  r, k, s.nsweeps, ext, ext + s.nsweeps, valid_p, dim_y, nranks, iters, fuse, kernel, halo, tail);
                            CHECK(ext + s.nsweeps - 1 <= valid_d, "rank %d step %zu: the right-hand side is read %ld rows deep, %ld were exchanged", r, k, ext + s.nsweeps - 1, valid_d);
                        }
                        valid_p = std::min<long>(g0 - s.g_begin + (r == 0 ? 1 << 20 : 0), s.g_end - g1 + (r == nranks - 1 ? 1 << 20 : 0));
                    }
                }
                CHECK(s.kind == first[k].kind && s.field == first[k].field && s.rows == first[k].rows && s.nsweeps == first[k].nsweeps,
                      "rank %d step %zu differs in shape from rank 0's", r, k);
                if (s.kind == SFL_STEP_EXCHANGE) {
                    CHECK(nranks > 1, "an exchange on a whole domain");
                    CHECK(s.rows >= 1 && s.g_begin >= 0, "exchange of %d rows at depth %d", s.rows, s.g_begin);
                    // (the executor refuses deeper ones -- transport.cpp exchange() -- so a plan for the product's own halo
                    // depths, which effective_halo clamps to the ghost rows and the thinnest slab, must stay inside both)
                    if (halo <= kGhost && halo <= thinnest && kernel >= 2)
                        CHECK(s.g_begin + s.rows <= std::max(halo, fuse) + tail, "exchange reaches %d rows deep at halo %d fuse %d tail %d", s.g_begin + s.rows, halo, fuse, tail);
                } else if (s.kind == SFL_STEP_SOR) {
                    passes += s.nsweeps;
                    CHECK(s.g_begin <= g0 && s.g_end >= g1, "launch [%d, %d) does not cover the owned rows [%d, %d)", s.g_begin, s.g_end, g0, g1);
                    CHECK(s.g_begin >= 0 && s.g_end <= dim_y, "launch rows [%d, %d) outside the domain of %d", s.g_begin, s.g_end, dim_y);
                    if (halo <= kGhost && halo <= thinnest && kernel >= 2)
                        CHECK(g0 - s.g_begin <= std::max(halo, fuse) + tail && s.g_end - g1 <= std::max(halo, fuse) + tail,
                              "launch [%d, %d) reaches beyond the halo of slab [%d, %d)", s.g_begin, s.g_end, g0, g1);
                    CHECK(s.nsweeps >= 1 && (kernel == 1 ? s.nsweeps == 1 : (s.nsweeps <= fuse)), "launch of %d passes at fuse %d", s.nsweeps, fuse);
                } else {
                    CHECK(s.kind == SFL_STEP_ZERO && kernel == 1, "unknown step kind %d", s.kind);
                }
            }
            CHECK(passes == 2 * iters, "%d colour passes planned for %d iterations", passes, iters);
        }
        int n = -1;
        std::vector<int> pl(64, -5);
        CHECK(sfl_sor_pass_plan(iters, fuse, &n, pl.data(), 8) == SFL_OK && n >= 0, "pass plan");
        for (int k = 8; k < 64; ++k) CHECK(pl[(size_t)k] == -5, "pass plan wrote beyond its capacity");
    }
}

static void bad_arguments()
{
    int a = 0, b = 0, n = 0;
    sfl_plan_step st[4];
    CHECK(sfl_slab_rows(0, 1, 0, &a, &b) == SFL_ERR_INVALID, "dim_y 0");
    CHECK(sfl_slab_rows(10, 0, 0, &a, &b) == SFL_ERR_INVALID, "nranks 0");
    CHECK(sfl_slab_rows(10, 2, 2, &a, &b) == SFL_ERR_INVALID, "rank == nranks");
    CHECK(sfl_slab_rows(10, 2, -1, &a, &b) == SFL_ERR_INVALID, "rank -1");
    CHECK(sfl_slab_rows(10, 2, 0, nullptr, &b) == SFL_ERR_INVALID && sfl_slab_rows(10, 2, 0, &a, nullptr) == SFL_ERR_INVALID, "NULL outputs");
    CHECK(sfl_sor_pass_plan(-1, 8, &n, nullptr, 0) == SFL_ERR_INVALID, "iters -1");
    CHECK(sfl_sor_pass_plan(4, 7, &n, nullptr, 0) == SFL_ERR_INVALID && sfl_sor_pass_plan(4, 0, &n, nullptr, 0) == SFL_ERR_INVALID &&
              sfl_sor_pass_plan(4, SFL_MAX_FUSE + 2, &n, nullptr, 0) == SFL_ERR_INVALID, "odd / zero / too deep fuse");
    CHECK(sfl_sor_pass_plan(4, 8, nullptr, nullptr, 0) == SFL_ERR_INVALID, "NULL n_passes");
    CHECK(sfl_plan_poisson(1, 1, 0, 4, 8, 2, 0, st, 4, &n) == SFL_ERR_INVALID, "dim_y 1");
    CHECK(sfl_plan_poisson(64, 2, 2, 4, 8, 2, 0, st, 4, &n) == SFL_ERR_INVALID, "rank out of range");
    CHECK(sfl_plan_poisson(64, 2, 0, -1, 8, 2, 0, st, 4, &n) == SFL_ERR_INVALID, "iters -1");
    CHECK(sfl_plan_poisson(64, 2, 0, 4, 8, 0, 0, st, 4, &n) == SFL_ERR_INVALID && sfl_plan_poisson(64, 2, 0, 4, 8, 4, 0, st, 4, &n) == SFL_ERR_INVALID, "kernel 0 / 4");
    CHECK(sfl_plan_poisson(64, 2, 0, 4, 3, 2, 0, st, 4, &n) == SFL_ERR_INVALID, "odd fuse");
    CHECK(sfl_plan_poisson(64, 2, 0, 4, 8, 2, -1, st, 4, &n) == SFL_ERR_INVALID, "negative halo");
    CHECK(sfl_plan_poisson(64, 2, 0, 4, 8, 2, 0, st, 4, nullptr) == SFL_ERR_INVALID, "NULL n_steps");
    CHECK(sfl_plan_poisson_tail(64, 2, 0, 4, 8, 2, 16, -1, st, 4, &n) == SFL_ERR_INVALID, "negative tail");
    CHECK(strlen(sfl_last_error()) > 0, "a failing call leaves a message");
    CHECK(sfl_abi_version() == SFL_ABI_VERSION, "ABI version");

    // contexts: argument checks come before the device is looked at; without a device the rest is SFL_ERR_HIP
    sfl_context *ctx = reinterpret_cast<sfl_context *>(0x1);
    CHECK(sfl_create(nullptr, 0, 64, 64) == SFL_ERR_INVALID, "NULL out");
    CHECK(sfl_create(&ctx, 0, 1, 64) == SFL_ERR_INVALID && ctx == nullptr, "dim_x 1 must be refused and *out cleared");
    ctx = reinterpret_cast<sfl_context *>(0x1);
    CHECK(sfl_create_slab(&ctx, 0, 64, 64, 3, 2) == SFL_ERR_INVALID && ctx == nullptr, "rank 3 of 2");
    CHECK(sfl_create_slab(&ctx, 0, 64, 4, 0, 8) == SFL_ERR_INVALID, "more slabs than rows");
    CHECK(sfl_create_slab(&ctx, 0, 32768, 16384, 0, 1) == SFL_ERR_INVALID, "2^29 cells in one context");
    CHECK(sfl_create_slab(&ctx, 0, 65536, 32768, 0, 8) == SFL_ERR_INVALID, "a domain beyond 2^30 cells");
    const int rc = sfl_create(&ctx, 0, 64, 64);
    CHECK((rc == SFL_OK) == (ctx != nullptr), "status and *out agree");
    if (rc == SFL_OK) {   // (a box with a GPU: the option checks run on a real context)
        int v = 0;
        CHECK(sfl_set_option(ctx, SFL_OPT_SOR_FUSE, 7) == SFL_ERR_INVALID && sfl_set_option(ctx, SFL_OPT_SOR_FUSE, 18) == SFL_ERR_INVALID, "fuse 7 / 18");
        CHECK(sfl_set_option(ctx, SFL_OPT_SOR_HALO, 161) == SFL_ERR_INVALID && sfl_set_option(ctx, SFL_OPT_ADVECT_HALO, 65) == SFL_ERR_INVALID, "halo limits");
        CHECK(sfl_set_option(ctx, SFL_OPT_TRANSPORT, 1) == SFL_ERR_INVALID && sfl_set_option(ctx, SFL_OPT_LAST_HALO, 1) == SFL_ERR_INVALID, "read-only options");
        CHECK(sfl_set_option(ctx, 999, 1) == SFL_ERR_INVALID && sfl_get_option(ctx, 999, &v) == SFL_ERR_INVALID, "unknown option");
        // round 6: one read / write option for the exchange schedule; the numbers of the three it replaces (and of the chained launch's
        // read-out) are refused; the opt-in arithmetic reads back
        CHECK(sfl_set_option(ctx, SFL_OPT_EXCHANGE_SCHEDULE, 4) == SFL_ERR_INVALID && sfl_set_option(ctx, SFL_OPT_EXCHANGE_SCHEDULE, -1) == SFL_ERR_INVALID, "schedule 4 / -1");
        for (int sch = 3; sch >= 0; --sch) CHECK(sfl_set_option(ctx, SFL_OPT_EXCHANGE_SCHEDULE, sch) == SFL_OK, "schedule %d", sch);
        CHECK(sfl_get_option(ctx, SFL_OPT_EXCHANGE_SCHEDULE, &v) == SFL_OK && v == 0, "a whole-domain context has nothing to order");
        for (int retired : {8, 13, 15, 16})
            CHECK(sfl_set_option(ctx, retired, 1) == SFL_ERR_INVALID && sfl_get_option(ctx, retired, &v) == SFL_ERR_INVALID, "retired option %d", retired);
        CHECK(sfl_get_option(ctx, SFL_OPT_SOR_FOLD, &v) == SFL_OK && v == 0, "the reference's two products are the default");
        CHECK(sfl_set_option(ctx, SFL_OPT_SOR_FOLD, 1) == SFL_OK && sfl_get_option(ctx, SFL_OPT_SOR_FOLD, &v) == SFL_OK && v == 1, "fold reads back");
        CHECK(sfl_get_option(ctx, SFL_OPT_MEASURED_WIRE_US, &v) == SFL_OK && v == -1, "nothing to measure on a whole domain, and the query measures nothing");
        CHECK(sfl_get_option(ctx, SFL_OPT_SOR_FUSE, nullptr) == SFL_ERR_INVALID, "NULL value");
        CHECK(sfl_destroy(ctx) == SFL_OK, "destroy");
    } else {
        CHECK(rc == SFL_ERR_HIP, "without a device: SFL_ERR_HIP (got %d: %s)", rc, sfl_last_error());
    }
    CHECK(sfl_destroy(nullptr) == SFL_OK, "destroying NULL is a no-op");
    CHECK(sfl_set_option(nullptr, SFL_OPT_SOR_FUSE, 8) == SFL_ERR_INVALID && sfl_synchronize(nullptr) == SFL_ERR_INVALID &&
              sfl_step(nullptr, 0.1f, 1.0f, 1, 1.9f) == SFL_ERR_INVALID && sfl_download(nullptr, 0, &a, 4) == SFL_ERR_INVALID, "NULL contexts");
    float f[4] = {0, 0, 0, 0};
    CHECK(sfl_host_poisson_solve(nullptr, f, 2, 2, 1.0f, 1, 1.9f) == SFL_ERR_INVALID && sfl_host_advect_vec2f(f, f, f, 2, 2, 0.1f, 1) == SFL_ERR_INVALID,
          "drop-ins: NULL field, next_p aliasing p");
    CHECK(sfl_host_release() == SFL_OK, "nothing to release");
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 4000;
    fuzz_plans(cases);
    bad_arguments();
    printf("host sanitizer driver: %d plan configurations x every rank (%ld launches checked against the ghost rows left exact), "
           "argument checks: %d failed checks\n", cases, validity_checks, failures);
    return failures ? 1 : 0;
}
