// host_tsan_driver.cpp -- TEST HARNESS (tests/cpp, `make tsan_host`; tests/test_host_sanitizers.py): the HOST side of the library
// -- contexts, options, groups of virtual ranks with their shared transport state, the three exchange executors, the one-rank
// RCCL transport, the host-pointer drop-ins with their per-thread context cache (csrc/host_dropin.cpp) -- RUN from four threads
// at once under ThreadSanitizer, over a HIP / RCCL runtime that lives on the host (fake_hip.cpp) and kernels that do nothing
// (launch_stubs_ok.cpp).  VERDICT r05 item 8.  What the ABI promises (include/sfl.h): a context (and a linked group) belongs to one
// thread at a time; different contexts, the drop-ins and the GPU-free queries may be used from different threads concurrently.
// Each thread therefore owns its contexts, and all threads hammer the shared parts: the error state, the drop-in caches, the
// plan queries.  Exit status 0 = every call returned what it should; ThreadSanitizer reports its findings itself (exit 66).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/sfl.h"

extern "C" long fake_hip_live_allocations();

static std::atomic<int> failures{0};
#define CHECK(cond, ...)                                                    \
    do {                                                                    \
        if (!(cond)) {                                                      \
            if (failures++ < 20) {                                          \
                fprintf(stderr, "CHECK failed (thread %d): %s -- ", tid, #cond); \
                fprintf(stderr, __VA_ARGS__);                               \
                fprintf(stderr, "\n");                                      \
            }                                                               \
        }                                                                   \
    } while (0)

static void one_thread(int tid, int rounds)
{
    for (int round = 0; round < rounds; ++round) {
        const int dim_x = 96 + 32 * tid, dim_y = 640 + 64 * ((round + tid) % 3), iters = 9 + tid;
        std::vector<float> v((size_t)dim_x * dim_y * 2, 0.25f), d((size_t)dim_x * dim_y, 0.5f), p((size_t)dim_x * dim_y);
        std::vector<uint32_t> col((size_t)dim_x * dim_y * 3, 7u), col2(col.size());

        // ---- a whole-domain context: options, field I/O, operators, steps, force queues
        sfl_context *c = nullptr;
        CHECK(sfl_create(&c, 0, dim_x, dim_y) == SFL_OK && c, "create: %s", sfl_last_error());
        if (c) {
            int val = -1;
            CHECK(sfl_set_option(c, SFL_OPT_SOR_FUSE, 8) == SFL_OK && sfl_set_option(c, SFL_OPT_SOR_FOLD, tid & 1) == SFL_OK, "options");
            CHECK(sfl_get_option(c, SFL_OPT_SOR_FOLD, &val) == SFL_OK && val == (tid & 1), "fold reads back per context");
            CHECK(sfl_upload(c, SFL_FIELD_VELOCITY, v.data(), v.size() * 4) == SFL_OK, "upload v: %s", sfl_last_error());
            CHECK(sfl_upload(c, SFL_FIELD_COLOR, col.data(), col.size() * 4) == SFL_OK, "upload colour");
            const int cells[4] = {3, 4, dim_x - 1, dim_y - 1};
            const float fv[4] = {1.f, 2.f, 3.f, 4.f};
            CHECK(sfl_queue_forces(c, cells, fv, 2) == SFL_OK, "forces");
            CHECK(sfl_step(c, 0.03f, 1.0f, iters, 1.96f) == SFL_OK, "step: %s", sfl_last_error());
            CHECK(sfl_step_n(c, 3, 0.03f, 1.0f, iters, 1.96f) == SFL_OK, "step_n: %s", sfl_last_error());
            CHECK(sfl_calculate_divergence(c, 1.0f) == SFL_OK && sfl_poisson_solve(c, 1.0f, iters, 1.96f) == SFL_OK &&
                      sfl_subtract_gradient(c, 1.0f) == SFL_OK, "operators: %s", sfl_last_error());
            CHECK(sfl_synchronize(c) == SFL_OK, "synchronize: %s", sfl_last_error());
            CHECK(sfl_download(c, SFL_FIELD_COLOR, col2.data(), col2.size() * 4) == SFL_OK, "download");
            CHECK(sfl_poisson_solve(c, 1.0f, -1, 1.96f) == SFL_ERR_INVALID && strstr(sfl_last_error(), "iters"), "this thread's own error message");
            CHECK(sfl_destroy(c) == SFL_OK, "destroy");
        }

        // ---- a group of virtual ranks (shared transport state: streams, halo tuner, epochs), every exchange schedule
        const int nranks = 2 + (tid + round) % 3;
        std::vector<sfl_context *> slabs((size_t)nranks, nullptr);
        bool ok = true;
        for (int r = 0; r < nranks; ++r) ok = ok && sfl_create_slab(&slabs[(size_t)r], 0, dim_x, dim_y, r, nranks) == SFL_OK;
        CHECK(ok, "slabs: %s", sfl_last_error());
        if (ok) {
            CHECK(sfl_group_link(slabs.data(), nranks) == SFL_OK, "link: %s", sfl_last_error());
            for (int schedule = 3; schedule >= 0; --schedule) {
                int got = -1;
                CHECK(sfl_set_option(slabs[0], SFL_OPT_EXCHANGE_SCHEDULE, schedule) == SFL_OK, "schedule %d", schedule);
                CHECK(sfl_get_option(slabs[(size_t)nranks - 1], SFL_OPT_EXCHANGE_SCHEDULE, &got) == SFL_OK && got >= 1 && got <= 3,
                      "group-wide schedule %d resolved to %d: %s", schedule, got, sfl_last_error());
                for (int r = 0; r < nranks; ++r) {
                    int b = 0, e = 0;
                    CHECK(sfl_slab_of(slabs[(size_t)r], &b, &e, nullptr, nullptr) == SFL_OK, "slab_of");
                    CHECK(sfl_upload(slabs[(size_t)r], SFL_FIELD_DIVERGENCE, d.data() + (size_t)b * dim_x, (size_t)(e - b) * dim_x * 4) == SFL_OK, "upload d");
                }
                for (int rep = 0; rep < 2; ++rep) CHECK(sfl_poisson_solve(slabs[0], 1.0f, iters, 1.96f) == SFL_OK, "slab solve: %s", sfl_last_error());
                int launches = 0, exchanges = 0, fuse = 0;
                CHECK(sfl_last_solve_info(slabs[(size_t)nranks / 2], &launches, &exchanges, &fuse) == SFL_OK && launches > 0, "solve info");
            }
            CHECK(sfl_step_n(slabs[0], 2, 0.03f, 1.0f, iters, 1.96f) == SFL_OK, "slab steps: %s", sfl_last_error());
            CHECK(sfl_synchronize(slabs[0]) == SFL_OK, "slab synchronize: %s", sfl_last_error());
        }
        for (sfl_context *s : slabs)
            if (s) CHECK(sfl_destroy(s) == SFL_OK, "destroy slab");

        // ---- one rank's program with RCCL (here: the host-side one-rank communicator) as the transport
        sfl_context *e = nullptr;
        CHECK(sfl_create_slab(&e, 0, dim_x, dim_y, 1, 4) == SFL_OK, "emulated rank: %s", sfl_last_error());
        if (e) {
            CHECK(sfl_comm_emulate_rccl(e) == SFL_OK, "comm_emulate_rccl: %s", sfl_last_error());
            CHECK(sfl_poisson_solve(e, 1.0f, iters, 1.96f) == SFL_OK && sfl_step(e, 0.03f, 1.0f, iters, 1.96f) == SFL_OK, "rccl rank: %s", sfl_last_error());
            CHECK(sfl_synchronize(e) == SFL_OK, "rccl rank synchronize: %s", sfl_last_error());
            CHECK(sfl_destroy(e) == SFL_OK, "destroy rccl rank");
        }

        // ---- the host-pointer drop-ins: a context cached PER THREAD (csrc/host_dropin.cpp), shapes that change under it
        CHECK(sfl_host_calculate_divergence(d.data(), v.data(), dim_x, dim_y, 1.0f) == SFL_OK, "host divergence: %s", sfl_last_error());
        CHECK(sfl_host_poisson_solve(p.data(), d.data(), dim_x, dim_y, 1.0f, iters, 1.96f) == SFL_OK, "host solve: %s", sfl_last_error());
        CHECK(sfl_host_subtract_gradient(v.data(), p.data(), dim_x, dim_y, 1.0f) == SFL_OK, "host gradient");
        CHECK(sfl_host_advect_vec3uq32(col2.data(), col.data(), v.data(), dim_x, dim_y, 0.03f, 0) == SFL_OK, "host dye");
        CHECK(sfl_host_poisson_solve(p.data(), d.data(), dim_x / 2, dim_y, 1.0f, 2, 1.96f) == SFL_OK, "host solve, another shape");
        if (round % 2) CHECK(sfl_host_release() == SFL_OK, "host release");

        // ---- GPU-free queries
        int n = 0, b = 0, en = 0;
        CHECK(sfl_plan_poisson(dim_y, 4, tid % 4, iters, 8, 3, 32, nullptr, 0, &n) == SFL_OK && n > 0, "plan");
        CHECK(sfl_slab_rows(dim_y, 4, tid % 4, &b, &en) == SFL_OK && en > b, "slab rows");
    }
    (void)sfl_host_release();
}

static int g_unguarded = 0;   // (`host_tsan_driver race`: proof that the harness SEES a race -- two threads, one plain int)

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "race")) {
        std::thread a([] { for (int k = 0; k < 100000; ++k) ++g_unguarded; }), b([] { for (int k = 0; k < 100000; ++k) ++g_unguarded; });
        a.join();
        b.join();
        printf("race self-test done (%d)\n", g_unguarded);
        return 0;
    }
    const int threads = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 6;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(one_thread, t, rounds);
    for (std::thread &t : pool) t.join();
    const long live = fake_hip_live_allocations();
    if (live != 0) {
        fprintf(stderr, "%ld fake-device allocations were never freed\n", live);
        ++failures;
    }
    printf("host_tsan_driver: %d threads x %d rounds, %d failed checks, %ld allocations left\n", threads, rounds, failures.load(), live);
    return failures.load() ? 1 : 0;
}
