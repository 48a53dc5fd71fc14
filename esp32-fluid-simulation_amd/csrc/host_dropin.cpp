// host_dropin.cpp -- the host-pointer drop-ins of the C ABI (sfl_host_*): the reference's operator signatures + status
// (finitediff.h:6-10, poisson.h:4-5, advect.h:74-76).  Each call uploads, runs the HIP kernels and downloads; the
// device-side context stays with the calling thread between calls.  Host C++ only.
#include "context.h"

using namespace sfl::host;

namespace {

int default_device()
{
    const char *e = getenv("SFL_DEVICE");
    return e ? atoi(e) : 0;
}

// Context of a host-pointer drop-in call.  The sketch's loop() calls five operators per frame
// on a 61 x 81 grid (ino:252-287): creating stream, events and buffers anew for each of them cost
// 3.2 ms per frame (profiles/r01_host_dropin_pcie.txt), ten times the reference's CPU time.  So
// the context of the last call stays with the calling thread -- for grids of up to
// kHostCacheCells cells (2^26 = 8192^2: at most 3.25 GB of fields on a 288 GB part) -- until the shape
// changes, a call fails, or sfl_host_release() is called.  At 8192^2 the set-up (768 MB of hipMalloc + fill,
// stream, events, hipFree) is 4 of the 15.6 ms a poisson_solve drop-in takes; the two 256 MB transfers already run
// at the PCIe rate from pageable memory (4.75 + 4.8 ms: the runtime pins the caller's pages after their first use,
// tools/ubench_host_register.hip), the solve takes 1.8 ms: 11.4 ms with the context retained
// (profiles/r03_host_dropin.txt).
constexpr int64_t kHostCacheCells = 1 << 26;
// (released when the thread ends -- a raw thread_local pointer kept up to 3.25 GB of device memory per exited
// thread, ADVICE r03 -- by sfl_host_release(), by a change of shape and by a failing call)
struct HostCache {
    sfl_context *ctx = nullptr;
    ~HostCache()
    {
        if (ctx) (void)sfl_destroy(ctx);
        ctx = nullptr;
    }
};
thread_local HostCache g_host_cache;

struct HostCtx {
    sfl_context *c = nullptr;
    bool cached = false, ok = false;

    int acquire(int dim_x, int dim_y)
    {
        const int dev = default_device();
        sfl_context *k = g_host_cache.ctx;
        if (k && k->device == dev && k->dim_x == dim_x && k->gdim_y == dim_y) {
            c = k;
            cached = true;
            sfl_context fresh;  // option defaults
            c->opt_sor_kernel = fresh.opt_sor_kernel;
            c->opt_sor_fold = fresh.opt_sor_fold;
            c->opt_sor_fuse = fresh.opt_sor_fuse;
            c->opt_sor_rows = fresh.opt_sor_rows;
            c->opt_sor_lane_cells = fresh.opt_sor_lane_cells;
            c->opt_advect_kernel = fresh.opt_advect_kernel;
            return SFL_OK;
        }
        if (k) {
            g_host_cache.ctx = nullptr;
            sfl_destroy(k);
        }
        SFL_TRY(sfl_create(&c, dev, dim_x, dim_y));
        if ((int64_t)dim_x * dim_y <= kHostCacheCells) {
            g_host_cache.ctx = c;
            cached = true;
        }
        return SFL_OK;
    }
    int done(int rc)
    {
        ok = rc == SFL_OK;
        return rc;
    }
    ~HostCtx()
    {
        if (!c) return;
        if (cached && ok) return;
        if (cached) g_host_cache.ctx = nullptr;  // unknown state after a failure: start afresh
        sfl_destroy(c);
    }
};

int check_channels(int channels, int kind)
{
    if (channels < 1 || channels > 3 || (kind != SFL_CHANNEL_F32 && kind != SFL_CHANNEL_UQ32))
        return fail(SFL_ERR_INVALID, "advect: element must be 1..3 channels of kind SFL_CHANNEL_F32 / SFL_CHANNEL_UQ32 "
                    "(got %d x kind %d)", channels, kind);
    return SFL_OK;
}

}  // namespace

extern "C" {

int sfl_host_advect_vec2f(float *next_p, const float *p, const float *vel, int dim_x, int dim_y,
                          float dt, int no_slip)
{
    if (!next_p || !p || !vel) return fail(SFL_ERR_INVALID, "NULL field pointer");
    if (next_p == p) return fail(SFL_ERR_INVALID, "advect: next_p must not alias p (advect.h:82)");
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    sfl_context *c = t.c;
    SFL_TRY(ensure(c, c->vel, 8, false));
    SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    SFL_TRY(upload_raw(c, c->vel, vel, 8));
    float *src = c->vel;
    if (p != vel) {  // advected field differs from the velocity: scratch field kept with the context
        SFL_TRY(ensure(c, c->host_scratch, 8, false));
        SFL_TRY(upload_raw(c, c->host_scratch, p, 8));
        src = c->host_scratch;
    }
    hipError_t e = sfl::launch_advect_vec2f(c->stream, c->vel_tmp, src, c->vel, c->geom, 0, dim_y, 0,
                                            dim_y, dt, no_slip != 0, nullptr, nullptr, c->opt_advect_kernel);
    int rc = e == hipSuccess ? download_raw(c, c->vel_tmp, next_p, 8)
                             : fail(SFL_ERR_HIP, "advect launch failed: %s", hipGetErrorString(e));
    return t.done(rc);
}

int sfl_host_advect_vec3uq32(uint32_t *next_p, const uint32_t *p, const float *vel, int dim_x,
                             int dim_y, float dt, int no_slip)
{
    if (!next_p || !p || !vel) return fail(SFL_ERR_INVALID, "NULL field pointer");
    if (next_p == p) return fail(SFL_ERR_INVALID, "advect: next_p must not alias p (advect.h:82)");
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_VELOCITY, vel, (size_t)dim_x * dim_y * 8));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_COLOR, p, (size_t)dim_x * dim_y * 12));
    SFL_TRY(sfl_advect_color(t.c, dt, no_slip));
    return t.done(sfl_download(t.c, SFL_FIELD_COLOR, next_p, (size_t)dim_x * dim_y * 12));
}

int sfl_host_advect_channels(void *next_p, const void *p, const float *vel, int dim_x, int dim_y, float dt,
                             int no_slip, int channels, int kind)
{
    if (!next_p || !p || !vel) return fail(SFL_ERR_INVALID, "NULL field pointer");
    if (next_p == p) return fail(SFL_ERR_INVALID, "advect: next_p must not alias p (advect.h:82)");
    SFL_TRY(check_channels(channels, kind));
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    sfl_context *c = t.c;
    // the 12-byte dye buffers double as staging for any element of up to three channels
    SFL_TRY(ensure(c, c->vel, 8, false));
    SFL_TRY(ensure(c, c->col, 12, false));
    SFL_TRY(ensure(c, c->col_tmp, 12, false));
    SFL_TRY(upload_raw(c, c->vel, vel, 8));
    SFL_TRY(upload_raw(c, c->col, p, (size_t)channels * 4));
    const hipError_t e = sfl::launch_advect_channels(c->stream, c->col_tmp, c->col, c->vel, dim_x, dim_y, dt,
                                                     no_slip != 0, channels, kind);
    const int rc = e == hipSuccess ? download_raw(c, c->col_tmp, next_p, (size_t)channels * 4)
                                   : fail(SFL_ERR_HIP, "advect launch failed: %s", hipGetErrorString(e));
    return t.done(rc);
}

int sfl_host_calculate_divergence(float *div, const float *v, int dim_x, int dim_y, float dx)
{
    if (!div || !v) return fail(SFL_ERR_INVALID, "NULL field pointer");
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_VELOCITY, v, (size_t)dim_x * dim_y * 8));
    SFL_TRY(sfl_calculate_divergence(t.c, dx));
    return t.done(sfl_download(t.c, SFL_FIELD_DIVERGENCE, div, (size_t)dim_x * dim_y * 4));
}

int sfl_host_subtract_gradient(float *v, const float *p, int dim_x, int dim_y, float dx)
{
    if (!v || !p) return fail(SFL_ERR_INVALID, "NULL field pointer");
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_VELOCITY, v, (size_t)dim_x * dim_y * 8));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_PRESSURE, p, (size_t)dim_x * dim_y * 4));
    SFL_TRY(sfl_subtract_gradient(t.c, dx));
    return t.done(sfl_download(t.c, SFL_FIELD_VELOCITY, v, (size_t)dim_x * dim_y * 8));
}

int sfl_host_poisson_solve(float *p, const float *div, int dim_x, int dim_y, float dx, int iters,
                           float omega)
{
    if (!p || !div) return fail(SFL_ERR_INVALID, "NULL field pointer");
    HostCtx t;
    SFL_TRY(t.acquire(dim_x, dim_y));
    const char *k = getenv("SFL_SOR_KERNEL");
    if (k) SFL_TRY(sfl_set_option(t.c, SFL_OPT_SOR_KERNEL, atoi(k)));
    const char *f = getenv("SFL_SOR_FUSE");
    if (f) SFL_TRY(sfl_set_option(t.c, SFL_OPT_SOR_FUSE, atoi(f)));
    SFL_TRY(sfl_upload(t.c, SFL_FIELD_DIVERGENCE, div, (size_t)dim_x * dim_y * 4));
    SFL_TRY(sfl_poisson_solve(t.c, dx, iters, omega));
    return t.done(sfl_download(t.c, SFL_FIELD_PRESSURE, p, (size_t)dim_x * dim_y * 4));
}

int sfl_host_release(void)
{
    sfl_context *k = g_host_cache.ctx;
    g_host_cache.ctx = nullptr;
    return k ? sfl_destroy(k) : SFL_OK;
}

}  // extern "C"
