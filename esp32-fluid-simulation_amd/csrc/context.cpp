// context.cpp -- contexts of the C ABI (include/sfl.h): error state, device / plan queries, create / destroy, options,
// field I/O, synchronize, timers.  Host C++ only; see context.h for the map of the library's host side.
#include "transport.h"

namespace sfl {
namespace host {

std::string &last_error()
{
    thread_local std::string g_error;
    return g_error;
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

size_t field_elem_bytes(int field)
{
    switch (field) {
        case SFL_FIELD_VELOCITY: return 8;
        case SFL_FIELD_COLOR: return 12;
        case SFL_FIELD_DIVERGENCE:
        case SFL_FIELD_PRESSURE: return 4;
    }
    return 0;
}

int use_device(sfl_context *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return SFL_OK;
}

int ensure_bytes(sfl_context *c, void **ptr, size_t elem_bytes, bool zero)
{
    SFL_TRY(use_device(c));
    void *m = nullptr;
    const size_t bytes = c->local_cells() * elem_bytes;
    HIP_TRY(hipMalloc(&m, bytes));
    if (zero) HIP_TRY(hipMemsetAsync(m, 0, bytes, c->stream));
    *ptr = m;
    return SFL_OK;
}

// divergence, pressure and the pressure's ping-pong partner: one block, zero-filled
int ensure_sor_block(sfl_context *c)
{
    if (c->sor_block) return SFL_OK;
    SFL_TRY(use_device(c));
    const size_t cells = (c->local_cells() + 63) & ~(size_t)63;  // thirds stay 256-byte aligned
    void *m = nullptr;
    HIP_TRY(hipMalloc(&m, 3 * cells * 4));
    HIP_TRY(hipMemsetAsync(m, 0, 3 * cells * 4, c->stream));
    c->sor_block = static_cast<float *>(m);
    c->div = c->sor_block;
    c->p = c->sor_block + cells;
    c->p_alt = c->sor_block + 2 * cells;
    return SFL_OK;
}

int ensure_field(sfl_context *c, int field)
{
    switch (field) {
        case SFL_FIELD_VELOCITY: return ensure(c, c->vel, 8, true);
        case SFL_FIELD_COLOR: return ensure(c, c->col, 12, true);
        case SFL_FIELD_DIVERGENCE:
        case SFL_FIELD_PRESSURE: return ensure_sor_block(c);
    }
    return fail(SFL_ERR_INVALID, "unknown field id %d", field);
}

void *field_ptr(sfl_context *c, int field)
{
    switch (field) {
        case SFL_FIELD_VELOCITY: return c->vel;
        case SFL_FIELD_COLOR: return c->col;
        case SFL_FIELD_DIVERGENCE: return c->div;
        case SFL_FIELD_PRESSURE: return c->p;
    }
    return nullptr;
}

std::vector<sfl_context *> peers_of(sfl_context *c)
{
    if (c->group) return c->group->members;
    return {c};
}

int min_owned_rows(const sfl_context *c)
{
    int m = c->gdim_y;
    for (int r = 0; r < c->nranks; ++r) {
        int b, e;
        sfl::slab_rows(c->gdim_y, c->nranks, r, &b, &e);
        if (e - b < m) m = e - b;
    }
    return m;
}

int check_dims(int dim_x, int dim_y)
{
    // with a dimension of 1 the reference's edge loops revisit cells: rejected
    if (dim_x < 2 || dim_y < 2)
        return fail(SFL_ERR_INVALID, "dim_x and dim_y must be >= 2 (got %d x %d)", dim_x, dim_y);
    if ((int64_t)dim_x * dim_y > (int64_t)1 << 30)
        return fail(SFL_ERR_INVALID, "domain of %d x %d cells exceeds the int index range of "
                    "operations.h:7", dim_x, dim_y);
    return SFL_OK;
}

// The kernels address a context's LOCAL arrays (owned + ghost rows) with 32-bit signed byte
// offsets; the widest element they index that way is the 8-byte velocity, so a local array may
// hold at most 2^28 cells (= 16384 x 16384, BASELINE config 5 on one GPU: 2 GiB of velocity).
// Larger domains need more slabs.
constexpr int64_t kMaxLocalCells = (int64_t)1 << 28;
int check_local_cells(int dim_x, int lrows)
{
    if ((int64_t)dim_x * lrows > kMaxLocalCells)
        return fail(SFL_ERR_INVALID, "a context holds at most 2^28 cells (%d x %d local rows asked): "
                    "split the domain into more slabs", dim_x, lrows);
    return SFL_OK;
}

int check_wait_error(sfl_context *c)
{
    for (sfl_context *m : peers_of(c))
        if (m->wait_error_seen)
            return fail(SFL_ERR_HIP, "slab %d/%d: a wait inside an earlier solve gave up (a halo message did not arrive in time): "
                        "the pressure field and everything computed from it are not valid; sfl_synchronize() reports and "
                        "clears the condition", m->rank, m->nranks);
    return SFL_OK;
}

// (slabs only) has a wait inside one of this context's launches given up?  The stream has just been drained.
static int look_for_wait_error(sfl_context *c)
{
    if (c->nranks < 2) return SFL_OK;
    int word = 0;
    HIP_TRY(hipMemcpy(&word, c->halo_flag + 2, sizeof word, hipMemcpyDeviceToHost));
    if (word) c->wait_error_seen = true;
    return check_wait_error(c);
}

int upload_raw(sfl_context *c, void *dev, const void *host, size_t elem_bytes)
{
    SFL_TRY(use_device(c));
    const size_t bytes = (size_t)(c->g1 - c->g0) * c->dim_x * elem_bytes;
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(dev) + c->owned_offset_cells() * elem_bytes, host,
                           bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SFL_OK;
}

int download_raw(sfl_context *c, const void *dev, void *host, size_t elem_bytes)
{
    SFL_TRY(use_device(c));
    const size_t bytes = (size_t)(c->g1 - c->g0) * c->dim_x * elem_bytes;
    HIP_TRY(hipMemcpyAsync(host, static_cast<const char *>(dev) + c->owned_offset_cells() * elem_bytes,
                           bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return look_for_wait_error(c);   // what was just handed out must not be a field a timed-out solve left behind
}

}  // namespace host
}  // namespace sfl

using namespace sfl::host;

// ==========================================================================================
// utilities
// ==========================================================================================
extern "C" {

int sfl_abi_version(void) { return SFL_ABI_VERSION; }

const char *sfl_last_error(void) { return last_error().c_str(); }

int sfl_device_count(int *count)
{
    if (!count) return fail(SFL_ERR_INVALID, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SFL_ERR_HIP, "no usable HIP device: %s", hipGetErrorString(e));
    *count = n;
    return SFL_OK;
}

int sfl_device_info(int device, char *name, size_t name_cap, int *compute_units,
                    size_t *total_mem_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_cap) {
        snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
    return SFL_OK;
}

int sfl_slab_rows(int dim_y, int nranks, int rank, int *row_begin, int *row_end)
{
    if (dim_y < 1 || nranks < 1 || rank < 0 || rank >= nranks || !row_begin || !row_end)
        return fail(SFL_ERR_INVALID, "bad slab query (dim_y %d, rank %d of %d)", dim_y, rank, nranks);
    sfl::slab_rows(dim_y, nranks, rank, row_begin, row_end);
    return SFL_OK;
}

int sfl_sor_pass_plan(int iters, int fuse, int *n_passes, int *passes, int cap)
{
    if (iters < 0 || fuse < 2 || (fuse & 1) || fuse > SFL_MAX_FUSE || !n_passes)
        return fail(SFL_ERR_INVALID, "bad pass plan query (iters %d, fuse %d)", iters, fuse);
    const std::vector<int> v = sfl::sor_pass_plan(iters, fuse);
    *n_passes = (int)v.size();
    if (passes)
        for (int k = 0; k < (int)v.size() && k < cap; ++k) passes[k] = v[k];
    return SFL_OK;
}

int sfl_plan_poisson_tail(int dim_y, int nranks, int rank, int iters, int fuse, int kernel, int halo, int tail,
                          sfl_plan_step *steps, int cap, int *n_steps)
{
    if (dim_y < 2 || nranks < 1 || rank < 0 || rank >= nranks || iters < 0 || !n_steps || halo < 0 || tail < 0 ||
        (kernel < 1 || kernel > 3) || (kernel >= 2 && (fuse < 2 || (fuse & 1) || fuse > SFL_MAX_FUSE)))
        return fail(SFL_ERR_INVALID, "bad plan query");
    const std::vector<sfl_plan_step> v = sfl::plan_poisson(dim_y, nranks, rank, iters, fuse, kernel, halo, tail);
    *n_steps = (int)v.size();
    if (steps)
        for (int k = 0; k < (int)v.size() && k < cap; ++k) steps[k] = v[k];
    return SFL_OK;
}

int sfl_plan_poisson(int dim_y, int nranks, int rank, int iters, int fuse, int kernel, int halo,
                     sfl_plan_step *steps, int cap, int *n_steps)
{
    if (dim_y < 2 || nranks < 1 || rank < 0 || rank >= nranks || iters < 0 || !n_steps ||
        (kernel < 1 || kernel > 3) || (kernel >= 2 && (fuse < 2 || (fuse & 1) || fuse > SFL_MAX_FUSE)))
        return fail(SFL_ERR_INVALID, "bad plan query");
    if (halo < 0) return fail(SFL_ERR_INVALID, "bad plan query");
    const std::vector<sfl_plan_step> v = sfl::plan_poisson(dim_y, nranks, rank, iters, fuse, kernel, halo);
    *n_steps = (int)v.size();
    if (steps)
        for (int k = 0; k < (int)v.size() && k < cap; ++k) steps[k] = v[k];
    return SFL_OK;
}

// ==========================================================================================
// contexts
// ==========================================================================================
int sfl_create_slab(sfl_context **out, int device, int dim_x, int dim_y, int rank, int nranks)
{
    if (!out) return fail(SFL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    SFL_TRY(check_dims(dim_x, dim_y));
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return fail(SFL_ERR_INVALID, "bad rank %d of %d", rank, nranks);
    if (nranks > dim_y) return fail(SFL_ERR_INVALID, "more slabs (%d) than rows (%d)", nranks, dim_y);
    {
        int b = 0, e = 0;
        sfl::slab_rows(dim_y, nranks, rank, &b, &e);
        SFL_TRY(check_local_cells(dim_x, (e - b) + (nranks > 1 ? 2 * kGhostRows : 0)));
    }
    int ndev = 0;
    SFL_TRY(sfl_device_count(&ndev));
    if (device < 0 || device >= ndev)
        return fail(SFL_ERR_HIP, "device %d not available (%d visible)", device, ndev);

    std::unique_ptr<sfl_context> c(new sfl_context);
    c->device = device;
    c->dim_x = dim_x;
    c->gdim_y = dim_y;
    c->rank = rank;
    c->nranks = nranks;
    sfl::slab_rows(dim_y, nranks, rank, &c->g0, &c->g1);
    c->ghost = nranks > 1 ? kGhostRows : 0;
    c->geom.dim_x = dim_x;
    c->geom.gdim_y = dim_y;
    c->geom.grow0 = c->g0 - c->ghost;
    c->geom.lrows = (c->g1 - c->g0) + 2 * c->ghost;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev_start));
    HIP_TRY(hipEventCreate(&c->ev_stop));
    void *flag = nullptr;
    HIP_TRY(hipMalloc(&flag, (4 + kCollectiveWords) * sizeof(int)));
    HIP_TRY(hipMemset(flag, 0, (4 + kCollectiveWords) * sizeof(int)));
    c->halo_flag = static_cast<int *>(flag);
    c->d_arrival = c->halo_flag + 1;
    c->d_done = c->halo_flag + 3;
    c->d_collective = c->halo_flag + 4;
    *out = c.release();
    return SFL_OK;
}

int sfl_create(sfl_context **out, int device, int dim_x, int dim_y)
{
    return sfl_create_slab(out, device, dim_x, dim_y, 0, 1);
}

int sfl_destroy(sfl_context *c)
{
    if (!c) return SFL_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->group) {
        // dissolve the group: the remaining members become plain slabs without a transport
        // (their collective operators then fail with SFL_ERR_STATE); the shared streams live
        // on through `keepalive` until the last member is destroyed
        const std::vector<sfl_context *> members = c->group->members;
        for (sfl_context *m : members) {
            m->keepalive = m->transport;
            m->transport.reset();
            m->group = nullptr;
        }
    }
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);  // nothing of a communicator may still be queued
    c->transport.reset();   // (an RCCL transport destroys its communicator)
    c->keepalive.reset();
    for (void *m : {(void *)c->vel, (void *)c->vel_tmp, (void *)c->col, (void *)c->col_tmp,
                    (void *)c->sor_block, (void *)c->halo_flag,
                    (void *)c->d_force_cells, (void *)c->d_force_vel, (void *)c->d_image,
                    (void *)c->host_scratch, (void *)c->d_reach, c->gather_buf})
        if (m) (void)hipFree(m);
    for (auto &st : c->force_stage) {
        if (st.cells) (void)hipHostFree(st.cells);
        if (st.vel) (void)hipHostFree(st.vel);
        if (st.copied) (void)hipEventDestroy(st.copied);
    }
    if (c->xstream) {
        (void)hipStreamSynchronize(c->xstream);
        (void)hipStreamDestroy(c->xstream);
    }
    if (c->d_report) (void)hipFree(c->d_report);
    if (c->h_report) (void)hipHostFree(c->h_report);
    if (c->ev_report) (void)hipEventDestroy(c->ev_report);
    if (c->ev_color_halo) (void)hipEventDestroy(c->ev_color_halo);
    if (c->ev_vel_final) (void)hipEventDestroy(c->ev_vel_final);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_arrived) (void)hipEventDestroy(c->ev_arrived);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SFL_OK;
}

// What one exchange costs depends on the protocol it was measured with: forget it -- and the halo depths that were chosen from it
// (a depth decided for another exchange cost must not outlive it) -- and measure again before the next solve.
static void invalidate_exchange_measurement(sfl_context *c)
{
    c->exchange_latency_us = -1;
    HaloTuner &t = c->group ? c->group->halo_tuner : c->halo_tuner;
    t.decided.clear();
    t.active = false;
}

static int set_option_one(sfl_context *c, int option, int value)
{
    switch (option) {
        case SFL_OPT_SOR_KERNEL:
            if (value < 0 || value > 2) return fail(SFL_ERR_INVALID, "SOR kernel must be 0, 1 or 2");
            c->opt_sor_kernel = value;
            return SFL_OK;
        case SFL_OPT_SOR_FUSE:
            if (value != 0 && (value < 2 || value > SFL_MAX_FUSE || (value & 1)))
                return fail(SFL_ERR_INVALID, "fuse must be 0 (auto) or even, 2..%d (got %d)", SFL_MAX_FUSE, value);
            c->opt_sor_fuse = value;
            return SFL_OK;
        case SFL_OPT_ADVECT_HALO:
            if (value < 0 || value > kAdvectGhostRows)
                return fail(SFL_ERR_INVALID, "advect halo must be 0 (auto) or 1..%d rows", kAdvectGhostRows);
            c->opt_advect_halo = value;
            return SFL_OK;
        case SFL_OPT_SOR_ROWS:
            if (value < 0) return fail(SFL_ERR_INVALID, "rows per chunk must be >= 0");
            if (value != c->opt_sor_rows) invalidate_exchange_measurement(c);   // (the timed solves the halo depth was chosen from ran on other tiles)
            c->opt_sor_rows = value;
            return SFL_OK;
        case SFL_OPT_TRANSPORT:
            return fail(SFL_ERR_INVALID, "SFL_OPT_TRANSPORT is read-only: use sfl_comm_attach / sfl_group_link");
        case SFL_OPT_LAST_EARLY_ROWS:
        case SFL_OPT_MEASURED_WIRE_US:
        case SFL_OPT_LAST_HALO:
            return fail(SFL_ERR_INVALID, "this option is read-only");
        case SFL_OPT_FUSE_PROJECTION:
            c->opt_fuse_projection = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_EXCHANGE_SCHEDULE: {   // 0 automatic, 1 in line, 2 one launch early behind events, 3 in time
            if (value < 0 || value > 3) return fail(SFL_ERR_INVALID, "exchange schedule must be 0 (auto), 1 (in line), 2 (behind events) or 3 (in time)");
            const int overlap = value == 1 ? 0 : 1, arrival = value == 2 ? 0 : value == 3 ? 1 : -1;
            if (overlap != c->opt_sor_overlap || arrival != c->opt_sor_arrival) invalidate_exchange_measurement(c);   // (measured with the other protocol)
            c->opt_sor_overlap = overlap;
            c->opt_sor_arrival = arrival;
            return SFL_OK;
        }
        case SFL_OPT_FUSE_DIVERGENCE:
            c->opt_fuse_divergence = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_SMALL_GRID:
            c->opt_small_grid = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_ADVECT_KERNEL:
            if (value < 0 || value > 2) return fail(SFL_ERR_INVALID, "advection kernel must be 0, 1 or 2");
            c->opt_advect_kernel = value;
            return SFL_OK;
        case SFL_OPT_EMULATE_WIRE_US:
            if (value < 0 || value > 10000) return fail(SFL_ERR_INVALID, "emulated wire delay must be 0..10000 us");
            c->opt_emulate_wire_us = value;
            invalidate_exchange_measurement(c);   // what an exchange costs is measured again before the next solve
            return SFL_OK;
        case SFL_OPT_HALO_TIMEOUT_MS:
            if (value < 0) return fail(SFL_ERR_INVALID, "halo timeout must be >= 0 ms (0 = the transport's default)");
            c->opt_halo_timeout_ms = value;
            return SFL_OK;
        case SFL_OPT_STEP_SEAMS:
            c->opt_step_seams = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_SOR_FOLD:
            c->opt_sor_fold = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_SOR_HALO:
            if (value != 0 && (value < 2 || value > kGhostRows))
                return fail(SFL_ERR_INVALID, "SOR halo must be 0 (auto) or 2..%d rows", kGhostRows);
            c->opt_sor_halo = value;
            return SFL_OK;
        case SFL_OPT_SOR_LANE_CELLS:
            if (value != 0 && value != 2)
                return fail(SFL_ERR_INVALID, "cells per lane must be 0 (auto) or 2 (the packed 4-cell "
                            "flavour of round 1 is gone: never faster)");
            c->opt_sor_lane_cells = value;
            return SFL_OK;
    }
    return fail(SFL_ERR_INVALID, "unknown option %d", option);
}

// Options of a linked group are GROUP-wide: the slabs execute one program in lock step, and a
// halo a peer trusts must be the halo that was exchanged (sfl_group_link aligns the members with
// slab 0 to begin with).  With RCCL every rank is its own process: set the same options on all.
int sfl_set_option(sfl_context *ctx, int option, int value)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    for (sfl_context *c : peers_of(ctx)) {
        SFL_TRY(set_option_one(c, option, value));
        // RCCL ranks: the option block is compared again, collectively, by the next operator (every rank that
        // changed an option does so at the same point of its program; sfl_comm_check_options does it at once)
        if (c->transport && c->transport->separate_processes()) c->options_dirty = true;
    }
    return SFL_OK;
}

int sfl_get_option(sfl_context *c, int option, int *value)
{
    if (!c || !value) return fail(SFL_ERR_INVALID, "NULL argument");
    switch (option) {
        case SFL_OPT_SOR_KERNEL: *value = c->opt_sor_kernel; return SFL_OK;
        case SFL_OPT_SOR_FOLD: *value = c->opt_sor_fold; return SFL_OK;
        case SFL_OPT_SOR_FUSE: *value = c->opt_sor_fuse; return SFL_OK;
        case SFL_OPT_ADVECT_HALO: *value = c->opt_advect_halo; return SFL_OK;
        case SFL_OPT_SOR_ROWS: *value = c->opt_sor_rows; return SFL_OK;
        case SFL_OPT_TRANSPORT: *value = c->transport ? c->transport->kind() : 0; return SFL_OK;
        case SFL_OPT_HALO_TIMEOUT_MS: *value = c->opt_halo_timeout_ms; return SFL_OK;
        case SFL_OPT_LAST_HALO: *value = c->last_halo; return SFL_OK;
        case SFL_OPT_MEASURED_WIRE_US:   // (a query only: measuring is a COLLECTIVE of the ranks and belongs to the next solve, resolve_schedule)
            *value = c->transport && c->nranks > 1 ? c->exchange_latency_us : -1;
            return SFL_OK;
        case SFL_OPT_EXCHANGE_SCHEDULE: {   // what the next solve will do: needs the streams' verdict (transport.cpp)
            if (!c->transport || c->nranks < 2 || c->opt_sor_kernel == 1) { *value = 0; return SFL_OK; }
            bool side_by_side = false;
            SFL_TRY(streams_run_concurrently(c, &side_by_side));
            const int asked = c->opt_sor_arrival >= 0 ? c->opt_sor_arrival : (c->transport->arrival_by_default() ? 1 : 0);
            *value = !c->opt_sor_overlap ? 1 : (asked && side_by_side ? 3 : 2);
            return SFL_OK;
        }
        case SFL_OPT_SOR_LANE_CELLS: *value = c->opt_sor_lane_cells; return SFL_OK;
        case SFL_OPT_SOR_HALO: *value = c->opt_sor_halo; return SFL_OK;
        case SFL_OPT_FUSE_PROJECTION: *value = c->opt_fuse_projection; return SFL_OK;
        case SFL_OPT_ADVECT_KERNEL: *value = c->opt_advect_kernel; return SFL_OK;
        case SFL_OPT_FUSE_DIVERGENCE: *value = c->opt_fuse_divergence; return SFL_OK;
        case SFL_OPT_SMALL_GRID: *value = c->opt_small_grid; return SFL_OK;
        case SFL_OPT_EMULATE_WIRE_US: *value = c->opt_emulate_wire_us; return SFL_OK;
        case SFL_OPT_STEP_SEAMS: *value = c->opt_step_seams; return SFL_OK;
        case SFL_OPT_LAST_EARLY_ROWS: *value = c->last_early_kept; return SFL_OK;
    }
    return fail(SFL_ERR_INVALID, "unknown option %d", option);
}

int sfl_slab_of(sfl_context *c, int *row_begin, int *row_end, int *rank, int *nranks)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (row_begin) *row_begin = c->g0;
    if (row_end) *row_end = c->g1;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return SFL_OK;
}

int sfl_upload(sfl_context *c, int field, const void *host, size_t bytes)
{
    if (!c || !host) return fail(SFL_ERR_INVALID, "NULL argument");
    SFL_TRY(settle_color(c));
    const size_t eb = field_elem_bytes(field);
    if (!eb) return fail(SFL_ERR_INVALID, "unknown field id %d", field);
    const size_t want = (size_t)(c->g1 - c->g0) * c->dim_x * eb;
    if (bytes != want) return fail(SFL_ERR_INVALID, "field %d: got %zu bytes, slab holds %zu", field, bytes, want);
    SFL_TRY(ensure_field(c, field));
    if (field == SFL_FIELD_VELOCITY) {
        ++c->vel_epoch;
        c->v_ghost_valid = 0;
    }
    if (field == SFL_FIELD_PRESSURE) c->p_ghost_valid = 0;
    return upload_raw(c, field_ptr(c, field), host, eb);
}

int sfl_download(sfl_context *c, int field, void *host, size_t bytes)
{
    if (!c || !host) return fail(SFL_ERR_INVALID, "NULL argument");
    SFL_TRY(settle_color(c));
    const size_t eb = field_elem_bytes(field);
    if (!eb) return fail(SFL_ERR_INVALID, "unknown field id %d", field);
    const size_t want = (size_t)(c->g1 - c->g0) * c->dim_x * eb;
    if (bytes != want) return fail(SFL_ERR_INVALID, "field %d: got %zu bytes, slab holds %zu", field, bytes, want);
    SFL_TRY(ensure_field(c, field));
    return download_raw(c, field_ptr(c, field), host, eb);
}

int sfl_field_device_ptr(sfl_context *c, int field, void **dev_ptr)
{
    if (!c || !dev_ptr) return fail(SFL_ERR_INVALID, "NULL argument");
    SFL_TRY(settle_color(c));
    const size_t eb = field_elem_bytes(field);
    if (!eb) return fail(SFL_ERR_INVALID, "unknown field id %d", field);
    SFL_TRY(ensure_field(c, field));
    // The pointer is writable: whatever was known about the field's ghost rows, or about the reach of the
    // velocity's back-traces, may be stale once the caller has used it (a velocity written through
    // the pointer was advected on the previous field's halo).  Treated like an upload.
    for (sfl_context *m : peers_of(c)) {
        if (field == SFL_FIELD_VELOCITY) {
            ++m->vel_epoch;
            m->v_ghost_valid = 0;
        }
        if (field == SFL_FIELD_PRESSURE) m->p_ghost_valid = 0;
    }
    *dev_ptr = static_cast<char *>(field_ptr(c, field)) + c->owned_offset_cells() * eb;
    return SFL_OK;
}

int sfl_synchronize(sfl_context *ctx)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx));
    int rc = SFL_OK;
    for (sfl_context *c : peers_of(ctx)) {
        SFL_TRY(use_device(c));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->nranks > 1) {
            int words[3] = {0, 0, 0};   // halo_flag, arrival count, a wait for it timed out
            HIP_TRY(hipMemcpy(words, c->halo_flag, sizeof words, hipMemcpyDeviceToHost));
            if (words[0]) {
                HIP_TRY(hipMemset(c->halo_flag, 0, sizeof(int)));
                rc = fail(SFL_ERR_HALO, "slab %d/%d: a back-trace left the %d-row advect halo; raise "
                          "SFL_OPT_ADVECT_HALO", c->rank, c->nranks, c->opt_advect_halo);
            }
            if (words[2]) {
                HIP_TRY(hipMemset(c->halo_flag + 2, 0, sizeof(int)));
                // bits: 1 a tile of a launch, 8 the exchange stream (for the sender count) waited for a halo message
                rc = fail(SFL_ERR_HIP, "slab %d/%d: a wait inside a solve lasted longer than %g s (waits 0x%x; arrival "
                          "count %d of %d): the pressure field is not valid", c->rank, c->nranks,
                          halo_timeout_us(c) / 1e6, words[2], words[1], c->arrival_epoch);
            }
            c->wait_error_seen = false;
        }
    }
    return rc;
}

int sfl_timer_start(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(use_device(c));
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
    return SFL_OK;
}

int sfl_timer_stop(sfl_context *c, float *elapsed_ms)
{
    if (!c || !elapsed_ms) return fail(SFL_ERR_INVALID, "NULL argument");
    SFL_TRY(use_device(c));
    HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_stop));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev_start, c->ev_stop));
    return SFL_OK;
}

int sfl_last_solve_info(sfl_context *c, int *launches, int *exchanges, int *fuse)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (launches) *launches = c->last_launches;
    if (exchanges) *exchanges = c->last_exchanges;
    if (fuse) *fuse = c->last_fuse;
    return SFL_OK;
}

}  // extern "C"
