// operators.cpp -- the operators of the C ABI on a context's resident fields: advection (advect.h:74-85), divergence and
// projection (finitediff.cpp:9-82), point forces (ino:264-269), the sketch's initial condition and dye visualiser.  On a
// slab each performs the halo exchanges it needs (transport.h).  Host C++ only; the kernels live in advect_tiled.hip /
// advect_generic.hip / stencil_kernels.hip.
#include "transport.h"

namespace sfl {
namespace host {

// ---- slab advection: which rows of the advected field does a slab need? --------------------------
// A back-trace reads the field up to |v_y| dt + 1 rows away from its cell (advect.h:81, :38-42).
// With a fixed halo (SFL_OPT_ADVECT_HALO = h >= 1) h rows are exchanged and a back-trace that leaves
// them raises SFL_ERR_HALO at the next sfl_synchronize.  With SFL_OPT_ADVECT_HALO = 0 the reach is
// MEASURED first (backtrace_reach_kernel over the owned cells, maximum over all slabs: every rank
// must exchange the same number of rows) and then
//   * reach <= ghost rows and <= the thinnest slab: exactly that many rows are exchanged;
//   * otherwise the whole field is gathered on every GPU (SURVEY 8e's all-gather fallback) and the
//     kernel samples the gathered copy -- correct for any velocity, at the price of the copy.
// The measurement costs a small kernel, a 2-int all-reduce and one host round trip per advection.

static int *advect_flag(sfl_context *c, const AdvectPlan &plan)
{
    if (c->nranks == 1 || plan.gather) return nullptr;
    if (plan.report) return c->d_report + 2;
    return plan.flag ? c->halo_flag : nullptr;
}

int launch_reach_set(sfl_context *c, int *words, float dt)
{
    SFL_TRY(use_device(c));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words, c->vel, c->geom, c->g0, c->g1, dt));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words + 4, c->vel, c->geom, c->g0, std::min(c->g0 + 1, c->g1), dt));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words + 6, c->vel, c->geom, std::max(c->g1 - 1, c->g0), c->g1, dt));
    return SFL_OK;
}

int measure_reach(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int *reach_out,
                  int *reach_ext_out)
{
    int reach = 0, reach_ext = 0;
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        if (!c->d_reach) {
            void *m = nullptr;
            HIP_TRY(hipMalloc(&m, kReachWords * sizeof(int)));
            c->d_reach = static_cast<int *>(m);
        }
        HIP_TRY(hipMemsetAsync(c->d_reach, 0, kReachWords * sizeof(int), c->stream));
        SFL_TRY(launch_reach_set(c, c->d_reach, dt));
    }
    SFL_TRY(reduce_max_inline(ctx, ctx->d_reach, kReachWords));   // maximum over the ranks (RCCL: on the exchange stream)
    for (sfl_context *c : peers) {
        int r[kReachWords] = {0};
        SFL_TRY(use_device(c));
        HIP_TRY(hipMemcpyAsync(r, c->d_reach, sizeof r, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        reach = std::max(reach, reach_own(r));
        reach_ext = std::max(reach_ext, reach_extended(r));
    }
    *reach_out = reach;
    if (reach_ext_out) *reach_ext_out = reach_ext;
    return SFL_OK;
}

static int plan_advect(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, AdvectPlan *plan)
{
    *plan = AdvectPlan{};
    if (ctx->nranks == 1) return SFL_OK;
    if (ctx->opt_advect_halo > 0) {
        plan->halo = ctx->opt_advect_halo;
        return SFL_OK;
    }
    int reach = 0;
    SFL_TRY(measure_reach(ctx, peers, dt, &reach));
    plan->flag = false;
    if (reach <= kAdvectGhostRows && reach <= min_owned_rows(ctx))
        plan->halo = reach;
    else
        plan->gather = true;
    return SFL_OK;
}

// `extend` = 1 (slab_step_auto; plan.halo then covers one row more than the reach): the ghost rows next to the cuts
// are advected as well, redundantly -- calculate_divergence then needs no exchange of its own.
// `interior_done` = L > 0 (slab_step_auto): rows [g0 + L, g1 - L) are in vel_tmp already (advect_interior_early): only the two
// bands next to the cuts, the rows that may need the halo, are advected here.
int advect_velocity_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                                   const AdvectPlan &plan, int extend, int interior_done)
{
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    }
    if (plan.gather)
        SFL_TRY(gather_field(ctx, peers, SFL_FIELD_VELOCITY));
    else   // (with early rows in vel_tmp: the velocity was final before they were queued, the halo need not wait for them)
        SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_VELOCITY, plan.halo, 0, interior_done > 0 ? ctx->ev_vel_final : nullptr));
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        const sfl::Slab whole{c->dim_x, c->gdim_y, 0, c->gdim_y};
        if (plan.gather)
            HIP_TRY(sfl::launch_advect_vec2f(c->stream, c->vel_tmp, static_cast<const float *>(c->gather_buf),
                                             c->vel, c->geom, c->g0, c->g1, 0, c->gdim_y, dt, no_slip != 0,
                                             nullptr, &whole, c->opt_advect_kernel));
        else if (interior_done > 0)   // both bands in one launch
            HIP_TRY(sfl::launch_advect_vec2f(c->stream, c->vel_tmp, c->vel, c->vel, c->geom, clip_lo(c, c->g0 - extend),
                                             c->g0 + interior_done, clip_lo(c, c->g0 - plan.halo),
                                             clip_hi(c, c->g1 + plan.halo), dt, no_slip != 0, advect_flag(c, plan), nullptr,
                                             c->opt_advect_kernel, c->g1 - interior_done, clip_hi(c, c->g1 + extend)));
        else
            HIP_TRY(sfl::launch_advect_vec2f(c->stream, c->vel_tmp, c->vel, c->vel, c->geom,
                                             clip_lo(c, c->g0 - extend), clip_hi(c, c->g1 + extend),
                                             clip_lo(c, c->g0 - plan.halo), clip_hi(c, c->g1 + plan.halo), dt,
                                             no_slip != 0, advect_flag(c, plan), nullptr, c->opt_advect_kernel));
        std::swap(c->vel, c->vel_tmp);  // ino:255
        ++c->vel_epoch;
        c->v_ghost_valid = plan.gather ? 0 : extend;
    }
    return SFL_OK;
}

int advect_color_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                                const AdvectPlan &plan)
{
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
        SFL_TRY(ensure(c, c->col_tmp, 12, false));
    }
    if (plan.gather) {
        SFL_TRY(gather_field(ctx, peers, SFL_FIELD_COLOR));
    } else if (plan.halo_sent) {
        SFL_TRY(use_device(ctx));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_color_halo, 0));
    } else {
        SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_COLOR, plan.halo));
    }
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        const sfl::Slab whole{c->dim_x, c->gdim_y, 0, c->gdim_y};
        if (plan.gather)
            HIP_TRY(sfl::launch_advect_vec3uq32(c->stream, c->col_tmp, static_cast<const uint32_t *>(c->gather_buf),
                                                c->vel, c->geom, c->g0, c->g1, 0, c->gdim_y, dt, no_slip != 0,
                                                nullptr, &whole, c->opt_advect_kernel));
        else
            HIP_TRY(sfl::launch_advect_vec3uq32(c->stream, c->col_tmp, c->col, c->vel, c->geom, c->g0, c->g1,
                                                clip_lo(c, c->g0 - plan.halo), clip_hi(c, c->g1 + plan.halo), dt,
                                                no_slip != 0, advect_flag(c, plan), nullptr, c->opt_advect_kernel));
        std::swap(c->col, c->col_tmp);  // ino:286
    }
    return SFL_OK;
}

// Copies the queued (cell, velocity) pairs to the device (asynchronously, through pinned staging) and empties the
// queue; *count = how many now wait in d_force_cells / d_force_vel for the kernel that applies them.
int stage_queued_forces(sfl_context *c, int *count)
{
    const int n = (int)(c->force_cells.size() / 2);
    *count = n;
    if (n == 0) return SFL_OK;
    SFL_TRY(use_device(c));
    if (n > c->d_force_cap) {
        // the previous step's kernel may still read the old arrays: drain once, on growth only
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_force_cells) (void)hipFree(c->d_force_cells);
        if (c->d_force_vel) (void)hipFree(c->d_force_vel);
        c->d_force_cells = nullptr;
        c->d_force_vel = nullptr;
        c->d_force_cap = 0;
        void *a = nullptr, *b = nullptr;
        HIP_TRY(hipMalloc(&a, sizeof(int) * 2 * n));
        c->d_force_cells = static_cast<int *>(a);
        HIP_TRY(hipMalloc(&b, sizeof(float) * 2 * n));
        c->d_force_vel = static_cast<float *>(b);
        c->d_force_cap = n;
    }
    // stage in pinned memory so that the copies are truly asynchronous and the host vectors can be
    // cleared at once; a slot is reused every second step, after its own copy has completed
    sfl_context::ForceStage &st = c->force_stage[c->force_slot];
    c->force_slot ^= 1;
    if (!st.copied) HIP_TRY(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
    if (st.pending) {
        HIP_TRY(hipEventSynchronize(st.copied));
        st.pending = false;
    }
    if (n > st.cap) {
        if (st.cells) (void)hipHostFree(st.cells);
        if (st.vel) (void)hipHostFree(st.vel);
        st.cells = nullptr;
        st.vel = nullptr;
        st.cap = 0;
        void *a = nullptr, *b = nullptr;
        HIP_TRY(hipHostMalloc(&a, sizeof(int) * 2 * n, hipHostMallocDefault));
        st.cells = static_cast<int *>(a);
        HIP_TRY(hipHostMalloc(&b, sizeof(float) * 2 * n, hipHostMallocDefault));
        st.vel = static_cast<float *>(b);
        st.cap = n;
    }
    memcpy(st.cells, c->force_cells.data(), sizeof(int) * 2 * n);
    memcpy(st.vel, c->force_vel.data(), sizeof(float) * 2 * n);
    c->force_cells.clear();
    c->force_vel.clear();
    HIP_TRY(hipMemcpyAsync(c->d_force_cells, st.cells, sizeof(int) * 2 * n, hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_force_vel, st.vel, sizeof(float) * 2 * n, hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipEventRecord(st.copied, c->stream));
    st.pending = true;
    return SFL_OK;
}

int apply_queued_forces(sfl_context *c)
{
    int n = 0;
    SFL_TRY(stage_queued_forces(c, &n));
    if (n > 0) {  // (the exact ghost rows, if any, receive the forces that fall into them as well: every rank
                  // queues the same global list, include/sfl.h)
        HIP_TRY(sfl::launch_apply_forces(c->stream, c->vel, c->geom, clip_lo(c, c->g0 - c->v_ghost_valid),
                                         clip_hi(c, c->g1 + c->v_ghost_valid), c->d_force_cells, c->d_force_vel, n));
        ++c->vel_epoch;
    }
    return SFL_OK;
}

// ino:276 + ino:281-287 in one pass: project each cell's own velocity, advect the dye with it.
// halo_sent: the dye's halo is already on its way / there (slab_step_auto sends it at the start of the step, behind
// ev_color_halo): wait for it instead of exchanging
int project_and_advect_color(sfl_context *ctx, float dt, float dx, int halo, bool report, bool halo_sent)
{
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure_field(c, SFL_FIELD_PRESSURE));
        SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
        SFL_TRY(ensure(c, c->col_tmp, 12, false));
    }
    if (ctx->p_ghost_valid < 1) SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_PRESSURE, 1));
    if (halo_sent) {
        SFL_TRY(use_device(ctx));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_color_halo, 0));
    } else {
        SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_COLOR, halo));
    }
    const float two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:78-79
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        const int h = c->nranks > 1 ? halo : 0;
        // (with a report the tile kernel also measures the reach of the projected velocity: post_reach_report)
        c->reach_in_report = false;
        HIP_TRY(sfl::launch_project_advect_vec3uq32(
            c->stream, c->col_tmp, c->col, c->vel, c->p, c->geom, c->g0, c->g1, clip_lo(c, c->g0 - h),
            clip_hi(c, c->g1 + h), dt, false, c->nranks > 1 ? (report ? c->d_report + 2 : c->halo_flag) : nullptr,
            two_dx_inv, c->opt_advect_kernel, c->nranks > 1 && report ? &c->reach_in_report : nullptr));
        std::swap(c->col, c->col_tmp);  // ino:286
        ++c->vel_epoch;                 // the projection rewrote the velocity
        c->v_ghost_valid = 0;
    }
    return SFL_OK;
}

// ino:252-256 + ino:274 in one pass: possible when nothing happens between the two (no queued drag
// forces, ino:264-269) and every neighbour of every cell is on this GPU (whole-domain context)
bool can_fuse_divergence(const sfl_context *c)
{
    if (!c->opt_fuse_divergence || c->nranks != 1 || c->transport || !c->force_cells.empty()) return false;
    const int64_t cells = (int64_t)c->dim_x * c->gdim_y;
    return c->opt_advect_kernel == 2 || (c->opt_advect_kernel == 0 && cells >= sfl::kAdvectTiledMinCells);
}

int advect_velocity_and_divergence(sfl_context *c, float dt, float dx)
{
    SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
    SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    SFL_TRY(ensure_field(c, SFL_FIELD_DIVERGENCE));
    SFL_TRY(use_device(c));
    const float two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:36
    HIP_TRY(sfl::launch_advect_divergence_tiled(c->stream, c->vel_tmp, c->div, c->vel, c->geom, dt, true, two_dx_inv));
    std::swap(c->vel, c->vel_tmp);  // ino:255
    return SFL_OK;
}

static int check_channels(int channels, int kind)
{
    if (channels < 1 || channels > 3 || (kind != SFL_CHANNEL_F32 && kind != SFL_CHANNEL_UQ32))
        return fail(SFL_ERR_INVALID, "advect: element must be 1..3 channels of kind SFL_CHANNEL_F32 / SFL_CHANNEL_UQ32 "
                    "(got %d x kind %d)", channels, kind);
    return SFL_OK;
}

}  // namespace host
}  // namespace sfl

using namespace sfl::host;

extern "C" {

int sfl_advect_velocity(sfl_context *ctx, float dt, int no_slip)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    }
    AdvectPlan plan;
    SFL_TRY(plan_advect(ctx, peers, dt, &plan));
    return advect_velocity_planned(ctx, peers, dt, no_slip, plan);
}

int sfl_advect_color(sfl_context *ctx, float dt, int no_slip)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
        SFL_TRY(ensure(c, c->col_tmp, 12, false));
    }
    AdvectPlan plan;
    SFL_TRY(plan_advect(ctx, peers, dt, &plan));
    return advect_color_planned(ctx, peers, dt, no_slip, plan);
}

int sfl_advect_external(sfl_context *c, void *next_p_dev, const void *p_dev, int channels, int kind, float dt,
                        int no_slip)
{
    if (!c || !next_p_dev || !p_dev) return fail(SFL_ERR_INVALID, "NULL argument");
    if (next_p_dev == p_dev) return fail(SFL_ERR_INVALID, "advect: next_p must not alias p (advect.h:82)");
    SFL_TRY(check_channels(channels, kind));
    if (c->nranks != 1)
        return fail(SFL_ERR_STATE, "sfl_advect_external needs a whole-domain context (slab %d/%d): the caller's array "
                    "has no ghost rows", c->rank, c->nranks);
    SFL_TRY(settle_color(c));
    SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
    SFL_TRY(use_device(c));
    HIP_TRY(sfl::launch_advect_channels(c->stream, next_p_dev, p_dev, c->vel, c->dim_x, c->gdim_y, dt, no_slip != 0,
                                        channels, kind));
    return SFL_OK;
}

int sfl_calculate_divergence(sfl_context *ctx, float dx)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure_field(c, SFL_FIELD_DIVERGENCE));
    }
    if (ctx->v_ghost_valid < 1) SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_VELOCITY, 1));
    const float two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:36
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        HIP_TRY(sfl::launch_divergence(c->stream, c->div, c->vel, c->geom, c->g0, c->g1, two_dx_inv,
                                       c->opt_advect_kernel));
    }
    return SFL_OK;
}

int sfl_poisson_solve(sfl_context *ctx, float dx, int iters, float omega)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    return run_poisson(ctx, dx, iters, omega);
}

int sfl_subtract_gradient(sfl_context *ctx, float dx)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure_field(c, SFL_FIELD_PRESSURE));
    }
    if (ctx->p_ghost_valid < 1) SFL_TRY(exchange_inline(ctx, peers, SFL_FIELD_PRESSURE, 1));
    const float two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:78-79
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        HIP_TRY(sfl::launch_subtract_gradient(c->stream, c->vel, c->p, c->geom, c->g0, c->g1,
                                              two_dx_inv, c->opt_advect_kernel));
        ++c->vel_epoch;
        c->v_ghost_valid = 0;
    }
    return SFL_OK;
}

int sfl_queue_forces(sfl_context *ctx, const int *cells_ij, const float *vel_xy, int n)
{
    if (!ctx || n < 0 || (n > 0 && (!cells_ij || !vel_xy))) return fail(SFL_ERR_INVALID, "bad arguments");
    for (sfl_context *c : peers_of(ctx)) {
        c->force_cells.insert(c->force_cells.end(), cells_ij, cells_ij + 2 * n);
        c->force_vel.insert(c->force_vel.end(), vel_xy, vel_xy + 2 * n);
    }
    return SFL_OK;
}

// The sketch's own message (ino:45-48) with the sketch's own transform (ino:264-269): the touch task speaks
// graphics coordinates, the sim Cartesian ones rotated by 90 degrees -- cell = index(coords.y, coords.x),
// velocity = (velocity.y, velocity.x).
int sfl_queue_drags(sfl_context *ctx, const sfl_drag *msgs, int n)
{
    if (!ctx || n < 0 || (n > 0 && !msgs)) return fail(SFL_ERR_INVALID, "bad arguments");
    std::vector<int> cells((size_t)2 * n);
    std::vector<float> vel((size_t)2 * n);
    for (int k = 0; k < n; ++k) {
        const int i = msgs[k].coord_y, j = msgs[k].coord_x;   // ino:265: index(msg.coords.y, msg.coords.x, N_ROWS)
        if (i >= ctx->dim_x || j >= ctx->gdim_y)
            return fail(SFL_ERR_INVALID, "drag %d: coords (x %d, y %d) address cell (i %d, j %d) outside the %d x %d "
                        "domain (the sketch would write out of bounds)", k, j, i, i, j, ctx->dim_x, ctx->gdim_y);
        cells[2 * k] = i;
        cells[2 * k + 1] = j;
        vel[2 * k] = msgs[k].vel_y;                           // ino:266: swapped(msg.velocity.y, msg.velocity.x)
        vel[2 * k + 1] = msgs[k].vel_x;
    }
    return sfl_queue_forces(ctx, cells.data(), vel.data(), n);
}

int sfl_setup_sketch_fields(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (c->nranks != 1) return fail(SFL_ERR_STATE, "setup needs a whole-domain context (slab %d/%d)", c->rank, c->nranks);
    SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
    SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
    SFL_TRY(use_device(c));
    HIP_TRY(sfl::launch_setup_sketch_fields(c->stream, c->vel, c->col, c->dim_x, c->gdim_y));
    return SFL_OK;
}

int sfl_render_rgb565(sfl_context *c, int scaling, int byteswap, uint16_t *host_image, size_t bytes)
{
    if (!c || !host_image) return fail(SFL_ERR_INVALID, "NULL argument");
    if (scaling < 1 || scaling > 64) return fail(SFL_ERR_INVALID, "scaling must be 1..64 (got %d)", scaling);
    if (c->nranks != 1) return fail(SFL_ERR_STATE, "render needs a whole-domain context (slab %d/%d)", c->rank, c->nranks);
    const size_t w = (size_t)scaling * (c->gdim_y - 1), h = (size_t)scaling * (c->dim_x - 1);
    if (bytes != w * h * 2) return fail(SFL_ERR_INVALID, "image is %zu x %zu uint16 = %zu bytes, got %zu", h, w, w * h * 2, bytes);
    SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
    SFL_TRY(use_device(c));
    if (bytes > c->d_image_bytes) {  // the frame buffer stays with the context between frames
        if (c->d_image) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_image);
            c->d_image = nullptr;
            c->d_image_bytes = 0;
        }
        void *img = nullptr;
        HIP_TRY(hipMalloc(&img, bytes));
        c->d_image = static_cast<uint16_t *>(img);
        c->d_image_bytes = bytes;
    }
    HIP_TRY(sfl::launch_render_rgb565(c->stream, c->d_image, c->col, c->dim_x, c->gdim_y, scaling,
                                      byteswap != 0));
    HIP_TRY(hipMemcpyAsync(host_image, c->d_image, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller reads host_image on return
    return SFL_OK;
}

}  // extern "C"
