// slab_step.cpp -- the sim task's step (ino:249-289) on a context: sfl_step, sfl_step_n (with the kernel that joins two
// steps of a whole-domain context), and a slab's step on the automatic advection halo: no host round trip inside a step,
// the dye's guessed halo checked one call late (settle_color).  Host C++ only.
#include "transport.h"

namespace sfl {
namespace host {

static int small_grid_step(sfl_context *c, float dt, float dx, int iters, float omega)
{
    if (iters < 0) return fail(SFL_ERR_INVALID, "iters must be >= 0 (got %d)", iters);
    SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
    SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
    SFL_TRY(ensure(c, c->col_tmp, 12, false));
    SFL_TRY(ensure_field(c, SFL_FIELD_DIVERGENCE));
    SFL_TRY(use_device(c));
    int n_forces = 0;
    SFL_TRY(stage_queued_forces(c, &n_forces));
    sfl::SmallStep a{};
    a.v_in = c->vel;
    a.v_out = c->vel_tmp;
    a.col_in = c->col;
    a.col_out = c->col_tmp;
    a.div = c->div;
    a.p = c->p;
    a.dim_x = c->dim_x;
    a.dim_y = c->gdim_y;
    a.iters = iters;
    a.dt = dt;
    a.two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:36, :78-79
    a.prm = sor_params(c, dx, omega);
    a.force_cells = c->d_force_cells;
    a.force_vel = c->d_force_vel;
    a.n_forces = n_forces;
    HIP_TRY(sfl::launch_small_step(c->stream, a));
    std::swap(c->vel, c->vel_tmp);  // ino:255
    std::swap(c->col, c->col_tmp);  // ino:286
    c->last_launches = 1;
    c->last_exchanges = 0;
    c->last_fuse = 2 * iters;
    return SFL_OK;
}

// ---- automatic advection halo without a host round trip inside the step ------------------------------------
// The reach of the back-traces depends on the velocity, known only on the device.  Two facts make a step without
// a mid-step read-back possible:
//   * the velocity advection of step k (ino:252-256) back-traces with the velocity step k - 1 left behind --
//     the very field step k - 1's dye advection (ino:281-287) back-traced with, at the same dt.  Its reach has
//     been measured by then: the halo of the velocity advection is EXACT, no guess;
//   * the dye advection is the LAST operator of a step and writes into the other colour buffer.  It runs on a
//     GUESSED halo (the reach known at the start of the step plus a margin); the kernel raises a flag when a
//     back-trace leaves it, a small kernel measures the true reach of the projected velocity, both are reduced
//     over the ranks on the exchange stream and copied to pinned host memory behind an event.  Whoever touches
//     the context next (the next step, a download, sfl_synchronize) looks at the report first: flag down = done,
//     reach recorded for the next step; flag up = the old colour buffer is still intact, the dye advection alone
//     is repeated with the exact reach (or the gathered field).  Nothing downstream ever saw the wrong dye.
static int ensure_report(sfl_context *c)
{
    if (c->d_report) return SFL_OK;
    SFL_TRY(use_device(c));
    void *d = nullptr, *h = nullptr;
    HIP_TRY(hipMalloc(&d, kReachWords * sizeof(int)));
    HIP_TRY(hipMemset(d, 0, kReachWords * sizeof(int)));
    HIP_TRY(hipHostMalloc(&h, kReportWords * sizeof(int), hipHostMallocDefault));
    memset(h, 0, kReportWords * sizeof(int));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_report, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_color_halo, hipEventDisableTiming));
    c->d_report = static_cast<int *>(d);
    c->h_report = static_cast<int *>(h);
    return SFL_OK;
}

// Measure the reach of the back-traces of the CURRENT velocity (the flag word of the report has been written by
// the advection kernel before), reduce over the ranks, start the copy to the host.  No host wait.
static int post_reach_report(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt)
{
    for (sfl_context *c : peers) {   // (unless the dye's kernel has measured it on the way: project_and_advect_color)
        if (!c->reach_in_report) SFL_TRY(launch_reach_set(c, c->d_report, dt));
        c->disp_in_report = c->reach_in_report;
        c->reach_in_report = false;
    }
    if (reduces_on_device(ctx)) {  // maximum over the ranks, on the exchange stream like every RCCL operation
        SFL_TRY(reduce_max_then_copy(ctx, ctx->d_report, kReachWords, ctx->h_report, ctx->ev_report));
    } else {
        for (sfl_context *c : peers) {
            SFL_TRY(use_device(c));
            HIP_TRY(hipMemcpyAsync(c->h_report, c->d_report, kReachWords * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            // the "a wait inside a solve gave up" word rides along: the next call on the context sees it
            HIP_TRY(hipMemcpyAsync(c->h_report + kReachWords, c->halo_flag + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(c->ev_report, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_report, 0, kReachWords * sizeof(int), c->stream));   // for the next step's dye kernel
            c->report_zeroed = true;
        }
    }
    for (sfl_context *c : peers) {
        c->color_unsettled = true;
        c->unsettled_dt = dt;
    }
    return SFL_OK;
}

// Examine the report of the last dye advection that ran on a guessed halo (see above); repeat it when the guess
// was short.  Cheap when nothing is pending.  Every entry point that reads or writes the fields calls it.
// The option blocks of RCCL ranks are compared again -- a collective -- only from operators every rank issues
// (`collective`): a rank-local sfl_download or sfl_synchronize right after sfl_set_option must not wait for peers
// that are not there; sfl_set_option on a communicator's context is to be called by all ranks.
int settle_color(sfl_context *ctx, bool collective)
{
    if (collective && ctx->options_dirty && ctx->transport && ctx->transport->separate_processes())
        SFL_TRY(sfl_comm_check_options(ctx));
    if (!ctx->color_unsettled) return SFL_OK;
    std::vector<sfl_context *> peers = peers_of(ctx);
    int reach = 0, reach_ext = 0, flag = 0, disp = 0;
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        HIP_TRY(hipEventSynchronize(c->ev_report));  // (the step it belongs to has long been queued; no stream is drained)
        reach = std::max(reach, reach_own(c->h_report));
        reach_ext = std::max(reach_ext, reach_extended(c->h_report));
        flag |= c->h_report[2];
        disp = c->disp_in_report && disp >= 0 ? std::max(disp, c->h_report[3]) : -1;
        if (c->h_report[kReachWords]) c->wait_error_seen = true;   // (reported by the operators' check_wait_error / sfl_synchronize)
        c->color_unsettled = false;
    }
    const float dt = ctx->unsettled_dt;
    for (sfl_context *c : peers) {  // the reach of the back-traces of the velocity as it stands now
        c->known_disp = disp;
        c->known_reach = reach;
        c->known_reach_ext = reach_ext;
        c->known_epoch = c->vel_epoch;
        c->known_dt = dt;
    }
    if (!flag) return SFL_OK;
    // the guess was short: back to the colour the step started with, advect again with what is now known
    AdvectPlan plan;
    plan.flag = false;
    if (reach <= kAdvectGhostRows && reach <= min_owned_rows(ctx))
        plan.halo = reach;
    else
        plan.gather = true;
    for (sfl_context *c : peers) std::swap(c->col, c->col_tmp);
    SFL_TRY(advect_color_planned(ctx, peers, dt, 0, plan));
    // The dye has just been rewritten on the compute stream.  A step that recorded "velocity and dye are final" before
    // this (advect_interior_early, called in front of settle_color) would let the dye's halo leave behind THAT event, i.e.
    // possibly before or while the repeat writes the rows it carries: the event is recorded again, behind
    // the repeat (later than necessary for the velocity's halo, which also starts behind it; this path is rare).
    if (ctx->vel_final_recorded) {
        SFL_TRY(use_device(ctx));
        HIP_TRY(hipEventRecord(ctx->ev_vel_final, ctx->stream));
    }
    return SFL_OK;
}

// One step of a slab group with the automatic advection halo (SFL_OPT_ADVECT_HALO = 0).
static int slab_step_auto(sfl_context *ctx, float dt, float dx, int iters, float omega)
{
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_report(c));
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    }
    const int limit = std::min(kAdvectGhostRows, min_owned_rows(ctx));
    const bool known = ctx->known_reach >= 0 && ctx->known_epoch == ctx->vel_epoch && ctx->known_dt == dt;
    int reach_v = 0, reach_v_ext = 0;  // halo for the owned rows' back-traces / for those of own +- 1 rows
    if (known) {
        reach_v = ctx->known_reach;
        reach_v_ext = ctx->known_reach_ext;
    } else {  // first step, or the velocity was written from outside: one measured advection (a host round trip)
        SFL_TRY(measure_reach(ctx, peers, dt, &reach_v, &reach_v_ext));
    }
    // Two of the step's small exchanges are traded for one redundant row each: the velocity advection also advects
    // the ghost row next to each cut (halo = reach_extended: the neighbours' edge rows trace into THEIR slabs), so calculate_divergence finds
    // its neighbours' rows in place; and the solve leaves one ghost row of p exact (plan tail), which is all
    // subtract_gradient reads beyond the cut.
    AdvectPlan pv;
    // exact by construction; armed all the same (a back-trace that leaves it -> SFL_ERR_HALO at sfl_synchronize) --
    // except on an emulated rank, whose ghost rows hold copies of its own rows: meaningless values, timing only
    pv.flag = ctx->transport && ctx->transport->kind() != 3 && ctx->transport->kind() != 4;
    int extend = 0;
    if (reach_v_ext <= limit) {
        pv.halo = std::max(reach_v_ext, 1);
        extend = 1;
    } else if (reach_v <= limit) {
        pv.halo = reach_v;
    } else {
        pv.gather = true;
    }
    // (rows out of the cuts' reach may be in vel_tmp already, advected while the host was waiting for the report: sfl_step)
    int interior_done = 0;
    if (known && extend && !pv.gather && ctx->early_rows > 0 && ctx->early_epoch == ctx->vel_epoch && ctx->early_dt == dt &&
        ctx->known_disp >= 0 && ctx->known_disp <= ctx->early_rows && pv.halo <= ctx->early_rows)
        interior_done = ctx->early_rows;   // no cell of those rows read beyond the slab: what is in vel_tmp is the advection
    for (sfl_context *c : peers) {
        c->early_rows = 0;
        c->last_early_kept = interior_done;
    }
    SFL_TRY(advect_velocity_planned(ctx, peers, dt, 1, pv, extend, interior_done));   // ino:252-256, exact halo
    for (sfl_context *c : peers) SFL_TRY(apply_queued_forces(c));           // ino:264-269
    SFL_TRY(sfl_calculate_divergence(ctx, dx));                             // ino:274
    for (sfl_context *c : peers) c->solve_tail = 1;
    const int rc_solve = sfl_poisson_solve(ctx, dx, iters, omega);          // ino:275
    for (sfl_context *c : peers) c->solve_tail = 0;
    SFL_TRY(rc_solve);
    // The dye is not touched before the end of the step and its halo is a guess made from what was known at the START of the
    // step (the projection changes the velocity a little, forces may change it a lot: checked after the step).  It is queued
    // HERE, behind the solve's exchanges on the exchange stream and after everything the GPU is waiting for has been queued:
    // the host needs 25 - 30 us for it, which used to stand between the velocity advection and the divergence.
    const int guess = std::min(limit, std::max(2, reach_v + 2 + reach_v / 4));
    {
        for (sfl_context *c : peers) {
            SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
            SFL_TRY(ensure(c, c->col_tmp, 12, false));
        }
        Overlap o;
        SFL_TRY(overlap_of(ctx, &o));
        if (ctx->vel_final_recorded) {   // the dye has been final since the step began: nothing of this step to wait for
            SFL_TRY(use_device(ctx));
            HIP_TRY(hipStreamWaitEvent(o.xstream, ctx->ev_vel_final, 0));
            SFL_TRY(exchange(peers, SFL_FIELD_COLOR, guess, o.xstream));
        } else {
            SFL_TRY(start_exchange(peers, o, SFL_FIELD_COLOR, guess, 0, false));
        }
        SFL_TRY(use_device(ctx));
        HIP_TRY(hipEventRecord(ctx->ev_color_halo, o.xstream));
        ctx->vel_final_recorded = false;
    }
    // dye advection on that guessed halo.  Its report words are zero: where the last report was copied to the host on the compute
    // stream they were zeroed right behind that copy (post_reach_report: the GPU is idle there and the host far ahead -- here the
    // memset stood between the solve and the dye's kernel, at the start of the step it would stand in the host's way), else now
    for (sfl_context *c : peers) {
        if (c->report_zeroed) continue;
        SFL_TRY(use_device(c));
        HIP_TRY(hipMemsetAsync(c->d_report, 0, kReachWords * sizeof(int), c->stream));
    }
    for (sfl_context *c : peers) c->report_zeroed = false;
    if (ctx->opt_fuse_projection) {
        SFL_TRY(project_and_advect_color(ctx, dt, dx, guess, true, true));  // ino:276 + ino:281-287, one pass over v
    } else {
        SFL_TRY(sfl_subtract_gradient(ctx, dx));                            // ino:276
        AdvectPlan pc;
        pc.halo = guess;
        pc.report = true;
        pc.halo_sent = true;
        SFL_TRY(advect_color_planned(ctx, peers, dt, 0, pc));               // ino:281-287
    }
    return post_reach_report(ctx, peers, dt);
}

// slab_step_auto's host has to read the last step's report (the reach of the projected velocity: the halo of this step's
// velocity advection; whether the dye's guess held) before it can queue the step -- 25 - 60 us in which the GPU has nothing to do.
// The rows further than the largest possible halo from both cuts need no halo at all: their advection is queued BEFORE the wait.
// The report also says how far from its own row any cell's sources lie (word [3], from the dye's kernel): within that halo,
// those rows never read beyond the slab and what was advected early stands; otherwise the step advects everything again.
static int advect_interior_early(sfl_context *ctx, float dt)
{
    std::vector<sfl_context *> peers = peers_of(ctx);
    const int limit = std::min(kAdvectGhostRows, min_owned_rows(ctx));
    for (sfl_context *c : peers)
        if (!c->vel || !c->vel_tmp || c->g1 - c->g0 < 2 * limit + 64) return SFL_OK;   // (nothing worth it, or not set up yet)
    SFL_TRY(use_device(ctx));
    if (!ctx->ev_vel_final) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_vel_final, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->ev_vel_final, ctx->stream));   // the velocity (and the dye) as the last step left them: what the halos will carry
    ctx->vel_final_recorded = true;
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        HIP_TRY(sfl::launch_advect_vec2f(c->stream, c->vel_tmp, c->vel, c->vel, c->geom, c->g0 + limit, c->g1 - limit, c->g0,
                                         c->g1, dt, true, nullptr, nullptr, c->opt_advect_kernel));
        c->early_rows = limit;
        c->early_epoch = c->vel_epoch;
        c->early_dt = dt;
    }
    return SFL_OK;
}

// ino:276 + ino:281-287 of one step and ino:252-256 + ino:274 of the next as one kernel (kernels.h launch_step_seam_tiled)
static int step_seam(sfl_context *c, float dt, float dx)
{
    SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
    SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    SFL_TRY(ensure_field(c, SFL_FIELD_COLOR));
    SFL_TRY(ensure(c, c->col_tmp, 12, false));
    SFL_TRY(ensure_field(c, SFL_FIELD_PRESSURE));
    SFL_TRY(use_device(c));
    const float two_dx_inv = 1.0f / (2.0f * dx);  // finitediff.cpp:36, :78-79
    HIP_TRY(sfl::launch_step_seam_tiled(c->stream, c->col_tmp, c->col, c->vel_tmp, c->div, c->vel, c->p, c->geom, dt,
                                        two_dx_inv));
    std::swap(c->col, c->col_tmp);  // ino:286
    std::swap(c->vel, c->vel_tmp);  // ino:255 of the next step (the projected velocity of this one was never stored)
    c->vel_epoch += 2;
    return SFL_OK;
}

}  // namespace host
}  // namespace sfl

using namespace sfl::host;

extern "C" {

int sfl_step(sfl_context *ctx, float dt, float dx, int iters, float omega)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (ctx->nranks > 1 && ctx->opt_advect_halo == 0 && ctx->color_unsettled && ctx->unsettled_dt == dt && ctx->force_cells.empty())
        SFL_TRY(advect_interior_early(ctx, dt));
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    if (small_grid(ctx)) return small_grid_step(ctx, dt, dx, iters, omega);
    if (ctx->nranks > 1 && ctx->opt_advect_halo == 0) return slab_step_auto(ctx, dt, dx, iters, omega);
    if (can_fuse_divergence(ctx)) {
        SFL_TRY(advect_velocity_and_divergence(ctx, dt, dx));  // ino:252-256 + ino:274
    } else {
        SFL_TRY(sfl_advect_velocity(ctx, dt, 1));              // ino:252-256
        for (sfl_context *c : peers_of(ctx)) SFL_TRY(apply_queued_forces(c));  // ino:264-269
        SFL_TRY(sfl_calculate_divergence(ctx, dx));            // ino:274
    }
    SFL_TRY(sfl_poisson_solve(ctx, dx, iters, omega));     // ino:275
    if (ctx->opt_fuse_projection) {
        SFL_TRY(project_and_advect_color(ctx, dt, dx, ctx->opt_advect_halo, false));  // ino:276 + ino:281-287, one pass over v
    } else {
        SFL_TRY(sfl_subtract_gradient(ctx, dx));           // ino:276
        SFL_TRY(sfl_advect_color(ctx, dt, 0));             // ino:281-287
    }
    return SFL_OK;
}

// The loop of the sim task (ino:249-289) calls the step back to back.  n steps in one call give the library the one
// fusion a per-step API has no place for: between two steps the projected velocity is written by the last kernel of
// one and read straight back by the first kernel of the next -- the seam kernel does both and never stores it
// (780 us at 8192^2 where the two kernels take 568 + 254; SFL_OPT_STEP_SEAMS, profiles/r04_step_seam.txt).
int sfl_step_n(sfl_context *ctx, int n, float dt, float dx, int iters, float omega)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (n < 0) return fail(SFL_ERR_INVALID, "n must be >= 0 (got %d)", n);
    SFL_TRY(settle_color(ctx, true));
    SFL_TRY(check_wait_error(ctx));
    const int64_t cells = (int64_t)ctx->dim_x * ctx->gdim_y;
    const bool tiled = ctx->opt_advect_kernel == 2 || (ctx->opt_advect_kernel == 0 && cells >= sfl::kAdvectTiledMinCells);
    const bool seams = n > 1 && ctx->opt_step_seams && ctx->nranks == 1 && !ctx->transport && !small_grid(ctx) && tiled &&
                       ctx->opt_fuse_projection && ctx->opt_fuse_divergence;
    if (!seams) {
        for (int k = 0; k < n; ++k) SFL_TRY(sfl_step(ctx, dt, dx, iters, omega));
        return SFL_OK;
    }
    // head of the first step: as sfl_step (queued forces go between its advection and its divergence, ino:264-269)
    if (can_fuse_divergence(ctx)) {
        SFL_TRY(advect_velocity_and_divergence(ctx, dt, dx));
    } else {
        SFL_TRY(sfl_advect_velocity(ctx, dt, 1));
        SFL_TRY(apply_queued_forces(ctx));
        SFL_TRY(sfl_calculate_divergence(ctx, dx));
    }
    for (int k = 0; k < n; ++k) {
        SFL_TRY(sfl_poisson_solve(ctx, dx, iters, omega));                                        // ino:275
        if (k + 1 < n)
            SFL_TRY(step_seam(ctx, dt, dx));                                                      // ino:276, :281-287 | :252-256, :274
        else
            SFL_TRY(project_and_advect_color(ctx, dt, dx, ctx->opt_advect_halo, false));          // ino:276 + ino:281-287
    }
    return SFL_OK;
}

}  // extern "C"
