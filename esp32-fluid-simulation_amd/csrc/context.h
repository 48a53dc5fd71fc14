// context.h -- what the host-side translation units of libsfl_hip.so share (internal; the public boundary is
// include/sfl.h): the context struct, the error plumbing and the helpers every unit needs.
//
//   context.cpp       errors, device / plan queries, create / destroy, options, field I/O, synchronize, timers
//   transport.{h,cpp} halo transports behind one interface (virtual-rank group, RCCL, emulated rank), the exchange
//                     protocol around them, communicator attach / option check
//   sor_executor.cpp  poisson_solve: walks slab_plan.h programs (in line, early exchanges behind events, exchanges in
//                     time counted on the device)
//   operators.cpp     advection, divergence, projection, forces, setup / render
//   slab_step.cpp     sfl_step / sfl_step_n, the automatic advection halo of a slab's step
//   host_dropin.cpp   the host-pointer drop-ins (sfl_host_*) and their per-thread context
//
// There is deliberately no CPU compute path in any of them: every operator ends in a kernel launch and fails with
// SFL_ERR_HIP when no device is usable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sfl.h"
#include "kernels.h"
#include "slab_plan.h"

namespace sfl {
namespace host {

// message of the last failing call on this thread (sfl_last_error)
std::string &last_error();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace host
}  // namespace sfl

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return ::sfl::host::fail(e_ == hipErrorOutOfMemory ? SFL_ERR_NOMEM : SFL_ERR_HIP,          \
                                     "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                                     __LINE__);                                                        \
    } while (0)

#define SFL_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != SFL_OK) return rc_; \
    } while (0)

namespace sfl {
namespace host {

constexpr int kGhostRows = 160;      // ghost rows allocated per side on a slab (nranks > 1): the deepest halo a solve's superstep may use
constexpr int kAdvectGhostRows = 64;  // ... of which an advection's halo may use this many (beyond: the field is gathered)
constexpr int kLegacySorHalo = 64;    // the solve's halo before it was chosen from the measured exchange (still its starting point)
constexpr size_t kAlternateSweepCells = 48u << 20;  // local cells from which successive SOR launches alternate direction
// Reach words of a slab (device ints, atomicMax'ed by backtrace_reach_kernel; zero them first):
//   [0] / [1]  rows the back-traces of the OWNED rows need below / above the slab;  [2] a back-trace left a guessed halo;
//   [3]        rows a cell's sources lie from its own row at most (the dye's tile kernel measures it);
//   [5]        rows the back-traces of the slab's FIRST row need above it, [6] those of its LAST row below it --
//              what the neighbour needs when it advects that row itself as a ghost row (slab_step_auto: the
//              velocity advection covers own +- 1 rows); [4], [7] come with them and are covered by [0] / [1].
// The pinned host copy of a report carries one word more: [8] = the context's "a wait inside a solve gave up" word.
constexpr int kReachWords = 8;
constexpr int kReportWords = kReachWords + 1;

class Transport;
class Group;

// The automatic halo depth of a slab's solves is DECIDED BY TIMING REAL SOLVES (sor_executor.cpp choose_halo): a model with the
// measured exchange as its input names up to three candidate depths, each of the first solves of a kind -- same iteration
// count, fuse depth, tail and schedule -- runs on one of them between two events (every depth gives the same bits), and the
// fastest is kept for that kind from then on (another depth has to beat the legacy one by 1.5 %).
constexpr int kCollectiveWords = 8;   // sfl_context::d_collective

struct HaloTuner {
    struct Kind {
        int iters, fuse, tail, in_time;
        bool operator==(const Kind &o) const { return iters == o.iters && fuse == o.fuse && tail == o.tail && in_time == o.in_time; }
    };
    struct Decided {
        Kind kind;
        int halo;
    };
    static constexpr int kCandidates = 3, kSolvesEach = 4;   // (the first round is not timed: new tilings, cold caches)
    static constexpr int kExplore = kCandidates * kSolvesEach;
    std::vector<Decided> decided;
    bool active = false;
    Kind kind{};
    int cand[kCandidates] = {0, 0, 0}, ncand = 0;
    int solve_no = 0;                    // exploratory solves issued so far
    int which[kExplore] = {0};           // the candidate exploratory solve n ran on
    // a pair of events around every exploratory solve, read TOGETHER when the kind is decided: a host wait between two solves
    // would time them in another regime than the one they run in afterwards (the GPU waiting for the host to queue launches
    // favours the plan with fewer launches: the deepest halo was chosen 4 % too slow)
    hipEvent_t ev[2 * kExplore] = {nullptr};
    ~HaloTuner()
    {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

}  // namespace host
}  // namespace sfl

struct sfl_context {
    int device = 0;
    int dim_x = 0, gdim_y = 0;
    int rank = 0, nranks = 1;
    int g0 = 0, g1 = 0;  // owned global rows
    int ghost = 0;       // ghost rows per side
    sfl::Slab geom{};
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // halo exchanges of a solve run on their own stream so that they overlap the launches that do
    // not depend on them (a slab with a transport of its own; a linked group keeps these in Group).  Created when the
    // transport is attached, not at the first solve; whether it really runs beside `stream` -- the runtime folds its streams
    // onto a few hardware queues, and a launch that waits for a message inside the kernel must not share a queue with the
    // stream that carries the message -- is MEASURED (transport.cpp streams_run_concurrently), never assumed.
    hipStream_t xstream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_arrived = nullptr;

    // fields: local arrays of geom.lrows rows (allocated on first use)
    float *vel = nullptr, *vel_tmp = nullptr;
    uint32_t *col = nullptr, *col_tmp = nullptr;
    // div, p and p_alt are three thirds of ONE allocation (sor_block)
    float *sor_block = nullptr;
    float *div = nullptr;
    float *p = nullptr, *p_alt = nullptr;  // p = current pressure, p_alt = ping-pong partner
    int *halo_flag = nullptr;      // device words: [0] a back-trace left a fixed advection halo
    int *d_arrival = nullptr;      // [1] halo messages arrived (and relaxed) so far: what cut-adjacent tiles poll inside a
                                   //     launch (kernels.h HaloWait); [2] such a wait timed out
    int arrival_epoch = 0;         // the last value queued for [1]
    int *d_collective = nullptr;   // [4 .. 4 + kCollectiveWords) scratch of the ranks' small reductions (measure_exchange, the halo tuner): allocated
                                   //     with the context, so that no rank can drop out of a collective over a failed allocation
    int *d_done = nullptr;         // [3] sender tiles finished so far (kernels.h HaloWait::done): what the exchange stream waits
    int done_target = 0;           //     for before a halo message leaves; done_target = the count the launches queued so far reach
    bool wait_error_seen = false;  // word [2] was found raised (download, a step's report): every call fails until sfl_synchronize
                                   // has reported and cleared it

    // queued point forces (ino:264-269)
    std::vector<int> force_cells;
    std::vector<float> force_vel;
    int *d_force_cells = nullptr;
    float *d_force_vel = nullptr;
    int d_force_cap = 0;
    // pinned staging of the queued forces, two slots used alternately: the copy of step k may
    // still be in flight while step k + 1 is being queued, never the one of step k - 1 (each
    // slot's last copy is fenced by its event before the slot is rewritten)
    struct ForceStage {
        int *cells = nullptr;
        float *vel = nullptr;
        int cap = 0;
        hipEvent_t copied = nullptr;
        bool pending = false;
    } force_stage[2];
    int force_slot = 0;

    // dye visualiser: device image + pinned host staging, kept between frames
    uint16_t *d_image = nullptr;
    size_t d_image_bytes = 0;

    // scratch field of sfl_host_advect_vec2f when the advected field is not the velocity
    float *host_scratch = nullptr;

    // automatic advection halo (SFL_OPT_ADVECT_HALO = 0): device scratch of the back-trace reach
    // {below, above}, and the whole advected field gathered on this GPU when the reach outruns the
    // ghost rows (allocated on first need; sized for the 12-byte dye element)
    int *d_reach = nullptr;
    void *gather_buf = nullptr;
    // ... without a host round trip inside sfl_step (slab_step_auto): the dye advection runs on a GUESSED halo,
    // the true reach of its back-traces and an "a back-trace left the halo" flag are reduced on the device,
    // land in pinned host memory behind ev_report, and are examined when the NEXT call touches the context
    int *d_report = nullptr;       // device reach words (launch_reach_set) with the flag in word [2]
    int *h_report = nullptr;       // pinned host copy (kReportWords)
    bool report_zeroed = false;    // d_report has been zeroed behind its copy to the host (post_reach_report)
    bool reach_in_report = false;  // the dye's kernel of this step has left the reach words in d_report already
    // slab_step_auto: the rows further than `early_rows` from both cuts were advected into vel_tmp BEFORE the host waited for the
    // last step's report (advect_interior_early); valid for the velocity of vel_epoch == early_epoch at early_dt
    int early_rows = 0;
    uint64_t early_epoch = 0;
    float early_dt = 0.0f;
    bool disp_in_report = false;   // the pending report carries word [3] (the tile kernel measured it)
    int last_early_kept = 0;       // slab_step_auto: rows from each cut beyond which the last step kept the early advection (0: none)
    int known_disp = -1;           // rows a cell's sources lie from its own row at most, for the velocity of known_epoch (-1: unknown)
    hipEvent_t ev_report = nullptr;
    hipEvent_t ev_color_halo = nullptr;  // the dye's halo, sent at the START of a step (slab_step_auto), has arrived
    hipEvent_t ev_vel_final = nullptr;   // recorded in front of the early interior advection (advect_interior_early)
    bool vel_final_recorded = false;     // ... in the step that is being queued
    bool color_unsettled = false;  // a dye advection on a guessed halo has not been checked yet
    float unsettled_dt = 0.0f;
    int known_reach = -1;          // reach of the back-traces of the CURRENT velocity at known_dt (-1: unknown)
    int known_reach_ext = -1;      // ... when own +- 1 rows are advected (reach_extended)
    uint64_t known_epoch = 0, vel_epoch = 1;   // vel_epoch counts the writes to the velocity field
    float known_dt = 0.0f;

    // the two halves of SFL_OPT_EXCHANGE_SCHEDULE: opt_sor_overlap 0 = in line (1); opt_sor_arrival -1 = automatic (the transport's choice,
    // Transport::arrival_by_default), 0 = behind events (2), 1 = in time (3)
    int opt_sor_kernel = 0, opt_sor_fuse = 0, opt_advect_halo = 0, opt_sor_rows = 0,
        opt_sor_lane_cells = 0, opt_sor_halo = 0, opt_fuse_projection = 1, opt_sor_overlap = 1,
        opt_advect_kernel = 0, opt_fuse_divergence = 1, opt_small_grid = 1, opt_emulate_wire_us = 0, opt_sor_arrival = -1,
        opt_step_seams = 1, opt_halo_timeout_ms = 0, opt_sor_fold = 0;

    // how halo rows reach the neighbouring slabs (transport.h); null on a whole-domain context and on a slab that has
    // not been attached / linked yet.  `group` = the same object when it is an in-process group of virtual ranks.
    std::shared_ptr<sfl::host::Transport> transport;
    sfl::host::Group *group = nullptr;
    std::shared_ptr<sfl::host::Transport> keepalive;   // a dissolved group's streams, while this context still runs on them
    bool options_dirty = false;         // an option changed since the ranks last compared their option blocks
    int streams_concurrent = -1;        // compute and exchange stream were seen to run side by side: 1, seen not to: 0, untested: -1
    // One halo exchange of this slab's transport, measured at attach / before the first solve (transport.cpp
    // measure_exchange): microseconds of a message of ~0 rows (latency: launch, protocol, wire) and nanoseconds per further
    // row of p (bandwidth).  Maximum over the ranks: every rank derives the same halo depth from it.  -1: not measured.
    int exchange_latency_us = -1, exchange_ns_per_row = 0;
    int last_halo = 0;                  // halo depth of the last solve's plan (SFL_OPT_LAST_HALO)
    sfl::host::HaloTuner halo_tuner;    // (a linked group keeps its own: Group::halo_tuner)

    int last_launches = 0, last_exchanges = 0, last_fuse = 0;
    int solve_tail = 0;   // ghost rows of p the next solve must leave exact (slab_step_auto: 1, for subtract_gradient)
    int p_ghost_valid = 0;  // ghost rows of p that are exact right now (set by the solve, cleared by whoever writes p)
    int v_ghost_valid = 0;  // ghost rows of the velocity that are exact right now (slab_step_auto advects own +- 1 rows)

    size_t local_cells() const { return (size_t)geom.lrows * dim_x; }
    size_t owned_offset_cells() const { return (size_t)ghost * dim_x; }
};

namespace sfl {
namespace host {

size_t field_elem_bytes(int field);
int use_device(sfl_context *c);
int ensure_bytes(sfl_context *c, void **ptr, size_t elem_bytes, bool zero);
template <class T>
int ensure(sfl_context *c, T *&ptr, size_t elem_bytes, bool zero)
{
    if (ptr) return SFL_OK;
    void *m = nullptr;
    SFL_TRY(ensure_bytes(c, &m, elem_bytes, zero));
    ptr = static_cast<T *>(m);
    return SFL_OK;
}
int ensure_field(sfl_context *c, int field);
void *field_ptr(sfl_context *c, int field);
// the contexts whose programs one host thread issues together: the members of a linked group, else the context itself
std::vector<sfl_context *> peers_of(sfl_context *c);
int min_owned_rows(const sfl_context *c);
inline int clip_lo(const sfl_context *, int g) { return g < 0 ? 0 : g; }
inline int clip_hi(const sfl_context *c, int g) { return g > c->gdim_y ? c->gdim_y : g; }
int upload_raw(sfl_context *c, void *dev, const void *host, size_t elem_bytes);
int download_raw(sfl_context *c, const void *dev, void *host, size_t elem_bytes);
// fails (SFL_ERR_HIP) when a wait inside one of the context's solves is known to have given up: p is not valid
int check_wait_error(sfl_context *c);

// ---- sor_executor.cpp ----
int run_poisson(sfl_context *ctx, float dx, int iters, float omega);
SorParams sor_params(const sfl_context *c, float dx, float omega);
// one workgroup, fields in LDS (small_grid.hip): may this context take that path?
bool small_grid(const sfl_context *c);
// halo timeout of the waits inside this context's launches, microseconds (kernels.h HaloWait::timeout_us)
int halo_timeout_us(const sfl_context *c);
// will the next solve count its halo exchanges on the device (SFL_OPT_EXCHANGE_SCHEDULE resolved; the streams' verdict is in)?
bool in_time_exchanges(const sfl_context *c);
// halo depth of a solve's supersteps: the option, or chosen from the measured exchange (sor_executor.cpp)
int effective_halo(const sfl_context *c, int fuse, int iters, bool in_time);

// ---- operators.cpp ----
struct AdvectPlan {
    int halo = 0;         // rows to exchange per side (fixed or measured)
    bool gather = false;  // sample a gathered copy of the whole field instead
    bool flag = true;     // fixed halo: let the kernel report a back-trace that leaves it
    bool report = false;  // ... into the context's reach report (a guessed halo, checked by settle_color)
                          // instead of the error flag sfl_synchronize turns into SFL_ERR_HALO
    bool halo_sent = false;  // the halo rows are already on their way (behind ev_color_halo): wait, do not exchange
};
int launch_reach_set(sfl_context *c, int *words, float dt);
inline int reach_own(const int *w) { return std::max(w[0], w[1]); }
inline int reach_extended(const int *w) { return std::max(reach_own(w), 1 + std::max(w[5], w[6])); }
int measure_reach(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int *reach_out,
                  int *reach_ext_out = nullptr);
int advect_velocity_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                            const AdvectPlan &plan, int extend = 0, int interior_done = 0);
int advect_color_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                         const AdvectPlan &plan);
int apply_queued_forces(sfl_context *c);
int stage_queued_forces(sfl_context *c, int *count);
int project_and_advect_color(sfl_context *ctx, float dt, float dx, int halo, bool report, bool halo_sent = false);
bool can_fuse_divergence(const sfl_context *c);
int advect_velocity_and_divergence(sfl_context *c, float dt, float dx);

// ---- slab_step.cpp ----
// Examine the report of the last dye advection that ran on a guessed halo; repeat it when the guess was short.  Cheap
// when nothing is pending.  Every entry point that reads or writes the fields calls it.  `collective` = the caller is an
// operator every rank of a communicator issues (a solve, a step, an advection): only those may compare option blocks.
int settle_color(sfl_context *ctx, bool collective = false);

}  // namespace host
}  // namespace sfl
