// sfl_api.cpp -- the C ABI of include/sfl.h: contexts (one GPU's row slab, fields resident in
// HBM), operator entry points, the executor that walks slab_plan.h programs, RCCL halo
// exchange, and the host-pointer drop-ins.  Host C++ only; the kernels live in
// stencil_kernels.hip / sor_fused.hip.
//
// There is deliberately no CPU compute path in this file: every operator ends in a kernel
// launch, and fails with SFL_ERR_HIP when no device is usable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
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

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SFL_ERR_NOMEM : SFL_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                        __LINE__);                                                         \
    } while (0)

#define NCCL_TRY(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess)                                                            \
            return fail(SFL_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

#define SFL_TRY(expr)          \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != SFL_OK) return rc_; \
    } while (0)

constexpr int kGhostRows = 64;  // ghost rows allocated per side on a slab (nranks > 1)
constexpr size_t kAlternateSweepCells = 48u << 20;  // local cells from which successive SOR launches alternate direction

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

struct Group;

}  // namespace

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
    // not depend on them (a slab with RCCL transport; a linked group keeps these in Group)
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
    int *d_done = nullptr;         // [3] sender tiles finished so far (kernels.h HaloWait::done): what the exchange stream waits
    int done_target = 0;           //     for before a halo message leaves; done_target = the count the launches queued so far reach
    int *d_chain = nullptr;        // one word per tile of a chained launch (kernels.h launch_sor_chain), allocated on first use
    int chain_words = 0;
    int chain_epoch = 0;           // the words only count up: the next chained launch starts from here

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
    int *h_report = nullptr;       // pinned host copy
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

    int opt_sor_kernel = 0, opt_sor_fuse = 0, opt_advect_halo = 0, opt_sor_rows = 0,
        opt_sor_lane_cells = 0, opt_sor_halo = 0, opt_fuse_projection = 1, opt_sor_overlap = 1,
        opt_advect_kernel = 0, opt_fuse_divergence = 1, opt_small_grid = 1, opt_emulate_wire_us = 0, opt_sor_arrival = 1, opt_step_seams = 1, opt_sor_chain = 0;

    ncclComm_t comm = nullptr;
    bool options_dirty = false;         // an option changed since the ranks last compared their option blocks
    bool emulated = false;              // sfl_comm_emulate: one rank's program with self-copies as transport
    std::shared_ptr<Group> group;       // collective membership (in-process virtual ranks)
    std::shared_ptr<Group> keepalive;   // keeps the group's shared stream alive

    int last_launches = 0, last_exchanges = 0, last_fuse = 0, last_chained = 0;
    int solve_tail = 0;   // ghost rows of p the next solve must leave exact (slab_step_auto: 1, for subtract_gradient)
    int p_ghost_valid = 0;  // ghost rows of p that are exact right now (set by the solve, cleared by whoever writes p)
    int v_ghost_valid = 0;  // ghost rows of the velocity that are exact right now (slab_step_auto advects own +- 1 rows)

    size_t local_cells() const { return (size_t)geom.lrows * dim_x; }
    size_t owned_offset_cells() const { return (size_t)ghost * dim_x; }
};

namespace {

// In-process virtual ranks: slabs of one domain living on ONE device, ordered by one stream.
struct Group {
    std::vector<sfl_context *> members;
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;  // in-process halo copies of a solve (see sfl_context::xstream)
    hipEvent_t ev_ready = nullptr, ev_arrived = nullptr;
    // chained launches (SFL_OPT_SOR_CHAIN) of virtual ranks run side by side: the first on `stream`, the others here.  Created
    // right behind `xstream`: the runtime deals streams to its hardware queues in turn, and these streams, `stream` and `xstream`
    // must not share one (a chain that waits for a message would sit in front of the copy that carries it)
    static constexpr int kSideChains = 2;
    hipStream_t chain_stream[kSideChains] = {nullptr, nullptr};
    hipEvent_t ev_chain[kSideChains] = {nullptr, nullptr};
    ~Group()
    {
        for (int k = 0; k < kSideChains; ++k) {
            if (ev_chain[k]) (void)hipEventDestroy(ev_chain[k]);
            if (chain_stream[k]) (void)hipStreamDestroy(chain_stream[k]);
        }
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_arrived) (void)hipEventDestroy(ev_arrived);
        if (xstream) (void)hipStreamDestroy(xstream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

int use_device(sfl_context *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return SFL_OK;
}

template <class T>
int ensure(sfl_context *c, T *&ptr, size_t elem_bytes, bool zero)
{
    if (ptr) return SFL_OK;
    SFL_TRY(use_device(c));
    void *m = nullptr;
    const size_t bytes = c->local_cells() * elem_bytes;
    HIP_TRY(hipMalloc(&m, bytes));
    if (zero) HIP_TRY(hipMemsetAsync(m, 0, bytes, c->stream));
    ptr = static_cast<T *>(m);
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

// ---- halo exchange ---------------------------------------------------------------------
// Every rank sends its `rows` lowest owned rows down and its `rows` highest owned rows up, and
// receives the neighbours' into the ghost rows adjacent to its owned block.
// `on` = stream to issue the transfers on (nullptr: the contexts' compute stream).
// `skip` > 0: only the rows at depth [skip, skip + rows) from the cuts travel (the ghost rows nearer the cut
// are still valid: early exchanges of slab_plan.cpp).
// `in_time` (run_poisson_in_time): the exchange is one step of the device-counted protocol -- it starts when the sender tiles
// of the launch in front of it have counted themselves (`wait_done`; the right-hand side's exchange starts behind an event
// instead) and ends by raising every receiver's arrival count, with a kernel behind the message's own.  (Raising it in the
// copy kernel itself -- written-through stores, the block that finishes last stores the word -- saves that kernel, 4 - 6 us,
// beside the solve's first launch, which only reads d, and costs more than it saves beside a launch at full memory traffic:
// the copy's stores then take 10 us to drain.  Measured, not kept: profiles/r04_exchanges_counted_on_the_device.txt.)
int exchange(const std::vector<sfl_context *> &peers, int field, int rows, hipStream_t on = nullptr, int skip = 0,
             bool in_time = false, bool wait_done = false)
{
    if (rows <= 0) return SFL_OK;
    sfl_context *any = peers[0];
    if (any->nranks == 1) return SFL_OK;
    if (skip < 0) return fail(SFL_ERR_INVALID, "negative halo offset");
    if (skip + rows > any->ghost)
        return fail(SFL_ERR_INVALID, "halo of %d rows exceeds the %d ghost rows of a slab", skip + rows,
                    any->ghost);
    if (skip + rows > min_owned_rows(any))
        return fail(SFL_ERR_INVALID, "halo of %d rows exceeds the thinnest slab (%d rows)", skip + rows,
                    min_owned_rows(any));
    const size_t eb = field_elem_bytes(field);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, field));
        ++c->last_exchanges;
    }
    const size_t row_bytes = (size_t)any->dim_x * eb;
    const size_t bytes = row_bytes * rows;
    auto row_ptr = [&](sfl_context *c, int g) {
        return static_cast<char *>(field_ptr(c, field)) + (size_t)(g - c->geom.grow0) * row_bytes;
    };

    if (any->group) {  // in-process transport: all virtual ranks share one stream
        if (in_time && wait_done)   // every slab's sender tiles first (a slab reads from its two neighbours)
            for (sfl_context *c : peers) {
                SFL_TRY(use_device(c));
                HIP_TRY(sfl::launch_wait_count(on ? on : c->stream, c->d_done, c->done_target, c->d_arrival + 1));
            }
        for (sfl_context *c : peers) {   // both bands of a slab in one launch
            SFL_TRY(use_device(c));
            sfl_context *lo = c->rank > 0 ? peers[c->rank - 1] : nullptr;
            sfl_context *hi = c->rank < c->nranks - 1 ? peers[c->rank + 1] : nullptr;
            void *dst_a = lo ? row_ptr(c, c->g0 - skip - rows) : nullptr, *dst_b = hi ? row_ptr(c, c->g1 + skip) : nullptr;
            const void *src_a = lo ? row_ptr(lo, lo->g1 - skip - rows) : nullptr, *src_b = hi ? row_ptr(hi, hi->g0 + skip) : nullptr;
            HIP_TRY(sfl::launch_copy_bands(on ? on : c->stream, dst_a, src_a, dst_b, src_b, bytes));
        }
        // every copy before any signal: a slab's arrival count then also says that its neighbours have READ what it sent
        // (the chained launch's guard, kernels.h ChainStep::guard_flag, relies on that)
        if (in_time)
            for (sfl_context *c : peers) {
                SFL_TRY(use_device(c));
                ++c->arrival_epoch;
                HIP_TRY(sfl::launch_signal_arrival(on ? on : c->stream, c->d_arrival, c->arrival_epoch));
            }
        return SFL_OK;
    }

    sfl_context *c = any;
    if (c->emulated) {
        // ONE rank of the group runs alone (bench.py --emulate-rank): every message it would send is copied,
        // same size, same stream, into the ghost rows it would receive into.  The bytes are this rank's own,
        // so results next to the cuts mean nothing; launches, copies and their ordering are the rank's program.
        SFL_TRY(use_device(c));
        hipStream_t st = on ? on : c->stream;
        const bool lo = c->rank > 0, hi = c->rank < c->nranks - 1;
        void *dst_a = lo ? row_ptr(c, c->g0 - skip - rows) : nullptr, *dst_b = hi ? row_ptr(c, c->g1 + skip) : nullptr;
        const void *src_a = lo ? row_ptr(c, c->g0 + skip) : nullptr, *src_b = hi ? row_ptr(c, c->g1 - skip - rows) : nullptr;
        if (in_time && wait_done) HIP_TRY(sfl::launch_wait_count(st, c->d_done, c->done_target, c->d_arrival + 1));
        HIP_TRY(sfl::launch_spin_us(st, c->opt_emulate_wire_us));   // the wire a self-copy does not have (0: none)
        HIP_TRY(sfl::launch_copy_bands(st, dst_a, src_a, dst_b, src_b, bytes));
        if (in_time) {
            ++c->arrival_epoch;
            HIP_TRY(sfl::launch_signal_arrival(st, c->d_arrival, c->arrival_epoch));
        }
        return SFL_OK;
    }
    if (!c->comm)
        return fail(SFL_ERR_STATE, "slab %d/%d has no communicator: call sfl_comm_attach() or "
                    "sfl_group_link() first", c->rank, c->nranks);
    SFL_TRY(use_device(c));
    hipStream_t st = on ? on : c->stream;
    if (in_time && wait_done) HIP_TRY(sfl::launch_wait_count(st, c->d_done, c->done_target, c->d_arrival + 1));
    NCCL_TRY(ncclGroupStart());
    if (c->rank > 0) {
        NCCL_TRY(ncclSend(row_ptr(c, c->g0 + skip), bytes, ncclChar, c->rank - 1, c->comm, st));
        NCCL_TRY(ncclRecv(row_ptr(c, c->g0 - skip - rows), bytes, ncclChar, c->rank - 1, c->comm, st));
    }
    if (c->rank < c->nranks - 1) {
        NCCL_TRY(ncclSend(row_ptr(c, c->g1 - skip - rows), bytes, ncclChar, c->rank + 1, c->comm, st));
        NCCL_TRY(ncclRecv(row_ptr(c, c->g1 + skip), bytes, ncclChar, c->rank + 1, c->comm, st));
    }
    NCCL_TRY(ncclGroupEnd());
    if (in_time) {
        ++c->arrival_epoch;
        HIP_TRY(sfl::launch_signal_arrival(st, c->d_arrival, c->arrival_epoch));
    }
    return SFL_OK;
}

int clip_lo(const sfl_context *c, int g) { return g < 0 ? 0 : g; }
int clip_hi(const sfl_context *c, int g) { return g > c->gdim_y ? c->gdim_y : g; }

sfl::SorParams sor_params(float dx, float omega)
{
    sfl::SorParams prm;
    prm.dx = dx;
    prm.omega = omega;
    prm.one_minus_omega = 1.0f - omega;  // (1 - omega) in float, poisson.cpp:98,111
    return prm;
}

// Fuse depth: explicit option, or auto from the slab size.  Measured on MI355X, ms per 80-iteration
// solve at fuse 8 / 10 / 12 / 14 / 16 (round 2, gpurun_out/r02_run13-14, auto rows per tile):
// 8192 x 8192: 3.6 / 3.0 / 2.51 / 2.39 / 1.95; 8192 x 4096: - / - / 1.26 / 1.14 / 1.08; 8192 x 2048: - / 0.70 /
// 0.64 / 0.60 / 0.60; 8192 x 1024: 0.458 / 0.398 / 0.425 / 0.57 / 0.64; 8192 x 512 (round 1): 0.39 / - / 0.42 /
// - / 0.45; 40 iterations: 4096^2 - / 0.343 / 0.327 / 0.308 / 0.294; 3072^2 0.245 / 0.208 / 0.223 / 0.294 / 0.297; 2048^2
// 0.155 / 0.143 / 0.175 / - / 0.217; 1024^2 0.086 / 0.090 / 0.097; 8192 x 768 (80): 0.366 / 0.333 / 0.393; 8192 x 512: 0.303 /
// 0.305 / 0.350.  Big slabs are bound by the pass over memory
// each launch makes and want the deepest fusion; small ones by the 2 * NS warm-up rows each tile
// re-streams.  Every rank of a group sees the same thinnest slab, so all ranks resolve the same value.
int effective_fuse(const sfl_context *c)
{
    int f = c->opt_sor_fuse;
    if (f == 0) {
        const int64_t cells = (int64_t)min_owned_rows(c) * c->dim_x;
        f = cells >= 12000000 ? 16 : cells >= 3000000 ? 10 : 8;
    }
    if (f < 2) f = 2;
    if (f > SFL_MAX_FUSE) f = SFL_MAX_FUSE;
    return f & ~1;
}

int effective_kernel(const sfl_context *c) { return c->opt_sor_kernel == 1 ? 1 : 2; }

// One workgroup, fields in LDS (small_grid.hip): whole-domain contexts of at most kSmallGridMaxCells cells whose
// kernel options are all automatic (an explicit kernel / fuse / tile choice is honoured as given).
bool small_grid(const sfl_context *c)
{
    return c->opt_small_grid && c->nranks == 1 && !c->group &&
           sfl::small_grid_fits(c->dim_x, c->gdim_y) && c->opt_sor_kernel == 0 &&
           c->opt_sor_fuse == 0 && c->opt_sor_rows == 0 && c->opt_sor_lane_cells == 0 && c->opt_advect_kernel == 0;
}


int effective_halo(const sfl_context *c, int fuse)
{
    // auto: 64 rows on slabs of >= 1024 rows (2-3 exchanges per 80-iteration solve, ~5 % extra
    // rows recomputed), 32 on thinner ones
    int h = c->opt_sor_halo ? c->opt_sor_halo : (min_owned_rows(c) >= 1024 ? 64 : 32);
    if (h > min_owned_rows(c)) h = min_owned_rows(c);  // a neighbour can only send rows it owns
    if (h > kGhostRows) h = kGhostRows;
    return h < fuse ? fuse : h;
}

// Exchanges IN TIME with everything counted on the device (run_poisson_in_time; SFL_OPT_SOR_ARRIVAL) instead of early exchanges
// behind cross-stream events: slabs with a transport, the fused kernel, exchanges overlapped.
bool in_time_exchanges(const sfl_context *c)
{
    return c->opt_sor_arrival && c->opt_sor_overlap && c->nranks > 1 && c->opt_sor_kernel != 1 &&
           (c->comm || c->group || c->emulated);
}

// ---- poisson_solve executor --------------------------------------------------------------
// One SOR launch of a plan step over output rows [g_begin, g_end) (a step may be issued in pieces:
// all pieces read c->p and write c->p_alt; the caller swaps once per step).
// `in` / `out` = the step's input and output arrays (null: c->p / c->p_alt); `on` = stream (null: the compute stream)
int launch_sor_rows(sfl_context *c, const sfl_plan_step &st, const sfl::SorParams &prm, int g_begin, int g_end,
                    int g2_begin = 0, int g2_end = 0, hipStream_t on = nullptr, const float *in = nullptr,
                    float *out = nullptr, const sfl::HaloWait *wait = nullptr, int *senders = nullptr)
{
    if (senders) *senders = 0;
    if (g_end <= g_begin && g2_end <= g2_begin) return SFL_OK;
    if (!in) in = c->p;
    if (!out) out = c->p_alt;
    SFL_TRY(use_device(c));
    // Slabs that outgrow the Infinity Cache (256 MB; p + d of 48 M cells = 384 MB) reverse the stream direction of
    // every tile from one launch to the next: a launch then begins on the rows its predecessor read and wrote
    // last, the only ones still cached (8192^2: -3 % per launch; no gain or a small loss on slabs that fit:
    // profiles/r03_alternate_sweep.txt).  last_launches counts the plan steps issued so far in this solve.
    const int sweep = c->local_cells() >= kAlternateSweepCells ? c->last_launches : 0;
    HIP_TRY(sfl::launch_sor_fused(on ? on : c->stream, out, st.from_zero ? nullptr : in, c->div, c->geom,
                                  sfl::SorRows{g_begin, g_end, g2_begin, g2_end}, st.nsweeps, st.first_colour,
                                  prm, c->opt_sor_rows, sweep, wait, senders));
    return SFL_OK;
}

// Device-side halo arrival (run_poisson_in_time, kernels.h HaloWait).  The exchange stream counts a context's arrived
// messages in a device word; the next launch on the compute stream is queued WITHOUT a cross-stream event and lets only
// its cut-adjacent tiles wait for the count.
sfl::HaloWait arrival_wait(const sfl_context *c)
{
    sfl::HaloWait w;
    w.flag = c->d_arrival;
    w.timed_out = c->d_arrival + 1;
    w.epoch = c->arrival_epoch;
    w.own_lo = c->rank > 0 ? c->g0 : -(1 << 30);                 // no cut on that side: nothing to wait for
    w.own_hi = c->rank < c->nranks - 1 ? c->g1 : (1 << 30);
    w.done = nullptr;
    w.send_lo_end = w.send_hi_begin = 0;
    return w;
}

int exec_sor_step(sfl_context *c, const sfl_plan_step &st, const sfl::SorParams &prm)
{
    SFL_TRY(use_device(c));
    if (st.kind == SFL_STEP_ZERO) {
        HIP_TRY(sfl::launch_zero_rows(c->stream, c->p, c->geom, clip_lo(c, c->geom.grow0),
                                      clip_hi(c, c->geom.grow0 + c->geom.lrows)));
        ++c->last_launches;
        return SFL_OK;
    }
    if (st.nsweeps == 1) {
        HIP_TRY(sfl::launch_sor_half_sweep(c->stream, c->p, c->div, c->geom, st.g_begin, st.g_end,
                                           st.first_colour, prm));
        ++c->last_launches;
        return SFL_OK;
    }
    SFL_TRY(launch_sor_rows(c, st, prm, st.g_begin, st.g_end));
    std::swap(c->p, c->p_alt);
    ++c->last_launches;  // plan steps, not pieces: an overlapped step counts once as well
    return SFL_OK;
}

// ---- chained supersteps (kernels.h launch_sor_chain; SFL_OPT_SOR_CHAIN) ---------------------------------------------------
// May plan steps [i, i + n) of `prog` go into one chained launch?  SOR steps of one supported fuse depth, none from zero.
int chainable_steps(const sfl_context *c, const std::vector<sfl_plan_step> &prog, size_t i, bool across_exchanges)
{
    if (c->opt_sor_chain <= 0 || effective_kernel(c) != 2) return 0;   // (automatic: only where exchanges run in time, below)
    int n = 0;
    size_t k = i;
    for (; k < prog.size() && n < sfl::kMaxChain; ++k) {
        const sfl_plan_step &st = prog[k];
        if (st.kind == SFL_STEP_EXCHANGE && across_exchanges && st.field == SFL_FIELD_PRESSURE && n > 0) continue;
        if (st.kind != SFL_STEP_SOR || st.from_zero || st.first_colour != 0 || st.nsweeps != prog[i].nsweeps ||
            st.g_end <= st.g_begin)
            break;
        ++n;
    }
    if (n < 2 || !sfl::sor_chain_supported(c->p, c->p_alt, c->div, c->geom, prog[i].nsweeps)) return 0;
    return n;
}

int ensure_chain_words(sfl_context *c)
{
    if (c->d_chain) return SFL_OK;
    // more words than any tiling of the slab has tiles: strips of >= 96 kept columns x chunks of >= kMinEdgeRows rows
    const int words = 32 * (c->dim_x / 96 + 3) * (c->geom.lrows / 8 + 4);   // a 128-byte line per tile
    void *m = nullptr;
    SFL_TRY(use_device(c));
    HIP_TRY(hipMalloc(&m, (size_t)words * sizeof(int)));
    HIP_TRY(hipMemsetAsync(m, 0, (size_t)words * sizeof(int), c->stream));
    c->d_chain = static_cast<int *>(m);
    c->chain_words = words;
    return SFL_OK;
}

// Plan steps [i, i + n) of a context whose launches need no halo protocol (whole domains, the in-line order), as one launch.
int exec_sor_chain(sfl_context *c, const std::vector<sfl_plan_step> &prog, size_t i, int n, const sfl::SorParams &prm)
{
    SFL_TRY(ensure_chain_words(c));
    sfl::ChainStep steps[sfl::kMaxChain];
    for (int k = 0; k < n; ++k) {
        const sfl_plan_step &st = prog[i + k];
        steps[k].g_begin = st.g_begin;
        steps[k].g_end = st.g_end;
        steps[k].sweep = c->local_cells() >= kAlternateSweepCells ? c->last_launches + k : 0;
        steps[k].hw = sfl::HaloWait{nullptr, nullptr, 0, 0, 0, nullptr, 0, 0};
        steps[k].guard_flag = nullptr;
        steps[k].guard_epoch = steps[k].guard_lo_end = steps[k].guard_hi_begin = 0;
    }
    HIP_TRY(sfl::launch_sor_chain(c->stream, c->p, c->p_alt, c->div, c->geom, steps, n, prog[i].nsweeps, prm, c->opt_sor_rows,
                                  c->d_chain, c->chain_words, c->chain_epoch, c->d_arrival + 1,
                                  c->opt_sor_chain >= 8 ? c->opt_sor_chain & ~3 : 0, nullptr));
    c->chain_epoch += n + 1;
    if (n & 1) std::swap(c->p, c->p_alt);
    c->last_launches += n;
    c->last_chained += n;
    return SFL_OK;
}

// The exchange stream and its two events (created on first use): the group's when the contexts are
// linked, the context's own otherwise.
struct Overlap {
    hipStream_t compute = nullptr, xstream = nullptr;
    hipEvent_t ready = nullptr, arrived = nullptr;
};

int overlap_of(sfl_context *c, Overlap *o)
{
    SFL_TRY(use_device(c));
    hipStream_t *xs = c->group ? &c->group->xstream : &c->xstream;
    hipEvent_t *e0 = c->group ? &c->group->ev_ready : &c->ev_ready;
    hipEvent_t *e1 = c->group ? &c->group->ev_arrived : &c->ev_arrived;
    if (!*xs) {
        HIP_TRY(hipStreamCreateWithFlags(xs, hipStreamNonBlocking));
        if (c->group)
            for (int k = 0; k < Group::kSideChains; ++k) {
                HIP_TRY(hipStreamCreateWithFlags(&c->group->chain_stream[k], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&c->group->ev_chain[k], hipEventDisableTiming));
            }
    }
    if (!*e0) HIP_TRY(hipEventCreateWithFlags(e0, hipEventDisableTiming));
    if (!*e1) HIP_TRY(hipEventCreateWithFlags(e1, hipEventDisableTiming));
    o->compute = c->stream;  // a linked group shares one compute stream
    o->xstream = *xs;
    o->ready = *e0;
    o->arrived = *e1;
    return SFL_OK;
}

// Halo exchange off the compute stream: starts once everything issued so far on the compute
// stream has completed, runs on the exchange stream; `arrived` marks its completion.
// `mark` = false: the caller queues more work behind the exchange on the exchange stream and records
// `arrived` itself (mark_arrived).
int start_exchange(const std::vector<sfl_context *> &peers, const Overlap &o, int field, int rows, int skip = 0,
                   bool mark = true, bool in_time = false)
{
    SFL_TRY(use_device(peers[0]));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    SFL_TRY(exchange(peers, field, rows, o.xstream, skip, in_time));
    SFL_TRY(use_device(peers[0]));
    if (mark) HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    return SFL_OK;
}

int mark_arrived(const std::vector<sfl_context *> &peers, const Overlap &o)
{
    SFL_TRY(use_device(peers[0]));
    HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    return SFL_OK;
}

int await_exchange(const std::vector<sfl_context *> &peers, const Overlap &o)
{
    SFL_TRY(use_device(peers[0]));
    HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    return SFL_OK;
}

// An exchange IN LINE with the compute stream's work.  With RCCL it still travels on the exchange
// stream -- every operation of a communicator is issued to ONE stream, whatever the operator --
// bracketed by the two events; in-process copies of a linked group go on the compute stream itself.
// `after` != nullptr: the exchanged rows were final when that event was recorded on the compute stream -- the exchange starts
// behind IT, not behind what has been queued on the compute stream since (slab_step_auto's early interior advection).
int exchange_inline(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field, int rows, int skip = 0,
                    hipEvent_t after = nullptr)
{
    if (rows <= 0 || ctx->nranks == 1) return SFL_OK;
    if (!ctx->comm && !ctx->emulated) return exchange(peers, field, rows, nullptr, skip);
    Overlap o;
    SFL_TRY(overlap_of(ctx, &o));
    if (after) {
        SFL_TRY(use_device(peers[0]));
        HIP_TRY(hipStreamWaitEvent(o.xstream, after, 0));
        SFL_TRY(exchange(peers, field, rows, o.xstream, skip));
        SFL_TRY(use_device(peers[0]));
        HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    } else {
        SFL_TRY(start_exchange(peers, o, field, rows, skip));
    }
    return await_exchange(peers, o);
}

// The rows the last p halo message of a solve was read from, and how many supersteps have been issued since: the superstep
// two behind a message overwrites its source (kernels.h ChainStep::guard_flag).
struct SentBand {
    bool valid = false;
    int epoch = 0, lo_end = 0, hi_begin = 0, age = 0;
};

// In-time exchanges with the launches CHAINED (SFL_OPT_SOR_CHAIN): the SOR steps from step i on -- up to kMaxChain, p exchanges
// between and behind them included -- as one chained launch per context, with the exchange stream's work (wait for the sender
// counts, copy / send, raise the arrival counts) queued behind them exactly as for single launches.  The chains of the virtual
// ranks of a group wait for each other's messages, so they must RUN side by side: the first on the compute stream, the others
// on a stream of their own each (joined back into the compute stream), all of them within a budget of waves that is resident
// at once.  *next = first plan step not consumed (== i: nothing was chained).
int chain_in_time(const std::vector<sfl_context *> &peers, const std::vector<std::vector<sfl_plan_step>> &progs, size_t i,
                  const sfl::SorParams &prm, const Overlap &o, bool *flagged, std::vector<SentBand> *bands, size_t *next)
{
    *next = i;
    const std::vector<sfl_plan_step> &prog = progs[0];   // every rank's program has the same shape
    std::vector<size_t> sor;
    std::vector<long> xch;   // the p exchange behind sor[k] (index into prog), or -1
    size_t k = i;
    const int ns = prog[i].nsweeps;
    while (k < prog.size() && (int)sor.size() < sfl::kMaxChain) {
        bool ok = true;
        for (size_t r = 0; r < peers.size(); ++r) {
            const sfl_plan_step &st = progs[r][k];
            ok = ok && st.kind == SFL_STEP_SOR && !st.from_zero && st.first_colour == 0 && st.nsweeps == ns && st.g_end > st.g_begin;
        }
        if (!ok) break;
        sor.push_back(k++);
        if (k < prog.size() && prog[k].kind == SFL_STEP_EXCHANGE && prog[k].field == SFL_FIELD_PRESSURE)
            xch.push_back((long)k++);
        else
            xch.push_back(-1);
    }
    const int n = (int)sor.size();
    if (n < 2) return SFL_OK;
    for (sfl_context *c : peers)
        if (!sfl::sor_chain_supported(c->p, c->p_alt, c->div, c->geom, ns)) return SFL_OK;
    // two waves per SIMD for all chains together: room for the exchange stream's kernels beside them, and the occupancy the
    // chain runs best at -- a thin slab's tiling has two tiles per SIMD; the slabs that touch the domain's boundary have three,
    // and their chains do better with two waves per SIMD that take a second tile (0.398 ms) than with three (0.444)
    int dev = 0, cus = 256;
    SFL_TRY(use_device(peers[0]));
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int budget = cus * 8 / (int)peers.size();
    if (peers[0]->opt_sor_chain >= 8 && peers[0]->opt_sor_chain < budget) budget = peers[0]->opt_sor_chain;
    budget -= budget % 4;
    // Side by side means a hardware queue each, for the compute stream, the exchange stream and every side stream; the runtime
    // folds its streams onto GPU_MAX_HW_QUEUES (default 4) of them in turn.  Two virtual ranks fit the default; three when the
    // process was started with more queues (tests/conftest.py does).  Folded streams are not a hang but a reported time-out.
    static const int side_chains = [] {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        return q && atoi(q) >= 6 ? Group::kSideChains : 1;
    }();
    if (budget < 8 || (int)peers.size() > 1 + side_chains || (peers.size() > 1 && !peers[0]->group)) return SFL_OK;
    for (sfl_context *c : peers) SFL_TRY(ensure_chain_words(c));   // (zeroed on the compute stream: before the event below)
    if (peers.size() > 1) HIP_TRY(hipEventRecord(o.ready, o.compute));   // the side streams start behind what is queued so far

    std::vector<std::vector<int>> senders(peers.size(), std::vector<int>(sfl::kMaxChain, 0));
    bool fl_out = *flagged;
    for (size_t r = 0; r < peers.size(); ++r) {
        sfl_context *c = peers[r];
        sfl::ChainStep steps[sfl::kMaxChain];
        int epoch = c->arrival_epoch;   // the value the arrival count reaches with the exchanges issued so far
        bool fl = *flagged;
        SentBand b = (*bands)[r];
        for (int q = 0; q < n; ++q) {
            const sfl_plan_step &st = progs[r][sor[q]];
            sfl::ChainStep &cs = steps[q];
            cs.g_begin = st.g_begin;
            cs.g_end = st.g_end;
            cs.sweep = c->local_cells() >= kAlternateSweepCells ? c->last_launches + q : 0;
            cs.hw = arrival_wait(c);
            cs.hw.epoch = epoch;
            if (!fl) cs.hw.flag = nullptr;
            ++b.age;
            cs.guard_flag = b.valid && b.age == 2 ? c->d_arrival : nullptr;
            cs.guard_epoch = b.epoch;
            cs.guard_lo_end = b.lo_end;
            cs.guard_hi_begin = b.hi_begin;
            fl = false;
            if (xch[q] >= 0) {
                const sfl_plan_step &x = progs[r][xch[q]];
                cs.hw.done = c->d_done;
                cs.hw.send_lo_end = c->rank > 0 ? c->g0 + x.g_begin + x.rows : -(1 << 30);
                cs.hw.send_hi_begin = c->rank < c->nranks - 1 ? c->g1 - x.g_begin - x.rows : (1 << 30);
                ++epoch;
                fl = true;
                b.valid = true;
                b.epoch = epoch;
                b.lo_end = cs.hw.send_lo_end;
                b.hi_begin = cs.hw.send_hi_begin;
                b.age = 0;
            }
        }
        hipStream_t on = c->stream;
        if (r > 0) {
            on = c->group->chain_stream[r - 1];
            HIP_TRY(hipStreamWaitEvent(on, o.ready, 0));
        }
        bool launched = false;
        HIP_TRY(sfl::launch_sor_chain(on, c->p, c->p_alt, c->div, c->geom, steps, n, ns, prm, c->opt_sor_rows, c->d_chain,
                                      c->chain_words, c->chain_epoch, c->d_arrival + 1, budget, senders[r].data(),
                                      c->opt_sor_chain < 0 ? cus * 13 : 0, &launched));
        if (!launched) return SFL_OK;   // (automatic mode: thin slabs only; first context: nothing has been changed yet)
        if (r > 0) HIP_TRY(hipEventRecord(c->group->ev_chain[r - 1], on));
        c->chain_epoch += n + 1;
        c->last_launches += n;
        c->last_chained += n;
        (*bands)[r] = b;
        fl_out = fl;
    }
    for (size_t r = 1; r < peers.size(); ++r) HIP_TRY(hipStreamWaitEvent(o.compute, peers[r]->group->ev_chain[r - 1], 0));
    for (int q = 0; q < n; ++q) {
        for (sfl_context *c : peers) std::swap(c->p, c->p_alt);   // c->p = what superstep q writes: the message's source
        for (size_t r = 0; r < peers.size(); ++r) peers[r]->done_target += senders[r][q];
        if (xch[q] < 0) continue;
        const sfl_plan_step &x = prog[xch[q]];
        SFL_TRY(exchange(peers, SFL_FIELD_PRESSURE, x.rows, o.xstream, x.g_begin, true, true));
    }
    *flagged = fl_out;
    *next = k;
    return SFL_OK;
}

// Exchanges IN TIME (slab_plan.cpp kernel 3; SFL_OPT_SOR_ARRIVAL): the halo of a superstep is sent after the launch that
// produces it, as in the textbook -- but nothing waits for a whole launch any more.  The launch in front of an exchange
// marks the tiles whose rows the message carries as SENDERS (top priority; each counts itself once its rows are written
// back); the exchange stream waits for that count, not for the launch, so the message leaves while the rest of the launch
// is still running; the launch behind the exchange is queued at once and only its cut-adjacent tiles wait for the arrival
// count.  No event on the compute stream, no ghost launch, no launch split.  (The right-hand side at the head of a solve
// was produced by other kernels: its exchange still starts behind an event.)
int run_poisson_in_time(sfl_context *ctx, const std::vector<sfl_context *> &peers,
                        const std::vector<std::vector<sfl_plan_step>> &progs, const sfl::SorParams &prm, const Overlap &o)
{
    bool flagged = false;   // the next launch's cut-adjacent tiles wait for the arrival count
    std::vector<SentBand> bands(peers.size());   // the last p message's source rows, per context (chained launches)
    bool chain_refused = false;
    const size_t n = progs[0].size();
    for (size_t i = 0; i < n; ++i) {
        const sfl_plan_step &st0 = progs[0][i];
        if (st0.kind == SFL_STEP_EXCHANGE) {   // the right-hand side (a p exchange is taken together with the launch before it)
            SFL_TRY(start_exchange(peers, o, st0.field, st0.rows, st0.g_begin, false, true));
            flagged = true;
            continue;
        }
        // automatic (-1): a context with a transport of its own (RCCL, the emulated rank) on slabs thin enough that every tile is
        // resident at two waves per SIMD; virtual ranks (a test transport) only when asked to
        if ((ctx->opt_sor_chain > 0 || (ctx->opt_sor_chain < 0 && !ctx->group && !chain_refused)) && !st0.from_zero) {
            size_t next = i;
            SFL_TRY(chain_in_time(peers, progs, i, prm, o, &flagged, &bands, &next));
            if (next > i) {
                i = next - 1;
                continue;
            }
            chain_refused = true;   // decided once per solve: the first attempt holds the widest row ranges
        }
        for (SentBand &b : bands) ++b.age;
        const bool sends = i + 1 < n && progs[0][i + 1].kind == SFL_STEP_EXCHANGE && progs[0][i + 1].field == SFL_FIELD_PRESSURE;
        for (size_t k = 0; k < peers.size(); ++k) {
            sfl_context *c = peers[k];
            const sfl_plan_step &st = progs[k][i];
            sfl::HaloWait w = arrival_wait(c);
            if (!flagged) w.flag = nullptr;
            if (sends) {
                const sfl_plan_step &x = progs[k][i + 1];
                w.done = c->d_done;
                w.send_lo_end = c->rank > 0 ? c->g0 + x.g_begin + x.rows : -(1 << 30);
                w.send_hi_begin = c->rank < c->nranks - 1 ? c->g1 - x.g_begin - x.rows : (1 << 30);
            }
            int senders = 0;
            SFL_TRY(launch_sor_rows(c, st, prm, st.g_begin, st.g_end, 0, 0, nullptr, nullptr, nullptr,
                                    (flagged || sends) ? &w : nullptr, &senders));
            c->done_target += senders;
            std::swap(c->p, c->p_alt);
            ++c->last_launches;
        }
        flagged = false;
        if (sends) {
            const sfl_plan_step &x = progs[0][i + 1];
            SFL_TRY(exchange(peers, SFL_FIELD_PRESSURE, x.rows, o.xstream, x.g_begin, true, true));
            flagged = true;
            for (size_t k = 0; k < peers.size(); ++k) {
                const sfl_context *c = peers[k];
                bands[k].valid = true;
                bands[k].epoch = c->arrival_epoch;
                bands[k].lo_end = c->rank > 0 ? c->g0 + x.g_begin + x.rows : -(1 << 30);
                bands[k].hi_begin = c->rank < c->nranks - 1 ? c->g1 - x.g_begin - x.rows : (1 << 30);
                bands[k].age = 0;
            }
            ++i;   // the exchange step has been issued
        }
    }
    return SFL_OK;
}

// The fused kernel's program on slabs, with the halo exchanges OVERLAPPED (cut-adjacent
// rows first, exchange on a second stream).  The plan is unchanged (slab_plan.cpp); what changes
// is the order in which the rows of two launches are issued around an exchange of H rows:
//
//   launch before the exchange    rows [g0, g0+H) and [g1-H, g1) -- what the neighbours will
//                                 receive -- go first; the exchange starts behind them on the
//                                 exchange stream; the launch's other rows follow on the compute
//                                 stream while the halos travel;
//   launch after the exchange     its rows [g0+ns, g1-ns) need no ghost row (a launch of ns passes
//                                 reads ns rows beyond its output) and go first; the compute
//                                 stream then waits for the halos and finishes with the two bands
//                                 next to the cuts.
//
// A slab at the bottom / top of the domain has no neighbour on that side: no band there.  Every
// piece of a step reads p and writes p_alt (one swap per step), the exchange reads the owned rows
// of the step's output and writes ghost rows nobody else touches meanwhile, so the pieces are
// independent and the result is the bits of the plain order.  With SFL_OPT_SOR_OVERLAP = 0, and for
// the baseline kernel, every step is issued whole on the compute stream.
int run_poisson_overlapped_steps(sfl_context *ctx, const std::vector<sfl_context *> &peers,
                                 const std::vector<std::vector<sfl_plan_step>> &progs, const sfl::SorParams &prm,
                                 const Overlap &o);

int run_poisson_overlapped(sfl_context *ctx, const std::vector<sfl_context *> &peers,
                           const std::vector<std::vector<sfl_plan_step>> &progs, const sfl::SorParams &prm)
{
    Overlap o;
    SFL_TRY(overlap_of(ctx, &o));
    const int rc = in_time_exchanges(ctx) ? run_poisson_in_time(ctx, peers, progs, prm, o)
                                          : run_poisson_overlapped_steps(ctx, peers, progs, prm, o);
    if (rc != SFL_OK) {
        // a launch or an RCCL call failed half way: nothing of this solve may still be in flight on the exchange
        // stream when the caller looks at (or destroys) the context; the error message of the failure is kept
        const std::string why = g_error;
        (void)hipStreamSynchronize(o.xstream);
        (void)hipStreamSynchronize(o.compute);
        g_error = why;
    }
    return rc;
}

int run_poisson_overlapped_steps(sfl_context *ctx, const std::vector<sfl_context *> &peers,
                                 const std::vector<std::vector<sfl_plan_step>> &progs, const sfl::SorParams &prm,
                                 const Overlap &o)
{
    bool pending = false;  // an exchange is in flight that the next launch's cut-adjacent rows need
    bool behind_early = false;  // ... an early one, with a launch queued behind it: the next launch needs all of it
    const size_t n = progs[0].size();
    for (size_t i = 0; i < n; ++i) {
        const sfl_plan_step &st0 = progs[0][i];
        if (st0.kind == SFL_STEP_EXCHANGE && st0.field == SFL_FIELD_PRESSURE && st0.g_begin > 0 && i + 1 < n &&
            progs[0][i + 1].kind == SFL_STEP_SOR && progs[0][i + 1].nsweeps <= st0.g_begin) {
            // EARLY exchange (slab_plan.cpp): the ghost rows are still valid as deep as the next launch needs for
            // the OWNED rows.  Compute stream: that launch, owned rows only, whole -- no piece of it waits for the
            // wire.  Exchange stream: the message (rows beyond that depth), then the same launch's passes on the
            // ghost rows it feeds (output rows [g_begin, g0) and [g1, g_end)).  Both read p and write p_alt, on
            // disjoint rows; the message lands in rows of p that the owned-row launch reads only into its
            // throw-away rim.  The launch AFTER this one waits for `arrived` -- WHOLE: it overwrites the owned rows this
            // rank's own outgoing message is still being read from (round 4 let only its cut-adjacent tiles wait, on a
            // device-side count: one solve in 26 000 came out wrong, profiles/r04_exchanges_counted_on_the_device.txt).
            if (pending) SFL_TRY(await_exchange(peers, o));
            pending = behind_early = false;
            SFL_TRY(start_exchange(peers, o, SFL_FIELD_PRESSURE, st0.rows, st0.g_begin, false));
            for (size_t k = 0; k < peers.size(); ++k) {
                sfl_context *c = peers[k];
                const sfl_plan_step &st = progs[k][i + 1];
                const int lo = c->rank > 0 ? c->g0 : st.g_begin;
                const int hi = c->rank < c->nranks - 1 ? c->g1 : st.g_end;
                SFL_TRY(launch_sor_rows(c, st, prm, lo, hi));                                      // compute stream
                SFL_TRY(launch_sor_rows(c, st, prm, st.g_begin, lo, hi, st.g_end, o.xstream));     // behind the message
            }
            SFL_TRY(mark_arrived(peers, o));
            for (sfl_context *c : peers) {
                std::swap(c->p, c->p_alt);
                ++c->last_launches;
            }
            pending = behind_early = true;
            ++i;  // the launch has been issued
            continue;
        }
        if (st0.kind == SFL_STEP_EXCHANGE) {  // (classic p exchanges are started by the launch before them)
            if (pending) SFL_TRY(await_exchange(peers, o));
            SFL_TRY(start_exchange(peers, o, st0.field, st0.rows, st0.g_begin));
            pending = true;
            continue;
        }
        if (pending && behind_early) {  // the ghost rows relaxed behind an early exchange: needed whole, now
            SFL_TRY(await_exchange(peers, o));
            pending = behind_early = false;
        }
        const bool sends_next = i + 1 < n && progs[0][i + 1].kind == SFL_STEP_EXCHANGE &&
                                progs[0][i + 1].field == SFL_FIELD_PRESSURE && progs[0][i + 1].g_begin == 0;
        const int send_rows = sends_next ? progs[0][i + 1].rows : 0;
        const int ns = st0.nsweeps;
        if (pending) {
            // rows that need no ghost row first, the cut-adjacent bands behind the exchange
            for (size_t k = 0; k < peers.size(); ++k) {
                sfl_context *c = peers[k];
                const sfl_plan_step &st = progs[k][i];
                const int lo = c->rank > 0 ? std::min(c->g0 + ns, st.g_end) : st.g_begin;
                const int hi = c->rank < c->nranks - 1 ? std::max(c->g1 - ns, lo) : st.g_end;
                SFL_TRY(launch_sor_rows(c, st, prm, lo, hi));
            }
            SFL_TRY(await_exchange(peers, o));
            pending = false;
            for (size_t k = 0; k < peers.size(); ++k) {
                sfl_context *c = peers[k];
                const sfl_plan_step &st = progs[k][i];
                const int lo = c->rank > 0 ? std::min(c->g0 + ns, st.g_end) : st.g_begin;
                const int hi = c->rank < c->nranks - 1 ? std::max(c->g1 - ns, lo) : st.g_end;
                SFL_TRY(launch_sor_rows(c, st, prm, st.g_begin, lo, hi, st.g_end));  // both bands, one launch
                std::swap(c->p, c->p_alt);
                ++c->last_launches;
            }
            if (sends_next) {  // a one-launch superstep: nothing left to overlap the next exchange with
                SFL_TRY(start_exchange(peers, o, SFL_FIELD_PRESSURE, send_rows));
                pending = true;
                ++i;
            }
            continue;
        }
        if (sends_next) {
            // the rows the neighbours will receive first, then the exchange, then the rest
            for (size_t k = 0; k < peers.size(); ++k) {
                sfl_context *c = peers[k];
                const sfl_plan_step &st = progs[k][i];
                const int lo = c->rank > 0 ? std::min(c->g0 + send_rows, st.g_end) : st.g_begin;
                const int hi = c->rank < c->nranks - 1 ? std::max(c->g1 - send_rows, lo) : st.g_end;
                SFL_TRY(launch_sor_rows(c, st, prm, st.g_begin, lo, hi, st.g_end));  // both bands, one launch
                std::swap(c->p, c->p_alt);  // ONE swap per step: the exchange sends from / receives into its output
            }
            SFL_TRY(start_exchange(peers, o, SFL_FIELD_PRESSURE, send_rows));
            pending = true;
            for (size_t k = 0; k < peers.size(); ++k) {
                sfl_context *c = peers[k];
                const sfl_plan_step &st = progs[k][i];
                const int lo = c->rank > 0 ? std::min(c->g0 + send_rows, st.g_end) : st.g_begin;
                const int hi = c->rank < c->nranks - 1 ? std::max(c->g1 - send_rows, lo) : st.g_end;
                // the remaining rows: from the step's input (now p_alt) into its output (now p)
                SFL_TRY(launch_sor_rows(c, st, prm, lo, hi, 0, 0, nullptr, c->p_alt, c->p));
                ++c->last_launches;
            }
            ++i;  // the exchange step has been issued
            continue;
        }
        for (size_t k = 0; k < peers.size(); ++k) SFL_TRY(exec_sor_step(peers[k], progs[k][i], prm));
    }
    if (pending) SFL_TRY(await_exchange(peers, o));
    return SFL_OK;
}

int run_poisson(sfl_context *ctx, float dx, int iters, float omega)
{
    if (iters < 0) return fail(SFL_ERR_INVALID, "iters must be >= 0 (got %d)", iters);
    std::vector<sfl_context *> peers = peers_of(ctx);
    const int fuse = effective_fuse(ctx), kernel = effective_kernel(ctx);
    std::vector<std::vector<sfl_plan_step>> progs;
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_DIVERGENCE));
        SFL_TRY(ensure_field(c, SFL_FIELD_PRESSURE));
        progs.push_back(sfl::plan_poisson(c->gdim_y, c->nranks, c->rank, iters, fuse,
                                          kernel == 2 && in_time_exchanges(ctx) ? 3 : kernel, effective_halo(ctx, fuse),
                                          ctx->solve_tail));
        c->last_launches = c->last_exchanges = c->last_chained = 0;
        c->p_ghost_valid = 0;
        c->last_fuse = kernel == 1 ? 1 : fuse;
    }
    const sfl::SorParams prm = sor_params(dx, omega);
    if (small_grid(ctx)) {  // one workgroup, p and d in LDS, every iteration in one launch
        SFL_TRY(use_device(ctx));
        HIP_TRY(sfl::launch_small_solve(ctx->stream, ctx->p, ctx->div, ctx->dim_x, ctx->gdim_y, iters, prm));
        ctx->last_launches = 1;
        ctx->last_fuse = 2 * iters;
        return SFL_OK;
    }
    if (iters == 0) {  // the reference still zero-fills p (poisson.cpp:117-119)
        for (sfl_context *c : peers) {
            SFL_TRY(use_device(c));
            HIP_TRY(sfl::launch_zero_rows(c->stream, c->p, c->geom, c->g0, c->g1));
        }
        return SFL_OK;
    }
    if (kernel == 2 && ctx->nranks > 1 && ctx->opt_sor_overlap && (ctx->comm || ctx->group || ctx->emulated)) {
        SFL_TRY(run_poisson_overlapped(ctx, peers, progs, prm));
    } else {
        for (size_t i = 0; i < progs[0].size(); ++i) {
            const sfl_plan_step &st0 = progs[0][i];
            if (st0.kind == SFL_STEP_EXCHANGE) {
                SFL_TRY(exchange_inline(ctx, peers, st0.field, st0.rows, st0.g_begin));
            } else if (const int n = chainable_steps(ctx, progs[0], i, false)) {   // same program shape on every peer
                for (size_t k = 0; k < peers.size(); ++k) SFL_TRY(exec_sor_chain(peers[k], progs[k], i, n, prm));
                i += n - 1;
            } else {
                for (size_t k = 0; k < peers.size(); ++k) SFL_TRY(exec_sor_step(peers[k], progs[k][i], prm));
            }
        }
    }
    // ghost rows of p the last launch left exact (the plan's tail): what subtract_gradient may read without an exchange
    int tail = ctx->nranks > 1 ? ctx->solve_tail : 0;
    for (size_t k = 0; k < peers.size() && tail > 0; ++k) {
        const sfl_context *c = peers[k];
        const sfl_plan_step &last = progs[k].back();
        if (last.kind != SFL_STEP_SOR) tail = 0;
        if (c->rank > 0) tail = std::min(tail, c->g0 - last.g_begin);
        if (c->rank < c->nranks - 1) tail = std::min(tail, last.g_end - c->g1);
    }
    for (sfl_context *c : peers) c->p_ghost_valid = tail > 0 ? tail : 0;
    return SFL_OK;
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
    return SFL_OK;
}

}  // namespace

// ==========================================================================================
// utilities
// ==========================================================================================
extern "C" {

int sfl_abi_version(void) { return SFL_ABI_VERSION; }

const char *sfl_last_error(void) { return g_error.c_str(); }

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
    HIP_TRY(hipMalloc(&flag, 4 * sizeof(int)));
    HIP_TRY(hipMemset(flag, 0, 4 * sizeof(int)));
    c->halo_flag = static_cast<int *>(flag);
    c->d_arrival = c->halo_flag + 1;
    c->d_done = c->halo_flag + 3;
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
        // (their collective operators then fail with SFL_ERR_STATE); the shared stream lives
        // on through `keepalive` until the last member is destroyed
        std::shared_ptr<Group> g = c->group;
        for (sfl_context *m : g->members) m->group.reset();
    }
    c->keepalive.reset();
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);  // nothing of the communicator may still be queued
    if (c->comm) (void)ncclCommDestroy(c->comm);
    for (void *m : {(void *)c->vel, (void *)c->vel_tmp, (void *)c->col, (void *)c->col_tmp,
                    (void *)c->sor_block, (void *)c->halo_flag, (void *)c->d_chain,
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
            if (value < 0 || value > kGhostRows)
                return fail(SFL_ERR_INVALID, "advect halo must be 0 (auto) or 1..%d rows", kGhostRows);
            c->opt_advect_halo = value;
            return SFL_OK;
        case SFL_OPT_SOR_ROWS:
            if (value < 0) return fail(SFL_ERR_INVALID, "rows per chunk must be >= 0");
            c->opt_sor_rows = value;
            return SFL_OK;
        case SFL_OPT_TRANSPORT:
            return fail(SFL_ERR_INVALID, "SFL_OPT_TRANSPORT is read-only: use sfl_comm_attach / sfl_group_link");
        case SFL_OPT_LAST_CHAINED:
        case SFL_OPT_LAST_EARLY_ROWS:
            return fail(SFL_ERR_INVALID, "this option is read-only");
        case SFL_OPT_FUSE_PROJECTION:
            c->opt_fuse_projection = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_SOR_OVERLAP:
            c->opt_sor_overlap = value ? 1 : 0;
            return SFL_OK;
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
            return SFL_OK;
        case SFL_OPT_SOR_ARRIVAL:
            c->opt_sor_arrival = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_STEP_SEAMS:
            c->opt_step_seams = value ? 1 : 0;
            return SFL_OK;
        case SFL_OPT_SOR_CHAIN:
            if (value < -1) return fail(SFL_ERR_INVALID, "SFL_OPT_SOR_CHAIN must be -1 (auto), 0, 1 or a number of waves >= 8");
            c->opt_sor_chain = value;   // >= 8: on, with at most that many waves per chain (several tiles per wave: a test aid)
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
        if (c->comm) c->options_dirty = true;
    }
    return SFL_OK;
}

int sfl_get_option(sfl_context *c, int option, int *value)
{
    if (!c || !value) return fail(SFL_ERR_INVALID, "NULL argument");
    switch (option) {
        case SFL_OPT_SOR_KERNEL: *value = c->opt_sor_kernel; return SFL_OK;
        case SFL_OPT_SOR_FUSE: *value = c->opt_sor_fuse; return SFL_OK;
        case SFL_OPT_ADVECT_HALO: *value = c->opt_advect_halo; return SFL_OK;
        case SFL_OPT_SOR_ROWS: *value = c->opt_sor_rows; return SFL_OK;
        case SFL_OPT_TRANSPORT: *value = c->comm ? 1 : (c->group ? 2 : (c->emulated ? 3 : 0)); return SFL_OK;
        case SFL_OPT_SOR_LANE_CELLS: *value = c->opt_sor_lane_cells; return SFL_OK;
        case SFL_OPT_SOR_HALO: *value = c->opt_sor_halo; return SFL_OK;
        case SFL_OPT_FUSE_PROJECTION: *value = c->opt_fuse_projection; return SFL_OK;
        case SFL_OPT_SOR_OVERLAP: *value = c->opt_sor_overlap; return SFL_OK;
        case SFL_OPT_ADVECT_KERNEL: *value = c->opt_advect_kernel; return SFL_OK;
        case SFL_OPT_FUSE_DIVERGENCE: *value = c->opt_fuse_divergence; return SFL_OK;
        case SFL_OPT_SMALL_GRID: *value = c->opt_small_grid; return SFL_OK;
        case SFL_OPT_EMULATE_WIRE_US: *value = c->opt_emulate_wire_us; return SFL_OK;
        case SFL_OPT_SOR_ARRIVAL: *value = c->opt_sor_arrival; return SFL_OK;
        case SFL_OPT_STEP_SEAMS: *value = c->opt_step_seams; return SFL_OK;
        case SFL_OPT_SOR_CHAIN: *value = c->opt_sor_chain; return SFL_OK;
        case SFL_OPT_LAST_CHAINED: *value = c->last_chained; return SFL_OK;
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

int sfl_comm_unique_id(void *id_out, size_t id_bytes)
{
    if (!id_out || id_bytes < sizeof(ncclUniqueId))
        return fail(SFL_ERR_INVALID, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return SFL_OK;
}

// Everything that must be identical on all ranks of a communicator for their programs to match: the domain,
// the group size and every option a plan or an exchange depends on.
constexpr int kOptionBlockInts = 16;
static void option_block(const sfl_context *c, int *b)
{
    const int v[kOptionBlockInts] = {SFL_ABI_VERSION, c->dim_x, c->gdim_y, c->nranks, c->opt_sor_kernel, c->opt_sor_fuse,
                                     c->opt_sor_halo, c->opt_sor_overlap, c->opt_advect_halo, c->opt_fuse_projection,
                                     c->opt_advect_kernel, c->opt_fuse_divergence, c->opt_small_grid, c->opt_sor_arrival, c->opt_sor_chain, 0};
    memcpy(b, v, sizeof v);
}

int sfl_comm_check_options(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (!c->comm) return fail(SFL_ERR_STATE, "no communicator attached");
    SFL_TRY(use_device(c));
    Overlap o;
    SFL_TRY(overlap_of(c, &o));
    int mine[kOptionBlockInts];
    option_block(c, mine);
    int *dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(int) * kOptionBlockInts * (size_t)(c->nranks + 1)));
    std::vector<int> all((size_t)kOptionBlockInts * c->nranks);
    int rc = SFL_OK;
    do {  // (single exit: the scratch buffer is freed on every path)
        if (hipMemcpy(dev, mine, sizeof mine, hipMemcpyHostToDevice) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = fail(SFL_ERR_HIP, "option block upload failed");
            break;
        }
        const ncclResult_t r = ncclAllGather(dev, dev + kOptionBlockInts, kOptionBlockInts, ncclInt32, c->comm, o.xstream);
        if (r != ncclSuccess) {
            rc = fail(SFL_ERR_RCCL, "ncclAllGather of the option block failed: %s", ncclGetErrorString(r));
            break;
        }
        if (hipStreamSynchronize(o.xstream) != hipSuccess ||
            hipMemcpy(all.data(), dev + kOptionBlockInts, all.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(SFL_ERR_HIP, "option block download failed");
            break;
        }
        for (int r2 = 0; r2 < c->nranks && rc == SFL_OK; ++r2)
            for (int k = 0; k < kOptionBlockInts; ++k)
                if (all[(size_t)r2 * kOptionBlockInts + k] != mine[k]) {
                    rc = fail(SFL_ERR_STATE, "rank %d and rank %d disagree on option-block word %d (%d vs %d): every rank "
                              "of a communicator must be created for the same domain and carry the same options",
                              c->rank, r2, k, mine[k], all[(size_t)r2 * kOptionBlockInts + k]);
                    break;
                }
    } while (false);
    (void)hipFree(dev);
    if (rc == SFL_OK) c->options_dirty = false;
    return rc;
}

int sfl_comm_attach(sfl_context *c, const void *id, size_t id_bytes)
{
    if (!c || !id || id_bytes < sizeof(ncclUniqueId)) return fail(SFL_ERR_INVALID, "bad arguments");
    if (c->comm || c->group || c->emulated) return fail(SFL_ERR_STATE, "context already has a transport");
    SFL_TRY(use_device(c));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCL_TRY(ncclCommInitRank(&c->comm, c->nranks, uid, c->rank));
    // the ranks are separate processes: a rank created for another domain or with other options would run a
    // different program (mismatched sends / receives: a hang or silently wrong halos) -- refuse it here, and
    // do not stay attached to a group this rank does not fit
    const int rc = sfl_comm_check_options(c);
    if (rc != SFL_OK) {
        const std::string why = g_error;
        if (c->xstream) (void)hipStreamSynchronize(c->xstream);
        (void)ncclCommDestroy(c->comm);
        c->comm = nullptr;
        g_error = why;
    }
    return rc;
}

int sfl_comm_emulate(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (c->comm || c->group) return fail(SFL_ERR_STATE, "context already has a transport");
    if (c->nranks < 2) return fail(SFL_ERR_STATE, "a whole-domain context has nothing to exchange");
    c->emulated = true;
    return SFL_OK;
}

int sfl_comm_loopback(sfl_context *c, int rows)
{
    if (!c || rows < 1 || rows > c->g1 - c->g0) return fail(SFL_ERR_INVALID, "bad loopback request");
    if (!c->comm) return fail(SFL_ERR_STATE, "no communicator attached");
    SFL_TRY(ensure_field(c, SFL_FIELD_DIVERGENCE));
    SFL_TRY(ensure_field(c, SFL_FIELD_PRESSURE));
    SFL_TRY(use_device(c));
    const size_t bytes = (size_t)rows * c->dim_x * 4;
    const size_t off = c->owned_offset_cells();
    Overlap o;  // like every operation of the communicator: on the exchange stream, between the two events
    SFL_TRY(overlap_of(c, &o));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    NCCL_TRY(ncclGroupStart());
    NCCL_TRY(ncclSend(c->div + off, bytes, ncclChar, c->rank, c->comm, o.xstream));
    NCCL_TRY(ncclRecv(c->p + off, bytes, ncclChar, c->rank, c->comm, o.xstream));
    c->p_ghost_valid = 0;
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    return SFL_OK;
}

int sfl_group_link(sfl_context **ctxs, int n)
{
    if (!ctxs || n < 1) return fail(SFL_ERR_INVALID, "bad group");
    for (int r = 0; r < n; ++r) {
        sfl_context *c = ctxs[r];
        if (!c || c->nranks != n || c->rank != r || c->device != ctxs[0]->device ||
            c->dim_x != ctxs[0]->dim_x || c->gdim_y != ctxs[0]->gdim_y || c->comm || c->group)
            return fail(SFL_ERR_INVALID, "ctxs[%d] is not slab %d of %d on the group's device", r, r, n);
    }
    auto g = std::make_shared<Group>();
    g->members.assign(ctxs, ctxs + n);
    HIP_TRY(hipSetDevice(ctxs[0]->device));
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    for (int r = 1; r < n; ++r) {  // group-wide options: slab 0's values
        sfl_context *c = ctxs[r], *z = ctxs[0];
        c->opt_sor_kernel = z->opt_sor_kernel;
        c->opt_sor_fuse = z->opt_sor_fuse;
        c->opt_advect_halo = z->opt_advect_halo;
        c->opt_sor_rows = z->opt_sor_rows;
        c->opt_sor_lane_cells = z->opt_sor_lane_cells;
        c->opt_sor_halo = z->opt_sor_halo;
        c->opt_fuse_projection = z->opt_fuse_projection;
        c->opt_sor_overlap = z->opt_sor_overlap;
        c->opt_advect_kernel = z->opt_advect_kernel;
        c->opt_fuse_divergence = z->opt_fuse_divergence;
        c->opt_small_grid = z->opt_small_grid;
        c->opt_sor_arrival = z->opt_sor_arrival;
        c->opt_sor_chain = z->opt_sor_chain;
    }
    for (int r = 0; r < n; ++r) {  // one stream orders the whole group
        sfl_context *c = ctxs[r];
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
        c->stream = g->stream;
        c->owns_stream = false;
        c->group = g;
        c->keepalive = g;
    }
    return SFL_OK;
}

static int settle_color(sfl_context *ctx);

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
struct AdvectPlan {
    int halo = 0;         // rows to exchange per side (fixed or measured)
    bool gather = false;  // sample a gathered copy of the whole field instead
    bool flag = true;     // fixed halo: let the kernel report a back-trace that leaves it
    bool report = false;  // ... into the context's reach report (a guessed halo, checked by settle_color)
                          // instead of the error flag sfl_synchronize turns into SFL_ERR_HALO
    bool halo_sent = false;  // the halo rows are already on their way (behind ev_color_halo): wait, do not exchange
};

int *advect_flag(sfl_context *c, const AdvectPlan &plan)
{
    if (c->nranks == 1 || plan.gather) return nullptr;
    if (plan.report) return c->d_report + 2;
    return plan.flag ? c->halo_flag : nullptr;
}

// Reach words of a slab (device ints, atomicMax'ed by backtrace_reach_kernel; zero them first):
//   [0] / [1]  rows the back-traces of the OWNED rows need below / above the slab;
//   [5]        rows the back-traces of the slab's FIRST row need above it, [6] those of its LAST row below it --
//              what the neighbour needs when it advects that row itself as a ghost row (slab_step_auto: the
//              velocity advection covers own +- 1 rows); [4], [7] come with them and are covered by [0] / [1].
constexpr int kReachWords = 8;
int launch_reach_set(sfl_context *c, int *words, float dt)
{
    SFL_TRY(use_device(c));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words, c->vel, c->geom, c->g0, c->g1, dt));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words + 4, c->vel, c->geom, c->g0, std::min(c->g0 + 1, c->g1), dt));
    HIP_TRY(sfl::launch_backtrace_reach(c->stream, words + 6, c->vel, c->geom, std::max(c->g1 - 1, c->g0), c->g1, dt));
    return SFL_OK;
}
// halo that covers the owned rows' back-traces / those of own +- 1 rows (the neighbours' edge rows included)
int reach_own(const int *w) { return std::max(w[0], w[1]); }
int reach_extended(const int *w) { return std::max(reach_own(w), 1 + std::max(w[5], w[6])); }

int measure_reach(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int *reach_out,
                  int *reach_ext_out = nullptr)
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
    if (ctx->comm) {  // maximum over the ranks, on the exchange stream like every RCCL operation
        Overlap o;
        SFL_TRY(overlap_of(ctx, &o));
        HIP_TRY(hipEventRecord(o.ready, o.compute));
        HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
        NCCL_TRY(ncclAllReduce(ctx->d_reach, ctx->d_reach, kReachWords, ncclInt32, ncclMax, ctx->comm, o.xstream));
        HIP_TRY(hipEventRecord(o.arrived, o.xstream));
        HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    }
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

int plan_advect(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, AdvectPlan *plan)
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
    if (reach <= kGhostRows && reach <= min_owned_rows(ctx))
        plan->halo = reach;
    else
        plan->gather = true;
    return SFL_OK;
}

// Gather the whole `field` (owned rows of every slab) into each context's gather_buf.
int gather_field(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field)
{
    const size_t eb = field_elem_bytes(field);
    const size_t row_bytes = (size_t)ctx->dim_x * eb;
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        if (!c->gather_buf) HIP_TRY(hipMalloc(&c->gather_buf, (size_t)c->gdim_y * c->dim_x * 12));
        ++c->last_exchanges;
    }
    auto owned = [&](sfl_context *c) {
        return static_cast<char *>(field_ptr(c, field)) + c->owned_offset_cells() * eb;
    };
    if (ctx->group) {
        for (sfl_context *c : peers)
            for (sfl_context *m : peers)
                HIP_TRY(hipMemcpyAsync(static_cast<char *>(c->gather_buf) + (size_t)m->g0 * row_bytes, owned(m),
                                       (size_t)(m->g1 - m->g0) * row_bytes, hipMemcpyDeviceToDevice, c->stream));
        return SFL_OK;
    }
    sfl_context *c = ctx;
    if (!c->comm) return fail(SFL_ERR_STATE, "slab %d/%d has no communicator", c->rank, c->nranks);
    Overlap o;
    SFL_TRY(overlap_of(c, &o));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(c->gather_buf) + (size_t)c->g0 * row_bytes, owned(c),
                           (size_t)(c->g1 - c->g0) * row_bytes, hipMemcpyDeviceToDevice, o.xstream));
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        int b = 0, e = 0;
        sfl::slab_rows(c->gdim_y, c->nranks, r, &b, &e);
        NCCL_TRY(ncclSend(owned(c), (size_t)(c->g1 - c->g0) * row_bytes, ncclChar, r, c->comm, o.xstream));
        NCCL_TRY(ncclRecv(static_cast<char *>(c->gather_buf) + (size_t)b * row_bytes, (size_t)(e - b) * row_bytes,
                          ncclChar, r, c->comm, o.xstream));
    }
    NCCL_TRY(ncclGroupEnd());
    HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    return SFL_OK;
}

// ---- operators ---------------------------------------------------------------------------
static int settle_color(sfl_context *ctx);

static int advect_velocity_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                                   const AdvectPlan &plan, int extend = 0, int interior_done = 0);

int sfl_advect_velocity(sfl_context *ctx, float dt, int no_slip)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx));
    std::vector<sfl_context *> peers = peers_of(ctx);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, SFL_FIELD_VELOCITY));
        SFL_TRY(ensure(c, c->vel_tmp, 8, false));
    }
    AdvectPlan plan;
    SFL_TRY(plan_advect(ctx, peers, dt, &plan));
    return advect_velocity_planned(ctx, peers, dt, no_slip, plan);
}

// `extend` = 1 (slab_step_auto; plan.halo then covers one row more than the reach): the ghost rows next to the cuts
// are advected as well, redundantly -- calculate_divergence then needs no exchange of its own.
// `interior_done` = L > 0 (slab_step_auto): rows [g0 + L, g1 - L) are in vel_tmp already (advect_interior_early): only the two
// bands next to the cuts, the rows that may need the halo, are advected here.
static int advect_velocity_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
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

static int advect_color_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
                                const AdvectPlan &plan);

int sfl_advect_color(sfl_context *ctx, float dt, int no_slip)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx));
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

static int advect_color_planned(sfl_context *ctx, const std::vector<sfl_context *> &peers, float dt, int no_slip,
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

static int check_channels(int channels, int kind);

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
    SFL_TRY(settle_color(ctx));
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
    SFL_TRY(settle_color(ctx));
    return run_poisson(ctx, dx, iters, omega);
}

int sfl_subtract_gradient(sfl_context *ctx, float dx)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx));
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

// Copies the queued (cell, velocity) pairs to the device (asynchronously, through pinned staging) and empties the
// queue; *count = how many now wait in d_force_cells / d_force_vel for the kernel that applies them.
static int stage_queued_forces(sfl_context *c, int *count)
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

static int apply_queued_forces(sfl_context *c)
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
static int project_and_advect_color(sfl_context *ctx, float dt, float dx, int halo, bool report, bool halo_sent = false)
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
static bool can_fuse_divergence(const sfl_context *c)
{
    if (!c->opt_fuse_divergence || c->nranks != 1 || c->group || !c->force_cells.empty()) return false;
    const int64_t cells = (int64_t)c->dim_x * c->gdim_y;
    return c->opt_advect_kernel == 2 || (c->opt_advect_kernel == 0 && cells >= sfl::kAdvectTiledMinCells);
}

static int advect_velocity_and_divergence(sfl_context *c, float dt, float dx)
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
    a.prm = sor_params(dx, omega);
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
    HIP_TRY(hipHostMalloc(&h, kReachWords * sizeof(int), hipHostMallocDefault));
    memset(h, 0, kReachWords * sizeof(int));
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
    if (ctx->comm) {  // maximum over the ranks, on the exchange stream like every RCCL operation
        Overlap o;
        SFL_TRY(overlap_of(ctx, &o));
        HIP_TRY(hipEventRecord(o.ready, o.compute));
        HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
        NCCL_TRY(ncclAllReduce(ctx->d_report, ctx->d_report, kReachWords, ncclInt32, ncclMax, ctx->comm, o.xstream));
        HIP_TRY(hipMemcpyAsync(ctx->h_report, ctx->d_report, kReachWords * sizeof(int), hipMemcpyDeviceToHost, o.xstream));
        HIP_TRY(hipEventRecord(ctx->ev_report, o.xstream));
    } else {
        for (sfl_context *c : peers) {
            SFL_TRY(use_device(c));
            HIP_TRY(hipMemcpyAsync(c->h_report, c->d_report, kReachWords * sizeof(int), hipMemcpyDeviceToHost, c->stream));
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
static int settle_color(sfl_context *ctx)
{
    if (ctx->comm && ctx->options_dirty) SFL_TRY(sfl_comm_check_options(ctx));   // (collective; see sfl_set_option)
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
    if (reach <= kGhostRows && reach <= min_owned_rows(ctx))
        plan.halo = reach;
    else
        plan.gather = true;
    for (sfl_context *c : peers) std::swap(c->col, c->col_tmp);
    return advect_color_planned(ctx, peers, dt, 0, plan);
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
    const int limit = std::min(kGhostRows, min_owned_rows(ctx));
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
    pv.flag = !ctx->emulated;
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
    const int limit = std::min(kGhostRows, min_owned_rows(ctx));
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

int sfl_step(sfl_context *ctx, float dt, float dx, int iters, float omega)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (ctx->nranks > 1 && ctx->opt_advect_halo == 0 && ctx->color_unsettled && ctx->unsettled_dt == dt && ctx->force_cells.empty())
        SFL_TRY(advect_interior_early(ctx, dt));
    SFL_TRY(settle_color(ctx));
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

// The loop of the sim task (ino:249-289) calls the step back to back.  n steps in one call give the library the one
// fusion a per-step API has no place for: between two steps the projected velocity is written by the last kernel of
// one and read straight back by the first kernel of the next -- the seam kernel does both and never stores it
// (780 us at 8192^2 where the two kernels take 568 + 254; SFL_OPT_STEP_SEAMS, profiles/r04_step_seam.txt).
int sfl_step_n(sfl_context *ctx, int n, float dt, float dx, int iters, float omega)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (n < 0) return fail(SFL_ERR_INVALID, "n must be >= 0 (got %d)", n);
    SFL_TRY(settle_color(ctx));
    const int64_t cells = (int64_t)ctx->dim_x * ctx->gdim_y;
    const bool tiled = ctx->opt_advect_kernel == 2 || (ctx->opt_advect_kernel == 0 && cells >= sfl::kAdvectTiledMinCells);
    const bool seams = n > 1 && ctx->opt_step_seams && ctx->nranks == 1 && !ctx->group && !small_grid(ctx) && tiled &&
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

int sfl_synchronize(sfl_context *ctx)
{
    if (!ctx) return fail(SFL_ERR_INVALID, "ctx is NULL");
    SFL_TRY(settle_color(ctx));
    int rc = SFL_OK;
    for (sfl_context *c : peers_of(ctx)) {
        SFL_TRY(use_device(c));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->nranks > 1 || c->d_chain) {
            int words[3] = {0, 0, 0};   // halo_flag, arrival count, a wait for it timed out
            HIP_TRY(hipMemcpy(words, c->halo_flag, sizeof words, hipMemcpyDeviceToHost));
            if (words[0]) {
                HIP_TRY(hipMemset(c->halo_flag, 0, sizeof(int)));
                rc = fail(SFL_ERR_HALO, "slab %d/%d: a back-trace left the %d-row advect halo; raise "
                          "SFL_OPT_ADVECT_HALO", c->rank, c->nranks, c->opt_advect_halo);
            }
            if (words[2]) {
                HIP_TRY(hipMemset(c->halo_flag + 2, 0, sizeof(int)));
                // bits: 1 a tile of a launch, 4 a tile of a chained launch, 8 the exchange stream (for the sender count) waited
                // for a halo message; 2 a tile of a chained launch for the tiles around it
                rc = fail(SFL_ERR_HIP, "slab %d/%d: a wait inside a solve lasted longer than %d s (waits 0x%x; arrival "
                          "count %d of %d): the pressure field is not valid", c->rank, c->nranks,
                          sfl::kHaloWaitTimeoutUs / 1000000, words[2], words[1], c->arrival_epoch);
            }
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

// ==========================================================================================
// host-pointer drop-ins
// ==========================================================================================
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

static int check_channels(int channels, int kind)
{
    if (channels < 1 || channels > 3 || (kind != SFL_CHANNEL_F32 && kind != SFL_CHANNEL_UQ32))
        return fail(SFL_ERR_INVALID, "advect: element must be 1..3 channels of kind SFL_CHANNEL_F32 / SFL_CHANNEL_UQ32 "
                    "(got %d x kind %d)", channels, kind);
    return SFL_OK;
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
