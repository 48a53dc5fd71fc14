// transport.cpp -- see transport.h: the three halo transports, the exchange protocol around them, communicator
// attach / option check / group link of the C ABI.  Host C++ only.
#include "transport.h"

#include <rccl/rccl.h>

#define NCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return ::sfl::host::fail(SFL_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_),     \
                                     __FILE__, __LINE__);                                                      \
    } while (0)

namespace sfl {
namespace host {

namespace {

char *row_ptr(sfl_context *c, int field, int g)
{
    const size_t row_bytes = (size_t)c->dim_x * field_elem_bytes(field);
    return static_cast<char *>(field_ptr(c, field)) + (size_t)(g - c->geom.grow0) * row_bytes;
}

char *owned_ptr(sfl_context *c, int field)
{
    return static_cast<char *>(field_ptr(c, field)) + c->owned_offset_cells() * field_elem_bytes(field);
}

size_t band_bytes(const sfl_context *c, const HaloBands &b) { return (size_t)c->dim_x * field_elem_bytes(b.field) * b.rows; }

// ---- RCCL: one process per GPU (or, with `self`, this rank talking to itself) --------------------------------------
class Rccl : public Transport {
public:
    ncclComm_t comm = nullptr;
    bool self = false;   // a one-rank communicator: every neighbour is this rank (sfl_comm_emulate_rccl)
    hipStream_t xstream = nullptr;   // the owning context's exchange stream: drained before the communicator goes

    ~Rccl() override
    {
        if (xstream) (void)hipStreamSynchronize(xstream);  // nothing of the communicator may still be queued
        if (comm) (void)ncclCommDestroy(comm);
    }
    int peer(int rank) const { return self ? 0 : rank; }
    int kind() const override { return self ? 4 : 1; }
    bool arrival_by_default() const override
    {
        // Real peers: a launch that waits inside the kernel for a message of another process is only as safe as that
        // process is punctual, and the scheme has never run on more than one GPU (tools/first_multi_gpu.sh): early
        // exchanges behind events unless SFL_OPT_EXCHANGE_SCHEDULE = 3 asks for it.  Talking to itself the rank is its own peer.
        return self;
    }
    bool separate_processes() const override { return !self; }
    int default_timeout_us() const override { return self ? 2000000 : 300000000; }

    int move(const std::vector<sfl_context *> &peers, const HaloBands &b, hipStream_t on) override
    {
        sfl_context *c = peers[0];
        SFL_TRY(use_device(c));
        hipStream_t st = on ? on : c->stream;
        const size_t bytes = band_bytes(c, b);
        if (self) HIP_TRY(launch_spin_us(st, c->opt_emulate_wire_us));   // (a rank talking to itself has no wire: SFL_OPT_EMULATE_WIRE_US adds one)
        NCCL_TRY(ncclGroupStart());
        if (c->rank > 0) {
            NCCL_TRY(ncclSend(row_ptr(c, b.field, c->g0 + b.skip), bytes, ncclChar, peer(c->rank - 1), comm, st));
            NCCL_TRY(ncclRecv(row_ptr(c, b.field, c->g0 - b.skip - b.rows), bytes, ncclChar, peer(c->rank - 1), comm, st));
        }
        if (c->rank < c->nranks - 1) {
            NCCL_TRY(ncclSend(row_ptr(c, b.field, c->g1 - b.skip - b.rows), bytes, ncclChar, peer(c->rank + 1), comm, st));
            NCCL_TRY(ncclRecv(row_ptr(c, b.field, c->g1 + b.skip), bytes, ncclChar, peer(c->rank + 1), comm, st));
        }
        NCCL_TRY(ncclGroupEnd());
        return SFL_OK;
    }
    int allreduce_max(sfl_context *c, int *dev_words, int n, hipStream_t on) override
    {
        SFL_TRY(use_device(c));
        NCCL_TRY(ncclAllReduce(dev_words, dev_words, n, ncclInt32, ncclMax, comm, on));
        return SFL_OK;
    }
    int gather(const std::vector<sfl_context *> &peers, int field, hipStream_t on) override
    {
        sfl_context *c = peers[0];
        SFL_TRY(use_device(c));
        const size_t row_bytes = (size_t)c->dim_x * field_elem_bytes(field);
        const int own = c->g1 - c->g0;
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(c->gather_buf) + (size_t)c->g0 * row_bytes, owned_ptr(c, field),
                               (size_t)own * row_bytes, hipMemcpyDeviceToDevice, on));
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < c->nranks; ++r) {
            if (r == c->rank) continue;
            int b = 0, e = 0;
            slab_rows(c->gdim_y, c->nranks, r, &b, &e);
            // (talking to itself a rank receives what it sends: the rows of the shorter of the two slabs)
            const int send_rows = self ? std::min(own, e - b) : own, recv_rows = self ? send_rows : e - b;
            NCCL_TRY(ncclSend(owned_ptr(c, field), (size_t)send_rows * row_bytes, ncclChar, peer(r), comm, on));
            NCCL_TRY(ncclRecv(static_cast<char *>(c->gather_buf) + (size_t)b * row_bytes, (size_t)recv_rows * row_bytes,
                              ncclChar, peer(r), comm, on));
        }
        NCCL_TRY(ncclGroupEnd());
        return SFL_OK;
    }
};

// ---- ONE rank of the group alone (bench.py --emulate-rank): every message it would send is copied, same size, same
// stream, into the ghost rows it would receive into.  The bytes are this rank's own, so results next to the cuts mean
// nothing; launches, copies and their ordering are the rank's program. ---------------------------------------------
class Emulated : public Transport {
public:
    int kind() const override { return 3; }
    bool arrival_by_default() const override { return true; }
    int move(const std::vector<sfl_context *> &peers, const HaloBands &b, hipStream_t on) override
    {
        sfl_context *c = peers[0];
        SFL_TRY(use_device(c));
        hipStream_t st = on ? on : c->stream;
        const bool lo = c->rank > 0, hi = c->rank < c->nranks - 1;
        void *dst_a = lo ? row_ptr(c, b.field, c->g0 - b.skip - b.rows) : nullptr, *dst_b = hi ? row_ptr(c, b.field, c->g1 + b.skip) : nullptr;
        const void *src_a = lo ? row_ptr(c, b.field, c->g0 + b.skip) : nullptr, *src_b = hi ? row_ptr(c, b.field, c->g1 - b.skip - b.rows) : nullptr;
        HIP_TRY(launch_spin_us(st, c->opt_emulate_wire_us));   // the wire a self-copy does not have (0: none)
        HIP_TRY(launch_copy_bands(st, dst_a, src_a, dst_b, src_b, band_bytes(c, b)));
        return SFL_OK;
    }
    int gather(const std::vector<sfl_context *> &peers, int, hipStream_t) override
    {
        return fail(SFL_ERR_STATE, "slab %d/%d has no communicator (an emulated rank has nobody to gather from)",
                    peers[0]->rank, peers[0]->nranks);
    }
};

Rccl *rccl_of(const sfl_context *c) { return c->transport ? dynamic_cast<Rccl *>(c->transport.get()) : nullptr; }

}  // namespace

// ---- the group of virtual ranks ---------------------------------------------------------------------------------------
Group::~Group()
{
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    if (ev_arrived) (void)hipEventDestroy(ev_arrived);
    if (xstream) (void)hipStreamDestroy(xstream);
    if (stream) (void)hipStreamDestroy(stream);
}

int Group::move(const std::vector<sfl_context *> &peers, const HaloBands &b, hipStream_t on)
{
    for (sfl_context *c : peers) {   // both bands of a slab in one launch
        SFL_TRY(use_device(c));
        sfl_context *lo = c->rank > 0 ? peers[c->rank - 1] : nullptr;
        sfl_context *hi = c->rank < c->nranks - 1 ? peers[c->rank + 1] : nullptr;
        void *dst_a = lo ? row_ptr(c, b.field, c->g0 - b.skip - b.rows) : nullptr, *dst_b = hi ? row_ptr(c, b.field, c->g1 + b.skip) : nullptr;
        const void *src_a = lo ? row_ptr(lo, b.field, lo->g1 - b.skip - b.rows) : nullptr, *src_b = hi ? row_ptr(hi, b.field, hi->g0 + b.skip) : nullptr;
        HIP_TRY(launch_copy_bands(on ? on : c->stream, dst_a, src_a, dst_b, src_b, band_bytes(c, b)));
    }
    return SFL_OK;
}

int Group::gather(const std::vector<sfl_context *> &peers, int field, hipStream_t)
{
    const size_t row_bytes = (size_t)peers[0]->dim_x * field_elem_bytes(field);
    for (sfl_context *c : peers)
        for (sfl_context *m : peers)
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(c->gather_buf) + (size_t)m->g0 * row_bytes, owned_ptr(m, field),
                                   (size_t)(m->g1 - m->g0) * row_bytes, hipMemcpyDeviceToDevice, c->stream));
    return SFL_OK;
}

// ---- the exchange protocol ----------------------------------------------------------------------------------------------
int overlap_of(sfl_context *c, Overlap *o)
{
    SFL_TRY(use_device(c));
    hipStream_t *xs = c->group ? &c->group->xstream : &c->xstream;
    hipEvent_t *e0 = c->group ? &c->group->ev_ready : &c->ev_ready;
    hipEvent_t *e1 = c->group ? &c->group->ev_arrived : &c->ev_arrived;
    if (!*xs) HIP_TRY(hipStreamCreateWithFlags(xs, hipStreamNonBlocking));   // (created with the context / the group: see there)
    if (!*e0) HIP_TRY(hipEventCreateWithFlags(e0, hipEventDisableTiming));
    if (!*e1) HIP_TRY(hipEventCreateWithFlags(e1, hipEventDisableTiming));
    o->compute = c->stream;  // a linked group shares one compute stream
    o->xstream = *xs;
    o->ready = *e0;
    o->arrived = *e1;
    return SFL_OK;
}

// (Raising the arrival count in the copy kernel itself -- written-through stores, the block that finishes last stores the
// word -- saves the signal kernel, 4 - 6 us, beside the solve's first launch, which only reads d, and costs more than it saves
// beside a launch at full memory traffic: the copy's stores then take 10 us to drain.  Measured, not kept:
// profiles/r04_exchanges_counted_on_the_device.txt.)
int exchange(const std::vector<sfl_context *> &peers, int field, int rows, hipStream_t on, int skip, bool in_time,
             bool wait_done)
{
    if (rows <= 0) return SFL_OK;
    sfl_context *any = peers[0];
    if (any->nranks == 1) return SFL_OK;
    if (skip < 0) return fail(SFL_ERR_INVALID, "negative halo offset");
    if (skip + rows > any->ghost)
        return fail(SFL_ERR_INVALID, "halo of %d rows exceeds the %d ghost rows of a slab", skip + rows, any->ghost);
    if (skip + rows > min_owned_rows(any))
        return fail(SFL_ERR_INVALID, "halo of %d rows exceeds the thinnest slab (%d rows)", skip + rows,
                    min_owned_rows(any));
    Transport *t = any->transport.get();
    if (!t)
        return fail(SFL_ERR_STATE, "slab %d/%d has no communicator: call sfl_comm_attach() or sfl_group_link() first",
                    any->rank, any->nranks);
    for (sfl_context *c : peers) {
        SFL_TRY(ensure_field(c, field));
        ++c->last_exchanges;
    }
    if (in_time && wait_done)   // every slab's sender tiles first (a slab reads from its two neighbours)
        for (sfl_context *c : peers) {
            SFL_TRY(use_device(c));
            HIP_TRY(launch_wait_count(on ? on : c->stream, c->d_done, c->done_target, c->d_arrival + 1, halo_timeout_us(c)));
        }
    SFL_TRY(t->move(peers, HaloBands{field, rows, skip}, on));
    // every message before any signal: a slab's arrival count then also says that its neighbours have READ what it sent
    if (in_time)
        for (sfl_context *c : peers) {
            SFL_TRY(use_device(c));
            ++c->arrival_epoch;
            HIP_TRY(launch_signal_arrival(on ? on : c->stream, c->d_arrival, c->arrival_epoch));
        }
    return SFL_OK;
}

int start_exchange(const std::vector<sfl_context *> &peers, const Overlap &o, int field, int rows, int skip, bool mark,
                   bool in_time)
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

// With a transport of its own stream the message still travels on the exchange stream -- every operation of a communicator
// is issued to ONE stream, whatever the operator -- bracketed by the two events; in-process copies of a linked group go on
// the compute stream itself.
int exchange_inline(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field, int rows, int skip,
                    hipEvent_t after)
{
    if (rows <= 0 || ctx->nranks == 1) return SFL_OK;
    if (!ctx->transport || !ctx->transport->own_stream()) return exchange(peers, field, rows, nullptr, skip);
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

int gather_field(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field)
{
    for (sfl_context *c : peers) {
        SFL_TRY(use_device(c));
        if (!c->gather_buf) HIP_TRY(hipMalloc(&c->gather_buf, (size_t)c->gdim_y * c->dim_x * 12));
        ++c->last_exchanges;
    }
    Transport *t = ctx->transport.get();
    if (!t) return fail(SFL_ERR_STATE, "slab %d/%d has no communicator", ctx->rank, ctx->nranks);
    if (!t->own_stream()) return t->gather(peers, field, nullptr);
    Overlap o;
    SFL_TRY(overlap_of(ctx, &o));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    SFL_TRY(t->gather(peers, field, o.xstream));
    HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    return SFL_OK;
}

bool reduces_on_device(const sfl_context *ctx) { return rccl_of(ctx) != nullptr; }

int reduce_max_inline(sfl_context *ctx, int *dev_words, int n)
{
    if (!reduces_on_device(ctx)) return SFL_OK;
    Overlap o;   // on the exchange stream like every RCCL operation
    SFL_TRY(overlap_of(ctx, &o));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    SFL_TRY(ctx->transport->allreduce_max(ctx, dev_words, n, o.xstream));
    HIP_TRY(hipEventRecord(o.arrived, o.xstream));
    HIP_TRY(hipStreamWaitEvent(o.compute, o.arrived, 0));
    return SFL_OK;
}

int reduce_max_then_copy(sfl_context *ctx, int *dev_words, int n, int *host_words, hipEvent_t done)
{
    Overlap o;
    SFL_TRY(overlap_of(ctx, &o));
    HIP_TRY(hipEventRecord(o.ready, o.compute));
    HIP_TRY(hipStreamWaitEvent(o.xstream, o.ready, 0));
    SFL_TRY(ctx->transport->allreduce_max(ctx, dev_words, n, o.xstream));
    HIP_TRY(hipMemcpyAsync(host_words, dev_words, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, o.xstream));
    // the "a wait inside a solve gave up" word rides along (kReportWords): the next call on the context sees it
    HIP_TRY(hipMemcpyAsync(host_words + kReachWords, ctx->halo_flag + 2, sizeof(int), hipMemcpyDeviceToHost, o.xstream));
    HIP_TRY(hipEventRecord(done, o.xstream));
    return SFL_OK;
}

// One wave on the compute stream waits (at most 200 ms) for a word that a kernel on the exchange stream raises.  Streams
// that share a hardware queue run in submission order: the raise then sits behind the wait, which gives up.  (The
// runtime folds its streams onto GPU_MAX_HW_QUEUES queues, 4 by default, in turn: an application with many streams
// of its own can put a context's two streams on one.)
int streams_run_concurrently(sfl_context *ctx, bool *yes)
{
    int &cached = ctx->group ? ctx->group->streams_concurrent : ctx->streams_concurrent;
    if (cached < 0) {
        Overlap o;
        SFL_TRY(overlap_of(ctx, &o));
        int *w = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w), 2 * sizeof(int)));
        int h[2] = {0, 1};
        // (both streams have run a kernel before the clock starts: the first use of a stream creates its hardware queue, which
        // can take longer than any sensible limit of the wait below -- the first group of a process was found "not concurrent")
        hipError_t e = hipMemsetAsync(w, 0, 2 * sizeof(int), o.compute);
        if (e == hipSuccess) e = launch_signal_arrival(o.compute, w, 0);
        if (e == hipSuccess) e = hipStreamSynchronize(o.compute);
        if (e == hipSuccess) e = launch_signal_arrival(o.xstream, w, 0);
        if (e == hipSuccess) e = hipStreamSynchronize(o.xstream);
        if (e == hipSuccess) e = hipEventRecord(o.ready, o.compute);
        if (e == hipSuccess) e = launch_wait_count(o.compute, w, 1, w + 1, 200000);
        if (e == hipSuccess) e = hipStreamWaitEvent(o.xstream, o.ready, 0);
        if (e == hipSuccess) e = launch_signal_arrival(o.xstream, w, 1);
        if (e == hipSuccess) e = hipStreamSynchronize(o.xstream);
        if (e == hipSuccess) e = hipStreamSynchronize(o.compute);
        if (e == hipSuccess) e = hipMemcpy(h, w, sizeof h, hipMemcpyDeviceToHost);
        (void)hipFree(w);
        if (e != hipSuccess) return fail(SFL_ERR_HIP, "stream concurrency probe failed: %s", hipGetErrorString(e));
        cached = h[1] == 0 ? 1 : 0;
    }
    *yes = cached == 1;
    return SFL_OK;
}

// kReps exchanges of `rows_a` rows of p, then kReps of `rows_b`, each one handed over from and to the compute stream BY THE
// PROTOCOL THE NEXT SOLVE WILL USE, timed by events on the compute stream: what an exchange costs a stream that has to wait
// for it.  Behind events: an event in front of the message, one behind it.  Counted on the device: a one-wave kernel on the
// compute stream stands in for the launch whose sender tiles count themselves, the exchange stream waits for that count,
// moves the message and raises the arrival count, a one-wave kernel on the compute stream stands in for the launch whose
// cut-adjacent tiles wait for it.  (Back-to-back messages on the exchange stream alone pipeline their launches and show a
// third of the cost: 10 us for RCCL's send / receive pair where a solve pays 17 more than for a copy; the event hand-overs
// cost 18 us that the counted protocol does not pay.)  latency = what a message of no rows would cost, per_row = the slope.
// The ghost rows of p are overwritten (nobody relies on them between solves: p_ghost_valid is cleared).
int measure_exchange(sfl_context *ctx)
{
    std::vector<sfl_context *> peers = peers_of(ctx);
    Transport *t = ctx->transport.get();
    if (!t || ctx->nranks < 2) return SFL_OK;
    const int deepest = std::min(kLegacySorHalo, min_owned_rows(ctx));
    const int rows_a = std::max(1, deepest / 8), rows_b = deepest;
    constexpr int kReps = 20;
    Overlap o;
    SFL_TRY(overlap_of(ctx, &o));   // (without an exchange stream there is nothing to reduce on either)
    SFL_TRY(use_device(ctx));
    // from here on a local failure is carried to the reduction at the end instead of returned (see there; a rank that cannot even
    // start its exchanges leaves its neighbours in ncclRecv all the same: RCCL has no time-out, the launcher's deadline is the net)
    int rc = SFL_OK;
    for (sfl_context *c : peers) {
        if (rc == SFL_OK) rc = ensure_field(c, SFL_FIELD_PRESSURE);
        c->p_ghost_valid = 0;
    }
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    for (hipEvent_t &e : ev)
        if (rc == SFL_OK && hipEventCreate(&e) != hipSuccess) rc = fail(SFL_ERR_HIP, "exchange measurement: no event");
    float ms_a = 0.0f, ms_b = 0.0f;
    do {
        if (rc != SFL_OK) break;
        const bool counted = in_time_exchanges(ctx);
        auto one = [&](int rows) {
            if (!counted) {
                int r = start_exchange(peers, o, SFL_FIELD_PRESSURE, rows);
                return r == SFL_OK ? await_exchange(peers, o) : r;
            }
            for (sfl_context *c : peers) {
                ++c->done_target;
                if (launch_signal_arrival(o.compute, c->d_done, c->done_target) != hipSuccess) return fail(SFL_ERR_HIP, "exchange measurement: launch failed");
            }
            int r = exchange(peers, SFL_FIELD_PRESSURE, rows, o.xstream, 0, true, true);
            for (sfl_context *c : peers)
                if (r == SFL_OK && launch_wait_count(o.compute, c->d_arrival, c->arrival_epoch, c->d_arrival + 1, halo_timeout_us(c)) != hipSuccess)
                    r = fail(SFL_ERR_HIP, "exchange measurement: launch failed");
            return r;
        };
        for (int k = 0; k < 3 && rc == SFL_OK; ++k) rc = one(rows_b);   // warm-up
        if (rc != SFL_OK) break;
        (void)hipEventRecord(ev[0], o.compute);
        for (int k = 0; k < kReps && rc == SFL_OK; ++k) rc = one(rows_a);
        (void)hipEventRecord(ev[1], o.compute);
        for (int k = 0; k < kReps && rc == SFL_OK; ++k) rc = one(rows_b);
        (void)hipEventRecord(ev[2], o.compute);
        if (rc != SFL_OK) break;
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_a, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_b, ev[1], ev[2]);
        if (e != hipSuccess) rc = fail(SFL_ERR_HIP, "exchange measurement: %s", hipGetErrorString(e));
    } while (false);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    for (sfl_context *c : peers) c->last_exchanges = 0;
    // A rank that failed locally still takes part in the reduction below, with a value that says so: its peers are in it, and RCCL
    // has no time-out.  Every rank then sees the failure and returns an error from the same call.
    constexpr int kFailed = 1 << 30;
    const std::string why = rc != SFL_OK ? last_error() : std::string();
    int words[2] = {kFailed, kFailed};
    if (rc == SFL_OK) {
        const double us_a = ms_a * 1e3 / kReps, us_b = ms_b * 1e3 / kReps;
        double per_row_ns = rows_b > rows_a ? (us_b - us_a) * 1e3 / (rows_b - rows_a) : 0.0;
        if (per_row_ns < 0.0) per_row_ns = 0.0;
        double lat = us_a - rows_a * per_row_ns * 1e-3;
        if (lat < 1.0) lat = 1.0;
        // (the slope is quoted per row of 8192 four-byte cells, so that grids of other widths compare)
        words[0] = (int)(lat + 0.5);
        words[1] = (int)(per_row_ns * 8192.0 / ctx->dim_x + 0.5);
    }
    if (reduces_on_device(ctx)) {   // the maximum over the ranks: every rank must derive the same plan from it
        int *dev = ctx->d_collective;
        if (hipMemcpy(dev, words, sizeof words, hipMemcpyHostToDevice) != hipSuccess)
            (void)hipMemset(dev, 0x7f, sizeof words);   // (whatever is there is reduced; a huge value reads as a failure)
        int rr = t->allreduce_max(ctx, dev, 2, o.xstream);
        if (rr == SFL_OK && (hipStreamSynchronize(o.xstream) != hipSuccess ||
                             hipMemcpy(words, dev, sizeof words, hipMemcpyDeviceToHost) != hipSuccess))
            rr = fail(SFL_ERR_HIP, "exchange measurement: reduction over the ranks failed");
        SFL_TRY(rr);
    }
    if (rc != SFL_OK) {
        last_error() = why;
        return rc;
    }
    if (words[0] >= kFailed) return fail(SFL_ERR_STATE, "exchange measurement failed on another rank of the communicator");
    for (sfl_context *c : peers) {
        c->exchange_latency_us = words[0];
        c->exchange_ns_per_row = words[1];
    }
    return SFL_OK;
}

}  // namespace host
}  // namespace sfl

// ==========================================================================================
// C ABI: communicators, groups
// ==========================================================================================
using namespace sfl::host;

extern "C" {

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
                                     c->opt_advect_kernel, c->opt_fuse_divergence, c->opt_small_grid, c->opt_sor_arrival,
                                     c->opt_sor_fold, c->streams_concurrent};   // (streams_concurrent LAST: sfl_comm_check_options)
    memcpy(b, v, sizeof v);
}

int sfl_comm_check_options(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    Rccl *t = rccl_of(c);
    if (!t) return fail(SFL_ERR_STATE, "no communicator attached");
    SFL_TRY(use_device(c));
    Overlap o;
    SFL_TRY(overlap_of(c, &o));
    const int world = t->self ? 1 : c->nranks;
    int mine[kOptionBlockInts];
    option_block(c, mine);
    int *dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), sizeof(int) * kOptionBlockInts * (size_t)(world + 1)));
    std::vector<int> all((size_t)kOptionBlockInts * world);
    int rc = SFL_OK;
    do {  // (single exit: the scratch buffer is freed on every path)
        if (hipMemcpy(dev, mine, sizeof mine, hipMemcpyHostToDevice) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = fail(SFL_ERR_HIP, "option block upload failed");
            break;
        }
        const ncclResult_t r = ncclAllGather(dev, dev + kOptionBlockInts, kOptionBlockInts, ncclInt32, t->comm, o.xstream);
        if (r != ncclSuccess) {
            rc = fail(SFL_ERR_RCCL, "ncclAllGather of the option block failed: %s", ncclGetErrorString(r));
            break;
        }
        if (hipStreamSynchronize(o.xstream) != hipSuccess ||
            hipMemcpy(all.data(), dev + kOptionBlockInts, all.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(SFL_ERR_HIP, "option block download failed");
            break;
        }
        for (int r2 = 0; r2 < world && rc == SFL_OK; ++r2)
            for (int k = 0; k < kOptionBlockInts - 1; ++k)
                if (all[(size_t)r2 * kOptionBlockInts + k] != mine[k]) {
                    rc = fail(SFL_ERR_STATE, "rank %d and rank %d disagree on option-block word %d (%d vs %d): every rank "
                              "of a communicator must be created for the same domain and carry the same options",
                              c->rank, r2, k, mine[k], all[(size_t)r2 * kOptionBlockInts + k]);
                    break;
                }
        // the last word is not an option but a finding (do this rank's two streams run side by side?): exchanges are counted
        // on the device only where they do on EVERY rank -- the ranks must walk the same plan
        for (int r2 = 0; r2 < world && rc == SFL_OK; ++r2)
            if (all[(size_t)r2 * kOptionBlockInts + kOptionBlockInts - 1] == 0) c->streams_concurrent = 0;
    } while (false);
    (void)hipFree(dev);
    if (rc == SFL_OK) c->options_dirty = false;
    return rc;
}

static int attach_rccl(sfl_context *c, const ncclUniqueId &uid, bool self)
{
    SFL_TRY(use_device(c));
    Overlap o;
    SFL_TRY(overlap_of(c, &o));
    auto t = std::make_shared<Rccl>();
    t->self = self;
    t->xstream = o.xstream;
    NCCL_TRY(ncclCommInitRank(&t->comm, self ? 1 : c->nranks, uid, self ? 0 : c->rank));
    c->transport = t;
    // the ranks are separate processes: a rank created for another domain or with other options would run a
    // different program (mismatched sends / receives: a hang or silently wrong halos) -- refuse it here, and
    // do not stay attached to a group this rank does not fit
    bool side_by_side = false;
    int rc = streams_run_concurrently(c, &side_by_side);
    if (rc == SFL_OK) rc = sfl_comm_check_options(c);
    if (rc == SFL_OK) rc = measure_exchange(c);   // (collective: every rank exchanges with its neighbours; after the option check)
    if (rc != SFL_OK) {
        const std::string why = last_error();
        c->transport.reset();
        last_error() = why;
    }
    return rc;
}

int sfl_comm_attach(sfl_context *c, const void *id, size_t id_bytes)
{
    if (!c || !id || id_bytes < sizeof(ncclUniqueId)) return fail(SFL_ERR_INVALID, "bad arguments");
    if (c->transport) return fail(SFL_ERR_STATE, "context already has a transport");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    return attach_rccl(c, uid, false);
}

int sfl_comm_emulate(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (c->transport) return fail(SFL_ERR_STATE, "context already has a transport");
    if (c->nranks < 2) return fail(SFL_ERR_STATE, "a whole-domain context has nothing to exchange");
    Overlap o;   // the exchange stream exists from here on, not from the first solve on (see sfl_context::xstream)
    SFL_TRY(overlap_of(c, &o));
    c->transport = std::make_shared<Emulated>();
    return SFL_OK;
}

int sfl_comm_emulate_rccl(sfl_context *c)
{
    if (!c) return fail(SFL_ERR_INVALID, "ctx is NULL");
    if (c->transport) return fail(SFL_ERR_STATE, "context already has a transport");
    if (c->nranks < 2) return fail(SFL_ERR_STATE, "a whole-domain context has nothing to exchange");
    ncclUniqueId uid;
    NCCL_TRY(ncclGetUniqueId(&uid));
    return attach_rccl(c, uid, true);
}

int sfl_comm_loopback(sfl_context *c, int rows)
{
    if (!c || rows < 1 || rows > c->g1 - c->g0) return fail(SFL_ERR_INVALID, "bad loopback request");
    Rccl *t = rccl_of(c);
    if (!t) return fail(SFL_ERR_STATE, "no communicator attached");
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
    NCCL_TRY(ncclSend(c->div + off, bytes, ncclChar, t->peer(c->rank), t->comm, o.xstream));
    NCCL_TRY(ncclRecv(c->p + off, bytes, ncclChar, t->peer(c->rank), t->comm, o.xstream));
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
            c->dim_x != ctxs[0]->dim_x || c->gdim_y != ctxs[0]->gdim_y || c->transport)
            return fail(SFL_ERR_INVALID, "ctxs[%d] is not slab %d of %d on the group's device", r, r, n);
    }
    auto g = std::make_shared<Group>();
    g->members.assign(ctxs, ctxs + n);
    HIP_TRY(hipSetDevice(ctxs[0]->device));
    // The members' own streams go FIRST: the runtime folds its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
    // default), and with the members' streams still alive the group's compute and exchange stream of the first group of a
    // process landed on ONE queue (found by streams_run_concurrently: that group fell back to exchanges behind events).
    for (int r = 0; r < n; ++r) {
        (void)hipStreamSynchronize(ctxs[r]->stream);
        (void)hipStreamDestroy(ctxs[r]->stream);
        ctxs[r]->stream = nullptr;
    }
    // the group's streams, one behind the other: the runtime deals streams to its hardware queues in turn
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g->xstream, hipStreamNonBlocking));
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
        c->opt_halo_timeout_ms = z->opt_halo_timeout_ms;
        c->opt_sor_fold = z->opt_sor_fold;
    }
    for (int r = 0; r < n; ++r) {  // one stream orders the whole group
        sfl_context *c = ctxs[r];
        c->stream = g->stream;
        c->owns_stream = false;
        c->transport = g;
        c->group = g.get();
    }
    return SFL_OK;
}

}  // extern "C"
