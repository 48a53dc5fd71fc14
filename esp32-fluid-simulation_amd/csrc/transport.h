// transport.h -- how halo rows travel between neighbouring row slabs, behind ONE interface, and the exchange
// protocol the executors build on it.  Three transports:
//
//   Group      virtual ranks: the slabs of one domain live on ONE device and are driven by one host thread; a halo
//              message is a device copy between two contexts' arrays (sfl_group_link; what the one-GPU tests run)
//   Rccl       one process per GPU: ncclSend / ncclRecv with the two neighbouring ranks inside one ncclGroup, every
//              operation of the communicator on the context's exchange stream (sfl_comm_attach; production).
//              The same class with `self` set sends every message to this rank ITSELF over a one-rank communicator
//              (sfl_comm_emulate_rccl): one rank's program on one GPU with RCCL's own kernels as the transport
//   Emulated   one rank's program alone, every message a self-copy of the same size (sfl_comm_emulate; timing only)
//
// The loop being sharded is poisson.cpp:121-124 (iters x two colour passes over the whole domain).
#pragma once
#include "context.h"

namespace sfl {
namespace host {

// The rows one halo message carries, per side: each rank sends its owned rows at depth [skip, skip + rows) from a cut
// and receives the neighbour's into the ghost rows at the same depth.
struct HaloBands {
    int field;
    int rows;
    int skip;
};

class Transport {
public:
    virtual ~Transport() {}
    virtual int kind() const = 0;   // the value SFL_OPT_TRANSPORT reads
    // Issue the message(s) of one exchange for every context of `peers` (peers_of: one context, or all virtual ranks)
    // to stream `on`.  Nothing else: what orders the message against the launches is the caller's business.
    virtual int move(const std::vector<sfl_context *> &peers, const HaloBands &b, hipStream_t on) = 0;
    // Element-wise maximum of `n` device ints over the ranks, in place, on stream `on`.  Transports whose ranks share a
    // host thread (or have no peers) return without doing anything: the caller takes the maximum over peers_of on the host.
    virtual int allreduce_max(sfl_context *c, int *dev_words, int n, hipStream_t on) { return SFL_OK; }
    // The owned rows of every rank's `field`, assembled in each context's gather_buf (issued to `on`).
    virtual int gather(const std::vector<sfl_context *> &peers, int field, hipStream_t on) = 0;
    // Does an exchange issued in line with the compute stream's work still travel on the exchange stream?  (RCCL: every
    // operation of a communicator goes to one stream; the emulated rank mimics that.  A group copies on its one stream.)
    virtual bool own_stream() const { return true; }
    // May the executor count halo arrivals on the device (SFL_OPT_EXCHANGE_SCHEDULE automatic)?  Launches then wait for a
    // message INSIDE the kernel, which is only safe where the peer that sends it cannot be arbitrarily late and where the
    // exchange stream is known to run beside the compute stream.
    virtual bool arrival_by_default() const = 0;
    // ranks whose hosts are separate processes: options are compared collectively, waits must outlast a late peer
    virtual bool separate_processes() const { return false; }
    // default limit of a wait inside a launch, microseconds
    virtual int default_timeout_us() const { return 2000000; }
};

// In-process virtual ranks: slabs of one domain living on ONE device, ordered by one stream.
class Group : public Transport {
public:
    std::vector<sfl_context *> members;
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;  // in-process halo copies of a solve (see sfl_context::xstream)
    hipEvent_t ev_ready = nullptr, ev_arrived = nullptr;
    int streams_concurrent = -1;   // as sfl_context::streams_concurrent, for the group's pair of streams
    HaloTuner halo_tuner;          // as sfl_context::halo_tuner, for the group's solves
    ~Group() override;
    int kind() const override { return 2; }
    int move(const std::vector<sfl_context *> &peers, const HaloBands &b, hipStream_t on) override;
    int gather(const std::vector<sfl_context *> &peers, int field, hipStream_t on) override;
    bool own_stream() const override { return false; }
    bool arrival_by_default() const override { return true; }
};

// The exchange stream and its two events: the group's when the contexts are linked, the context's own otherwise.
struct Overlap {
    hipStream_t compute = nullptr, xstream = nullptr;
    hipEvent_t ready = nullptr, arrived = nullptr;
};
int overlap_of(sfl_context *c, Overlap *o);

// Every rank sends its `rows` lowest owned rows down and its `rows` highest owned rows up, and
// receives the neighbours' into the ghost rows adjacent to its owned block.
// `on` = stream to issue the transfers on (nullptr: the contexts' compute stream).
// `skip` > 0: only the rows at depth [skip, skip + rows) from the cuts travel (the ghost rows nearer the cut
// are still valid: early exchanges of slab_plan.cpp).
// `in_time` (run_poisson_in_time): the exchange is one step of the device-counted protocol -- it starts when the sender tiles
// of the launch in front of it have counted themselves (`wait_done`; the right-hand side's exchange starts behind an event
// instead) and ends by raising every receiver's arrival count, with a kernel behind the message's own.
int exchange(const std::vector<sfl_context *> &peers, int field, int rows, hipStream_t on = nullptr, int skip = 0,
             bool in_time = false, bool wait_done = false);
// Halo exchange off the compute stream: starts once everything issued so far on the compute stream has completed, runs
// on the exchange stream; `arrived` marks its completion (`mark` = false: the caller queues more work behind the
// exchange on the exchange stream and records `arrived` itself, mark_arrived).
int start_exchange(const std::vector<sfl_context *> &peers, const Overlap &o, int field, int rows, int skip = 0,
                   bool mark = true, bool in_time = false);
int mark_arrived(const std::vector<sfl_context *> &peers, const Overlap &o);
int await_exchange(const std::vector<sfl_context *> &peers, const Overlap &o);
// An exchange IN LINE with the compute stream's work.  `after` != nullptr: the exchanged rows were final when that event was
// recorded on the compute stream -- the exchange starts behind IT, not behind what has been queued on the compute stream since.
int exchange_inline(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field, int rows, int skip = 0,
                    hipEvent_t after = nullptr);
// Gather the whole `field` (owned rows of every slab) into each context's gather_buf, in line with the compute stream.
int gather_field(sfl_context *ctx, const std::vector<sfl_context *> &peers, int field);
// Maximum over the ranks of `n` device words of `ctx` (in line with the compute stream; nothing where the ranks share a host).
int reduce_max_inline(sfl_context *ctx, int *dev_words, int n);
// ... on the exchange stream behind what the compute stream holds so far, followed there by a copy of the words to pinned
// host memory and `done`; false where the transport has nothing to reduce (the caller copies on the compute stream).
bool reduces_on_device(const sfl_context *ctx);
int reduce_max_then_copy(sfl_context *ctx, int *dev_words, int n, int *host_words, hipEvent_t done);
// Do the compute and the exchange stream of this context run side by side?  (A launch that waits for a message inside the
// kernel would otherwise sit in front of the kernels that deliver it.)  Measured once per context / group, cached.
int streams_run_concurrently(sfl_context *ctx, bool *yes);
// What does one halo exchange of this context's transport cost?  Times back-to-back exchanges of p at two depths on the
// exchange stream (the protocol a solve uses, nothing else running), fits latency + per-row cost, takes the maximum over
// the ranks and stores it in every context of peers_of(ctx) (exchange_latency_us, exchange_ns_per_row).  Collective on
// RCCL ranks: called at attach there, before the first solve elsewhere.
int measure_exchange(sfl_context *ctx);

}  // namespace host
}  // namespace sfl
