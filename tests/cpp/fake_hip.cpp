// fake_hip.cpp -- TEST HARNESS ONLY (tests/cpp, `make tsan_host` / `make san_host_fake`): a HIP runtime and an RCCL that live entirely
// on the host, so that the HOST side of the library -- contexts, options, virtual-rank groups, the three exchange executors, the
// one-rank RCCL transport, the host-pointer drop-ins with their per-thread context cache -- can RUN on a box without a GPU, under
// ThreadSanitizer or AddressSanitizer.  "Device" memory is malloc'ed host memory, streams and events are tokens, every call
// completes before it returns; the kernels are no-op stubs (launch_stubs_ok.cpp), so the fields hold garbage: what runs is the
// host logic, not the arithmetic.  Never part of the product; nothing here is reachable from libsfl_hip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
std::atomic<long> g_live_allocations{0};
struct Pending {
    void *recv;
    const void *send;
    size_t bytes;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_sends, g_recvs;
size_t nccl_bytes(size_t count, ncclDataType_t t) { return count * (t == ncclChar || t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : 4); }
void flush_group()
{
    // a one-rank communicator: the k-th receive of a group is fed by its k-th send (the rank talks to itself)
    for (size_t k = 0; k < g_recvs.size() && k < g_sends.size(); ++k) memmove(g_recvs[k].recv, g_sends[k].send, g_recvs[k].bytes);
    g_sends.clear();
    g_recvs.clear();
}
}  // namespace

extern "C" long fake_hip_live_allocations() { return g_live_allocations.load(); }

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof *p);
    strcpy(p->name, "fake gfx950 (host memory)");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) ++g_live_allocations; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { if (p) --g_live_allocations; free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    *e = reinterpret_cast<hipEvent_t>(malloc(sizeof(double)));
    *reinterpret_cast<double *>(*e) = 0.0;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    *reinterpret_cast<double *>(e) = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = (float)(*reinterpret_cast<double *>(b) - *reinterpret_cast<double *>(a)) + 0.001f;
    return hipSuccess;
}

// ---- RCCL: a communicator of ONE rank (what sfl_comm_emulate_rccl builds) ----
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId, int rank)
{
    if (n != 1 || rank != 0) return ncclInvalidArgument;   // (nobody else to talk to on the host)
    *c = reinterpret_cast<ncclComm_t>(malloc(8));
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }
ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_group_depth == 0) flush_group(); return ncclSuccess; }
ncclResult_t ncclSend(const void *b, size_t count, ncclDataType_t t, int, ncclComm_t, hipStream_t)
{
    g_sends.push_back({nullptr, b, nccl_bytes(count, t)});
    if (g_group_depth == 0) flush_group();
    return ncclSuccess;
}
ncclResult_t ncclRecv(void *b, size_t count, ncclDataType_t t, int, ncclComm_t, hipStream_t)
{
    g_recvs.push_back({b, nullptr, nccl_bytes(count, t)});
    if (g_group_depth == 0) flush_group();
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *s, void *r, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t)
{
    if (s != r) memmove(r, s, nccl_bytes(count, t));
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *s, void *r, size_t count, ncclDataType_t t, ncclComm_t, hipStream_t)
{
    if (s != r) memmove(r, s, nccl_bytes(count, t));
    return ncclSuccess;
}
}
