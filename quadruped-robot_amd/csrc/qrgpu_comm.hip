// ============================================================================
// libqrgpu.so, multi-GPU: the one exchange of the path -- the all-gather of per-robot torques over RCCL / xGMI
//  -- behind the C ABI, so that a C++ caller needs neither torch nor its own collective code.
//
// Robots are independent: rank r owns a contiguous shard and there is no exchange inside a tick.  After a tick every
// rank holds tau[12][n_local]; one ncclAllGather leaves tau_all[nranks][12][n_local] on every rank.  48 KB per rank at 1024
// robots: latency bound, so it runs on a stream of its own, behind an event of the compute stream, while the next tick computes;
// the compute stream only waits (qrgpu_allgather_fence) for the gather that still reads the buffer it is about to overwrite.
//
// librccl is opened at first use (dlopen): single-GPU users of libqrgpu.so never load it.
// ============================================================================
#include <dlfcn.h>
#include <cstring>
#include <cstdlib>
#include <rccl/rccl.h>

#include "qrgpu_ctx.h"

namespace qrgpu {
__global__ void qr_gate_kernel(int *counter, int expected_total, long long max_ticks, int *timed_out, int timed_out_value, int *bump);   // qr_mpc_kernel.hip
}

namespace {

// How the streams of a context learn of a gather.  Two forms:
//   events (stream-ordered RCCL, nothing else): the gather waits for an event of the compute stream, the fence makes the compute stream wait for an
//          event of the communication stream.  The DEFAULT whenever the communicator has more than one rank: no run on several GPUs has ever been
//          available to the builder, and the first one must not depend on anything cleverer than stream order.
//   polls: one-thread launches poll counts (the tick's join bumps one for the gather, a launch behind the gather bumps one for the fence) -- a wait
//          for an event of another stream costs the waiting stream 5-8 us on this pool even when the event has long happened, 4 % of a tick
//          (4.47 against 4.25 M ticks/s with a one-rank communicator on one GPU).  The default for a ONE-rank communicator (where it is tested:
//          tests/test_gpu_comm.py, bench.py QRGPU_BENCH_FORCE_COMM=1) and opt-in for more: QRGPU_COMM_EVENTS=0.  (QRGPU_COMM_EVENTS=1: events always.)
bool comm_polls(const qrgpu_ctx *c)
{
    static const int ev = [] { const char *e = getenv("QRGPU_COMM_EVENTS"); return e ? atoi(e) : -1; }();
    if (ev >= 0) return ev == 0;
    return !(c && c->comm_nranks > 1);
}

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl R;
    if (R.handle || !R.err.empty()) return &R;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.handle) break;
    }
    if (!R.handle) { R.err = std::string("dlopen librccl: ") + dlerror(); return &R; }
    auto sym = [&](const char *s) { void *p = dlsym(R.handle, s); if (!p && R.err.empty()) R.err = std::string("librccl lacks ") + s; return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
    R.CommCount = (decltype(R.CommCount))sym("ncclCommCount");
    R.CommUserRank = (decltype(R.CommUserRank))sym("ncclCommUserRank");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    return &R;
}

#define NCCLCHK(ctx, R, call)                                                                         \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + ((R)->GetErrorString ? (R)->GetErrorString(r_) : "rccl error"); \
            return QRGPU_ERR_COMM;                                                                    \
        }                                                                                             \
    } while (0)

int ensure_comm_stream(qrgpu_ctx *c)
{
    if (c->comm_stream) return QRGPU_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_tick, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_gather[i], hipEventDisableTiming));
    if (!c->d_gather_done) {
        HIPCHK(c, hipMalloc(&c->d_gather_done, 2 * sizeof(int)));
        HIPCHK(c, hipMemset(c->d_gather_done, 0, 2 * sizeof(int)));
        HIPCHK(c, hipDeviceSynchronize());
        c->gather_total[0] = c->gather_total[1] = 0;
    }
    return QRGPU_OK;
}

}  // namespace

extern "C" {

int qrgpu_comm_unique_id(unsigned char id[QRGPU_COMM_ID_BYTES])
{
    static_assert(QRGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id blob size");
    if (!id) return QRGPU_ERR_BAD_ARG;
    Rccl *R = rccl();
    if (!R->err.empty()) return QRGPU_ERR_COMM;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return QRGPU_ERR_COMM;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return QRGPU_OK;
}

int qrgpu_comm_init_rank(qrgpu_ctx *c, const unsigned char id[QRGPU_COMM_ID_BYTES], int nranks, int rank)
{
    if (!c || !id || nranks <= 0 || rank < 0 || rank >= nranks) return QRGPU_ERR_BAD_ARG;
    if (c->comm) return QRGPU_ERR_BAD_ARG;            // one communicator per context
    Rccl *R = rccl();
    if (!R->err.empty()) { c->err = R->err; return QRGPU_ERR_COMM; }
    int rc = ensure_comm_stream(c);
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, R, R->CommInitRank(&comm, nranks, u, rank));
    c->comm = comm; c->comm_owned = true; c->comm_nranks = nranks; c->comm_rank = rank;
    return QRGPU_OK;
}

int qrgpu_comm_destroy(qrgpu_ctx *c)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    if (c->comm && c->comm_owned) { Rccl *R = rccl(); if (R->CommDestroy) R->CommDestroy((ncclComm_t)c->comm); }
    c->comm = nullptr; c->comm_owned = false; c->comm_nranks = 0; c->comm_rank = 0;
    if (c->comm_stream) {
        hipSetDevice(c->device);
        hipEventDestroy(c->ev_tick);
        for (int i = 0; i < 2; ++i) hipEventDestroy(c->ev_gather[i]);
        hipStreamDestroy(c->comm_stream);
        c->comm_stream = nullptr; c->ev_tick = nullptr; c->ev_gather[0] = c->ev_gather[1] = nullptr;
        c->ev_gather_pending[0] = c->ev_gather_pending[1] = false;
    }
    return QRGPU_OK;
}

int qrgpu_comm_info(const qrgpu_ctx *c, int *nranks, int *rank)
{
    if (!c || !c->comm) return QRGPU_ERR_NOT_SETUP;
    if (nranks) *nranks = c->comm_nranks;
    if (rank) *rank = c->comm_rank;
    return QRGPU_OK;
}

static int allgather_tau(qrgpu_ctx *c, void *nccl_comm, const float *d_tau, int n_local, float *d_tau_all, int slot, bool after_tick)
{
    if (!c || !d_tau || !d_tau_all || n_local <= 0 || slot < 0 || slot > 1) return QRGPU_ERR_BAD_ARG;
    ncclComm_t comm = nccl_comm ? (ncclComm_t)nccl_comm : (ncclComm_t)c->comm;
    if (!comm) return QRGPU_ERR_NOT_SETUP;
    Rccl *R = rccl();
    if (!R->err.empty()) { c->err = R->err; return QRGPU_ERR_COMM; }
    int rc = ensure_comm_stream(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    // the gather starts when everything queued on the compute stream so far (this tick's torques) is complete ...
    if (after_tick) {
        // ... of the context's last pipelined tick: a one-thread launch on the communication stream polls the count that tick's join bumps -- no
        // event on the compute stream (recording one there costs that stream 7 us a tick on this pool).  Bounded (30 s), and a gate that gives up
        // says so through the word qrgpu_sync looks at: a gather of torques that are not there yet must not pass silently.
        hipLaunchKernelGGL(qrgpu::qr_gate_kernel, dim3(1), dim3(64), 0, c->comm_stream, c->d_tick_done, c->tick_done_total, (long long)3000000000LL, c->lane[0].d_pre_hint + 2, 1,
                           (int *)nullptr);
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipEventRecord(c->ev_tick, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_tick, 0));
    }
    NCCLCHK(c, R, R->AllGather(d_tau, d_tau_all, (size_t)12 * (size_t)n_local, ncclFloat, comm, c->comm_stream));
    // ... and whoever overwrites d_tau (buffer `slot`) later fences on this event
    HIPCHK(c, hipEventRecord(c->ev_gather[slot], c->comm_stream));
    if (comm_polls(c)) {
        // (one thread behind the gather on its stream: bumps the slot's count -- and, the count being what it then expects, leaves at once)
        ++c->gather_total[slot];
        hipLaunchKernelGGL(qrgpu::qr_gate_kernel, dim3(1), dim3(64), 0, c->comm_stream, c->d_gather_done + slot, c->gather_total[slot], (long long)0, (int *)nullptr, 0,
                           c->d_gather_done + slot);
        HIPCHK(c, hipGetLastError());
    }
    c->ev_gather_pending[slot] = true;
    return QRGPU_OK;
}

int qrgpu_allgather_tau(qrgpu_ctx *c, void *nccl_comm, const float *d_tau, int n_local, float *d_tau_all, int slot)
{
    return allgather_tau(c, nccl_comm, d_tau, n_local, d_tau_all, slot, false);
}

int qrgpu_allgather_tau_of_tick(qrgpu_ctx *c, void *nccl_comm, const float *d_tau, int n_local, float *d_tau_all, int slot)
{
    // the gather of the torques of the context's LAST qrgpu_tick_batch: when that was a pipelined tick the gather waits for that tick, not for the stream
    return allgather_tau(c, nccl_comm, d_tau, n_local, d_tau_all, slot, c && c->last_tick_piped && comm_polls(c) && c->d_tick_done != nullptr);
}

int qrgpu_allgather_fence(qrgpu_ctx *c, int slot)
{
    if (!c || slot < 0 || slot > 1) return QRGPU_ERR_BAD_ARG;
    if (!c->comm_stream || !c->ev_gather_pending[slot]) return QRGPU_OK;
    // (overlapped ticks run on streams of their own: the next one must wait for this gather too before it overwrites the buffer -- qrgpu_tick_batch)
    if (c->overlap) c->ov_fence_slots |= 1 << slot;
    // (A wait for an event of another stream costs the waiting stream several microseconds even when the event has long happened -- 8 us between
    //  two launches on this pool, against 2 without.  The gather of two steps ago normally HAS happened by the time the host queues this step:
    //  ask first, and put the wait on the stream only when it is still running.  hipErrorNotReady is not an error here.)
    if (comm_polls(c) && c->gather_joined[slot] == c->gather_total[slot]) {     // a pipelined tick's join has waited for this gather already
        c->ev_gather_pending[slot] = false;
        return QRGPU_OK;
    }
    const hipError_t q = hipEventQuery(c->ev_gather[slot]);
    if (q != hipSuccess) {
        (void)hipGetLastError();
        if (comm_polls(c)) {
            // bounded (30 s: another rank may be late with its side of the collective, and the first gather also sets up RCCL's connections; beyond
            // that it is a hung collective, the stream goes on and qrgpu_sync reports it)
            hipLaunchKernelGGL(qrgpu::qr_gate_kernel, dim3(1), dim3(64), 0, c->stream, c->d_gather_done + slot, c->gather_total[slot], (long long)3000000000LL,
                               c->lane[0].d_pre_hint + 2, 1, (int *)nullptr);
            HIPCHK(c, hipGetLastError());
        } else HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gather[slot], 0));
    }
    c->ev_gather_pending[slot] = false;
    return QRGPU_OK;
}

int qrgpu_allgather_wait(qrgpu_ctx *c, int slot)
{
    // device-side wait for a CONSUMER of d_tau_all queued on the compute stream: no host block, and the pending flag stays (the fence
    // in front of the next overwrite of the source buffer is still due)
    if (!c || slot < 0 || slot > 1) return QRGPU_ERR_BAD_ARG;
    if (!c->comm_stream || !c->ev_gather_pending[slot]) return QRGPU_OK;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gather[slot], 0));
    return QRGPU_OK;
}

int qrgpu_comm_sync(qrgpu_ctx *c)
{
    if (!c) return QRGPU_ERR_BAD_ARG;
    if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    return QRGPU_OK;
}

}  // extern "C"
