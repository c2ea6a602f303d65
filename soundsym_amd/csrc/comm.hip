// comm.hip -- the source-sharded match with its collectives inside the library (RCCL over xGMI).
//
// Role on the path: the reference's loop over targets (SoundSequence::clone_from_dictionary,
// src/sound.rs:451-455, one SoundDictionary::at_distance per target, :351-370) for a dictionary whose
// entries are split over the GPUs of one node.  Every rank searches its shard; two small collectives
// make the answer global: an all-reduce(MIN) of the per-target bounds between filter and selection
// (so that a rank without a target's neighbour re-scores nothing for it) and an all-gather of every
// rank's per-target (cost, global index), merged by the first-minimum rule (smallest key, lowest
// index on ties -- the rule of src/sound.rs:361-367, preserved because shards are ordered).
// Everything is enqueued on the context's stream; the host synchronises once per step.
//
// RCCL is bound at run time: a process that already carries it (a PyTorch process does) is used as
// it is, otherwise librccl.so.1 is opened.  No RCCL, no sharded match -- there is no substitute path.
#include "ssym_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

using namespace ssym;

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

void *find_symbol(void *&handle, const char *name)
{
    void *p = dlsym(RTLD_DEFAULT, name);
    if (p)
        return p;
    if (!handle) {
        const char *env = getenv("SSYM_RCCL_LIB");
        const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *c : cands) {
            if (!c || !*c)
                continue;
            handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
            if (handle)
                break;
        }
    }
    return handle ? dlsym(handle, name) : nullptr;
}

Rccl load_rccl()
{
    Rccl r;
    void *h = nullptr;
#define SSYM_BIND(field, sym)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(find_symbol(h, sym));             \
    if (!r.field) {                                                                  \
        r.why = std::string("RCCL symbol ") + sym + " not found (librccl.so.1 / $SSYM_RCCL_LIB)"; \
        return r;                                                                    \
    }
    SSYM_BIND(GetUniqueId, "ncclGetUniqueId")
    SSYM_BIND(CommInitRank, "ncclCommInitRank")
    SSYM_BIND(CommDestroy, "ncclCommDestroy")
    SSYM_BIND(CommAbort, "ncclCommAbort")
    SSYM_BIND(CommGetAsyncError, "ncclCommGetAsyncError")
    SSYM_BIND(AllReduce, "ncclAllReduce")
    SSYM_BIND(AllGather, "ncclAllGather")
    SSYM_BIND(GetErrorString, "ncclGetErrorString")
#undef SSYM_BIND
    r.ok = true;
    return r;
}

const Rccl &rccl()
{
    static const Rccl r = load_rccl();
    return r;
}

}  // namespace

// The ranks of ONE process (a thread per rank, each with its own context -- on one GPU or several): the exchange
// without RCCL.  RCCL refuses two ranks on one device, so this is how the multi-rank logic of ssym_match_sharded
// (block layout and strides of the gather, merge over G shards, the overflow repeat agreed through the gathered status,
// an empty shard among full ones) runs on a one-GPU test box; it is not stream-ordered (host barriers around device
// copies) and not what the multi-GPU bench uses.
struct ssym_local_group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    bool aborted = false;               // a rank gave up: every barrier, now and later, returns false at once
    std::vector<const void *> slot;     // per rank: the buffer it offers in the current collective
    // false: the group was aborted, or a rank did not arrive within the deadline (which aborts it)
    bool barrier(int64_t timeout_ms)
    {
        std::unique_lock<std::mutex> lk(m);
        if (aborted)
            return false;
        const unsigned long long g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return gen != g || aborted; });
        if (ok && gen != g)
            return true;            // (a barrier that completed counts even if somebody aborted right after it)
        aborted = true;
        cv.notify_all();
        return false;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

// words of list status every rank's gathered block ends with: entries list 1 wanted, "it did not fit", the rank's
// status code (0 = fine, else -SSYM_E_*) and the phase its local work failed in
constexpr int kStatusWords = 4;

struct ssym_comm {
    ncclComm_t nccl = nullptr;
    ssym_local_group *local = nullptr;   // non-NULL: the in-process transport above instead of RCCL
    DeviceBuf local_tmp;                 // in-process all-reduce: every rank's values side by side
    int rank = 0, world = 1;
    DeviceBuf bounds;      // M f64: per-target bounds (and, before them, the candidates' costs of a pruned step)
    DeviceBuf cand;        // M f64: candidates' costs
    DeviceBuf send, recv;  // one block per rank: [M f64 cost][Mpad u32 index][u32 wanted, u32 overflow]
    uint32_t *status_host = nullptr;    // pinned: world x kStatusWords + {list-2 count, 0} + the 8 give-up counters of the exact kernel
    bool dead = false;                  // aborted: every further call answers SSYM_E_COMM
    int64_t timeout_ms = 60000;         // deadline of one step (ssym_comm_set_timeout, $SSYM_COMM_TIMEOUT_MS)
    int fault_phase = 0, fault_kind = 0;   // ssym_comm_inject_fault, one shot
    const double *replay_bounds = nullptr; // ssym_comm_replay_bounds: joins every bound exchange as one more rank's offer
    uint32_t replay_n = 0;
    hipEvent_t ev[8]{};    // 0-1 all-reduce of the candidates' costs, 2-3 of the bounds, 4 step start, 5-6 all-gather, 7 step end
};

#define SSYM_NCCL_CHECK(ctx, call)                                                        \
    do {                                                                                  \
        ncclResult_t r__ = (call);                                                        \
        if (r__ != ncclSuccess) {                                                         \
            (ctx)->err = std::string(#call) + ": " + rccl().GetErrorString(r__);          \
            return SSYM_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

__global__ void comm_fill_block_kernel(double *cost, uint32_t *idx, uint32_t *status, double v, uint32_t base, uint32_t n,
                                       uint32_t code, uint32_t phase)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        cost[i] = v;
        idx[i] = base;
    }
    if (i == 0) {
        status[0] = 0;
        status[1] = 0;
        status[2] = code;
        status[3] = phase;
    }
}

__global__ void comm_fill_f64_kernel(double *p, double v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = v;
}

__global__ void comm_status_kernel(const uint32_t *__restrict__ hdr1, uint32_t *__restrict__ status)
{
    status[0] = hdr1 ? hdr1[0] : 0u;     // entries list 1 wanted
    status[1] = hdr1 ? hdr1[1] : 0u;     // it did not fit
    status[2] = 0u;                      // this rank's local work went through
    status[3] = 0u;
}

static float ev_ms2(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f;
}

// out[i] = min(out[i], other[i]): the replayed bounds of a larger world join the exchange (ssym_comm_replay_bounds)
__global__ void comm_min_with_kernel(double *__restrict__ out, const double *__restrict__ other, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = fmin(out[i], other[i]);
}

__global__ void comm_min_rows_kernel(const double *__restrict__ rows, int nRows, uint32_t n, double *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double m = rows[i];
    for (int r = 1; r < nRows; ++r)
        m = fmin(m, rows[(size_t)r * n + i]);      // (bounds and costs are never NaN: fmin is a plain minimum here)
    out[i] = m;
}

// Give the communicator up: RCCL's kernels of this rank leave their loops (ncclCommAbort), the ranks of an in-process
// group see the flag at their next barrier.  Peers over RCCL notice nothing by themselves -- they meet their deadline.
static void comm_abort(ssym_comm *c)
{
    if (c->dead)
        return;
    c->dead = true;
    if (c->local)
        c->local->abort();
    else if (c->nccl && rccl().ok) {
        (void)rccl().CommAbort(c->nccl);     // frees the communicator as ncclCommDestroy would
        c->nccl = nullptr;
    }
}

static int32_t local_timeout(ssym_ctx *ctx)
{
    ctx->err = "ssym_match_sharded: a rank of the in-process group did not arrive within the deadline (or gave the "
               "group up); the communicator is aborted";
    return SSYM_E_TIMEOUT;
}

// in-process all-gather: every rank copies every rank's block into its own receive buffer
static int32_t local_all_gather(ssym_ctx *ctx, ssym_comm *c, const void *send, void *recv, size_t bytes)
{
    ssym_local_group *g = c->local;
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));          // my block is complete
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->slot[c->rank] = send;
    }
    if (!g->barrier(c->timeout_ms))                                  // everybody's block is complete and published
        return local_timeout(ctx);
    for (int r = 0; r < c->world; ++r)
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync((char *)recv + (size_t)r * bytes, g->slot[r], bytes, hipMemcpyDeviceToDevice,
                                           ctx->stream));
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (!g->barrier(c->timeout_ms))                                  // everybody has read: the blocks may change again
        return local_timeout(ctx);
    return SSYM_OK;
}

static int32_t comm_all_gather(ssym_ctx *ctx, ssym_comm *c, const void *send, void *recv, size_t bytes)
{
    if (c->local)
        return local_all_gather(ctx, c, send, recv, bytes);
    SSYM_NCCL_CHECK(ctx, rccl().AllGather(send, recv, bytes, ncclUint8, c->nccl, ctx->stream));
    return SSYM_OK;
}

static int32_t comm_all_reduce_min(ssym_ctx *ctx, ssym_comm *c, double *buf, uint32_t n)
{
    if (c->local) {
        int32_t rc = ensure(ctx, c->local_tmp, sizeof(double) * (size_t)n * c->world);
        if (rc == SSYM_OK)
            rc = local_all_gather(ctx, c, buf, c->local_tmp.ptr, sizeof(double) * n);
        if (rc != SSYM_OK)
            return rc;
        comm_min_rows_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>((const double *)c->local_tmp.ptr, c->world, n, buf);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        return SSYM_OK;
    }
    SSYM_NCCL_CHECK(ctx, rccl().AllReduce(buf, buf, n, ncclFloat64, ncclMin, c->nccl, ctx->stream));
    return SSYM_OK;
}

// After an abort the stream should drain: the aborted collectives' kernels leave their loops.  That is what RCCL
// promises and what a world-1 communicator has been seen to do; with more ranks it is unverified (no multi-GPU box has
// run this), so the drain is bounded as well: the stream's tail event is polled for a short second deadline, and a
// stream that does not drain leaves the context's stream in an unknown state -- reported, not waited for.
static bool comm_drain_after_abort(ssym_ctx *ctx, int64_t limit_ms = 5000)
{
    hipEvent_t tail = nullptr;
    if (hipEventCreateWithFlags(&tail, hipEventDisableTiming) != hipSuccess)
        return false;
    bool drained = false;
    if (hipEventRecord(tail, ctx->stream) == hipSuccess) {
        using clock = std::chrono::steady_clock;
        const clock::time_point t0 = clock::now();
        for (;;) {
            const hipError_t e = hipEventQuery(tail);
            if (e == hipSuccess) {
                drained = true;
                break;
            }
            if (e != hipErrorNotReady || clock::now() - t0 > std::chrono::milliseconds(limit_ms))
                break;
            std::this_thread::yield();
        }
    }
    (void)hipEventDestroy(tail);
    if (!drained)
        ctx->err += "; the stream did not drain after the abort (its remaining work is abandoned: destroy the context)";
    return drained;
}

// The step's one synchronisation, under the communicator's deadline.  With RCCL the stream holds collectives that only
// finish when every rank has enqueued its part, so a peer that never arrives would keep hipStreamSynchronize forever:
// the end-of-step event is polled instead, RCCL's asynchronous error state with it; on expiry the communicator is
// aborted (its kernels then leave the stream) and the call fails with SSYM_E_TIMEOUT.  The in-process transport has
// nothing but local work on the stream at this point (its barriers carry the deadline).
static int32_t comm_wait_step(ssym_ctx *ctx, ssym_comm *c, hipEvent_t done)
{
    if (c->local) {
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return SSYM_OK;
    }
    using clock = std::chrono::steady_clock;
    const clock::time_point t0 = clock::now();
    const clock::duration limit = std::chrono::milliseconds(c->timeout_ms);
    for (unsigned spin = 1;; ++spin) {
        const hipError_t e = hipEventQuery(done);
        if (e == hipSuccess)
            return SSYM_OK;
        if (e != hipErrorNotReady) {
            ctx->err = std::string("ssym_match_sharded: hipEventQuery: ") + hipGetErrorString(e);
            return SSYM_E_HIP;
        }
        if ((spin & 63u) != 0)
            continue;
        const clock::duration waited = clock::now() - t0;
        ncclResult_t async = ncclSuccess;
        if (rccl().CommGetAsyncError(c->nccl, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            ctx->err = std::string("ssym_match_sharded: RCCL reported an asynchronous error: ") + rccl().GetErrorString(async) +
                       "; the communicator is aborted";
            comm_abort(c);
            (void)comm_drain_after_abort(ctx);
            return SSYM_E_COMM;
        }
        if (waited > limit) {
            ctx->err = "ssym_match_sharded: the step did not complete within " + std::to_string(c->timeout_ms) +
                       " ms (a rank failed to take part); the communicator is aborted";
            comm_abort(c);
            (void)comm_drain_after_abort(ctx);            // the aborted collectives leave the stream (bounded wait)
            return SSYM_E_TIMEOUT;
        }
        if (waited > std::chrono::milliseconds(2))
            std::this_thread::yield();                    // a healthy step is over long before; stop burning the core
    }
}

static const char *status_name(int32_t rc)
{
    switch (rc) {
    case SSYM_E_INVALID: return "SSYM_E_INVALID";
    case SSYM_E_EMPTY_DICT: return "SSYM_E_EMPTY_DICT";
    case SSYM_E_NO_DEVICE: return "SSYM_E_NO_DEVICE";
    case SSYM_E_HIP: return "SSYM_E_HIP";
    case SSYM_E_NOMEM: return "SSYM_E_NOMEM";
    case SSYM_E_UNSUPPORTED: return "SSYM_E_UNSUPPORTED";
    case SSYM_E_TIMEOUT: return "SSYM_E_TIMEOUT";
    case SSYM_E_COMM: return "SSYM_E_COMM";
    default: return "an unknown status";
    }
}

static int64_t default_timeout_ms()
{
    const char *e = getenv("SSYM_COMM_TIMEOUT_MS");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? v : 60000;
}

static int32_t comm_alloc_host(ssym_ctx *ctx, ssym_comm *c, const char *who)
{
    c->timeout_ms = default_timeout_ms();
    hipError_t e = hipHostMalloc((void **)&c->status_host, sizeof(uint32_t) * (kStatusWords * (size_t)c->world + 2 + 8),
                                 hipHostMallocDefault);
    for (auto &ev : c->ev)
        if (e == hipSuccess)
            e = hipEventCreate(&ev);
    if (e != hipSuccess) {
        ctx->err = std::string(who) + ": " + hipGetErrorString(e);
        return SSYM_E_HIP;
    }
    return SSYM_OK;
}

extern "C" {

int32_t ssym_comm_available(void)
{
    return guarded(nullptr, [&]() -> int32_t { return rccl().ok ? 1 : 0; });
}

int32_t ssym_comm_unique_id(void *out_id)
{
    if (!out_id)
        return SSYM_E_INVALID;
    return guarded(nullptr, [&]() -> int32_t {
    if (!rccl().ok)
        return SSYM_E_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess)
        return SSYM_E_HIP;
    static_assert(sizeof(id) == SSYM_COMM_ID_BYTES, "RCCL unique id size");
    memcpy(out_id, &id, sizeof(id));
    return SSYM_OK;
    });
}

int32_t ssym_comm_create(ssym_ctx *ctx, const void *id, int32_t rank, int32_t world, ssym_comm **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (!id || !out || world < 1 || rank < 0 || rank >= world) {
        ctx->err = "ssym_comm_create: bad arguments";
        return SSYM_E_INVALID;
    }
    *out = nullptr;
    if (!rccl().ok) {
        ctx->err = "ssym_comm_create: " + rccl().why;
        return SSYM_E_UNSUPPORTED;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ssym_comm *c = new (std::nothrow) ssym_comm();
    if (!c)
        return SSYM_E_NOMEM;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, uid, rank);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
        delete c;
        return SSYM_E_HIP;
    }
    const int32_t rc = comm_alloc_host(ctx, c, "ssym_comm_create");
    if (rc != SSYM_OK) {
        const std::string keep = ctx->err;
        ssym_comm_destroy(ctx, c);
        ctx->err = keep;
        return rc;
    }
    *out = c;
    return SSYM_OK;
    });
}

int32_t ssym_comm_destroy(ssym_ctx *ctx, ssym_comm *c)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!c)
        return SSYM_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (c->dead)
            (void)comm_drain_after_abort(ctx);   // (a dead communicator: never an unbounded wait behind its collectives)
        else
            (void)hipStreamSynchronize(ctx->stream);
    }
    if (c->nccl && rccl().ok)
        (void)rccl().CommDestroy(c->nccl);       // (an aborted communicator is gone already: nccl == NULL)
    for (DeviceBuf *b : {&c->bounds, &c->cand, &c->send, &c->recv, &c->local_tmp})
        if (b->ptr)
            (void)hipFree(b->ptr);
    if (c->status_host)
        (void)hipHostFree(c->status_host);
    for (auto &ev : c->ev)
        if (ev)
            (void)hipEventDestroy(ev);
    delete c;
    return SSYM_OK;
    });
}

int32_t ssym_comm_set_timeout(ssym_comm *comm, int64_t milliseconds)
{
    if (!comm || milliseconds <= 0)
        return SSYM_E_INVALID;
    comm->timeout_ms = milliseconds;
    return SSYM_OK;
}

int32_t ssym_comm_is_dead(const ssym_comm *comm)
{
    return comm ? (comm->dead ? 1 : 0) : SSYM_E_INVALID;
}

// The two hooks below exist for tests and measurements; a process has to ask for them (SSYM_TEST_HOOKS=1 in its
// environment when the call is made), otherwise they refuse.
static bool test_hooks_enabled()
{
    const char *e = getenv("SSYM_TEST_HOOKS");
    return e && atoi(e) != 0;
}

int32_t ssym_comm_replay_bounds(ssym_comm *comm, const double *bounds_dev, uint32_t n)
{
    if (!comm || (bounds_dev && n == 0))
        return SSYM_E_INVALID;
    if (!test_hooks_enabled())
        return SSYM_E_UNSUPPORTED;
    comm->replay_bounds = bounds_dev;
    comm->replay_n = bounds_dev ? n : 0;
    return SSYM_OK;
}

int32_t ssym_comm_inject_fault(ssym_comm *comm, int32_t phase, int32_t kind)
{
    if (!comm || phase < 0 || phase > 2 || kind < 0 || kind > 1)
        return SSYM_E_INVALID;
    if (!test_hooks_enabled())
        return SSYM_E_UNSUPPORTED;
    comm->fault_phase = phase;
    comm->fault_kind = kind;
    return SSYM_OK;
}

int32_t ssym_local_group_create(int32_t world, ssym_local_group **out)
{
    if (!out || world < 1)
        return SSYM_E_INVALID;
    *out = nullptr;
    return guarded(nullptr, [&]() -> int32_t {
        ssym_local_group *g = new ssym_local_group();
        g->world = world;
        g->slot.assign((size_t)world, nullptr);
        *out = g;
        return SSYM_OK;
    });
}

int32_t ssym_local_group_destroy(ssym_local_group *group)
{
    delete group;
    return SSYM_OK;
}

int32_t ssym_comm_create_local(ssym_ctx *ctx, ssym_local_group *group, int32_t rank, ssym_comm **out)
{
    if (!ctx)
        return SSYM_E_INVALID;
    return guarded(ctx, [&]() -> int32_t {
    if (!group || !out || rank < 0 || rank >= group->world) {
        ctx->err = "ssym_comm_create_local: bad arguments";
        return SSYM_E_INVALID;
    }
    *out = nullptr;
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ssym_comm *c = new ssym_comm();
    c->local = group;
    c->rank = rank;
    c->world = group->world;
    const int32_t rc = comm_alloc_host(ctx, c, "ssym_comm_create_local");
    if (rc != SSYM_OK) {
        const std::string keep = ctx->err;
        ssym_comm_destroy(ctx, c);
        ctx->err = keep;
        return rc;
    }
    *out = c;
    return SSYM_OK;
    });
}

}  // extern "C"

namespace {

struct StreamOnly {      // the phases below only enqueue; restored on every way out
    ssym_ctx *c;
    explicit StreamOnly(ssym_ctx *ctx) : c(ctx) { c->stream_only = true; c->so_cap = 0; c->so_hdr1 = c->so_hdr2 = nullptr; }
    ~StreamOnly() { c->stream_only = false; c->so_cap = 0; }
};

// One step.  *agreed = true on return means every rank of the communicator leaves the step with this same status
// (success, or the failure one of them reported through the gathered block) and the communicator is still in step;
// any other way out (*agreed false and a status != SSYM_OK) makes the caller abort the communicator.
int32_t match_sharded_step(ssym_ctx *ctx, ssym_comm *comm, const ssym_dict *dict, const ssym_queries *q,
                           const double *distance, uint32_t index_base, uint32_t *out_idx, double *out_cost,
                           uint32_t flags, bool *agreed)
{
    if (!dict || !q || !out_idx) {
        ctx->err = "ssym_match_sharded: NULL argument";
        return SSYM_E_INVALID;
    }
    const uint32_t M = q->set.n;
    ssym_timings tm{};
    tm.n_pairs = (uint64_t)dict->set.n * M;
    if (M == 0) {               // (the targets are the same on every rank: everybody returns here)
        ctx->timings = tm;
        *agreed = true;
        return SSYM_OK;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int G = comm->world;
    const bool outDev = (flags & SSYM_OUT_DEVICE) != 0;
    const bool refcos = ctx->metric == SSYM_METRIC_REFCOS;
    const bool emptyShard = dict->set.n == 0;
    const double foldStart = refcos ? 2.0 : (double)INFINITY;
    const int faultPhase = comm->fault_phase, faultKind = comm->fault_kind;
    comm->fault_phase = comm->fault_kind = 0;       // one shot

    // one block per rank: costs, indices (padded to an even count so that blocks stay 8-byte aligned), list status
    const size_t mPad = ((size_t)M + 1) & ~(size_t)1;
    const size_t blk = sizeof(double) * M + sizeof(uint32_t) * mPad + kStatusWords * sizeof(uint32_t);
    int32_t rc = ensure(ctx, comm->bounds, sizeof(double) * M);
    if (rc == SSYM_OK)
        rc = ensure(ctx, comm->cand, sizeof(double) * M);
    if (rc == SSYM_OK)
        rc = ensure(ctx, comm->send, blk);
    if (rc == SSYM_OK)
        rc = ensure(ctx, comm->recv, blk * G);
    if (rc != SSYM_OK)
        return rc;               // (without its exchange buffers a rank cannot take part: the caller aborts)
    double *bounds = (double *)comm->bounds.ptr;
    double *sendCost = (double *)comm->send.ptr;
    uint32_t *sendIdx = (uint32_t *)((char *)comm->send.ptr + sizeof(double) * M);
    uint32_t *sendStatus = sendIdx + mPad;
    uint32_t *userIdx = out_idx;
    double *userCost = out_cost;
    if (!outDev) {
        rc = ensure(ctx, ctx->out_idx, sizeof(uint32_t) * M);
        if (rc == SSYM_OK)
            rc = ensure(ctx, ctx->out_cost, sizeof(double) * M);
        if (rc != SSYM_OK)
            return rc;
        userIdx = (uint32_t *)ctx->out_idx.ptr;
        userCost = (double *)ctx->out_cost.ptr;
    }

    // refcos: the integer filter's records of the two sets are built HERE, on their first sharded step (it synchronises
    // once; inside the stream-only phases nothing may) -- where they cannot be built the step takes the f64 matrix pipe
    if (refcos && dict->set.n && dict->set.dim == q->set.dim)
        (void)refcos_q8_ready(ctx, dict->set, q->set);
    StageScope stageScope(ctx);
    StreamOnly streamOnly(ctx);
    hipEvent_t *cev = comm->ev;
    float coll_ms = 0.f;
    const bool wantPrune = ((flags & SSYM_DTW_PRUNE) || ctx->prune_default) && !distance && !refcos;

    // This rank's LOCAL work runs through here.  The first failure is remembered -- status, phase, message --, every
    // later piece of local work is skipped, and the collectives go on with neutral blocks: the status travels in the
    // gathered block and every rank returns it after the step's one synchronisation.
    int32_t failed = SSYM_OK;
    int failedPhase = 0;
    std::string failedMsg;
    auto local = [&](int phase, auto &&work) {
        if (failed != SSYM_OK)
            return;
        int32_t r;
        try {
            if (faultPhase == phase && faultKind == 0) {
                ctx->err = "injected failure (ssym_comm_inject_fault)";
                r = SSYM_E_NOMEM;
            } else {
                r = work();
            }
        } catch (const std::bad_alloc &) {
            ctx->err = "out of host memory";
            r = SSYM_E_NOMEM;
        } catch (...) {
            ctx->err = "unexpected C++ exception inside the library";
            r = SSYM_E_HIP;
        }
        if (r != SSYM_OK) {
            failed = r;
            failedPhase = phase;
            failedMsg = ctx->err;
        }
    };
    auto left_the_step = [&](int phase) {          // ssym_comm_inject_fault, kind 1
        if (faultPhase != phase || faultKind != 1)
            return false;
        ctx->err = "injected failure (ssym_comm_inject_fault): the rank leaves the step without its collectives";
        return true;
    };
    const uint32_t fillGrid = (M + 255) / 256;

    // ---- phase 1: candidates (pruned steps), filter, per-target bounds ------------------------------------
    if (dict->set.n && dict->set.dim != q->set.dim)
        local(1, [&]() -> int32_t {
            ctx->err = "dim mismatch between dictionary and targets";
            return SSYM_E_INVALID;
        });
    SSYM_HIP_CHECK(ctx, hipEventRecord(cev[4], st));
    if (left_the_step(1))
        return SSYM_E_HIP;
    bool pruned = false;
    if (wantPrune) {
        double *cand = (double *)comm->cand.ptr;
        if (!emptyShard)
            local(1, [&] { return match_candidates_impl(ctx, dict, q, cand); });
        if (emptyShard || failed != SSYM_OK) {
            comm_fill_f64_kernel<<<fillGrid, 256, 0, st>>>(cand, (double)INFINITY, M);
            SSYM_HIP_CHECK(ctx, hipGetLastError());
        }
        SSYM_HIP_CHECK(ctx, hipEventRecord(cev[0], st));
        rc = comm_all_reduce_min(ctx, comm, cand, M);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipEventRecord(cev[1], st));
        pruned = true;
    }
    if (!emptyShard)
        local(1, [&] {
            return match_begin_impl(ctx, dict, q, distance, index_base, bounds, pruned ? (const double *)comm->cand.ptr : nullptr);
        });
    if (emptyShard || failed != SSYM_OK) {
        comm_fill_f64_kernel<<<fillGrid, 256, 0, st>>>(bounds, (double)INFINITY, M);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
    }
    const bool filterPath = !emptyShard && failed == SSYM_OK && ctx->pending.filter;
    tm.used_filter = filterPath ? 1 : 0;
    tm.pruned = filterPath && ctx->pending.pruned ? 1 : 0;
    SSYM_HIP_CHECK(ctx, hipEventRecord(cev[2], st));
    rc = comm_all_reduce_min(ctx, comm, bounds, M);
    if (rc != SSYM_OK)
        return rc;
    if (comm->replay_bounds) {       // measurement hook: the bounds the ranks of a larger world would have agreed on
        if (comm->replay_n != M) {
            ctx->err = "ssym_match_sharded: ssym_comm_replay_bounds holds bounds for another number of targets";
            return SSYM_E_INVALID;
        }
        comm_min_with_kernel<<<(M + 255) / 256, 256, 0, st>>>(bounds, comm->replay_bounds, M);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
    }
    SSYM_HIP_CHECK(ctx, hipEventRecord(cev[3], st));

    // ---- phase 2 + exchange; repeated once by EVERY rank when any rank's candidate list overflowed -----------
    uint32_t *status = comm->status_host;
    uint32_t *ownCount = status + kStatusWords * G;          // {list-2 count, 0}
    uint32_t *gaveUp = ownCount + 2;                         // the exact kernel's 8 give-up counters
    // (the merge compares |cost - distance| for dtw; refcos shards report the key |sim - distance| itself, which the
    //  merge must compare as it is)
    const double *distDev = nullptr;
    if (distance && !refcos) {
        if (emptyShard || failed != SSYM_OK) {      // (otherwise phase 1 has uploaded them)
            rc = ensure(ctx, ctx->dist, sizeof(double) * M);
            if (rc == SSYM_OK)
                rc = stage_h2d(ctx, ctx->dist.ptr, distance, sizeof(double) * M);
            if (rc != SSYM_OK)
                return rc;
        }
        distDev = (const double *)ctx->dist.ptr;
    }
    float sel_ms = 0.f, ref_ms = 0.f, red_ms = 0.f;
    ssym_ctx::Pending keep = ctx->pending;          // (finish consumes it; a second attempt needs it again)
    for (int attempt = 0; attempt < 2; ++attempt) {
        tm.attempts = attempt + 1;
        if (left_the_step(2))
            return SSYM_E_HIP;
        bool finished = false;
        if (!emptyShard)
            local(2, [&]() -> int32_t {
                ctx->pending = keep;
                ctx->so_hdr1 = ctx->so_hdr2 = nullptr;
                ctx->so_filter = ctx->so_refcos = false;
                const int32_t r = match_finish_impl(ctx, bounds, sendIdx, sendCost, SSYM_OUT_DEVICE | (flags & SSYM_DTW_FORCE_EXACT));
                finished = r == SSYM_OK;
                return r;
            });
        if (finished) {
            comm_status_kernel<<<1, 1, 0, st>>>((ctx->so_filter || ctx->so_refcos) ? ctx->so_hdr1 : nullptr, sendStatus);
        } else {          // an empty shard reports the fold start; so does a rank whose local work failed, with its status
            comm_fill_block_kernel<<<fillGrid, 256, 0, st>>>(sendCost, sendIdx, sendStatus, foldStart, index_base, M,
                                                             (uint32_t)(-failed), (uint32_t)failedPhase);
        }
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        SSYM_HIP_CHECK(ctx, hipEventRecord(cev[5], st));
        rc = comm_all_gather(ctx, comm, comm->send.ptr, comm->recv.ptr, blk);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipEventRecord(cev[6], st));
        rc = launch_merge_shards(ctx, (uint32_t)G, M, (const double *)comm->recv.ptr,
                                 (const uint32_t *)((const char *)comm->recv.ptr + sizeof(double) * M), distDev, userIdx,
                                 userCost, blk / sizeof(double), blk / sizeof(uint32_t));
        if (rc != SSYM_OK)
            return rc;
        // every rank's status words, this rank's own list-2 count and the exact kernel's give-up counters
        SSYM_HIP_CHECK(ctx, hipMemcpy2DAsync(status, kStatusWords * sizeof(uint32_t),
                                             (const char *)comm->recv.ptr + blk - kStatusWords * sizeof(uint32_t), blk,
                                             kStatusWords * sizeof(uint32_t), (size_t)G, hipMemcpyDeviceToHost, st));
        ownCount[0] = ownCount[1] = 0;
        const bool ownLists = finished && (ctx->so_filter || ctx->so_refcos) && ctx->so_hdr2;
        if (ownLists)
            SSYM_HIP_CHECK(ctx, hipMemcpyAsync(ownCount, ctx->so_hdr2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        const unsigned pipeMask = ctx->pipe_mask;
        if (pipeMask && ctx->pipe_flag.ptr)
            SSYM_HIP_CHECK(ctx, hipMemcpyAsync(gaveUp, ctx->pipe_flag.ptr, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (!outDev) {
            rc = stage_d2h(ctx, out_idx, userIdx, sizeof(uint32_t) * M);
            if (rc == SSYM_OK && out_cost)
                rc = stage_d2h(ctx, out_cost, userCost, sizeof(double) * M);
            if (rc != SSYM_OK)
                return rc;
        }
        SSYM_HIP_CHECK(ctx, hipEventRecord(cev[7], st));
        rc = comm_wait_step(ctx, comm, cev[7]);             // the step's one synchronisation, under the deadline
        if (rc != SSYM_OK)
            return rc;
        release_deferred(ctx);                              // (blocks ensure() replaced while the step was being enqueued)
        // ---- from here on every rank looks at the same gathered words and takes the same way ----
        for (int g = 0; g < G; ++g)
            if (status[kStatusWords * g + 2]) {
                const int32_t code = -(int32_t)status[kStatusWords * g + 2];
                const std::string where = "rank " + std::to_string(g) + " of " + std::to_string(G) + ", phase " +
                                          std::to_string(status[kStatusWords * g + 3]);
                if (g == comm->rank)
                    ctx->err = "ssym_match_sharded failed on this rank (" + where + "): " + failedMsg +
                               "; every rank returns " + status_name(code);
                else
                    ctx->err = "ssym_match_sharded failed on " + where + " with " + status_name(code) +
                               " (ssym_last_error there has the cause); every rank returns it, this rank's results are void";
                ctx->pending_d2h.clear();
                ctx->pending.valid = false;
                *agreed = true;
                return code;
            }
        if (pipeMask) {
            for (int i = 0; i < 8; ++i)
                if ((pipeMask >> i & 1u) && gaveUp[i])
                    ++tm.exact_redone;
            ctx->pipe_mask = 0;
        }
        if (attempt == 0)
            coll_ms += ev_ms2(cev[2], cev[3]) + (pruned ? ev_ms2(cev[0], cev[1]) : 0.f);
        coll_ms += ev_ms2(cev[5], cev[6]);
        if (finished && ctx->so_filter) {
            hipEvent_t *ev = ctx->ev;
            sel_ms += ev_ms2(ev[0], ev[3]);
            ref_ms += ev_ms2(ev[3], ev[4]);
            red_ms += ev_ms2(ev[4], ev[5]);
            tm.n_refined = ownCount[0];
        } else if (finished && ctx->so_refcos) {       // refcos through the matrix pipe: main kernel | exact keys and fold
            tm.used_filter = 1;
            tm.refcos_filter = ctx->timings.refcos_filter;     // (which filter the enqueueing call took: capi.hip)
            tm.main_launches = 1;
            tm.main_ms = ev_ms2(ctx->ev[0], ctx->ev[1]);
            red_ms += ev_ms2(ctx->ev[1], ctx->ev[2]);
            tm.n_refined = ownCount[0];
        }
        bool anyOverflow = false;
        for (int g = 0; g < G; ++g)
            anyOverflow |= status[kStatusWords * g + 1] != 0;
        if (!anyOverflow)
            break;
        *agreed = true;                 // (the two ways out below are taken by every rank alike)
        if (attempt == 1) {
            ctx->err = "dtw: candidate list overflow on a rank of the sharded match";
            return SSYM_E_NOMEM;
        }
        for (int g = 0; g < G; ++g)
            if (status[kStatusWords * g + 1] && status[kStatusWords * g] >= 0xffffffffu) {
                ctx->err = "dtw: too many near-tied candidates for one batch on a rank of the sharded match";
                return SSYM_E_UNSUPPORTED;
            }
        *agreed = false;
        if (status[kStatusWords * comm->rank + 1])
            ctx->so_cap = status[kStatusWords * comm->rank];    // this rank's list wanted that much
    }
    *agreed = true;
    stage_finish(ctx);
    if (filterPath) {
        tm.main_launches = ctx->filter_launches;
        tm.main_ms = ev_ms2(ctx->ev[6], ctx->ev[1]);
        if (!tm.pruned && ctx->band < 0)
            tm.n_filter_cells = ctx->launched_cells * 64ull;
        if (tm.pruned) {
            tm.n_filter_cells = ctx->pruned_cells * 64ull;
            if (ctx->band < 0) {
                const SegmentSet &src = dict->set, &tgt = q->set;
                const double full = (double)src.n_pad * tgt.n_pad * src.frames_pad * std::max<uint32_t>(tgt.max_frames, 1);
                ctx->prune_swept = (float)std::min(1.0, (double)tm.n_filter_cells / full);
            }
        }
    } else if (!emptyShard && !refcos) {
        tm.n_refined = tm.n_pairs;
    }
    tm.select_ms = sel_ms;
    tm.refine_ms = ref_ms;
    tm.reduce_ms = red_ms;
    tm.collective_ms = coll_ms;
    tm.total_ms = ev_ms2(cev[4], cev[7]);
    if (pruned)
        tm.prune_ms = ev_ms2(cev[4], cev[0]);
    ctx->timings = tm;
    return SSYM_OK;
}

int32_t match_sharded_impl(ssym_ctx *ctx, ssym_comm *comm, const ssym_dict *dict, const ssym_queries *q,
                           const double *distance, uint32_t index_base, uint32_t *out_idx, double *out_cost,
                           uint32_t flags)
{
    if (!comm) {
        ctx->err = "ssym_match_sharded: NULL argument";
        return SSYM_E_INVALID;
    }
    if (comm->dead) {
        ctx->err = "ssym_match_sharded: the communicator was aborted by an earlier failure; destroy it and create a new one";
        return SSYM_E_COMM;
    }
    bool agreed = false;
    int32_t rc;
    try {
        rc = match_sharded_step(ctx, comm, dict, q, distance, index_base, out_idx, out_cost, flags, &agreed);
    } catch (const std::bad_alloc &) {
        ctx->err = "out of host memory";
        rc = SSYM_E_NOMEM;
    } catch (...) {
        ctx->err = "unexpected C++ exception inside the library";
        rc = SSYM_E_HIP;
    }
    if (rc != SSYM_OK && !agreed) {
        // this rank left the step on a way its peers do not know of: whatever it has enqueued may never complete, and
        // whatever they have enqueued waits for it.  Give the communicator up (RCCL: this rank's collectives leave the
        // stream; the peers meet their deadline and abort theirs) and drain the stream.
        const std::string keep = ctx->err;
        comm_abort(comm);
        ctx->stream_only = false;
        ctx->err.clear();
        (void)comm_drain_after_abort(ctx);
        const std::string drainNote = ctx->err;
        release_deferred(ctx);
        ctx->pending_d2h.clear();
        ctx->pending.valid = false;
        ctx->err = keep + " [this rank left the sharded step: its communicator is aborted]" + drainNote;
    }
    return rc;
}

}  // namespace

extern "C" int32_t ssym_match_sharded(ssym_ctx *ctx, ssym_comm *comm, const ssym_dict *dict, const ssym_queries *q,
                                      const double *distance, uint32_t index_base, uint32_t *out_idx,
                                      double *out_cost, uint32_t flags)
{
    if (!ctx)
        return SSYM_E_INVALID;
    return guarded(ctx, [&]() -> int32_t {
        return match_sharded_impl(ctx, comm, dict, q, distance, index_base, out_idx, out_cost, flags);
    });
}
