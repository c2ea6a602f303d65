// capi.hip -- the C ABI of include/soundsym_amd.h on top of the kernels.
#include "ssym_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

using namespace ssym;

// the tail behind the refcos search's results (refcos_mfma.hip, refcos_pack_tail): header words of its lists + timestamps
constexpr size_t kTailBytes = 4 * sizeof(uint32_t) + 3 * sizeof(unsigned long long);

static thread_local std::string g_create_err;

extern "C" {

int32_t ssym_abi_version(void) { return SSYM_ABI_VERSION; }

const char *ssym_last_error(const ssym_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int32_t ssym_ctx_create(const ssym_config *cfg, ssym_ctx **out)
{
    return guarded((ssym_ctx *)nullptr, [&]() -> int32_t {
    if (!cfg || !out || cfg->struct_size != sizeof(ssym_config)) {
        g_create_err = "ssym_ctx_create: bad config (NULL or struct_size mismatch)";
        return SSYM_E_INVALID;
    }
    *out = nullptr;
    if ((cfg->metric != SSYM_METRIC_REFCOS && cfg->metric != SSYM_METRIC_DTW) ||
        (cfg->dtype != SSYM_DTYPE_F64 && cfg->dtype != SSYM_DTYPE_F32) || cfg->band < -1) {
        g_create_err = "ssym_ctx_create: bad metric / dtype / band";
        return SSYM_E_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("ssym_ctx_create: no HIP device (") +
                       (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                       "); this library has no CPU path";
        return SSYM_E_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        g_create_err = "ssym_ctx_create: device ordinal out of range";
        return SSYM_E_INVALID;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) {
        g_create_err = "ssym_ctx_create: hipGetDeviceProperties failed";
        return SSYM_E_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("ssym_ctx_create: device is ") + prop.gcnArchName +
                       ", kernels are built for gfx950 only";
        return SSYM_E_NO_DEVICE;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) {
        g_create_err = "ssym_ctx_create: hipSetDevice failed";
        return SSYM_E_HIP;
    }
    ssym_ctx *ctx = new (std::nothrow) ssym_ctx();
    if (!ctx) {
        g_create_err = "out of memory";
        return SSYM_E_NOMEM;
    }
    ctx->device = cfg->device;
    ctx->metric = cfg->metric;
    ctx->dtype = cfg->dtype;
    ctx->band = cfg->band;
    ctx->squared = cfg->dtw_squared ? 1 : 0;
    ctx->prune_default = cfg->dtw_prune != 0;
    ctx->num_cus = prop.multiProcessorCount;
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) == hipSuccess && khz > 0)
            ctx->wall_clock_khz = (double)khz;
    }
    if (cfg->stream) {
        ctx->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            g_create_err = "hipStreamCreate failed";
            delete ctx;
            return SSYM_E_HIP;
        }
        ctx->owns_stream = true;
    }
    for (auto &ev : ctx->ev) {
        if (hipEventCreate(&ev) != hipSuccess) {
            g_create_err = "hipEventCreate failed";
            delete ctx;
            return SSYM_E_HIP;
        }
    }
    *out = ctx;
    return SSYM_OK;
    });
}

int32_t ssym_ctx_destroy(ssym_ctx *ctx)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DeviceBuf *bufs[] = {&ctx->handoff, &ctx->cmat, &ctx->tmin, &ctx->cand, &ctx->cand2, &ctx->cand_xmin,
                         &ctx->cand_cost, &ctx->best, &ctx->selmask, &ctx->selcnt, &ctx->topk,
                         &ctx->abandon, &ctx->prune_pairs, &ctx->prune_cost, &ctx->one_ticket, &ctx->dist, &ctx->part, &ctx->out_idx, &ctx->out_cost,
                         &ctx->pipe_flag, &ctx->tmin2, &ctx->zeros, &ctx->stamps};
    for (DeviceBuf *b : bufs)
        if (b->ptr)
            (void)hipFree(b->ptr);
    release_deferred(ctx);
    dev_cache_release(ctx);
    if (ctx->stage)
        (void)hipHostFree(ctx->stage);
    for (auto &ev : ctx->ev)
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->owns_stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SSYM_OK;
    });
}

int32_t ssym_ctx_synchronize(ssym_ctx *ctx)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SSYM_OK;
    });
}

int32_t ssym_get_timings(const ssym_ctx *ctx, ssym_timings *out)
{
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = ctx->timings;
    return SSYM_OK;
}

// ---- dictionary / queries -----------------------------------------------------------------------
static int32_t make_set(ssym_ctx *ctx, SegmentSet &set, const void *feats, bool on_device,
                        const uint64_t *off, uint32_t n, uint32_t dim, bool is_source)
{
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int32_t rc = pack_segments(ctx, set, feats, on_device, off, n, dim, is_source);
    if (rc != SSYM_OK)
        free_segments(ctx, set);
    return rc;
}

int32_t ssym_dict_create(ssym_ctx *ctx, const void *feats, const uint64_t *frame_offsets,
                         uint32_t n_segments, uint32_t dim, ssym_dict **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = nullptr;
    ssym_dict *d = new (std::nothrow) ssym_dict();
    if (!d)
        return SSYM_E_NOMEM;
    int32_t rc = make_set(ctx, d->set, feats, false, frame_offsets, n_segments, dim, true);
    if (rc != SSYM_OK) {
        delete d;
        return rc;
    }
    *out = d;
    return SSYM_OK;
    });
}

int32_t ssym_dict_create_device(ssym_ctx *ctx, const void *feats_dev, const uint64_t *frame_offsets,
                                uint32_t n_segments, uint32_t dim, ssym_dict **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = nullptr;
    ssym_dict *d = new (std::nothrow) ssym_dict();
    if (!d)
        return SSYM_E_NOMEM;
    int32_t rc = make_set(ctx, d->set, feats_dev, true, frame_offsets, n_segments, dim, true);
    if (rc != SSYM_OK) {
        delete d;
        return rc;
    }
    *out = d;
    return SSYM_OK;
    });
}

int32_t ssym_dict_append(ssym_ctx *ctx, ssym_dict *dict, const void *feats,
                         const uint64_t *frame_offsets, uint32_t n_segments)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !dict)
        return SSYM_E_INVALID;
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return append_segments(ctx, dict->set, feats, frame_offsets, n_segments);
    });
}

int32_t ssym_dict_size(const ssym_dict *dict, uint32_t *out_n_segments)
{
    if (!dict || !out_n_segments)
        return SSYM_E_INVALID;
    *out_n_segments = dict->set.n;
    return SSYM_OK;
}

int32_t ssym_dict_destroy(ssym_ctx *ctx, ssym_dict *dict)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!dict)
        return SSYM_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    free_segments(ctx, dict->set);
    if (dict->selfsim.ptr)
        (void)hipFree(dict->selfsim.ptr);
    delete dict;
    return SSYM_OK;
    });
}

int32_t ssym_queries_create(ssym_ctx *ctx, const void *feats, const uint64_t *frame_offsets,
                            uint32_t n_targets, uint32_t dim, ssym_queries **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = nullptr;
    ssym_queries *q = new (std::nothrow) ssym_queries();
    if (!q)
        return SSYM_E_NOMEM;
    int32_t rc = make_set(ctx, q->set, feats, false, frame_offsets, n_targets, dim, false);
    if (rc != SSYM_OK) {
        delete q;
        return rc;
    }
    *out = q;
    return SSYM_OK;
    });
}

int32_t ssym_queries_create_device(ssym_ctx *ctx, const void *feats_dev, const uint64_t *frame_offsets,
                                   uint32_t n_targets, uint32_t dim, ssym_queries **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = nullptr;
    ssym_queries *q = new (std::nothrow) ssym_queries();
    if (!q)
        return SSYM_E_NOMEM;
    int32_t rc = make_set(ctx, q->set, feats_dev, true, frame_offsets, n_targets, dim, false);
    if (rc != SSYM_OK) {
        delete q;
        return rc;
    }
    *out = q;
    return SSYM_OK;
    });
}

int32_t ssym_queries_destroy(ssym_ctx *ctx, ssym_queries *q)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!q)
        return SSYM_OK;
    if (ctx)
        (void)hipSetDevice(ctx->device);
    free_segments(ctx, q->set);      // blocks go back to the context's cache; reuse is stream-ordered
    delete q;
    return SSYM_OK;
    });
}

// ---- the hot path -----------------------------------------------------------------------------------
static float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess)
        return 0.f;
    return ms;
}

static int32_t check_match_args(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q)
{
    if (!ctx)
        return SSYM_E_INVALID;
    if (!dict || !q) {
        ctx->err = "dictionary or queries handle is NULL";
        return SSYM_E_INVALID;
    }
    if (dict->set.n == 0) {
        // the reference indexes sounds[0] of an empty Vec and panics (src/sound.rs:369)
        ctx->err = "empty dictionary";
        return SSYM_E_EMPTY_DICT;
    }
    if (dict->set.dim != q->set.dim) {
        ctx->err = "dim mismatch between dictionary and targets";
        return SSYM_E_INVALID;
    }
    return SSYM_OK;
}

}  // extern "C"

// filter costs live in record-slot coordinates; the caller sees segments in its own order
__global__ void f32_to_f64_matrix_kernel(const float *__restrict__ in, uint32_t rows, uint32_t cols,
                                         uint32_t ld, const uint32_t *__restrict__ permRow,
                                         const uint32_t *__restrict__ permCol, double *__restrict__ out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (c < cols && r < rows)
        out[(size_t)permRow[r] * cols + permCol[c]] = (double)in[(size_t)r * ld + c];
}

// per-target values between slot order (inside) and the caller's target order (outside)
__global__ void slots_to_targets_kernel(const double *__restrict__ bySlot, const uint32_t *__restrict__ perm,
                                        uint32_t n, double *__restrict__ byTarget)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        byTarget[perm[i]] = bySlot[i];
}
__global__ void targets_to_slots_kernel(const double *__restrict__ byTarget, const uint32_t *__restrict__ perm,
                                        uint32_t n, double *__restrict__ bySlot)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        bySlot[i] = byTarget[perm[i]];
}

// k_top = 1: ssym_match_queries (outputs [M]); k_top > 1: ssym_match_topk (outputs [M][k_top]).
// phase 0: the whole match.  Phases 1 / 2 are ssym_match_begin / ssym_match_finish: phase 1 stops
// after the filter and the per-target threshold (copied to bounds_dev), phase 2 takes the threshold
// back from bounds_dev (after the ranks' all-reduce) and runs selection, re-scoring and the fold.
constexpr uint32_t kFlagFewTargets = 0x80000000u;      // internal: set by ssym_match_batch / ssym_match_one for <= 4 targets

static int32_t match_impl(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                          const double *distance, uint32_t index_base, uint32_t k_top, uint32_t *out_idx,
                          double *out_cost, uint32_t flags, int phase = 0, double *bounds_dev = nullptr,
                          const double *prune_cost_dev = nullptr /* phase 1: reduced candidate costs, by target */)
{
    StageScope stageScope(ctx);
    int32_t rc = check_match_args(ctx, dict, q);
    if (rc != SSYM_OK)
        return rc;
    const SegmentSet &src = dict->set;
    const SegmentSet &tgt = q->set;
    const uint32_t N = src.n, M = tgt.n;
    ssym_timings tm{};
    tm.n_pairs = (uint64_t)N * M;
    if (M == 0) {
        ctx->timings = tm;
        return SSYM_OK;
    }
    if (!out_idx && phase != 1) {
        ctx->err = "out_idx is NULL";
        return SSYM_E_INVALID;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool outDev = (flags & SSYM_OUT_DEVICE) != 0;
    if (phase != 2)
        ctx->pipe_mask = 0;            // give-up counters of the exact kernel's pipelined variant: this call's start here
    if (phase == 0) {
        // any other call that uses the context's scratch ends a begin .. finish in progress (finish then reports
        // "without begin") and drops the candidates of ssym_match_candidates: their buffers are shared
        ctx->pending.cand = false;
        ctx->pending.valid = false;
    }
    if (ctx->prune_default && phase == 0 && M >= 64)      // (a handful of targets: the extra launches cost more than they save)
        flags |= SSYM_DTW_PRUNE;

    // per-target distance (morph_to, src/sound.rs:440-446)
    const double *distDev = nullptr;
    if (phase == 2) {
        distDev = ctx->pending.has_dist ? (const double *)ctx->dist.ptr : nullptr;   // uploaded by phase 1
    } else if (distance) {
        rc = ensure(ctx, ctx->dist, sizeof(double) * M);
        if (rc != SSYM_OK)
            return rc;
        rc = stage_h2d(ctx, ctx->dist.ptr, distance, sizeof(double) * M);
        if (rc != SSYM_OK)
            return rc;
        distDev = (const double *)ctx->dist.ptr;
    }
    uint32_t *idxDev = out_idx;
    double *costDev = out_cost;
    // refcos with host outputs: values, indices and four header words of the search's lists in ONE device block, so that
    // one copy brings back everything the call synchronises for (a search of 0.23 ms notices four)
    const size_t costBytes = sizeof(double) * (size_t)M * k_top;
    const size_t idxBytes = (sizeof(uint32_t) * (size_t)M * k_top + 7) & ~(size_t)7;      // (the tail's stamps: 8-byte aligned)
    uint32_t *hdrTail = nullptr;
    if (!outDev && ctx->metric == SSYM_METRIC_REFCOS) {
        rc = ensure(ctx, ctx->out_cost, costBytes + idxBytes + kTailBytes);
        if (rc != SSYM_OK)
            return rc;
        costDev = (double *)ctx->out_cost.ptr;
        idxDev = (uint32_t *)((char *)ctx->out_cost.ptr + costBytes);
        hdrTail = (uint32_t *)((char *)idxDev + idxBytes);
    } else if (!outDev) {
        rc = ensure(ctx, ctx->out_idx, idxBytes);
        if (rc != SSYM_OK)
            return rc;
        rc = ensure(ctx, ctx->out_cost, costBytes);
        if (rc != SSYM_OK)
            return rc;
        idxDev = (uint32_t *)ctx->out_idx.ptr;
        costDev = (double *)ctx->out_cost.ptr;
    }

    hipEvent_t *ev = ctx->ev;
    bool outputsStaged = false;          // host outputs already copied and synchronised (refcos filter path)
    bool stampsValid = false;            // refcos filter path: phase times from device timestamps instead of events
    float stampMs[2] = {0.f, 0.f};
    if (ctx->metric == SSYM_METRIC_REFCOS) {
        // The plain first-minimum search goes through the f64 matrix pipe (refcos_mfma.hip): every pair's dot as a
        // GEMM, a rigorous interval per key, and the reference's own arithmetic only on the few pairs that can
        // hold a target's minimum (top-k: one of its k smallest keys) -- same bits out.  Small problems keep the exact
        // tile kernel on every pair; so does a call whose candidate list overflowed.
        // (a sharded step only enqueues: its candidate list's header travels in the gathered status like the dtw
        //  lists', and the attempt every rank repeats after an overflow -- so_cap set -- takes the exact tile kernel)
        // Top-k goes that way up to k = 64: the waves' own thresholds (the k-th smallest bound of 64 rows each) only decide
        // what is LISTED; the threshold that selects the candidates is the k-th smallest bound over all of a target's listed
        // pairs (refcos_mfma.hip, refcos_topk_*).  4096 x 4096 x 128f x 12d, k = 2 / 4 / 8 / 16 / 64: 0.41 / 0.47 / 0.63 / 0.93 /
        // 2.9 ms against 3.2 / 3.1 / 3.3 / 3.9 / 7.0 ms on the exact tile kernel (tools/refcos_topk_timing.py).
        const char *kmaxKnob = ssym_knob("SSYM_REFCOS_TOPK_MAX");                       // (measurements: where the filters stop paying)
        const uint32_t kFilterMax = kmaxKnob ? (uint32_t)std::max(1, atoi(kmaxKnob)) : 64u;
        bool viaMfma = !(ctx->stream_only && (ctx->so_cap || k_top > 1)) && k_top <= kFilterMax && refcos_mfma_supported(ctx, src, tgt);
        ctx->so_refcos = false;
        // Which filter: the integer one where both sets have its records; should ITS list overflow -- values so close that
        // 23 bits of fixed point cannot tell them apart -- the f64 filter gets the search before the exact tile kernel does
        // (a sharded step repeats with the tile kernel at once: one agreed repeat per step).
        bool q8 = viaMfma && refcos_q8_ready(ctx, src, tgt);
        // Timing without events: an event record between two kernels costs ~7 us of gap on the stream, three of them a
        // tenth of a search of 0.2 ms; the search's first kernel, the first one after the main kernel and the last one
        // read the device's wall clock instead (a sharded step keeps the events: comm.hip reads them).
        const bool useStamps = !ctx->stream_only && ctx->wall_clock_khz > 0;
        unsigned long long *stampsDev = nullptr;
        if (viaMfma && useStamps) {
            rc = ensure(ctx, ctx->stamps, 128);
            if (rc != SSYM_OK)
                return rc;
            stampsDev = (unsigned long long *)ctx->stamps.ptr;
        }
        while (viaMfma) {
            const uint32_t *h1dev = nullptr, *h2dev = nullptr;
            uint32_t h1[2] = {0, 0}, h2[2] = {0, 0};
            if (!stampsDev)
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], st));
            // (where the tail goes: behind the host outputs' device block, or -- device outputs -- behind the timestamps)
            char *packed = (!ctx->stream_only && !outDev && hdrTail) ? stage_take(ctx, costBytes + idxBytes + kTailBytes) : nullptr;
            uint32_t *tailDev = packed ? hdrTail : (stampsDev ? (uint32_t *)(stampsDev + 4) : nullptr);
            rc = launch_refcos_match_mfma(ctx, src, tgt, distDev, index_base, idxDev, costDev, &h1dev, &h2dev, k_top, q8, stampsDev,
                                          tailDev);
            if (rc == SSYM_E_NOMEM && !ctx->stream_only) {
                // the filters' lists did not fit (a top-k list of a large grid is a few GB): the exact tile kernel needs
                // N x M x 8 bytes only and was the path of these calls before the filters took them
                ctx->err.clear();
                viaMfma = false;
                break;
            }
            if (rc != SSYM_OK)
                return rc;
            tm.refcos_filter = q8 ? 2 : 1;
            if (!stampsDev)
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[2], st));
            if (ctx->stream_only) {              // ssym_match_sharded reads the headers after the step's one synchronisation
                ctx->so_hdr1 = h1dev;
                ctx->so_hdr2 = h2dev;
                ctx->so_refcos = true;
                ctx->so_filter = false;
                tm.used_filter = 1;
                tm.main_launches = 1;
                ctx->timings = tm;
                return SSYM_OK;                  // (device outputs: the sharded step's send block)
            }
            // host outputs: results, headers and timestamps come back in one copy and under one synchronisation; should
            // the list have overflowed the results are dropped and the exact kernel's staged instead
            const size_t pendingBefore = ctx->pending_d2h.size();
            unsigned long long tailHost[(kTailBytes + 7) / 8] = {0};
            const unsigned char *tailAt = nullptr;
            if (packed) {
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(packed, costDev, costBytes + idxBytes + kTailBytes, hipMemcpyDeviceToHost, st));
                if (out_cost)
                    ctx->pending_d2h.push_back({out_cost, packed, costBytes});
                ctx->pending_d2h.push_back({out_idx, packed + costBytes, sizeof(uint32_t) * (size_t)M * k_top});
                outputsStaged = true;
                tailAt = (const unsigned char *)packed + costBytes + idxBytes;
            } else if (stampsDev) {              // device outputs (or no staging window): the tail alone comes back
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(tailHost, tailDev, kTailBytes, hipMemcpyDeviceToHost, st));
                tailAt = (const unsigned char *)tailHost;
                if (!outDev) {
                    rc = stage_d2h(ctx, out_idx, idxDev, sizeof(uint32_t) * (size_t)M * k_top);
                    if (rc == SSYM_OK && out_cost)
                        rc = stage_d2h(ctx, out_cost, costDev, costBytes);
                    if (rc != SSYM_OK)
                        return rc;
                    outputsStaged = true;
                }
            } else {
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(h1, h1dev, sizeof(h1), hipMemcpyDeviceToHost, st));
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(h2, h2dev, sizeof(h2), hipMemcpyDeviceToHost, st));
                if (!outDev) {
                    rc = stage_d2h(ctx, out_idx, idxDev, sizeof(uint32_t) * (size_t)M * k_top);
                    if (rc == SSYM_OK && out_cost)
                        rc = stage_d2h(ctx, out_cost, costDev, costBytes);
                    if (rc != SSYM_OK)
                        return rc;
                    outputsStaged = true;
                }
            }
            SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            if (tailAt) {
                const uint32_t *t = (const uint32_t *)tailAt;
                h1[0] = t[0]; h1[1] = t[1]; h2[0] = t[2]; h2[1] = t[3];
                if (stampsDev) {
                    unsigned long long ts[3];
                    memcpy(ts, tailAt + 4 * sizeof(uint32_t), sizeof(ts));
                    stampMs[0] = (float)((double)(ts[1] - ts[0]) / ctx->wall_clock_khz);       // main kernel (+ its init)
                    stampMs[1] = (float)((double)(ts[2] - ts[1]) / ctx->wall_clock_khz);       // selection, exact keys, fold
                    stampsValid = true;
                }
            }
            if (h1[1]) {                     // more near-ties than the list holds
                ctx->pending_d2h.resize(pendingBefore);
                outputsStaged = false;
                if (q8) {
                    q8 = false;              // ... for the integer filter: the f64 filter next
                    continue;
                }
                viaMfma = false;             // ... for the f64 filter too: the exact kernel on every pair
                tm.refcos_filter = 0;
                stampsValid = false;
            } else {
                tm.used_filter = 1;
                tm.n_refined = h2[0];
            }
            break;
        }
        if (!viaMfma) {
        rc = ensure(ctx, ctx->cmat, sizeof(double) * (size_t)N * M);
        if (rc != SSYM_OK)
            return rc;
        double *sims = (double *)ctx->cmat.ptr;
        SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], st));
        rc = launch_refcos_sims(ctx, src, tgt, sims);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipEventRecord(ev[1], st));
        rc = launch_refcos_argmin(ctx, N, M, sims, distDev, index_base, k_top, idxDev, costDev);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipEventRecord(ev[2], st));
        }
        tm.main_launches = 1;          // event times are read after the one synchronisation below
    } else {
        // frames wider than the filter's 42 values: the filter scores the first 42 and bounds the cost from
        // below; that supports the plain first-minimum search (no per-target distances, k = 1)
        const bool wide = filter_lower_bound_only(ctx, src, tgt);
        // few short queries against a small dictionary: the exact kernel on every pair is one launch of a few
        // thousand waves, the filter path a chain of ~20 launches (1 query x 1024 entries of 5...40 frames:
        // 81 us against 227 us per call; from 16 queries on the filter path is the shorter one)
        const bool fewPairs = (flags & kFlagFewTargets) && (uint64_t)N * M <= 8192 && src.max_frames + tgt.max_frames <= 128;
        // (sharded runs on wide frames exchange bounds in cost space: no per-target distances there)
        const bool useFilter = !(flags & SSYM_DTW_FORCE_EXACT) && !fewPairs && filter_supported(ctx, src, tgt) &&
                               (!wide || phase == 0 || !(phase == 2 ? ctx->pending.has_dist : distance != nullptr));
        tm.used_filter = useFilter ? 1 : 0;
        if (useFilter) {
            rc = ensure(ctx, ctx->cmat, sizeof(float) * (size_t)src.n_pad * tgt.n_pad);
            if (rc != SSYM_OK)
                return rc;
            float *cmat = (float *)ctx->cmat.ptr;
            // early abandoning applies to the plain first-minimum search of one unsharded call
            // (phase 1: the candidates were scored by ssym_match_candidates, their costs reduced over the ranks;
            //  phase 2: what phase 1 did)
            const bool prune = phase == 2 ? ctx->pending.pruned
                                          : (flags & SSYM_DTW_PRUNE) && k_top == 1 && !distDev &&
                                                (phase == 0 ? true : (prune_cost_dev != nullptr && !wide));
            if (phase != 2) {
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], st));
                const float *abandon = nullptr;
                unsigned long long *colCtr = nullptr;
                if (prune) {
                    rc = phase == 0 ? launch_dtw_prune_candidates(ctx, src, tgt) : SSYM_OK;
                    if (rc == SSYM_OK)
                        rc = launch_dtw_prune_thresholds(ctx, src, tgt, prune_cost_dev, &abandon);
                    if (rc != SSYM_OK)
                        return rc;
                    colCtr = (unsigned long long *)((char *)ctx->abandon.ptr + ctx->abandon.bytes) - 1;
                    rc = zero_words(ctx, colCtr, sizeof(*colCtr));
                    if (rc != SSYM_OK)
                        return rc;
                }
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[6], st));
                rc = launch_dtw_filter(ctx, src, tgt, cmat, abandon, colCtr, prune ? prune_cand_slots(ctx, tgt) : nullptr);
                if (rc != SSYM_OK)
                    return rc;
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[1], st));
                if (prune && ctx->stream_only) {     // through the pinned window: a pageable destination would block the host
                    rc = stage_d2h(ctx, &ctx->pruned_cells, colCtr, sizeof(*colCtr));
                    if (rc != SSYM_OK)
                        return rc;
                } else if (prune) {
                    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(&ctx->pruned_cells, colCtr, sizeof(*colCtr),
                                                       hipMemcpyDeviceToHost, st));
                }
                rc = wide ? launch_dtw_bounds_partial(ctx, src, tgt, cmat, prune ? (const double *)ctx->prune_cost.ptr : nullptr, k_top, distDev)
                          : launch_dtw_bounds(ctx, src, tgt, cmat, distDev, k_top,
                                              prune ? (const double *)ctx->prune_cost.ptr : nullptr);
                if (rc != SSYM_OK)
                    return rc;
            }
            tm.pruned = prune ? 1 : 0;
            tm.main_launches = ctx->filter_launches;     // one per class of source lengths (dtw_filter.hip)
            if (phase == 1) {
                // hand the threshold out: non-negative doubles (or +inf), bit for bit what stage 1 uses
                slots_to_targets_kernel<<<(M + 255) / 256, 256, 0, st>>>((const double *)ctx->tmin.ptr, tgt.perm, M,
                                                                         bounds_dev);
                SSYM_HIP_CHECK(ctx, hipGetLastError());
                ctx->pending.pruned = prune;
                if (ctx->stream_only) {          // ssym_match_sharded: the times are read after the step's one synchronisation
                    ctx->timings = tm;
                    return SSYM_OK;
                }
                SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
                ctx->pending.main_ms = ev_ms(ev[6], ev[1]);
                tm.main_ms = ctx->pending.main_ms;
                ctx->timings = tm;
                return SSYM_OK;
            }
            if (phase == 2) {
                targets_to_slots_kernel<<<(M + 255) / 256, 256, 0, st>>>(bounds_dev, tgt.perm, M,
                                                                         (double *)ctx->tmin.ptr);
                SSYM_HIP_CHECK(ctx, hipGetLastError());
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], st));
            }
            // list 1 (worst-case margin) is a few pairs per target when near-duplicates exist and
            // ~10^2 when they do not; on overflow stage 1 reports the size it wanted, the later
            // stages see the flag and do nothing, and the selection is redone with that room
            // (exactness never depends on the capacity)
            uint64_t cap = std::max<uint64_t>((256ull + 16ull * (k_top - 1)) * M, 65536);
            if (ctx->stream_only && ctx->so_cap)
                cap = ctx->so_cap;               // the size a previous attempt of this step asked for
            cap = std::min<uint64_t>(cap, (uint64_t)N * M);
            float sel_ms = 0.f, ref_ms = 0.f, red_ms = 0.f;
            for (int attempt = 0; attempt < 2; ++attempt) {
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[2], st));
                rc = launch_dtw_select(ctx, src, tgt, cmat, distDev, (uint32_t)cap);          // stage 1
                if (rc != SSYM_OK)
                    return rc;
                uint32_t *hdr1 = (uint32_t *)ctx->cand.ptr;
                rc = ensure(ctx, ctx->cand_xmin, sizeof(float) * cap);
                if (rc != SSYM_OK)
                    return rc;
                rc = launch_certify(ctx, src, tgt, hdr1, (const uint2 *)(hdr1 + 2), (uint32_t)cap,
                                    (float *)ctx->cand_xmin.ptr);                               // certificates
                if (rc != SSYM_OK)
                    return rc;
                const uint32_t *knownSrc =
                    prune ? (const uint32_t *)((const uint2 *)((const uint32_t *)ctx->prune_pairs.ptr + 2) + M) : nullptr;
                rc = launch_dtw_select2(ctx, src, tgt, cmat, (const float *)ctx->cand_xmin.ptr, distDev,
                                        (uint32_t)cap, k_top, wide, knownSrc);                  // stage 2
                if (rc != SSYM_OK)
                    return rc;
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[3], st));
                rc = ensure(ctx, ctx->cand_cost, sizeof(double) * (cap + M));
                if (rc != SSYM_OK)
                    return rc;
                uint32_t *hdr2 = (uint32_t *)ctx->cand2.ptr;
                rc = launch_dtw_exact(ctx, src, tgt, (const uint2 *)(hdr2 + 2), hdr2, (uint32_t)cap,
                                      (double *)ctx->cand_cost.ptr);
                if (rc != SSYM_OK)
                    return rc;
                if (prune) {
                    rc = launch_prune_append_known(ctx, M);
                    if (rc != SSYM_OK)
                        return rc;
                }
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[4], st));
                rc = launch_dtw_final(ctx, src, tgt, distDev, (uint32_t)(cap + (prune ? M : 0)), index_base, k_top,
                                      idxDev, costDev);
                if (rc != SSYM_OK)
                    return rc;
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[5], st));
                if (ctx->stream_only) {          // one attempt, enqueued only: the caller looks at the headers later
                    ctx->so_hdr1 = hdr1;
                    ctx->so_hdr2 = hdr2;
                    ctx->so_cap = cap;
                    ctx->so_filter = true;
                    ctx->timings = tm;
                    return SSYM_OK;
                }
                // ONE synchronisation per attempt: the lists' header words land in the pinned window (a copy into
                // pageable memory is a host round trip of its own: three of them and a second synchronisation for the
                // results were 60-80 us of a call), and the host results are requested in front of it -- an
                // overflowing list 1 (rare) drops them and asks again after the repeat
                uint32_t h1s[2] = {0, 0}, h2s[2] = {0, 0};
                unsigned gaves[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t *hw = ctx->api_depth > 0 ? (uint32_t *)stage_take(ctx, 12 * sizeof(uint32_t)) : nullptr;
                uint32_t *h1 = hw ? hw : h1s, *h2 = hw ? hw + 2 : h2s;
                unsigned *gave = hw ? hw + 4 : gaves;
                if (hw)
                    memset(hw, 0, 12 * sizeof(uint32_t));
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(h1, hdr1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(h2, hdr2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                if (ctx->pipe_mask)
                    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(gave, ctx->pipe_flag.ptr, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                const size_t pendingBefore = ctx->pending_d2h.size();
                bool stagedHere = false;
                if (!outDev && ctx->api_depth > 0) {
                    rc = stage_d2h(ctx, out_idx, idxDev, sizeof(uint32_t) * (size_t)M * k_top);
                    if (rc == SSYM_OK && out_cost)
                        rc = stage_d2h(ctx, out_cost, costDev, sizeof(double) * (size_t)M * k_top);
                    if (rc != SSYM_OK)
                        return rc;
                    // (a result too large for the window went straight to the caller's memory: still behind this
                    //  synchronisation, and harmlessly overwritten by a repeat)
                    stagedHere = true;
                }
                SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
                for (int i = 0; i < 8; ++i)
                    if ((ctx->pipe_mask >> i & 1u) && gave[i])
                        ++tm.exact_redone;
                ctx->pipe_mask = 0;
                if (stagedHere) {
                    if (h1[1] && !(attempt == 1 || cap == (uint64_t)N * M))
                        ctx->pending_d2h.resize(pendingBefore);      // the repeat's results are the ones to hand over
                    else
                        outputsStaged = true;
                }
                if (attempt == 0 && phase != 2)
                    sel_ms += ev_ms(ev[1], ev[2]);      // the per-target threshold (bounds) belongs to selection
                sel_ms += ev_ms(ev[2], ev[3]);
                ref_ms += ev_ms(ev[3], ev[4]);
                red_ms += ev_ms(ev[4], ev[5]);
                tm.n_refined = h2[0];
                if (!h1[1])
                    break;
                if (attempt == 1 || cap == (uint64_t)N * M) {
                    ctx->err = "dtw: candidate list overflow";
                    return SSYM_E_NOMEM;
                }
                cap = h1[0];
                if (cap >= 0xffffffffull) {
                    ctx->err = "dtw: too many near-tied candidates for one batch";
                    return SSYM_E_UNSUPPORTED;
                }
            }
            tm.main_ms = phase == 2 ? ctx->pending.main_ms : ev_ms(ev[6], ev[1]);
            if (!tm.pruned && ctx->band < 0)
                tm.n_filter_cells = ctx->launched_cells * 64ull;      // from the launches' geometry (dtw_filter.hip)
            if (tm.pruned) {
                tm.prune_ms = phase == 2 ? 0.f : ev_ms(ev[0], ev[6]);
                tm.n_filter_cells = ctx->pruned_cells * 64ull;
                if (ctx->band < 0) {
                    const double full = (double)src.n_pad * tgt.n_pad * src.frames_pad * std::max<uint32_t>(tgt.max_frames, 1);
                    ctx->prune_swept = (float)std::min(1.0, (double)tm.n_filter_cells / full);
                }
            }
            tm.select_ms = sel_ms;
            tm.refine_ms = ref_ms;
            tm.reduce_ms = red_ms;
            tm.total_ms = ev_ms(ev[0], ev[5]) + (phase == 2 ? ctx->pending.main_ms : 0.f);
        } else {
            rc = ensure(ctx, ctx->cmat, sizeof(double) * (size_t)N * M);
            if (rc != SSYM_OK)
                return rc;
            double *costs = (double *)ctx->cmat.ptr;
            SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], st));
            rc = launch_dtw_exact(ctx, src, tgt, nullptr, nullptr, 0, costs);
            if (rc != SSYM_OK)
                return rc;
            SSYM_HIP_CHECK(ctx, hipEventRecord(ev[1], st));
            rc = launch_dtw_final_allpairs(ctx, N, M, costs, distDev, index_base, k_top, idxDev, costDev);
            if (rc != SSYM_OK)
                return rc;
            SSYM_HIP_CHECK(ctx, hipEventRecord(ev[2], st));
            if (ctx->stream_only) {
                ctx->so_filter = false;
                ctx->timings = tm;
                return SSYM_OK;
            }
            unsigned gave[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ctx->pipe_mask)
                SSYM_HIP_CHECK(ctx, hipMemcpyAsync(gave, ctx->pipe_flag.ptr, sizeof(gave), hipMemcpyDeviceToHost, st));
            SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            for (int i = 0; i < 8; ++i)
                if ((ctx->pipe_mask >> i & 1u) && gave[i])
                    ++tm.exact_redone;
            tm.refine_ms = ev_ms(ev[0], ev[1]);
            tm.reduce_ms = ev_ms(ev[1], ev[2]);
            tm.total_ms = ev_ms(ev[0], ev[2]);
            tm.n_refined = (uint64_t)N * M;
        }
    }

    if (!outDev && !outputsStaged) {
        rc = stage_d2h(ctx, out_idx, idxDev, sizeof(uint32_t) * (size_t)M * k_top);
        if (rc == SSYM_OK && out_cost)
            rc = stage_d2h(ctx, out_cost, costDev, sizeof(double) * (size_t)M * k_top);
        if (rc != SSYM_OK)
            return rc;
    }
    if (ctx->stream_only) {              // refcos through ssym_match_sharded (device outputs)
        ctx->so_filter = false;
        ctx->timings = tm;
        return SSYM_OK;
    }
    if ((!outDev || ctx->metric == SSYM_METRIC_REFCOS) && !outputsStaged)      // (staged: already synchronised above)
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    stage_finish(ctx);
    if (ctx->metric == SSYM_METRIC_REFCOS && stampsValid) {
        tm.main_ms = stampMs[0];
        tm.reduce_ms = stampMs[1];
        tm.total_ms = stampMs[0] + stampMs[1];
    } else if (ctx->metric == SSYM_METRIC_REFCOS) {
        tm.main_ms = ev_ms(ev[0], ev[1]);
        tm.reduce_ms = ev_ms(ev[1], ev[2]);
        tm.total_ms = ev_ms(ev[0], ev[2]);
    }
    ctx->timings = tm;
    return SSYM_OK;
}

extern "C" {

int32_t ssym_match_queries(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                           const double *distance, uint32_t index_base, uint32_t *out_idx,
                           double *out_cost, uint32_t flags)
{
    return guarded(ctx, [&]() -> int32_t {
    return match_impl(ctx, dict, q, distance, index_base, 1, out_idx, out_cost, flags);
    });
}

int32_t ssym_match_topk(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, const double *distance,
                        uint32_t k, uint32_t index_base, uint32_t *out_idx, double *out_cost, uint32_t flags)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (k == 0 || k > SSYM_TOPK_MAX) {
        ctx->err = "ssym_match_topk: k must be in 1..SSYM_TOPK_MAX";
        return SSYM_E_INVALID;
    }
    return match_impl(ctx, dict, q, distance, index_base, k, out_idx, out_cost, flags);
    });
}

// Two-phase match for source-sharded runs (see the header).  A tiny kernel-free helper fills the
// bounds with +inf when the filter does not apply; the all-reduce then changes nothing.
static bool prune_applies(const ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q)
{
    const bool wideFrames = filter_lower_bound_only(ctx, dict->set, q->set);
    return ctx->metric == SSYM_METRIC_DTW && q->set.n > 0 && !wideFrames && filter_supported(ctx, dict->set, q->set);
}

__global__ void fill_f64_kernel(double *p, double v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = v;
}

}  // extern "C"

namespace ssym {

int32_t match_candidates_impl(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, double *cost_dev)
{
    int32_t rc = check_match_args(ctx, dict, q);
    if (rc != SSYM_OK)
        return rc;
    if (!cost_dev) {
        ctx->err = "ssym_match_candidates: cost_dev is NULL";
        return SSYM_E_INVALID;
    }
    ssym_ctx::Pending &pd = ctx->pending;
    pd = ssym_ctx::Pending{};
    const uint32_t M = q->set.n;
    if (M == 0)
        return SSYM_OK;
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (prune_applies(ctx, dict, q)) {
        rc = launch_dtw_prune_candidates(ctx, dict->set, q->set);
        if (rc != SSYM_OK)
            return rc;
        slots_to_targets_kernel<<<(M + 255) / 256, 256, 0, ctx->stream>>>((const double *)ctx->prune_cost.ptr,
                                                                          q->set.perm, M, cost_dev);
        pd.cand = true;
        pd.dict = dict;
        pd.q = q;
    } else {
        fill_f64_kernel<<<(M + 255) / 256, 256, 0, ctx->stream>>>(cost_dev, (double)INFINITY, M);
    }
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    if (!ctx->stream_only)
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SSYM_OK;
}

int32_t match_begin_impl(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, const double *distance,
                         uint32_t index_base, double *bounds_dev, const double *prune_cost_dev)
{
    int32_t rc = check_match_args(ctx, dict, q);
    if (rc != SSYM_OK)
        return rc;
    if (!bounds_dev) {
        ctx->err = "ssym_match_begin: bounds_dev is NULL";
        return SSYM_E_INVALID;
    }
    ssym_ctx::Pending &pd = ctx->pending;
    // the reduced candidate costs are only usable when THIS context scored its candidates for the same sets
    // (finish appends them); otherwise the call is a plain begin
    if (prune_cost_dev && !(pd.cand && pd.dict == dict && pd.q == q && !distance))
        prune_cost_dev = nullptr;
    pd = ssym_ctx::Pending{};
    pd.dict = dict;
    pd.q = q;
    pd.index_base = index_base;
    pd.has_dist = distance != nullptr;
    const uint32_t M = q->set.n;
    if (distance)
        pd.dist_host.assign(distance, distance + M);
    const bool wideFrames = filter_lower_bound_only(ctx, dict->set, q->set);
    pd.filter = ctx->metric == SSYM_METRIC_DTW && M > 0 && filter_supported(ctx, dict->set, q->set) &&
                (!wideFrames || !distance);
    if (pd.filter) {
        rc = match_impl(ctx, dict, q, distance, index_base, 1, nullptr, nullptr, prune_cost_dev ? SSYM_DTW_PRUNE : 0u, 1,
                        bounds_dev, prune_cost_dev);
        if (rc != SSYM_OK)
            return rc;
    } else if (M > 0) {
        SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        fill_f64_kernel<<<(M + 255) / 256, 256, 0, ctx->stream>>>(bounds_dev, (double)INFINITY, M);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        if (!ctx->stream_only)
            SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    pd.valid = true;
    return SSYM_OK;
}

int32_t match_finish_impl(ssym_ctx *ctx, const double *bounds_dev, uint32_t *out_idx, double *out_cost, uint32_t flags)
{
    ssym_ctx::Pending &pd = ctx->pending;
    if (!pd.valid) {
        ctx->err = "ssym_match_finish without ssym_match_begin";
        return SSYM_E_INVALID;
    }
    pd.valid = false;
    if (!bounds_dev) {
        ctx->err = "ssym_match_finish: bounds_dev is NULL";
        return SSYM_E_INVALID;
    }
    const double *dist = pd.has_dist ? pd.dist_host.data() : nullptr;
    if (!pd.filter)
        return match_impl(ctx, pd.dict, pd.q, dist, pd.index_base, 1, out_idx, out_cost, flags);
    return match_impl(ctx, pd.dict, pd.q, dist, pd.index_base, 1, out_idx, out_cost, flags, 2,
                      const_cast<double *>(bounds_dev));
}

}  // namespace ssym

extern "C" {

int32_t ssym_match_candidates(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, double *cost_dev)
{
    return guarded(ctx, [&]() -> int32_t {
    return match_candidates_impl(ctx, dict, q, cost_dev);
    });
}

int32_t ssym_match_begin(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, const double *distance,
                         uint32_t index_base, double *bounds_dev)
{
    return guarded(ctx, [&]() -> int32_t {
    return match_begin_impl(ctx, dict, q, distance, index_base, bounds_dev, nullptr);
    });
}

int32_t ssym_match_begin_pruned(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, uint32_t index_base,
                                const double *cost_dev, double *bounds_dev)
{
    return guarded(ctx, [&]() -> int32_t {
    if (ctx && !cost_dev) {
        ctx->err = "ssym_match_begin_pruned: cost_dev is NULL";
        return SSYM_E_INVALID;
    }
    return match_begin_impl(ctx, dict, q, nullptr, index_base, bounds_dev, cost_dev);
    });
}

int32_t ssym_match_finish(ssym_ctx *ctx, const double *bounds_dev, uint32_t *out_idx, double *out_cost,
                          uint32_t flags)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    return match_finish_impl(ctx, bounds_dev, out_idx, out_cost, flags);
    });
}

int32_t ssym_match_batch(ssym_ctx *ctx, const ssym_dict *dict, const void *tgt_feats,
                         const uint64_t *tgt_frame_offsets, uint32_t n_targets, const double *distance,
                         uint32_t *out_idx, double *out_cost)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (!dict) {
        ctx->err = "dictionary handle is NULL";
        return SSYM_E_INVALID;
    }
    if (dict->set.n == 0) {
        ctx->err = "empty dictionary";
        return SSYM_E_EMPTY_DICT;
    }
    StageScope stageScope(ctx);       // one staging window for the pack and the match
    // refcos, up to 64 short queries -- or dtw, up to 4 short queries against short entries: the whole call in ONE launch
    // (refcos_match_one_kernel / dtw_match_few_kernel); queries, offsets and distances go into the pinned window,
    // which the kernel reads and answers into directly
    const bool fewRefcos = out_idx && tgt_feats && refcos_few_supported(ctx, dict->set, tgt_frame_offsets, n_targets);
    const bool fewDtw = out_idx && tgt_feats && dtw_few_supported(ctx, dict->set, tgt_frame_offsets, n_targets);
    if (fewRefcos || fewDtw) {
        const size_t esz = ctx->dtype == SSYM_DTYPE_F32 ? sizeof(float) : sizeof(double);
        const uint64_t f0 = tgt_frame_offsets[0], f1 = tgt_frame_offsets[n_targets];
        const size_t qBytes = (size_t)(f1 - f0) * dict->set.dim * esz;
        char *qP = stage_take(ctx, qBytes ? qBytes : 8);
        uint64_t *offP = (uint64_t *)stage_take(ctx, sizeof(uint64_t) * (n_targets + 1));
        double *distP = distance ? (double *)stage_take(ctx, sizeof(double) * n_targets) : nullptr;
        double *valP = (double *)stage_take(ctx, sizeof(double) * n_targets);
        uint32_t *idxP = (uint32_t *)stage_take(ctx, sizeof(uint32_t) * n_targets);
        if (qP && offP && valP && idxP && (distP || !distance)) {
            SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
            ctx->pending.cand = false;
            ctx->pending.valid = false;
            if (qBytes)
                memcpy(qP, (const char *)tgt_feats + (size_t)f0 * dict->set.dim * esz, qBytes);
            for (uint32_t i = 0; i <= n_targets; ++i)
                offP[i] = tgt_frame_offsets[i] - f0;
            if (distP)
                memcpy(distP, distance, sizeof(double) * n_targets);
            hipEvent_t *ev = ctx->ev;
            // refcos: the kernel reads the device's wall clock into the pinned window itself (two event records cost 2.5 us
            // of a 32 us call); dtw keeps the events
            unsigned long long *tsP = (fewRefcos && ctx->wall_clock_khz > 0) ? (unsigned long long *)stage_take(ctx, 8 * ((size_t)n_targets + 1)) : nullptr;
            const bool noEv = tsP != nullptr;
            if (tsP)
                memset(tsP, 0, 8 * ((size_t)n_targets + 1));
            if (!noEv)
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[0], ctx->stream));
            int32_t rcf = fewRefcos ? launch_refcos_match_few(ctx, dict->set, qP, offP, n_targets, distP, 1.0, valP, idxP, tsP)
                                    : launch_dtw_match_few(ctx, dict->set, qP, offP, n_targets, distP, valP, idxP);
            if (rcf != SSYM_OK)
                return rcf;
            if (!noEv)
                SSYM_HIP_CHECK(ctx, hipEventRecord(ev[1], ctx->stream));
            SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(out_idx, idxP, sizeof(uint32_t) * n_targets);
            if (out_cost)
                memcpy(out_cost, valP, sizeof(double) * n_targets);
            ssym_timings tm{};
            tm.n_pairs = (uint64_t)dict->set.n * n_targets;
            if (noEv) {
                unsigned long long tEnd = tsP[0];
                for (uint32_t i = 0; i < n_targets; ++i)
                    tEnd = std::max(tEnd, tsP[1 + i]);
                tm.main_ms = tm.total_ms = (float)((double)(tEnd - tsP[0]) / ctx->wall_clock_khz);
            } else {
                tm.main_ms = tm.total_ms = ev_ms(ev[0], ev[1]);
            }
            tm.main_launches = 1;
            ctx->timings = tm;
            return SSYM_OK;
        }
    }
    ssym_queries *q = nullptr;
    hipEvent_t e0 = ctx->ev[8], e1 = ctx->ev[9];      // (the match below records ev[0..6] itself)
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SSYM_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
    ctx->defer_sync = true;        // this call synchronises once, at the end of the match
    // a handful of short dtw queries go to the exact kernel on every pair (match_impl, kFlagFewTargets): their
    // pack can leave out everything only the filter needs
    uint64_t maxQ = 0;
    if (tgt_frame_offsets)
        for (uint32_t i = 0; i < n_targets; ++i)
            maxQ = std::max<uint64_t>(maxQ, tgt_frame_offsets[i + 1] - tgt_frame_offsets[i]);
    ctx->pack_light = ctx->metric == SSYM_METRIC_DTW && n_targets <= 4 && (uint64_t)dict->set.n * n_targets <= 8192 &&
                      dict->set.max_frames + maxQ <= 128;
    int32_t rc = ssym_queries_create(ctx, tgt_feats, tgt_frame_offsets, n_targets, dict->set.dim, &q);
    ctx->pack_light = false;
    ctx->defer_sync = false;
    if (rc != SSYM_OK) {
        (void)hipStreamSynchronize(ctx->stream);      // the caller's buffers may go after an error too
        return rc;
    }
    SSYM_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
    // a handful of queries at a time is the reference's own call pattern (match_sound per target,
    // src/sound.rs:453-454): what counts then is the length of the launch chain, see match_impl
    rc = match_impl(ctx, dict, q, distance, 0, 1, out_idx, out_cost, n_targets <= 4 ? kFlagFewTargets : 0u);
    if (rc == SSYM_OK)
        ctx->timings.pack_ms = ev_ms(e0, e1);
    else
        (void)hipStreamSynchronize(ctx->stream);
    ssym_queries_destroy(ctx, q);
    return rc;
    });
}

int32_t ssym_match_one(ssym_ctx *ctx, const ssym_dict *dict, const void *feats, uint64_t n_frames,
                       double distance, uint32_t *out_idx, double *out_cost)
{
    return guarded(ctx, [&]() -> int32_t {
    const uint64_t off[2] = {0, n_frames};
    return ssym_match_batch(ctx, dict, feats, off, 1, &distance, out_idx, out_cost);
    });
}

/* from_distances (src/sound.rs:405-417) on the device; see chain.hip. */
int32_t ssym_chain(ssym_ctx *ctx, ssym_dict *dict, const void *start_feats, uint64_t start_frames,
                   const double *distances, uint32_t n_steps, uint32_t *out_idx, double *out_cost)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (!dict) {
        ctx->err = "dictionary handle is NULL";
        return SSYM_E_INVALID;
    }
    if (n_steps == 0)
        return SSYM_OK;
    if (!distances || !out_idx || (!start_feats && start_frames)) {
        ctx->err = "ssym_chain: NULL argument";
        return SSYM_E_INVALID;
    }
    const SegmentSet &src = dict->set;
    const uint32_t N = src.n;
    if (N == 0) {
        ctx->err = "empty dictionary";          // the reference panics at the first step (:369)
        return SSYM_E_EMPTY_DICT;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ctx->pending.cand = false;          // (the chain uses the scratch a begin .. finish would still need)
    ctx->pending.valid = false;
    hipStream_t st = ctx->stream;
    const bool refcos = ctx->metric == SSYM_METRIC_REFCOS;
    const double init = refcos ? 2.0 : (double)INFINITY;

    // device outputs + the current index
    int32_t rc = ensure(ctx, ctx->out_idx, sizeof(uint32_t) * ((size_t)n_steps + 1));
    if (rc != SSYM_OK)
        return rc;
    rc = ensure(ctx, ctx->out_cost, sizeof(double) * n_steps);
    if (rc != SSYM_OK)
        return rc;
    uint32_t *idxDev = (uint32_t *)ctx->out_idx.ptr;
    uint32_t *cur = idxDev + n_steps;
    double *costDev = (double *)ctx->out_cost.ptr;

    // step 0: the start sound against the whole dictionary (N values)
    const uint64_t off[2] = {0, start_frames};
    ssym_queries *q = nullptr;
    rc = ssym_queries_create(ctx, start_feats, off, 1, src.dim, &q);
    if (rc != SSYM_OK)
        return rc;
    rc = ensure(ctx, ctx->cmat, sizeof(double) * N);
    if (rc == SSYM_OK)
        rc = refcos ? launch_refcos_sims(ctx, src, q->set, (double *)ctx->cmat.ptr)
                    : launch_dtw_exact(ctx, src, q->set, nullptr, nullptr, 0, (double *)ctx->cmat.ptr);
    if (rc == SSYM_OK)
        rc = launch_chain_argmin(ctx, (const double *)ctx->cmat.ptr, 0, nullptr, N, distances[0], init, !refcos,
                                 0, cur, idxDev, costDev);
    if (rc != SSYM_OK) {
        ssym_queries_destroy(ctx, q);
        return rc;
    }

    if (n_steps > 1 && refcos) {
        // later queries are dictionary entries: rows of the self-similarity matrix
        if (dict->selfsim_n != N) {
            rc = ensure(ctx, dict->selfsim, sizeof(double) * (size_t)N * N);
            if (rc == SSYM_OK)
                rc = launch_refcos_sims(ctx, src, src, (double *)dict->selfsim.ptr);
            if (rc != SSYM_OK) {
                ssym_queries_destroy(ctx, q);
                return rc;
            }
            dict->selfsim_n = N;
        }
        for (uint32_t i = 1; i < n_steps && rc == SSYM_OK; ++i)
            rc = launch_chain_argmin(ctx, (const double *)dict->selfsim.ptr, N, cur, N, distances[i], init, false, i,
                                     cur, idxDev, costDev);
    } else if (n_steps > 1) {
        rc = ensure(ctx, ctx->cand, sizeof(uint32_t) * 2 + sizeof(uint2) * (size_t)N);
        if (rc == SSYM_OK)
            rc = ensure(ctx, ctx->cand_cost, sizeof(double) * N);
        uint32_t *hdr = (uint32_t *)ctx->cand.ptr;
        for (uint32_t i = 1; i < n_steps && rc == SSYM_OK; ++i) {
            rc = launch_chain_pairs(ctx, cur, N, (uint2 *)(hdr + 2), hdr);
            if (rc == SSYM_OK)
                rc = launch_dtw_exact(ctx, src, src, (const uint2 *)(hdr + 2), hdr, N, (double *)ctx->cand_cost.ptr);
            if (rc == SSYM_OK)
                rc = launch_chain_argmin(ctx, (const double *)ctx->cand_cost.ptr, 0, nullptr, N, distances[i], init,
                                         true, i, cur, idxDev, costDev);
        }
    }
    if (rc == SSYM_OK) {
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync(out_idx, idxDev, sizeof(uint32_t) * n_steps, hipMemcpyDeviceToHost, st));
        if (out_cost)
            SSYM_HIP_CHECK(ctx, hipMemcpyAsync(out_cost, costDev, sizeof(double) * n_steps, hipMemcpyDeviceToHost,
                                               st));
    }
    hipError_t e = hipStreamSynchronize(st);
    ssym_queries_destroy(ctx, q);
    if (rc == SSYM_OK && e != hipSuccess) {
        ctx->err = std::string("ssym_chain: ") + hipGetErrorString(e);
        return SSYM_E_HIP;
    }
    return rc;
    });
}

int32_t ssym_pair_matrix(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, int32_t exact,
                         double *out_matrix)
{
    return guarded(ctx, [&]() -> int32_t {
    int32_t rc = check_match_args(ctx, dict, q);
    if (rc != SSYM_OK)
        return rc;
    if (!out_matrix) {
        ctx->err = "out_matrix is NULL";
        return SSYM_E_INVALID;
    }
    ctx->pending.cand = false;          // (the matrix lands in the scratch a begin .. finish would still need)
    ctx->pending.valid = false;
    const SegmentSet &src = dict->set;
    const SegmentSet &tgt = q->set;
    const uint32_t N = src.n, M = tgt.n;
    if (M == 0)
        return SSYM_OK;
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    rc = ensure(ctx, ctx->part, sizeof(double) * (size_t)N * M);
    if (rc != SSYM_OK)
        return rc;
    double *mat = (double *)ctx->part.ptr;
    if (ctx->metric == SSYM_METRIC_REFCOS) {
        if (exact == 3 && !refcos_q8_ready(ctx, src, tgt)) {
            ctx->err = "the integer filter does not take these sets (a value that is not finite or far out of range, a "
                       "segment of more than 32768 values, SSYM_REFCOS_Q8=0): ask for exact = 2";
            return SSYM_E_UNSUPPORTED;
        }
        rc = exact >= 2 ? launch_refcos_mfma_sims(ctx, src, tgt, mat, exact == 3) : launch_refcos_sims(ctx, src, tgt, mat);
    } else if (exact) {
        rc = launch_dtw_exact(ctx, src, tgt, nullptr, nullptr, 0, mat);
    } else {
        if (!filter_supported(ctx, src, tgt)) {
            ctx->err = "dtw filter does not cover this shape (band / frames / dim); ask for exact = 1";
            return SSYM_E_UNSUPPORTED;
        }
        rc = ensure(ctx, ctx->cmat, sizeof(float) * (size_t)src.n_pad * tgt.n_pad);
        if (rc != SSYM_OK)
            return rc;
        rc = launch_dtw_filter(ctx, src, tgt, (float *)ctx->cmat.ptr);
        if (rc == SSYM_OK) {
            dim3 grid((M + 255) / 256, N);
            f32_to_f64_matrix_kernel<<<grid, 256, 0, st>>>((const float *)ctx->cmat.ptr, N, M, tgt.n_pad, src.perm,
                                                           tgt.perm, mat);
            SSYM_HIP_CHECK(ctx, hipGetLastError());
        }
    }
    if (rc != SSYM_OK)
        return rc;
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(out_matrix, mat, sizeof(double) * (size_t)N * M,
                                       hipMemcpyDeviceToHost, st));
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return SSYM_OK;
    });
}

int32_t ssym_merge_shards_at(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets, const double *costs_dev,
                             const uint32_t *idx_dev, const double *distance, uint32_t *out_idx_dev,
                             double *out_cost_dev)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (n_shards == 0 || !costs_dev || !idx_dev || !out_idx_dev) {
        ctx->err = "ssym_merge_shards: bad arguments";
        return SSYM_E_INVALID;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    StageScope stageScope(ctx);
    const double *distDev = nullptr;
    if (distance && n_targets) {
        int32_t rc = ensure(ctx, ctx->dist, sizeof(double) * n_targets);
        if (rc == SSYM_OK)
            rc = stage_h2d(ctx, ctx->dist.ptr, distance, sizeof(double) * n_targets);
        if (rc != SSYM_OK)
            return rc;
        distDev = (const double *)ctx->dist.ptr;
    }
    int32_t rc = launch_merge_shards(ctx, n_shards, n_targets, costs_dev, idx_dev, distDev, out_idx_dev, out_cost_dev);
    if (rc != SSYM_OK)
        return rc;
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SSYM_OK;
    });
}

int32_t ssym_merge_shards(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets, const double *costs_dev,
                          const uint32_t *idx_dev, uint32_t *out_idx_dev, double *out_cost_dev)
{
    return guarded(ctx, [&]() -> int32_t {
    return ssym_merge_shards_at(ctx, n_shards, n_targets, costs_dev, idx_dev, nullptr, out_idx_dev, out_cost_dev);
    });
}

}  // extern "C"
