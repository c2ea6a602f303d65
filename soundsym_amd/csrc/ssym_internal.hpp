// Internal declarations shared by the HIP translation units of libsoundsym_amd.so.
// Nothing here is part of the C ABI (include/soundsym_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "soundsym_amd.h"

// Measurement and test knobs (SSYM_FILTER_*, SSYM_REFCOS_*, SSYM_EXACT_*, SSYM_CELLS_*, SSYM_PRUNE_NT ...: none changes a
// result) are read only in a process that asked for the library's test hooks with SSYM_TEST_HOOKS=1 -- a production
// process cannot be steered by a stray variable.  (SSYM_COMM_TIMEOUT_MS and SSYM_RCCL_LIB are configuration, not knobs.)
inline const char *ssym_knob(const char *name)
{
    const char *h = getenv("SSYM_TEST_HOOKS");
    return (h && atoi(h) != 0) ? getenv(name) : nullptr;
}

namespace ssym {

// dtw filter geometry (see dtw_filter_kernel.hpp): a source slot holds 16*nt*rb rows, processed by
// one wave in rb passes of nt 32x32 tiles each.
struct FilterShape {
    int nt = 0, rb = 0;
    int rows() const { return 16 * nt * rb; }
};
inline FilterShape filter_shape(int max_frames)
{
    if (max_frames <= 16) return FilterShape{1, 1};
    if (max_frames <= 32) return FilterShape{2, 1};
    if (max_frames <= 48) return FilterShape{3, 1};
    if (max_frames <= 4096) return FilterShape{4, (max_frames + 63) / 64};
    return FilterShape{};   // beyond the filter's reach: exact kernel only
}

// How a frame's values are laid into the filter records (dtw_filter_kernel.hpp):
//   1  one f16 piece per value, frames of 14...42 values (42 product slots + 6 norm slots of K = 48);
//   2  two pieces per value on BOTH sides, up to 13 values (3 cross products per value: 39 + 6 slots of K = 48);
//   3  up to 13 values in K = 32: the SOURCE in two pieces, the TARGET in one (two for its first two values), the
//      norms those of the represented frames in two pieces each -- two MFMAs per tile instead of three (the third plane
//      of the records stays zero).  The default for up to 13 values; SSYM_FILTER_K48=1 keeps layout 2 (measurements).
int filter_pieces(int dim);
// MFMAs the unbanded filter issues per 32 x 32 tile for that layout
// (layout 1 needs dim + 6 slots: frames of 14...26 values fit two planes as well)
inline int filter_mfmas(int pieces, int dim) { return (pieces == 3 || (pieces == 1 && dim + 6 <= 32)) ? 2 : 3; }
// values per frame the filter sees: frames wider than 42 values enter with their first 42 only, which
// makes the filter's cost a LOWER bound of the pair's cost ("wide frames")
inline int filter_dim_used(int dim) { return dim <= 42 ? dim : 42; }

// doubles allocated behind a set's values: 16-byte loads that start on a segment's last value stay inside the buffer
constexpr size_t kRawTailPad = 2;

struct DeviceBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct SegmentSet {
    // raw features, always f64 on device (f32 inputs are widened exactly)
    double *raw = nullptr;          // [total_frames * dim]
    uint64_t *off = nullptr;        // device copy of frame offsets, rebased to 0, [n+1]
    std::vector<uint64_t> h_off;    // host copy, rebased to 0
    uint32_t n = 0;
    uint32_t dim = 0;
    uint64_t total_frames = 0;
    uint32_t max_frames = 0;
    // refcos
    double *norm = nullptr;         // [n] norm(me) of src/sound.rs:35-38 per segment, then [n] sqrt(norm) rounded up, [n] 1 / norm
    // dtw filter
    int32_t *len = nullptr;         // [n_pad] frames per segment (0 for padding segments)
    float *max_sqnorm = nullptr;    // [n_pad] max_f ||frame||^2 per segment (f32, rounded up)
    double max_sqnorm_all = 0.0;    // host: max over all segments
    double max_abs = 0.0;           // host: max |value| over the set (inf if any value is not finite)
    uint32_t n_pad = 0;
    uint32_t frames_pad = 0;        // record slots per segment
    bool is_source = false;
    bool light = false;             // dtw targets packed without slot order and statistics: exact kernels only
    // f16 operand records, (re)built by dtw_filter.hip whenever the common scale changes
    mutable void *rec = nullptr;    // [n_pad][frames_pad][48] _Float16
    mutable double rec_scale = 0.0;
    // dtw: record slot p holds segment perm[p]; slots are ordered by segment length so that the
    // segments sharing a wave / workgroup / target group need about the same rows and columns
    // (ragged real-world segmentations otherwise pay for the longest member).  len / max_sqnorm
    // are indexed by slot; pad slots map to 0xffffffff.  Everything outside the filter, certify and
    // selection kernels keeps the caller's indices.
    uint32_t *perm = nullptr;
    std::vector<uint32_t> h_perm;
    mutable size_t rec_bytes = 0;
    mutable uint32_t rec_slots = 0;   // record slots per segment in `rec`
    mutable int rec_lead = 0;         // slot of frame 0 (-1: end-aligned)
    size_t raw_capacity_vals = 0;
    // SSYM_DTW_PRUNE: mean frame per segment [n][dim] (caller order), built on first use (prune.hip)
    mutable float *centroid = nullptr;
    mutable uint32_t centroid_n = 0;
    // refcos tile kernel (refcos.hip): the segments ordered by length, built on first use (the 4 x 8 pairs an 8-lane group
    // accumulates then end within a few blocks of each other); results stay in the caller's order
    mutable uint32_t *len_order = nullptr;
    mutable uint32_t len_order_n = 0;
    // refcos integer filter (refcos_q8.hip): every value as three signed bytes of a 23-bit fixed-point number scaled per
    // segment, rows padded with zeros to a common length, built on first use.  Row g, group k of 32 elements: 128 bytes =
    // [digit 1: 32 bytes][digit 2][digit 3][32 bytes of zeros].
    mutable int8_t *q8 = nullptr;         // [q8_rows][q8_groups][128]
    mutable double *q8_info = nullptr;    // [q8_rows][4]: error mass A1, scale 2^-E, length term A3, 0 (refcos_q8.hip)
    mutable uint32_t q8_rows = 0;         // n rounded up to the 128 segments of a tile (rows beyond n: zeros)
    mutable uint32_t q8_groups = 0;
    mutable int q8_state = 0;             // 0: not tried, 1: built, -1: this set cannot take the integer filter
};

}  // namespace ssym

struct ssym_ctx {
    int device = 0;
    int metric = SSYM_METRIC_DTW;
    int dtype = SSYM_DTYPE_F32;
    int band = -1;
    int squared = 0;
    bool prune_default = false;     // ssym_config.dtw_prune
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string err;
    ssym_timings timings{};
    int num_cus = 256;
    // scratch (grown on demand, reused across calls)
    ssym::DeviceBuf handoff;    // dtw filter: per-wave row hand-off between row-block passes
    ssym::DeviceBuf cmat;       // dtw filter costs f32 [n_pad][m_pad]  /  refcos sims f64
    ssym::DeviceBuf tmin;       // per-target stage-1 threshold (smallest worst-case upper key bound, f64 bits)
    ssym::DeviceBuf tmin2;      // per-target stage-2 threshold (per-pair certificates)
    ssym::DeviceBuf cand;       // candidate pairs (uint2) + counter + overflow flag (list 1)
    ssym::DeviceBuf selmask;    // stage-1 hit masks, one u64 per (64-source chunk, target)
    ssym::DeviceBuf selcnt;     // stage-1 per-target counts / segment starts / fill cursors
    ssym::DeviceBuf cand2;      // list 2: pairs that survive the per-pair certificates
    ssym::DeviceBuf cand_xmin;  // certificate (smallest cell) per list-1 pair
    ssym::DeviceBuf cand_cost;  // exact f64 cost per candidate
    ssym::DeviceBuf best;       // per-target best bits / idx
    ssym::DeviceBuf topk;       // top-k rounds: previous round's key bits / index per target
    ssym::DeviceBuf prune_pairs, prune_cost;   // SSYM_DTW_PRUNE: candidate pair per target (+ source by target), exact costs
    ssym::DeviceBuf abandon;    // SSYM_DTW_PRUNE: per-target-slot thresholds (f32, accumulator units) + cell counter
    ssym::DeviceBuf dist;       // per-target distance (f64)
    ssym::DeviceBuf part;       // refcos partial argmin
    ssym::DeviceBuf one_ticket; // refcos_match_one_kernel: the "last workgroup" counter (zero between calls)
    ssym::DeviceBuf out_idx, out_cost;  // staging for host outputs
    ssym::DeviceBuf zeros;      // 256 bytes of 0.0: what refcos_mfma_kernel's staging DMA reads beyond a segment's end
    ssym::DeviceBuf stamps;     // refcos search: three device timestamps (start, main kernel done, all done) + header words
    double wall_clock_khz = 0;  // rate of wall_clock64() on this device (hipDeviceAttributeWallClockRate)
    // dtw_exact_pipe_kernel: 8 give-up counters, one per launch (round robin); pipe_mask = the slots this API
    // call used, read back with the call's results into ssym_timings.exact_redone
    ssym::DeviceBuf pipe_flag;
    int pipe_slot = 0;
    unsigned pipe_mask = 0;
    hipEvent_t ev[10]{};        // 0-6: phases of a match; 8-9: ssym_match_batch's pack
    float prune_swept = 1.f;               // share of the filter's cells the last pruned call swept (picks the pass height)
    unsigned long long pruned_cells = 0;   // SSYM_DTW_PRUNE: the filter's counter of the last call (host copy)
    int filter_launches = 1;               // kernel launches of the last dtw filter call (one per class of source lengths)
    unsigned long long launched_cells = 0; // unbanded filter without early abandoning: DP cells (per lane) of the last call's launches, padding included
    // set by ssym_match_batch / ssym_match_one around their internal pack: the call synchronises
    // once at its end, so the pack stages need not wait for their copies individually
    bool defer_sync = false;
    bool pack_light = false;        // set by ssym_match_batch around the pack of a handful of short dtw queries
    // set by ssym_match_sharded (comm.hip) around the phases of a source-sharded match: everything is only ENQUEUED
    // on the stream -- no host synchronisation, no read-back; phase 2 runs ONE selection attempt with `so_cap`
    // entries of room (0 = the default) and leaves the device headers of its two candidate lists in so_hdr1 /
    // so_hdr2 for the caller, who reads them after the step's single synchronisation
    bool stream_only = false;
    std::vector<void *> deferred_free;      // blocks ensure() replaced during a stream-only step: freed after its synchronisation
    uint64_t so_cap = 0;
    const uint32_t *so_hdr1 = nullptr, *so_hdr2 = nullptr;
    bool so_filter = false;         // the last stream-only phase ran the filter path (events ev[1..6] are its)
    bool so_refcos = false;         // ... ran the refcos search through the matrix pipe (so_hdr1 / so_hdr2 are its lists, ev[0..2] its events)
    // ssym_match_begin .. ssym_match_finish (two-phase match of a source-sharded run)
    struct Pending {
        bool valid = false, filter = false, has_dist = false;
        bool cand = false;          // ssym_match_candidates has scored this (dict, q): begin may prune
        bool pruned = false;        // begin ran the filter with early abandoning: finish appends the candidates
        const ssym_dict *dict = nullptr;
        const ssym_queries *q = nullptr;
        uint32_t index_base = 0;
        std::vector<double> dist_host;
        float main_ms = 0.f;
    } pending;
    // pinned staging for the small transfers of one API call (a single-query match moves a few KB in
    // four copies; from pageable memory every one of them blocks the host).  Uploads are copied here
    // and sent asynchronously; results land here and reach the caller's buffers after the call's one
    // synchronisation.  The cursor restarts when an outermost API call begins (api_depth).
    char *stage = nullptr;
    size_t stage_cap = 0, stage_cur = 0;
    int api_depth = 0;
    struct PendingD2H { void *user; const void *pinned; size_t bytes; };
    std::vector<PendingD2H> pending_d2h;
    // small-block cache for per-call segment sets (ssym_match_one packs one query per call; going
    // to the driver for every hipMalloc / hipFree -- the latter synchronises the device -- cost more
    // than the match itself).  All users run on `stream`, so reuse is stream-ordered.
    std::unordered_map<void *, size_t> live_blocks;      // pointer -> rounded size
    std::multimap<size_t, void *> free_blocks;           // rounded size -> cached pointer
    size_t free_bytes = 0;
};

struct ssym_dict {
    ssym::SegmentSet set;
    // refcos self-similarity matrix [n][n] for ssym_chain, valid while selfsim_n == set.n
    ssym::DeviceBuf selfsim;
    uint32_t selfsim_n = 0;
};

struct ssym_queries {
    ssym::SegmentSet set;
};

namespace ssym {

#define SSYM_HIP_CHECK(ctx, call)                                                         \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);              \
            return SSYM_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

int32_t ensure(ssym_ctx *ctx, DeviceBuf &b, size_t bytes);
void release_deferred(ssym_ctx *ctx);
int32_t zero_words(ssym_ctx *ctx, void *p, size_t bytes);     // a multiple of 4 bytes, zeroed by a kernel on the context's stream        // after a stream-only step's synchronisation (comm.hip)

// Every extern "C" entry point runs its body through this: the header promises that no C++ exception crosses
// the boundary (a Rust or C caller cannot unwind through it).  std::bad_alloc from the host-side containers
// becomes SSYM_E_NOMEM, anything else SSYM_E_HIP; the per-call mode flags of the context are put back.
template <class F>
inline int32_t guarded(ssym_ctx *ctx, F &&body) noexcept
{
    int32_t rc;
    const char *what = nullptr;
    try {
        return body();
    } catch (const std::bad_alloc &) {
        rc = SSYM_E_NOMEM;
        what = "out of host memory";
    } catch (const std::exception &e) {
        rc = SSYM_E_HIP;
        what = "unexpected C++ exception inside the library";
    } catch (...) {
        rc = SSYM_E_HIP;
        what = "unexpected C++ exception inside the library";
    }
    if (ctx) {
        ctx->stream_only = ctx->defer_sync = ctx->pack_light = false;
        try {
            ctx->err = what;
        } catch (...) {
        }
    }
    return rc;
}

// pack.hip
int32_t pack_segments(ssym_ctx *ctx, SegmentSet &set, const void *feats, bool feats_on_device,
                      const uint64_t *frame_offsets, uint32_t n, uint32_t dim, bool is_source);
int32_t append_segments(ssym_ctx *ctx, SegmentSet &set, const void *feats,
                        const uint64_t *frame_offsets, uint32_t n);
void free_segments(ssym_ctx *ctx, SegmentSet &set);
// cached device blocks (pack.hip): ctx may be NULL for dev_free (falls back to hipFree)
int32_t dev_alloc(ssym_ctx *ctx, void **p, size_t bytes);
void dev_free(ssym_ctx *ctx, void *p);
void dev_cache_release(ssym_ctx *ctx);
// staged host <-> device copies (pack.hip); fall back to plain hipMemcpyAsync when the transfer is large
int32_t stage_h2d(ssym_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int32_t stage_d2h(ssym_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
void stage_finish(ssym_ctx *ctx);      // after the stream is synchronised: hand the staged results over
struct StageScope {                    // one per public entry point that moves host data
    ssym_ctx *c;
    explicit StageScope(ssym_ctx *ctx) : c(ctx)
    {
        if (c && c->api_depth++ == 0) {
            c->stage_cur = 0;
            c->pending_d2h.clear();      // leftovers of a call that failed before its synchronisation
        }
    }
    ~StageScope() { if (c) --c->api_depth; }
};

// dtw_filter.hip
bool filter_supported(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt);
// a band the banded kernel cannot take: the unbanded filter runs instead and bounds the banded cost from below
bool filter_band_as_bound(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt);
// the filter's cost is only a lower bound of the pair's cost (frames wider than 42 values, or such a band)
inline bool filter_lower_bound_only(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    return (int)src.dim > filter_dim_used((int)src.dim) || filter_band_as_bound(ctx, src, tgt);
}
// abandon != NULL: early abandoning against per-target-slot thresholds in accumulator units (prune.hip)
int32_t launch_dtw_filter(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                          float *cmat /*[src.n_pad][tgt.n_pad]*/, const float *abandon = nullptr,
                          unsigned long long *colCtr = nullptr /* PRUNE: += column steps x rows per pass */,
                          const uint32_t *candSlot = nullptr /* PRUNE: per target slot, the source slot scored already */);
// prune.hip: candidate per target -> exact cost (ctx->prune_pairs / prune_cost, by target slot) ...
int32_t launch_dtw_prune_candidates(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt);
uint32_t *prune_cand_slots(ssym_ctx *ctx, const SegmentSet &tgt);   // candidate's source slot per target slot
// ... -> thresholds in ctx->abandon, from the own costs or from cost_by_target (caller's target order)
int32_t launch_dtw_prune_thresholds(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                                    const double *cost_by_target, const float **abandon_out);
// the candidates (pairs + exact costs) join list 2 behind the entries dtw_exact has filled in
int32_t launch_prune_append_known(ssym_ctx *ctx, uint32_t n_tgt);
// the common scale the (unbanded) filter runs the two sets with; builds the records if needed
int32_t ensure_filter_records(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *scale_out);

// certify.hip: smallest cell of every listed pair (per-pair error certificate)
int32_t launch_certify(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const uint32_t *candHdr,
                       const uint2 *pairs, uint32_t cap, float *xmin);

// dtw_exact.hip
// pairs == nullptr: every (s,t) pair, out[s*n_tgt + t]; else out[k] for pairs[k] with k < *count
int32_t launch_dtw_exact(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                         const uint2 *pairs, const uint32_t *count_dev, uint32_t max_pairs,
                         double *out);

// select.hip.  k_top = 1: the reference's first-minimum fold; k_top > 1: ssym_match_topk, outputs
// [n_tgt][k_top] (rounds of the same fold, each above the previous round's (key, index))
// stage 1: worst-case margin over the whole filter matrix -> ctx->cand (list 1)
int32_t launch_dtw_bounds(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                          const double *dist_dev, uint32_t k_top,       // threshold per target -> ctx->tmin
                          const double *seed_by_slot = nullptr /* exact costs of pairs known already (k_top = 1) */);
// wide frames: the filter cost is only a lower bound; the threshold is the EXACT cost of the pair the
// filter likes best per target (one exact evaluation per target) -> ctx->tmin
int32_t launch_dtw_bounds_partial(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                                  const double *seed_by_slot = nullptr, uint32_t k_top = 1,
                                  const double *dist_dev = nullptr /* per-target distances, caller's order */);
int32_t launch_dtw_select(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                          const float *cmat, const double *dist_dev, uint32_t cap);
// stage 2: per-pair intervals from the certificates of list 1 -> ctx->cand2 (list 2, same capacity)
int32_t launch_dtw_select2(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                           const float *xmin, const double *dist_dev, uint32_t cap, uint32_t k_top,
                           bool lower_bound_only = false,
                           const uint32_t *known_src = nullptr /* per target: a pair scored already, kept out */);
int32_t launch_dtw_final(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                         const double *dist_dev, uint32_t cap, uint32_t index_base, uint32_t k_top,
                         uint32_t *out_idx_dev, double *out_cost_dev);
int32_t launch_dtw_final_allpairs(ssym_ctx *ctx, uint32_t n_src, uint32_t n_tgt,
                                  const double *costs, const double *dist_dev,
                                  uint32_t index_base, uint32_t k_top, uint32_t *out_idx_dev,
                                  double *out_cost_dev);
// shard g's costs start at costs + g * cost_stride (doubles), its indices at idx + g * idx_stride (u32);
// strides 0 = n_targets (the dense [n_shards][n_targets] layout of ssym_merge_shards)
int32_t launch_merge_shards(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets,
                            const double *costs, const uint32_t *idx, const double *dist_dev,
                            uint32_t *out_idx, double *out_cost, size_t cost_stride = 0, size_t idx_stride = 0);

// capi.hip: the phases of a source-sharded match, shared with comm.hip.  With ctx->stream_only they only enqueue.
int32_t match_candidates_impl(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, double *cost_dev);
int32_t match_begin_impl(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q, const double *distance,
                         uint32_t index_base, double *bounds_dev, const double *prune_cost_dev);
int32_t match_finish_impl(ssym_ctx *ctx, const double *bounds_dev, uint32_t *out_idx, double *out_cost, uint32_t flags);

// chain.hip
int32_t launch_chain_argmin(ssym_ctx *ctx, const double *base, size_t row_stride, const uint32_t *row_sel,
                            uint32_t n, double distance, double init, bool report_value, uint32_t step,
                            uint32_t *cur, uint32_t *out_idx, double *out_cost);
int32_t launch_chain_pairs(ssym_ctx *ctx, const uint32_t *cur, uint32_t n, uint2 *pairs, uint32_t *count);

// refcos.hip
int32_t launch_refcos_sims(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                           double *sims /*[n_src][n_tgt]*/);
// one query, one launch (ssym_match_one with the refcos metric)
bool refcos_few_supported(const ssym_ctx *ctx, const SegmentSet &src, const uint64_t *q_off, uint32_t n_queries);
int32_t launch_refcos_match_few(ssym_ctx *ctx, const SegmentSet &src, const void *queries, const uint64_t *q_off,
                                uint32_t n_queries, const double *distances, double default_dist, double *out_val,
                                uint32_t *out_idx, unsigned long long *stamps = nullptr);
bool dtw_few_supported(const ssym_ctx *ctx, const SegmentSet &src, const uint64_t *q_off, uint32_t n_queries);
int32_t launch_dtw_match_few(ssym_ctx *ctx, const SegmentSet &src, const void *queries, const uint64_t *q_off,
                             uint32_t n_queries, const double *distances, double *out_cost, uint32_t *out_idx);
// refcos_mfma.hip: the search through the f64 matrix pipe (filter) + exact keys of the candidates; bit-exact results
bool refcos_mfma_supported(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt);
// the int8 records of a set (built on first use); true when BOTH sets have them and the pair of sets fits the kernel
int32_t refcos_q8_ensure(ssym_ctx *ctx, const SegmentSet &set);
bool refcos_q8_ready(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt);
void refcos_q8_release(ssym_ctx *ctx, const SegmentSet &set);
// the integer filter's main kernel: thresholds and list 1 in the layout of refcos_mfma.hip (sims: ssym_pair_matrix(exact = 3))
int32_t launch_refcos_q8_kernel(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const double *dist_dev,
                                unsigned long long *thr, uint32_t *hdr1, void *list1, uint32_t cap, uint32_t k_top,
                                double *sims);
int32_t launch_refcos_match_mfma(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const double *dist_dev,
                                 uint32_t index_base, uint32_t *out_idx_dev, double *out_cost_dev,
                                 const uint32_t **list1_hdr, const uint32_t **list2_hdr, uint32_t k_top = 1, bool integer_filter = false,
                                 unsigned long long *stamps = nullptr, uint32_t *tail = nullptr);
int32_t launch_refcos_mfma_sims(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *sims,
                                bool integer_filter = false);
char *stage_take(ssym_ctx *ctx, size_t bytes);      // pack.hip: room in the call's pinned window (NULL: none)
int32_t launch_refcos_argmin(ssym_ctx *ctx, uint32_t n_src, uint32_t n_tgt, const double *sims,
                             const double *dist_dev, uint32_t index_base, uint32_t k_top,
                             uint32_t *out_idx_dev, double *out_cost_dev);

}  // namespace ssym
