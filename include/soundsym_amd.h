/*
 * soundsym_amd.h -- C ABI of the MI355X-native segment-distance matcher.
 *
 * This is the drop-in boundary for ONE hot path of andrewcsmith/soundsym: the all-pairs
 * segment-distance search + per-target argmin.  The reference has no FFI of its own (pure Rust,
 * no `extern "C"`); every entry point below names the reference interface it replaces so that a
 * maintainer can bind it 1:1 from Rust (`INTEGRATION.md` shows the binding).  Reference citations
 * are relative to the upstream repository root.
 *
 * Conventions
 *   - plain C: opaque handles, plain pointers and sizes, no C++ types, no exceptions;
 *   - every function returns an int32 status (0 = SSYM_OK, negative = error); the message for the
 *     last failure on a context is ssym_last_error(ctx);
 *   - a "segment" is what the reference calls a Sound inside a SoundDictionary / SoundSequence:
 *     `frames x dim` feature values, frame-major, contiguous (Sound::mfccs(), src/sound.rs:189-193;
 *     segment slicing src/sound.rs:330-343).  A set of segments is handed over as ONE flat value
 *     buffer plus `n+1` FRAME offsets: segment i = values [off[i]*dim, off[i+1]*dim);
 *   - the library COPIES caller memory at create time (the caller keeps ownership, like the Rust
 *     side keeps its Vec<Arc<Sound>>); results are indices into the dictionary, the Rust side
 *     then does `dict.sounds[idx].clone()` exactly as src/sound.rs:369 does;
 *   - calls are synchronous on the caller's thread, like the reference (no threads there).  One
 *     context per thread, or an external lock.  A context owns one HIP stream on one GPU;
 *   - there is NO CPU fallback: without a usable gfx950 device ssym_ctx_create fails with
 *     SSYM_E_NO_DEVICE.
 *
 * Metric modes (SURVEY.md section 0 / DESIGN.md)
 *   SSYM_METRIC_REFCOS  the reference's own segment distance: cosine_sim (src/sound.rs:22-33,
 *                       prefix dot / product of squared norms, f64) searched by at_distance
 *                       (src/sound.rs:351-370): argmin_i |sim_i - distance|, first minimum wins,
 *                       fold start (0, 2.0).  Arithmetic order follows the reference bit for bit.
 *   SSYM_METRIC_DTW     dynamic time warping over frame-wise L2 local costs with min-of-three
 *                       recurrence and optional Sakoe-Chiba band (not in the reference; defined
 *                       in DESIGN.md).  argmin_i |cost_i - distance| (distance NULL = 0), first
 *                       minimum wins, fold start (0, +inf).
 */
#ifndef SOUNDSYM_AMD_H
#define SOUNDSYM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSYM_ABI_VERSION 3

#if defined(__GNUC__)
#define SSYM_API __attribute__((visibility("default")))
#else
#define SSYM_API
#endif

typedef struct ssym_ctx ssym_ctx;         /* one GPU + one stream + scratch                      */
typedef struct ssym_dict ssym_dict;       /* SoundDictionary's feature side (src/sound.rs:290)   */
typedef struct ssym_queries ssym_queries; /* the targets of one batch (SoundSequence::sounds)    */
typedef struct ssym_samples ssym_samples; /* the dictionary sounds' SAMPLES, resident on the GPU   */
typedef struct ssym_comm ssym_comm;       /* one rank of a source-sharded run: an RCCL communicator  */

enum {
    SSYM_OK = 0,
    SSYM_E_INVALID = -1,     /* bad argument (NULL, dim mismatch, non-monotonic offsets, ...)    */
    SSYM_E_EMPTY_DICT = -2,  /* replaces the reference's panic at src/sound.rs:369               */
    SSYM_E_NO_DEVICE = -3,   /* no usable gfx950 device / HIP runtime -- there is no CPU path    */
    SSYM_E_HIP = -4,         /* a HIP call failed; see ssym_last_error                           */
    SSYM_E_NOMEM = -5,
    SSYM_E_UNSUPPORTED = -6, /* shape outside what the kernels handle (see DESIGN.md limits)     */
    SSYM_E_TIMEOUT = -7,     /* ssym_match_sharded: a rank did not arrive within the communicator's
                                deadline; the communicator has been ABORTED (ncclCommAbort) and can
                                only be destroyed                                                */
    SSYM_E_COMM = -8         /* the communicator is dead (aborted by an earlier failure on this or
                                another rank, or RCCL reported an asynchronous error)            */
};

enum { SSYM_METRIC_REFCOS = 0, SSYM_METRIC_DTW = 1 };
enum { SSYM_DTYPE_F64 = 0, SSYM_DTYPE_F32 = 1 };

/* flags for ssym_match_queries */
enum {
    SSYM_OUT_DEVICE = 1u,      /* out_idx / out_cost are device pointers on ctx's GPU            */
    SSYM_DTW_FORCE_EXACT = 2u, /* skip the f32 MFMA filter: exact f64 kernel on every pair       */
    SSYM_DTW_PRUNE = 4u        /* dtw first-minimum search (k = 1, no per-target distances): one
                                  candidate pair per target is scored first and the filter
                                  abandons pairs that are provably above it; same indices and
                                  costs, the time then depends on the data; ignored where it does
                                  not apply                                                       */
};

typedef struct ssym_config {
    uint32_t struct_size;  /* = sizeof(ssym_config)                                              */
    int32_t device;        /* HIP device ordinal                                                 */
    int32_t metric;        /* SSYM_METRIC_*                                                      */
    int32_t dtype;         /* SSYM_DTYPE_* of every feature buffer handed to this context        */
    int32_t band;          /* dtw: Sakoe-Chiba radius in frames, -1 = none                       */
    int32_t dtw_squared;   /* dtw: 0 = L2 local cost (default), 1 = squared L2                   */
    void *stream;          /* hipStream_t to enqueue on; NULL = the library creates one          */
    int32_t dtw_prune;     /* dtw: 1 = every batched match of at least 64 targets runs as if          */
                           /* SSYM_DTW_PRUNE were passed (the entry points without a flags argument,  */
                           /* ssym_match_batch, included); 0 = only where the flag is given           */
    int32_t reserved;      /* 0                                                                  */
} ssym_config;

/* Per-phase device time of the LAST ssym_match_* call on the context, measured with HIP events
 * recorded on the context's stream (milliseconds; 0 when a phase did not run).  One exception: a refcos search
 * through a filter (refcos_filter 1 or 2) outside ssym_match_sharded is timed by the device's wall clock, read by
 * its first kernel, the first kernel after the main one and its last kernel -- an event record between two kernels
 * costs several microseconds of gap on the stream, which a search of 0.2 ms notices. */
typedef struct ssym_timings {
    float pack_ms;      /* target packing (only when the call packed targets itself)             */
    float main_ms;      /* dtw: MFMA filter kernel / refcos: similarity tile kernel              */
    float select_ms;    /* dtw: bounds, two-stage candidate selection, certificates              */
    float refine_ms;    /* dtw: exact f64 re-scoring of candidates (or of every pair)            */
    float reduce_ms;    /* final per-target argmin                                               */
    float total_ms;     /* first event to last event                                             */
    uint64_t n_pairs;   /* n_sources * n_targets of the call                                     */
    uint64_t n_refined; /* dtw: pairs re-scored exactly                                          */
    int32_t main_launches; /* kernel launches that made up main_ms                               */
    int32_t used_filter;   /* dtw: 1 = MFMA filter + refine, 0 = exact kernel on every pair      */
    float prune_ms;        /* SSYM_DTW_PRUNE: candidate search + exact scores + thresholds        */
    int32_t pruned;        /* 1 = the filter ran with early abandoning                           */
    uint64_t n_filter_cells; /* dtw, unbanded filter: DP cells it evaluated, padding included (pruned runs: counted by
                                the kernel; full runs: from the launches' geometry); 0 elsewhere   */
    float collective_ms;   /* ssym_match_sharded: the RCCL all-reduce(s) + all-gather, device time  */
    int32_t attempts;      /* ssym_match_sharded: selection attempts of the step (1 unless a rank's
                              candidate list overflowed and every rank redid the tail)             */
    int32_t exact_redone;  /* dtw: launches of the exact kernel whose wave-pipelined variant gave up
                              waiting and whose list was scored again by the plain variant (0 in any
                              healthy run; results are correct either way)                          */
    int32_t refcos_filter; /* refcos searches: 0 = the exact tile kernel on every pair, 1 = the f64 matrix-pipe
                              filter, 2 = the integer (i8 matrix-pipe) filter; exact keys on the candidates
                              either way (this word was `reserved` before: same size and place)        */
} ssym_timings;

SSYM_API int32_t ssym_abi_version(void);

/* Context ------------------------------------------------------------------------------------ */
SSYM_API int32_t ssym_ctx_create(const ssym_config *cfg, ssym_ctx **out);
SSYM_API int32_t ssym_ctx_destroy(ssym_ctx *ctx);
SSYM_API const char *ssym_last_error(const ssym_ctx *ctx); /* ctx may be NULL: last create failure       */
SSYM_API int32_t ssym_ctx_synchronize(ssym_ctx *ctx);
SSYM_API int32_t ssym_get_timings(const ssym_ctx *ctx, ssym_timings *out);

/* Dictionary: replaces SoundDictionary::{new, from_segments, add_segments}
 * (src/sound.rs:296, 323, 330) as far as features are concerned.
 *   feats          flat values, dtype per cfg.dtype (HOST memory)
 *   frame_offsets  n_segments + 1 offsets in FRAMES, non-decreasing, frame_offsets[0] may be > 0
 *   dim            values per frame (NCOEFFS = 12 in the reference, src/lib.rs:22)
 * n_segments = 0 creates an empty dictionary (SoundDictionary::new); matching against it fails
 * with SSYM_E_EMPTY_DICT. */
SSYM_API int32_t ssym_dict_create(ssym_ctx *ctx, const void *feats, const uint64_t *frame_offsets,
                         uint32_t n_segments, uint32_t dim, ssym_dict **out);
/* Same, but `feats` is a DEVICE pointer on the context's GPU (offsets stay on the host). */
SSYM_API int32_t ssym_dict_create_device(ssym_ctx *ctx, const void *feats_dev,
                                const uint64_t *frame_offsets, uint32_t n_segments, uint32_t dim,
                                ssym_dict **out);
/* add_segments (src/sound.rs:330): appended segments get the next indices. */
SSYM_API int32_t ssym_dict_append(ssym_ctx *ctx, ssym_dict *dict, const void *feats,
                         const uint64_t *frame_offsets, uint32_t n_segments);
SSYM_API int32_t ssym_dict_size(const ssym_dict *dict, uint32_t *out_n_segments);
SSYM_API int32_t ssym_dict_destroy(ssym_ctx *ctx, ssym_dict *dict);

/* Targets of one batch, made resident once (the `for sound in self.sounds` side of
 * clone_from_dictionary, src/sound.rs:453). */
SSYM_API int32_t ssym_queries_create(ssym_ctx *ctx, const void *feats, const uint64_t *frame_offsets,
                            uint32_t n_targets, uint32_t dim, ssym_queries **out);
SSYM_API int32_t ssym_queries_create_device(ssym_ctx *ctx, const void *feats_dev,
                                   const uint64_t *frame_offsets, uint32_t n_targets,
                                   uint32_t dim, ssym_queries **out);
SSYM_API int32_t ssym_queries_destroy(ssym_ctx *ctx, ssym_queries *q);

/* The hot path.  Replaces the loop of clone_from_dictionary (src/sound.rs:451-455: one
 * match_sound per target) and of morph_to (src/sound.rs:440-446: one at_distance per target).
 *   distance    NULL: 1.0 per target in refcos (match_sound, src/sound.rs:346-348), 0.0 in dtw;
 *               else n_targets values in HOST memory (morph_to's distances)
 *   index_base  added to every returned index (a rank holding the source shard [base, base+n)
 *               returns global indices)
 *   out_idx     n_targets u32: chosen dictionary index per target (+ index_base)
 *   out_cost    nullable, n_targets f64: refcos -> the winning |sim - distance| (the reference's
 *               discarded `min_distance`, src/sound.rs:361-368); dtw -> the winner's DTW cost.
 *               When nothing beats the fold start the index is 0 (+ index_base) and the value is
 *               the fold start (2.0 / +inf), as in src/sound.rs:361-367.
 *   flags       SSYM_OUT_DEVICE, SSYM_DTW_FORCE_EXACT
 * Returns after the results are written (the stream is synchronised). */
SSYM_API int32_t ssym_match_queries(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                           const double *distance, uint32_t index_base, uint32_t *out_idx,
                           double *out_cost, uint32_t flags);

/* The k best dictionary entries per target ("top-k candidates"; the
 * reference itself only ever takes the first: at_distance, src/sound.rs:351-370).  Same inputs as
 * ssym_match_queries; entry r of target t is at [t * k + r].  Entries are ordered by
 * (|value - distance|, index) ascending, value = cosine_sim in refcos, DTW cost in dtw -- so entry 0
 * is ssym_match_queries' answer whenever anything beats the fold start (key < 2.0 in refcos,
 * finite cost in dtw; NaN keys never enter, as in src/sound.rs:362).  When fewer than k entries
 * qualify the rest of the row is SSYM_NO_MATCH with cost NaN.  out_cost as in ssym_match_queries
 * (refcos: the key |sim - distance|; dtw: the cost).  1 <= k <= SSYM_TOPK_MAX.  dtw results are
 * those of an exact f64 evaluation of every pair, as for k = 1. */
#define SSYM_TOPK_MAX 64u
#define SSYM_NO_MATCH 0xffffffffu
SSYM_API int32_t ssym_match_topk(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                        const double *distance, uint32_t k, uint32_t index_base, uint32_t *out_idx,
                        double *out_cost, uint32_t flags);

/* Convenience: pack host targets, match, release.  Same contract as ssym_match_queries with
 * host outputs. */
SSYM_API int32_t ssym_match_batch(ssym_ctx *ctx, const ssym_dict *dict, const void *tgt_feats,
                         const uint64_t *tgt_frame_offsets, uint32_t n_targets,
                         const double *distance, uint32_t *out_idx, double *out_cost);

/* One query: SoundDictionary::at_distance(distance, other) (src/sound.rs:351) /
 * match_sound(other) (src/sound.rs:346, pass distance = 1.0 in refcos). */
SSYM_API int32_t ssym_match_one(ssym_ctx *ctx, const ssym_dict *dict, const void *feats,
                       uint64_t n_frames, double distance, uint32_t *out_idx, double *out_cost);

/* SoundSequence::from_distances (src/sound.rs:405-417): starting from `start`, step i matches the
 * previous step's result (the start sound for i = 0) with at_distance(distances[i], .) and the
 * match becomes the next query.  out_idx[i] / out_cost[i] (nullable) are step i's result, as
 * ssym_match_one would return them.  The chain runs on the device without a host round trip per
 * step: in refcos the later steps are row lookups in the dictionary's self-similarity matrix,
 * which is computed on first use and kept with the dictionary (hence the non-const handle;
 * n^2 f64 of device memory) until ssym_dict_append changes it; in dtw every step re-scores the
 * dictionary against the current entry with the exact f64 kernel. */
SSYM_API int32_t ssym_chain(ssym_ctx *ctx, ssym_dict *dict, const void *start_feats, uint64_t start_frames,
                   const double *distances, uint32_t n_steps, uint32_t *out_idx, double *out_cost);

/* The whole [n_sources][n_targets] matrix in HOST memory, row-major, f64:
 *   refcos: cosine_sim(source, target) (src/sound.rs:22-33), bit for bit (exact = 0 or 1); exact = 2 -> the
 *           similarities the f64 matrix pipe's dots give (the filter of the refcos search: FMA chains in another
 *           summation order, within (3 L + 16) 2^-53 sqrt(norm(me) norm(you)) / nrm of the reference's);
 *           exact = 3 -> the similarities of the integer filter (the refcos search's default filter: every value as a
 *           23-bit fixed-point number per segment, exact integer products; SSYM_E_UNSUPPORTED where that filter does
 *           not take the sets, e.g. values that are not finite);
 *   dtw:    exact = 0 -> the f32 MFMA filter's costs (frames wider than 42 values: of their first 42
 *           values only; a Sakoe-Chiba band beyond the banded kernel, r > 47: the unbanded cost --
 *           either way a lower bound of every pair's cost); exact = 1 -> the exact f64 costs. */
SSYM_API int32_t ssym_pair_matrix(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                         int32_t exact, double *out_matrix);

/* Source-sharded multi-GPU, dtw metric: the one real exchange the path has.  Each rank's filter gives,
 * per target, an upper bound on the best key in ITS shard; a rank whose shard does not hold a
 * target's neighbour would otherwise re-score ~10^2 of its own pairs per target for nothing.
 *   ssym_match_begin   runs the filter and writes the per-target bound to bounds_dev (n_targets f64,
 *                      DEVICE memory of the caller, e.g. a torch tensor); the stream is synchronised
 *   (caller)           all-reduce(MIN) of bounds_dev over the ranks (RCCL; n_targets * 8 bytes)
 *   ssym_match_finish  selects candidates against the reduced bounds, re-scores them exactly and
 *                      writes what ssym_match_queries would (index 0 + base / +inf for a target
 *                      none of whose pairs in this shard can win -- ssym_merge_shards then takes
 *                      another shard's entry)
 * dict / q / distance must stay alive between the two calls; flags as for ssym_match_queries.
 * Any other matching call on the context in between (it would use the same scratch) ends the pair:
 * ssym_match_finish then fails with "without ssym_match_begin".
 * Where the filter does not apply (refcos, shapes outside its limits) begin writes +inf and finish
 * is a plain ssym_match_queries, so callers need no second code path.  With one rank, or without
 * the all-reduce, the pair is equivalent to ssym_match_queries. */
SSYM_API int32_t ssym_match_begin(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                         const double *distance, uint32_t index_base, double *bounds_dev);
SSYM_API int32_t ssym_match_finish(ssym_ctx *ctx, const double *bounds_dev, uint32_t *out_idx,
                          double *out_cost, uint32_t flags);

/* Early abandoning (SSYM_DTW_PRUNE) in a source-sharded run: only the rank that holds a target's
 * neighbour knows a tight bound before the filter, so the candidates' costs are exchanged first.
 *   ssym_match_candidates    scores this shard's candidate pair per target exactly and writes the costs
 *                            to cost_dev (n_targets f64, DEVICE memory; +inf where pruning does not
 *                            apply: refcos, frames wider than 42 values, shapes outside the filter)
 *   (caller)                 all-reduce(MIN) of cost_dev over the ranks
 *   ssym_match_begin_pruned  ssym_match_begin (no per-target distances) whose filter abandons pairs
 *                            that are provably above the reduced costs; then the all-reduce of the
 *                            bounds and ssym_match_finish as before.  Without a preceding
 *                            ssym_match_candidates on the same (dict, q) it is a plain ssym_match_begin.
 * Results are those of the unpruned sequence, bit for bit. */
SSYM_API int32_t ssym_match_candidates(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                                       double *cost_dev);
SSYM_API int32_t ssym_match_begin_pruned(ssym_ctx *ctx, const ssym_dict *dict, const ssym_queries *q,
                                         uint32_t index_base, const double *cost_dev, double *bounds_dev);

/* Source-sharded multi-GPU: after an all-gather of every shard's (cost, global index) per target
 * (n_shards x n_targets each, shard-major, DEVICE memory), pick per target the shard entry with
 * the smallest cost, lowest global index on equal cost -- the same first-minimum rule as
 * src/sound.rs:361-367 because shards are ordered by index.  Outputs are DEVICE memory. */
SSYM_API int32_t ssym_merge_shards(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets,
                          const double *costs_dev, const uint32_t *idx_dev, uint32_t *out_idx_dev,
                          double *out_cost_dev);
/* The same for matches made with per-target distances (morph_to, src/sound.rs:440-446): the shards
 * folded on |cost - distance|, so the merge does too.  `distance`: n_targets f64 in HOST memory, or
 * NULL (= ssym_merge_shards).  dtw costs only: refcos shards report the key itself, which
 * ssym_merge_shards already compares correctly. */
SSYM_API int32_t ssym_merge_shards_at(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets,
                             const double *costs_dev, const uint32_t *idx_dev, const double *distance,
                             uint32_t *out_idx_dev, double *out_cost_dev);

/* Source-sharded multi-GPU with the collectives INSIDE the library (RCCL over xGMI, enqueued on the context's
 * stream): the north star's "sharding the source-segment axis with an RCCL all-gather of per-target argmin
 * indices".  The reference has no counterpart -- its loop (src/sound.rs:451-455) is serial -- so these entry
 * points replace that loop for a dictionary that is split over the GPUs of one node, one process (or thread)
 * per GPU, each with its own context:
 *   ssym_comm_unique_id  rank 0 obtains the 128-byte RCCL id (ncclGetUniqueId) and hands it to the other
 *                        ranks by any means the host has (a file, a pipe, MPI, torch.distributed ...)
 *   ssym_comm_create     ncclCommInitRank on the context's device; collective over all `world` ranks
 *   ssym_match_sharded   rank g holds the dictionary shard [index_base, index_base + n) and ALL targets
 *                        (the same targets, in the same order, on every rank).  One step =
 *                          filter over the shard                      (ssym_match_begin, no host sync)
 *                          ncclAllReduce(MIN) of the M per-target bounds            (M x 8 bytes)
 *                          selection, exact re-scoring, fold                (ssym_match_finish, no host sync)
 *                          ncclAllGather of (cost f64, global index u32) per target + list status (12 M + 8 bytes)
 *                          merge: smallest key, lowest global index on ties      (ssym_merge_shards_at)
 *                        all enqueued back to back on the context's stream; the host synchronises ONCE, at the
 *                        end, and every rank returns the same, complete answer -- bit for bit what
 *                        ssym_match_queries returns for the unsharded dictionary.  Should a rank's candidate list
 *                        overflow (the gathered status says so to everyone) all ranks repeat the tail once with
 *                        the room asked for.  With SSYM_DTW_PRUNE the candidates' costs are all-reduced first
 *                        (ssym_match_candidates / ssym_match_begin_pruned).  An empty local shard takes part and
 *                        reports the fold start; a dictionary that is empty on EVERY rank is the caller's to
 *                        reject (the reference panics, src/sound.rs:369).
 *                        out_idx / out_cost / flags as for ssym_match_queries (SSYM_OUT_DEVICE honoured).
 * FAILURE on one rank is part of the protocol (the reference fails on its one thread, src/sound.rs:369,440-449; a
 * sharded replacement has to fail on ALL ranks, and may never leave a rank waiting):
 *   - a rank whose LOCAL work fails (out of memory, a HIP error, a shape the kernels reject, a C++ exception) still
 *     takes part in every collective of the step with neutral blocks and puts its status code into the block the
 *     all-gather carries anyway; after the step's one synchronisation EVERY rank returns that same code (the lowest
 *     failing rank's), ssym_last_error names the rank and the phase, and the communicator stays usable;
 *   - a rank that cannot take part at all (it cannot allocate its exchange buffers, a collective cannot be enqueued,
 *     its caller never makes the call) leaves its peers waiting: they wait under a DEADLINE (ssym_comm_set_timeout,
 *     default 60 s, $SSYM_COMM_TIMEOUT_MS), then abort their communicator (ncclCommAbort) and return
 *     SSYM_E_TIMEOUT; the rank that could not take part aborts its own and returns its error.  An aborted communicator
 *     answers every further call with SSYM_E_COMM and can only be destroyed; RCCL's asynchronous errors
 *     (ncclCommGetAsyncError) end the wait the same way.  After an abort the context's stream is drained under a
 *     second, short deadline (5 s): that ncclCommAbort makes the queued collectives leave the stream has been seen with a
 *     world-1 communicator only -- no run over more than one GPU exists, so the behaviour at world > 1 is UNVERIFIED; a
 *     stream that does not drain is reported in ssym_last_error ("did not drain"), never waited for without bound, and
 *     the context should then be destroyed.
 * RCCL is looked up at run time (symbols already in the process, else librccl.so.1 / $SSYM_RCCL_LIB), so a
 * single-GPU user needs no RCCL at all; without it the three calls fail with SSYM_E_UNSUPPORTED. */
#define SSYM_COMM_ID_BYTES 128
SSYM_API int32_t ssym_comm_unique_id(void *out_id /* SSYM_COMM_ID_BYTES */);
SSYM_API int32_t ssym_comm_create(ssym_ctx *ctx, const void *id, int32_t rank, int32_t world, ssym_comm **out);
SSYM_API int32_t ssym_comm_destroy(ssym_ctx *ctx, ssym_comm *comm);
SSYM_API int32_t ssym_match_sharded(ssym_ctx *ctx, ssym_comm *comm, const ssym_dict *dict, const ssym_queries *q,
                                    const double *distance, uint32_t index_base, uint32_t *out_idx,
                                    double *out_cost, uint32_t flags);
/* 1 when this library could bind RCCL (every symbol it calls), 0 otherwise: what the ranks of a job agree on BEFORE
 * any of them enters ssym_comm_create (a rank without RCCL would leave the others inside ncclCommInitRank). */
SSYM_API int32_t ssym_comm_available(void);
/* Deadline of one ssym_match_sharded step in milliseconds (> 0); see the failure rules above. */
SSYM_API int32_t ssym_comm_set_timeout(ssym_comm *comm, int64_t milliseconds);
/* 1 = the communicator was aborted (every call on it fails with SSYM_E_COMM), 0 = usable. */
SSYM_API int32_t ssym_comm_is_dead(const ssym_comm *comm);
/* TEST AND MEASUREMENT HOOKS.  Both refuse with SSYM_E_UNSUPPORTED unless the calling process has SSYM_TEST_HOOKS=1 in its
 * environment at the time of the call: nothing a production caller can trip over.  The same variable gates the library's
 * measurement knobs (SSYM_FILTER_*, SSYM_REFCOS_*, SSYM_EXACT_*, SSYM_CELLS_*, SSYM_PRUNE_NT: A/B switches between kernels
 * that return the same bits): without it none of them is read.  SSYM_COMM_TIMEOUT_MS and SSYM_RCCL_LIB
 * are configuration and always honoured.
 *
 * Fault injection for the containment tests (tests/test_gpu_comm.py), one shot: the NEXT ssym_match_sharded on this
 * communicator fails in `phase` (1 = the filter phase before the bound exchange, 2 = selection / re-scoring before the
 * gather).  kind 0: the local work reports SSYM_E_NOMEM (the rank takes part, every rank returns SSYM_E_NOMEM);
 * kind 1: the rank leaves the step there without its collectives (its peers meet the deadline). */
SSYM_API int32_t ssym_comm_inject_fault(ssym_comm *comm, int32_t phase, int32_t kind);
/* Replay of a larger world on one GPU (bench.py --replay-world): from now on the per-target bounds every
 * ssym_match_sharded step agrees on are the element-wise minimum of the all-reduce's result and these `n` device doubles
 * -- the bounds ssym_match_begin returns for the FULL dictionary are exactly what the ranks of a run over its shards
 * would have all-reduced -- so that a one-rank step on one shard selects and re-scores what that rank would in the
 * larger run.  The buffer stays the caller's and must outlive the steps; n = the steps' number of targets; NULL clears. */
SSYM_API int32_t ssym_comm_replay_bounds(ssym_comm *comm, const double *bounds_dev, uint32_t n);

/* The ranks of ONE process (a thread per rank, every rank its own context, on one GPU or several) without RCCL:
 * the same ssym_match_sharded, its two exchanges done with host barriers around device copies instead of
 * stream-ordered RCCL calls.  RCCL refuses two ranks on one device; this transport is how the multi-rank logic
 * (gather layout, merge over G shards, the agreed repeat after an overflow, empty shards) is exercised on a one-GPU
 * box (tests/test_gpu_comm.py).  Every rank's thread must be inside its ssym_match_sharded call at the same time. */
typedef struct ssym_local_group ssym_local_group;
SSYM_API int32_t ssym_local_group_create(int32_t world, ssym_local_group **out);
SSYM_API int32_t ssym_local_group_destroy(ssym_local_group *group);
SSYM_API int32_t ssym_comm_create_local(ssym_ctx *ctx, ssym_local_group *group, int32_t rank, ssym_comm **out);

/* Reconstruction tail (the step right after the hot path): the samples of every dictionary sound,
 * resident on the GPU (Sound::samples(), src/sound.rs:181; `sample_offsets` = n_sounds+1 SAMPLE
 * offsets into `samples`, f64, HOST memory). */
SSYM_API int32_t ssym_samples_create(ssym_ctx *ctx, const double *samples, const uint64_t *sample_offsets,
                            uint32_t n_sounds, ssym_samples **out);
SSYM_API int32_t ssym_samples_destroy(ssym_ctx *ctx, ssym_samples *s);

/* clone_from_dictionary's length fit (src/sound.rs:456-465) + to_sound's concatenation (:475-480):
 * for target t the samples of dictionary sound idx[t], zero-padded or truncated to
 * out_offsets[t+1]-out_offsets[t] samples, written at out_offsets[t].
 *   idx          n_targets dictionary indices (HOST), as returned by ssym_match_*
 *   out_offsets  n_targets+1 SAMPLE offsets of the output (HOST); out_offsets[0] must be 0
 *   out_samples  nullable, out_offsets[n] f64 (HOST)
 *   out_pcm32    nullable, out_offsets[n] i32 (HOST): Sound::write_file's conversion
 *                `(i32::MAX as f64 * sample) as i32` (src/sound.rs:139; truncating, saturating,
 *                NaN -> 0) */
SSYM_API int32_t ssym_reconstruct(ssym_ctx *ctx, const ssym_samples *s, const uint32_t *idx,
                         const uint64_t *out_offsets, uint32_t n_targets, double *out_samples,
                         int32_t *out_pcm32);

/* Feature front-end, the step before the hot path: what
 * Sound::from_samples(.., None, ..) computes through analyze_mfccs (src/sound.rs:215-242) --
 * 1024-sample Hanning windows hopped by 256 (src/lib.rs:24-25), per window `n_coeffs` MFCCs between
 * f_lo and f_hi Hz (the reference: 12, 100..8000, src/sound.rs:218), frame-major.
 * PARITY UNPINNED: the reference's arithmetic is in un-vendored crates (vox_box, sample); this is a
 * self-consistent extractor whose definition is in csrc/mfcc.hip and DESIGN.md.
 *   samples      n_samples f64 (HOST)
 *   flags        SSYM_MFCC_PAD_TAIL: n_samples / 256 frames, samples past the end read as 0
 *                (default: full windows only, (n_samples - 1024) / 256 + 1);
 *                SSYM_OUT_DEVICE: out_mfccs is a device pointer (feeds ssym_*_create_device)
 *   out_mfccs    frames * n_coeffs f64, frames as ssym_mfcc_num_frames reports
 *   out_mean     nullable, n_coeffs f64 (HOST): analyze_mean_mfccs (src/sound.rs:271-286) */
#define SSYM_MFCC_BIN 1024
#define SSYM_MFCC_HOP 256
#define SSYM_MFCC_PAD_TAIL 4u
SSYM_API int32_t ssym_mfcc_num_frames(uint64_t n_samples, uint32_t flags, uint64_t *out_frames);
SSYM_API int32_t ssym_mfcc(ssym_ctx *ctx, const double *samples, uint64_t n_samples, double sample_rate,
                  uint32_t n_coeffs, double f_lo, double f_hi, uint32_t flags, double *out_mfccs,
                  double *out_mean);

#ifdef __cplusplus
}
#endif
#endif /* SOUNDSYM_AMD_H */
