// prune.hip -- per-target thresholds for the early-abandoning dtw filter (SSYM_DTW_PRUNE).
//
// The first-minimum search (SoundDictionary::at_distance, src/sound.rs:351-370, with the dtw metric
// and distance 0) only needs the cost of pairs that can still win.  Before the filter runs, ONE
// candidate source per target is scored exactly; its cost bounds the target's minimum from above, and
// the filter kernel drops a wave's task as soon as every one of its 64 pairs is provably above its
// target's bound (dtw_filter_kernel.hpp, PRUNE).  The result is the same index and the same cost as
// without pruning, whatever the candidates are -- they only decide how early the filter can stop:
//
//   1. centroid (mean frame) per segment, cached with the segment set;
//   2. candidate of a target = the source with the nearest centroid (a time warp moves the mean of
//      a segment very little, so a warped copy of a source is found; when nothing is close the
//      candidate is still a valid pair and merely gives a weak bound);
//   3. exact f64 cost of the M candidate pairs (dtw_exact.hip);
//      -- the candidate pairs themselves are finished: their lanes count as dead in the filter (their
//      exact cost seeds the selection's per-target bound instead of their filter value) and they join the
//      final fold directly, so no task has to run to its end for their sake;
//   4. threshold in the filter's accumulator units: a filter value D~ stands for a true prefix cost
//      D >= D~ (1 - (L+6) u) - 1.02 L cell  (dtw_margin.hpp with the worst-case cell error over the
//      dictionary), so "D~ > (c + 1.02 L cell) / ((1 - (L+6) u) outScale)" proves cost > c.
#include "ssym_internal.hpp"
#include "dtw_margin.hpp"

#include <algorithm>
#include <cmath>

namespace ssym {

__global__ __launch_bounds__(64) void segment_centroid_kernel(const double *__restrict__ raw,
                                                              const uint64_t *__restrict__ off, uint32_t n,
                                                              uint32_t dim, float *__restrict__ cen)
{
    const uint32_t s = blockIdx.x;
    if (s >= n)
        return;
    const uint64_t f0 = off[s], f1 = off[s + 1];
    for (uint32_t e = threadIdx.x; e < dim; e += 64) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        uint64_t f = f0;
        for (; f + 4 <= f1; f += 4) {
            a0 += raw[(f + 0) * dim + e];
            a1 += raw[(f + 1) * dim + e];
            a2 += raw[(f + 2) * dim + e];
            a3 += raw[(f + 3) * dim + e];
        }
        for (; f < f1; ++f)
            a0 += raw[f * dim + e];
        const double sum = (a0 + a1) + (a2 + a3);
        cen[(size_t)s * dim + e] = f1 > f0 ? (float)(sum / (double)(f1 - f0)) : 0.0f;
    }
}

constexpr int kCandTargets = 16;   // targets per workgroup of the candidate search

// pairs[t] = (source with the nearest centroid among the non-empty ones, target of slot t), t < nTgt
__global__ __launch_bounds__(256) void prune_candidate_kernel(const float *__restrict__ cenS,
                                                              const uint64_t *__restrict__ offS,
                                                              const uint32_t *__restrict__ permS, uint32_t nSrc,
                                                              const float *__restrict__ cenT,
                                                              const uint64_t *__restrict__ offT,
                                                              const uint32_t *__restrict__ permT, uint32_t nTgt,
                                                              uint32_t nTgtPad, uint32_t dim, int band,
                                                              uint32_t *__restrict__ hdr,
                                                              uint2 *__restrict__ pairs,
                                                              uint32_t *__restrict__ knownSrc,
                                                              uint32_t *__restrict__ candSlot)
{
    extern __shared__ float tm[];                       // [dim][kCandTargets]
    __shared__ float redD[4][kCandTargets];
    __shared__ uint32_t redI[4][kCandTargets];
    __shared__ int lenT[kCandTargets];
    const uint32_t t0 = blockIdx.x * kCandTargets;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        hdr[0] = nTgt, hdr[1] = 0;
    if (blockIdx.x == 0)
        for (uint32_t t = nTgt + threadIdx.x; t < nTgtPad; t += 256)
            candSlot[t] = 0xffffffffu;                  // pad target slots have no candidate
    for (uint32_t i = threadIdx.x; i < dim * kCandTargets; i += 256) {
        const uint32_t k = i / kCandTargets, tt = i % kCandTargets;
        const uint32_t slot = t0 + tt;
        tm[i] = slot < nTgt ? cenT[(size_t)permT[slot] * dim + k] : 0.0f;
    }
    if (threadIdx.x < kCandTargets) {
        const uint32_t slot = t0 + threadIdx.x;
        lenT[threadIdx.x] = slot < nTgt ? (int)(offT[permT[slot] + 1] - offT[permT[slot]]) : 0;
    }
    __syncthreads();
    float best[kCandTargets];
    uint32_t bi[kCandTargets];
#pragma unroll
    for (int tt = 0; tt < kCandTargets; ++tt)
        best[tt] = __builtin_inff(), bi[tt] = 0xffffffffu;
    // sources are walked in record-slot order (slots [0, nSrc) hold the real segments): the filter wants the
    // candidate's slot, everything else its index as the caller counts
    for (uint32_t p = threadIdx.x; p < nSrc; p += 256) {
        const uint32_t s = permS[p];
        const int ls = (int)(offS[s + 1] - offS[s]);
        if (ls == 0)
            continue;                                   // an empty source matches nothing
        float d[kCandTargets];
#pragma unroll
        for (int tt = 0; tt < kCandTargets; ++tt)
            d[tt] = 0.0f;
        for (uint32_t k = 0; k < dim; ++k) {
            const float v = cenS[(size_t)s * dim + k];
#pragma unroll
            for (int tt = 0; tt < kCandTargets; ++tt) {
                const float x = v - tm[k * kCandTargets + tt];
                d[tt] += x * x;
            }
        }
#pragma unroll
        for (int tt = 0; tt < kCandTargets; ++tt) {
            // inside a Sakoe-Chiba band a pair whose lengths differ by more than r has no path at all
            const float dd = (band >= 0 && abs(ls - lenT[tt]) > band) ? __builtin_inff() : d[tt];
            if (dd < best[tt] || bi[tt] == 0xffffffffu)        // (also takes a NaN distance when nothing else came)
                best[tt] = dd, bi[tt] = p;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int tt = 0; tt < kCandTargets; ++tt) {
        float b = best[tt];
        uint32_t i = bi[tt];
        for (int o = 32; o >= 1; o >>= 1) {
            const float ob = __shfl_xor(b, o);
            const uint32_t oi = __shfl_xor(i, o);
            const bool take = oi != 0xffffffffu && (i == 0xffffffffu || ob < b || (ob == b && oi < i));
            b = take ? ob : b;
            i = take ? oi : i;
        }
        if (lane == 0)
            redD[wave][tt] = b, redI[wave][tt] = i;
    }
    __syncthreads();
    if (threadIdx.x < kCandTargets) {
        const uint32_t slot = t0 + threadIdx.x;
        float b = redD[0][threadIdx.x];
        uint32_t i = redI[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) {
            const float ob = redD[w][threadIdx.x];
            const uint32_t oi = redI[w][threadIdx.x];
            const bool take = oi != 0xffffffffu && (i == 0xffffffffu || ob < b || (ob == b && oi < i));
            b = take ? ob : b;
            i = take ? oi : i;
        }
        if (slot < nTgt) {
            const uint32_t p = i == 0xffffffffu ? 0u : i;          // all sources empty: cost +inf anyway
            const uint32_t s = permS[p];
            pairs[slot] = make_uint2(s, permT[slot]);
            knownSrc[permT[slot]] = s;
            candSlot[slot] = p;
        }
    }
}

// abandon[slot] in accumulator units; pad slots get -inf (their lanes never hold a wave back)
__global__ void prune_threshold_kernel(const double *__restrict__ exact, const uint32_t *__restrict__ permT,
                                       uint32_t nTgt, uint32_t nTgtPad,
                                       const int *__restrict__ tgtLen, const float *__restrict__ tgtMaxSq,
                                       double srcMaxSq, int srcMaxFrames, MarginParams mp, double outScale,
                                       float *__restrict__ abandon)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgtPad)
        return;
    if (t >= nTgt || tgtLen[t] == 0) {          // nothing to find for a pad slot or an empty target
        abandon[t] = -__builtin_inff();
        return;
    }
    const double c = exact[permT ? permT[t] : t];
    float thr = __builtin_inff();                       // no finite bound: nothing is dropped for this target
    if (c < __builtin_inf()) {
        const double u = 5.9604644775390625e-8;         // 2^-24
        const double L = (double)(srcMaxFrames + tgtLen[t] - 1);
        const double cell = dtw_cell_error(mp, 0.0, srcMaxSq, (double)tgtMaxSq[t]);
        double v = (fmax(c, 0.0) + 1.02 * L * cell + 1e-300) / ((1.0 - (L + 6.0) * u) * outScale);
        v *= 1.000004;                                  // f32 roundings of the threshold and of res * outScale
        thr = (float)v;
        if ((double)thr < v)
            thr = __uint_as_float(__float_as_uint(thr) + 1u);
    }
    abandon[t] = thr;
}

// The candidates' exact costs are final: stage 2 leaves them out of list 2 (select.hip, knownSrc) and they
// join the re-scored pairs here, behind the entries the exact kernel has just filled in.
__global__ void prune_append_known_kernel(const uint2 *__restrict__ pairsK, const double *__restrict__ costK,
                                          uint32_t n, const uint32_t *__restrict__ hdr2, uint2 *__restrict__ pairs2,
                                          double *__restrict__ costs2)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    const uint32_t base = hdr2[0];
    pairs2[base + t] = pairsK[t];
    costs2[base + t] = costK[t];
}
__global__ void prune_bump_kernel(uint32_t *hdr2, uint32_t n) { hdr2[0] += n; }

int32_t launch_prune_append_known(ssym_ctx *ctx, uint32_t n_tgt)
{
    const uint32_t *hdrK = (const uint32_t *)ctx->prune_pairs.ptr;
    uint32_t *hdr2 = (uint32_t *)ctx->cand2.ptr;
    prune_append_known_kernel<<<(n_tgt + 255) / 256, 256, 0, ctx->stream>>>(
        (const uint2 *)(hdrK + 2), (const double *)ctx->prune_cost.ptr, n_tgt, hdr2, (uint2 *)(hdr2 + 2),
        (double *)ctx->cand_cost.ptr);
    prune_bump_kernel<<<1, 1, 0, ctx->stream>>>(hdr2, n_tgt);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

static int32_t ensure_centroids(ssym_ctx *ctx, const SegmentSet &set)
{
    if (set.centroid && set.centroid_n == set.n)
        return SSYM_OK;
    if (set.centroid)
        dev_free(ctx, set.centroid);
    set.centroid = nullptr;
    set.centroid_n = 0;
    int32_t rc = dev_alloc(ctx, (void **)&set.centroid, sizeof(float) * std::max<size_t>((size_t)set.n * set.dim, 1));
    if (rc != SSYM_OK)
        return rc;
    if (set.n)
        segment_centroid_kernel<<<set.n, 64, 0, ctx->stream>>>(set.raw, set.off, set.n, set.dim, set.centroid);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    set.centroid_n = set.n;
    return SSYM_OK;
}

// the candidate's source SLOT per target slot (pad slots: none), behind the pairs and the per-target sources
uint32_t *prune_cand_slots(ssym_ctx *ctx, const SegmentSet &tgt)
{
    return (uint32_t *)((uint2 *)((uint32_t *)ctx->prune_pairs.ptr + 2) + tgt.n) + tgt.n;
}

// step 1-3 of the header comment: candidate pair per target slot and its exact cost
// (ctx->prune_pairs: hdr | pairs[M] by slot | source by target; ctx->prune_cost[M] by slot)
int32_t launch_dtw_prune_candidates(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    hipStream_t st = ctx->stream;
    int32_t rc = ensure_centroids(ctx, src);
    if (rc == SSYM_OK)
        rc = ensure_centroids(ctx, tgt);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->prune_pairs,
                    sizeof(uint32_t) * 2 + (sizeof(uint2) + sizeof(uint32_t)) * (size_t)tgt.n + sizeof(uint32_t) * tgt.n_pad);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->prune_cost, sizeof(double) * tgt.n);
    if (rc != SSYM_OK)
        return rc;
    uint32_t *hdr = (uint32_t *)ctx->prune_pairs.ptr;
    uint2 *pairs = (uint2 *)(hdr + 2);
    const unsigned nb = (tgt.n + kCandTargets - 1) / kCandTargets;
    prune_candidate_kernel<<<nb, 256, sizeof(float) * src.dim * kCandTargets, st>>>(
        src.centroid, src.off, src.perm, src.n, tgt.centroid, tgt.off, tgt.perm, tgt.n, tgt.n_pad, src.dim, ctx->band, hdr,
        pairs, (uint32_t *)(pairs + tgt.n), prune_cand_slots(ctx, tgt));
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return launch_dtw_exact(ctx, src, tgt, pairs, hdr, tgt.n, (double *)ctx->prune_cost.ptr);
}

// step 4: thresholds in accumulator units -> ctx->abandon.  cost_by_target == NULL: this context's own
// candidate costs; else per-target costs in the CALLER's order (a source-sharded run hands in the minimum
// over all ranks' candidates: any pair above it loses to another shard's pair)
int32_t launch_dtw_prune_thresholds(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                                    const double *cost_by_target, const float **abandon_out)
{
    hipStream_t st = ctx->stream;
    double scale = 1.0;
    int32_t rc = ensure_filter_records(ctx, src, tgt, &scale);      // fixes the scale the thresholds are expressed in
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->abandon, (sizeof(float) * tgt.n_pad + 15) / 8 * 8 + sizeof(unsigned long long));
    if (rc != SSYM_OK)
        return rc;
    const MarginParams mp = margin_params(ctx, src, tgt);
    const double outScale = ctx->squared ? 1.0 / (scale * scale) : 1.0 / scale;
    prune_threshold_kernel<<<(tgt.n_pad + 255) / 256, 256, 0, st>>>(
        cost_by_target ? cost_by_target : (const double *)ctx->prune_cost.ptr, cost_by_target ? tgt.perm : nullptr,
        tgt.n, tgt.n_pad, tgt.len, tgt.max_sqnorm, src.max_sqnorm_all, (int)src.max_frames, mp, outScale,
        (float *)ctx->abandon.ptr);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    *abandon_out = (const float *)ctx->abandon.ptr;
    return SSYM_OK;
}

}  // namespace ssym
