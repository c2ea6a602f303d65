// select.hip -- per-target reductions of the matching path.
//
// Replaces the fold of SoundDictionary::at_distance (src/sound.rs:359-367):
//     .map(|v| (v - distance).abs()).enumerate().fold((0usize, 2f64), |..| if d < min {..})
// i.e. first minimum of |value - distance| with a strict '<', start value INIT, index 0 when
// nothing beats INIT.  INIT = 2.0 for refcos (the reference's literal), +inf for dtw.
//
// dtw has two stages around the exact kernel:
//   1. candidate selection on the f32 filter costs with a rigorous margin (derivation below),
//   2. final first-minimum over the exactly re-scored candidates.
#include "ssym_internal.hpp"
#include "dtw_margin.hpp"

#include <algorithm>
#include <cmath>

namespace ssym {

constexpr unsigned long long kInfBitsSel = 0x7ff0000000000000ull;      // +inf as order-preserving key bits

// ---------------------------------------------------------------------------------------------
// Error bound of the f16-split MFMA filter (dtw_filter_kernel.hpp), u = 2^-24, s = common scale.
//
//  Per cell the accumulator holds x~ for the true x = |a - b|^2 (both in units scaled by s^2):
//    * operand split: every scaled value v is fed as H1 + H2 with |v - H1 - H2| <= max(2^-22 |v|, 2^-25)
//      (f16 pieces, the second may be subnormal); the product H2a*H2b is dropped.  Over the 13
//      dims this moves x by at most 12 u (|a|^2 + |b|^2) + 2^-12            (|s v| < 64);
//    * norms are fed in three f16 pieces (error < 2^-33 relative, negligible);
//    * accumulation of the 45 products inside three chained K=16 MFMAs, f32 accumulator: at most
//      48 roundings (counted twice in case the matrix pipe truncates): 2 * 48 u * 2(|a|^2+|b|^2);
//      together  |x~ - x| <= 204 u (|a|^2 + |b|^2) + 2^-12   -- the code uses
//          E = 256 u (max|a|^2 of the source + max|b|^2 of the target) + 2^-12 / s^2   (unscaled).
//  local cost c~ = v_sqrt_f32(|x~|):  |c~ - c| = |x~ - x| / (c~ + c).  certify.hip reports, for
//      the pairs that matter, m = the smallest value of the same MFMA expression over all of the
//      pair's cells (recomputed, so within E of x like the filter's own value): every cell has
//      x >= m - E and x~ >= m - 2E:
//          m > 6E :  |c~ - c| <= E / (2 sqrt(m - 2E))         (no cell is near zero)
//          else   :  |c~ - c| <= sqrt(E)                      (|sqrt(y) - sqrt(x)| <= sqrt|y - x|)
//      (m = 0 stands for "no certificate": the first, worst-case selection over all N x M pairs)
//      squared-L2 mode: |c~ - c| <= E.   The norms in the records are those of the REPRESENTED
//      frames, so x is the squared distance of the rounded frames; rounding to the f16 piece(s)
//      moves every frame by <= rho |frame| (rho = 2^-22 with two pieces, 2^-11 with one, dims 14..42)
//      and therefore c by <= rho (|a| + |b|) -- an absolute term, no square root involved.
//    * record layout 3 (up to 13 values in K = 32, dtw_filter.hip build_filter_records_kernel): the SOURCE in two
//      pieces (rho_a = 2^-22), the TARGET in one (rho_b = 2^-11; two for its first two values, where the product
//      H2a*H2b is dropped as above: <= 4 u (|a|^2 + |b|^2)), norms of the represented frames in TWO f16 pieces
//      (<= 2^-22 relative each: 4 u (|a|^2 + |b|^2) + 2^-25 absolute), 32 products in two chained MFMAs
//      (2 * 32 u * 2 (|a|^2 + |b|^2)): together 140 u (|a|^2 + |b|^2) + 2^-12, inside the same E; the rounding
//      term of a cell is rho_a |a| + rho_b |b| (dtw_margin.hpp in_round_a / in_round_b).
//  DP: min is exact; each of the <= L = Fa+Fb-1 additions along a path rounds once and v_sqrt_f32 is
//      within 1 ulp, and DTW is monotone and 1-Lipschitz in the cell costs along the optimal path
//      of either side, so for EVERY pair
//          |C~ - C| <= err := L * cell + (L + 6) u * C~ .
//  Selection (rigorous), two stages: C lies in [C~ - err, C~ + err], hence key = |C - delta| lies in
//      [key_lo, key_hi]; the exact first minimum over s is attained by some s with
//      key_lo(s) <= min_s' key_hi(s').  Stage 1 applies this to all N x M pairs with the worst-case
//      err (m = 0) and keeps list 1; certify.hip computes m for list 1; stage 2 applies it again
//      inside list 1 with the per-pair err and keeps list 2, which is re-scored exactly.
// ---------------------------------------------------------------------------------------------
// SSYM_SELECT_PRETEST=0 (tests, measurements): every pair forms its interval in dtw_colmin_kernel / dtw_mark_kernel, as
// before round 4; the lists are the same either way
static int select_pretest()
{
    const char *k = ssym_knob("SSYM_SELECT_PRETEST");
    return !(k && atoi(k) == 0);
}

__global__ void fill_u64_kernel(unsigned long long *p, unsigned long long v, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = v;
}

// one launch for the two (or three) small blocks a stage starts from: n keys at +inf, optionally n indices at
// "none", optionally a two-word list header at zero
__global__ void init_best_kernel(unsigned long long *key, uint32_t *idx, uint32_t n, uint32_t *hdr)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        key[i] = kInfBitsSel;
        if (idx)
            idx[i] = 0xffffffffu;
    }
    if (hdr && i < 2)
        hdr[i] = 0;
}

constexpr int kSelChunk = 64;   // sources per (target, chunk) hit mask
constexpr int kSelTgt = 64;     // targets per workgroup of the two kernels below
constexpr int kSelSub = 4;      // threads per (target, chunk): 16 sources each (one thread per 64 sources left the
                                // 8-GPU share of configs[2], 512 sources, with 128 workgroups of dependent f64 chains:
                                // 38 + 27 us for the two kernels; four threads per chunk: see DESIGN.md 7)

// ub[t] = min_s key_hi(s,t) as order-preserving u64 bits
__global__ __launch_bounds__(kSelTgt * kSelSub) void dtw_colmin_kernel(
    const float *__restrict__ cmat, uint32_t nSrc, uint32_t nTgt,
    uint32_t mPad, const double *__restrict__ dist, const int *__restrict__ srcLen,
    const float *__restrict__ srcMaxSq, const int *__restrict__ tgtLen, const float *__restrict__ tgtMaxSq,
    MarginParams mp, const uint32_t *__restrict__ permT, const unsigned long long *__restrict__ prev,
    unsigned long long *__restrict__ ub, int pretest)
{
    // s, t are record SLOTS (the filter's coordinates); the caller's per-target distance is looked up
    // through the slot's segment
    __shared__ unsigned long long sBest[kSelTgt];
    const uint32_t tx = threadIdx.x % kSelTgt, ty = threadIdx.x / kSelTgt;
    const uint32_t t = blockIdx.x * kSelTgt + tx;
    if (ty == 0)
        sBest[tx] = kInfBitsSel;
    __syncthreads();
    if (t < nTgt) {
        const double delta = dist ? dist[permT[t]] : 0.0;
        const double nb = (double)tgtMaxSq[t], rb = (double)tgtMaxSq[mp.tgt_resid_off + t];
        const int fb = tgtLen[t];
        const uint32_t s0 = blockIdx.y * kSelChunk + ty * (kSelChunk / kSelSub);
        const uint32_t s1 = min(s0 + kSelChunk / kSelSub, nSrc);
        // top-k rounds (prev != NULL): the smallest bound strictly above the previous round's
        const double floorv = prev ? __longlong_as_double((long long)prev[t]) : -1.0;
        // ... starting from what the workgroups before this one have already found for the target (any bound in ub[t] is a
        // key_hi of the same minimum -- or the caller's seed, which takes part in it --, however stale the read)
        double best = pretest ? __longlong_as_double((long long)ub[t]) : __builtin_inf();
        // the thread's costs first, all loads in flight together (one dependent load per loop trip left a 16384 x 16384
        // matrix at 1 TB/s)
        float cv[kSelChunk / kSelSub];
#pragma unroll
        for (uint32_t i = 0; i < kSelChunk / kSelSub; ++i)
            cv[i] = s0 + i < s1 ? cmat[(size_t)(s0 + i) * mPad + t] : __builtin_inff();
#pragma unroll
        for (uint32_t i = 0; i < kSelChunk / kSelSub; ++i) {
            const uint32_t s = s0 + i;
            if (s >= s1)
                break;
            // key_hi is the larger distance of delta from the ends of an interval that holds the filter cost, so it is at
            // least |cost - delta|: a pair that far out cannot lower the minimum, and its interval (some 150 f64
            // instructions, three square roots) is not formed -- all but a few pairs per thread (NaN: skipped as well;
            // its interval is +inf)
            const double c = (double)cv[i];
            if (pretest && !(fabs(c - delta) < best))
                continue;
            double klo, khi;
            dtw_key_interval(mp, c, 0.0, (double)srcMaxSq[s], nb, srcLen[s], fb, delta, klo, khi,
                             (double)srcMaxSq[mp.src_resid_off + s], rb);
            if (khi < best && khi > floorv)
                best = khi;
        }
        if (best < __builtin_inf())
            atomicMin(&sBest[tx], (unsigned long long)__double_as_longlong(best));
    }
    __syncthreads();
    if (ty == 0 && t < nTgt && sBest[tx] != kInfBitsSel)
        atomicMin(&ub[t], sBest[tx]);
}

// Stage-1 list, grouped by target (certify.hip keeps a target's records in registers across its
// run of sources), built without a contended counter in three passes:
//   mark:    four threads per (target, 64-source chunk) -> u64 hit mask, per-target count += popcount
//   scan:    segment start per target = exclusive sum of the counts; hdr[0] = total, hdr[1] = total > cap
//   scatter: every thread with hits claims popcount slots inside its target's segment
// cand layout: [0] = count (the number wanted, even past cap), [1] = overflow flag, pairs from cand + 2
__global__ __launch_bounds__(kSelTgt * kSelSub) void dtw_mark_kernel(
    const float *__restrict__ cmat, uint32_t nSrc, uint32_t nTgt,
    uint32_t mPad, const double *__restrict__ dist, const int *__restrict__ srcLen,
    const float *__restrict__ srcMaxSq, const int *__restrict__ tgtLen, const float *__restrict__ tgtMaxSq,
    MarginParams mp, const uint32_t *__restrict__ permT, const unsigned long long *__restrict__ ub,
    unsigned long long *__restrict__ mask, uint32_t *__restrict__ cnt, double srcMaxSqAll, int srcMaxFrames, int pretest)
{
    __shared__ unsigned long long sHits[kSelTgt];
    const uint32_t tx = threadIdx.x % kSelTgt, ty = threadIdx.x / kSelTgt;
    const uint32_t t = blockIdx.x * kSelTgt + tx;
    if (ty == 0)
        sHits[tx] = 0;
    __syncthreads();
    if (t < nTgt) {
        unsigned long long hits = 0;
        const double thr = __longlong_as_double((long long)ub[t]);
        if (thr < __builtin_inf()) {     // else no finite cost for this target: the fold keeps (0, +inf)
            const double delta = dist ? dist[permT[t]] : 0.0;
            const double nb = (double)tgtMaxSq[t], rb = (double)tgtMaxSq[mp.tgt_resid_off + t];
            const int fb = tgtLen[t];
            const uint32_t c0 = blockIdx.y * kSelChunk;
            const uint32_t s0 = c0 + ty * (kSelChunk / kSelSub);
            const uint32_t s1 = min(s0 + kSelChunk / kSelSub, nSrc);
            // A cheap NECESSARY condition first, per target: key_lo <= thr needs lo <= delta + thr and hi >= delta - thr
            // with lo >= cost - err, hi <= cost + err, and err is at most errMax = A + rel * cost for the longest source
            // with the largest frame (dtw_cell_error grows with na and with the residual; no certificate, xmin = 0) --
            // so cost must lie in [loCut, hiCut].  Only pairs that pass form their own interval (the exact test, same
            // list as before): on unrelated data a few per target instead of all of them.
            const double u = 5.9604644775390625e-8;
            const double Lmax = (double)(srcMaxFrames + fb - 1);
            const double A = 1.02 * Lmax * dtw_cell_error(mp, 0.0, srcMaxSqAll, nb, -1.0, rb) * (1.0 + 1e-9) + 1e-300;
            const double rel = (Lmax + 6.0) * u * (1.0 + 1e-9);
            const double hiCut = rel < 0.5 ? (delta + thr + A) / (1.0 - rel) * (1.0 + 1e-9) + 1e-290 : __builtin_inf();
            const double loRaw = (delta - thr - A) / (1.0 + rel);
            const double loCut = mp.lower_only ? -__builtin_inf() : (loRaw > 0.0 ? loRaw * (1.0 - 1e-9) - 1e-290 : -__builtin_inf());
            float cv[kSelChunk / kSelSub];                           // all of the thread's loads in flight together
#pragma unroll
            for (uint32_t i = 0; i < kSelChunk / kSelSub; ++i)
                cv[i] = s0 + i < s1 ? cmat[(size_t)(s0 + i) * mPad + t] : __builtin_inff();
#pragma unroll
            for (uint32_t i = 0; i < kSelChunk / kSelSub; ++i) {
                const uint32_t s = s0 + i;
                if (s >= s1)
                    break;
                const double c = (double)cv[i];
                if (pretest && !(c <= hiCut && c >= loCut))        // (NaN fails both: its interval is +inf, never a hit)
                    continue;
                double klo, khi;
                dtw_key_interval(mp, c, 0.0, (double)srcMaxSq[s], nb, srcLen[s], fb, delta, klo, khi,
                                 (double)srcMaxSq[mp.src_resid_off + s], rb);
                if (klo <= thr)
                    hits |= 1ull << (s - c0);
            }
        }
        if (hits)
            atomicOr(&sHits[tx], hits);
    }
    __syncthreads();
    if (ty == 0 && t < nTgt) {
        const unsigned long long hits = sHits[tx];
        mask[(size_t)blockIdx.y * nTgt + t] = hits;
        if (hits)
            atomicAdd(&cnt[t], (uint32_t)__popcll(hits));
    }
}

// one block: cnt[t] -> exclusive prefix (in place), total into hdr[0], overflow flag into hdr[1]
__global__ __launch_bounds__(1024) void dtw_scan_kernel(uint32_t *__restrict__ cnt, uint32_t nTgt, uint32_t cap,
                                                        uint32_t *__restrict__ hdr)
{
    __shared__ uint32_t waveSum[16];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nTgt; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < nTgt ? cnt[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        if (lane == 63)
            waveSum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave)
                before += waveSum[w];
            all += waveSum[w];
        }
        const unsigned long long c = carry;
        if (t < nTgt)      // starts past 2^32 - 1 only arise together with the overflow flag
            cnt[t] = (uint32_t)min(c + before + incl - v, 0xffffffffull);
        __syncthreads();
        if (threadIdx.x == 0)
            carry = c + all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hdr[0] = (uint32_t)min(carry, 0xffffffffull);
        hdr[1] = carry > cap ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void dtw_scatter_kernel(const unsigned long long *__restrict__ mask,
                                                          uint32_t nTgt, const uint32_t *__restrict__ start,
                                                          uint32_t *__restrict__ fill,
                                                          const uint32_t *__restrict__ hdr,
                                                          uint2 *__restrict__ candPairs)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt || hdr[1])
        return;
    unsigned long long hits = mask[(size_t)blockIdx.y * nTgt + t];
    if (!hits)
        return;
    uint32_t slot = start[t] + atomicAdd(&fill[t], (uint32_t)__popcll(hits));
    const uint32_t s0 = blockIdx.y * kSelChunk;
    while (hits) {
        const int b = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        candPairs[slot++] = make_uint2(s0 + b, t);
    }
}

// ---- stage 2: inside list 1, with the per-pair certificate --------------------------------------
__global__ void dtw_stage2_ub_kernel(const uint32_t *__restrict__ hdr1, const uint2 *__restrict__ pairs1,
                                     const float *__restrict__ xmin, uint32_t cap, const float *__restrict__ cmat,
                                     uint32_t mPad, const double *__restrict__ dist, const int *__restrict__ srcLen,
                                     const float *__restrict__ srcMaxSq, const int *__restrict__ tgtLen,
                                     const float *__restrict__ tgtMaxSq, MarginParams mp,
                                     const uint32_t *__restrict__ permT,
                                     const unsigned long long *__restrict__ prev,
                                     unsigned long long *__restrict__ ub)
{
    const uint32_t n = hdr1[1] ? 0u : min(hdr1[0], cap);     // overflowed list 1: the host redoes stage 1
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint2 p = pairs1[k];
        double klo, khi;
        dtw_key_interval(mp, (double)cmat[(size_t)p.x * mPad + p.y], (double)xmin[k], (double)srcMaxSq[p.x],
                         (double)tgtMaxSq[p.y], srcLen[p.x], tgtLen[p.y], dist ? dist[permT[p.y]] : 0.0, klo, khi,
                         (double)srcMaxSq[mp.src_resid_off + p.x], (double)tgtMaxSq[mp.tgt_resid_off + p.y]);
        const double floorv = prev ? __longlong_as_double((long long)prev[p.y]) : -1.0;
        if (khi < __builtin_inf() && khi > floorv)
            atomicMin(&ub[p.y], (unsigned long long)__double_as_longlong(khi));
    }
}

__global__ void dtw_stage2_keep_kernel(const uint32_t *__restrict__ hdr1, const uint2 *__restrict__ pairs1,
                                       const float *__restrict__ xmin, uint32_t cap, const float *__restrict__ cmat,
                                       uint32_t mPad, const double *__restrict__ dist, const int *__restrict__ srcLen,
                                       const float *__restrict__ srcMaxSq, const int *__restrict__ tgtLen,
                                       const float *__restrict__ tgtMaxSq, MarginParams mp,
                                       const uint32_t *__restrict__ permS, const uint32_t *__restrict__ permT,
                                       const unsigned long long *__restrict__ ub,
                                       const unsigned long long *__restrict__ ub1,
                                       const uint32_t *__restrict__ knownSrc, uint32_t *__restrict__ hdr2,
                                       uint2 *__restrict__ pairs2)
{
    const uint32_t n = hdr1[1] ? 0u : min(hdr1[0], cap);     // overflowed list 1: the host redoes stage 1
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint2 p = pairs1[k];
        double klo, khi;
        dtw_key_interval(mp, (double)cmat[(size_t)p.x * mPad + p.y], (double)xmin[k], (double)srcMaxSq[p.x],
                         (double)tgtMaxSq[p.y], srcLen[p.x], tgtLen[p.y], dist ? dist[permT[p.y]] : 0.0, klo, khi,
                         (double)srcMaxSq[mp.src_resid_off + p.x], (double)tgtMaxSq[mp.tgt_resid_off + p.y]);
        // ub1 = the stage-1 threshold: never above ub on one GPU, but in a source-sharded run it is
        // the minimum over ALL ranks (ssym_match_begin / _finish) and may undercut this shard's best
        const double thr = fmin(__longlong_as_double((long long)ub[p.y]), __longlong_as_double((long long)ub1[p.y]));
        if (klo <= thr) {    // list 2 leaves the filter's slot coordinates: (segment, segment) as the caller counts them
            const uint2 o = make_uint2(permS[p.x], permT[p.y]);
            if (!knownSrc || knownSrc[o.y] != o.x)      // (early abandoning: the candidate pair has its exact cost already)
                pairs2[atomicAdd(&hdr2[0], 1u)] = o;    // list 1's capacity: cannot overflow
        }
    }
}

// ---- top-k rounds (ssym_match_topk) --------------------------------------------------------------
// The k best per target are found in k rounds of the same first-minimum machinery, each round
// restricted to entries lexicographically above the previous round's winner (key, index).
// For the candidate THRESHOLD a round takes the smallest upper bound strictly above the previous
// one: after k rounds that is the k-th smallest DISTINCT upper bound, which is >= the k-th smallest
// with multiplicity, so at least k pairs have their exact key below it and every pair of the exact
// top k (key <= the k-th exact key <= threshold) is kept.
constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;
constexpr unsigned long long kDblMaxBits = 0x7fefffffffffffffull;

// after a threshold round: nothing above prev -> fewer than k distinct bounds -> keep every finite pair
__global__ void topk_advance_kernel(unsigned long long *__restrict__ cur, unsigned long long *__restrict__ prev,
                                    uint32_t n, int round, int last)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    unsigned long long v = cur[t];
    if (v == kInfBits && round > 0)
        v = kDblMaxBits;
    prev[t] = v;
    cur[t] = last ? v : kInfBits;
}

__device__ __forceinline__ bool above_prev(double key, uint32_t s, const unsigned long long *prevKey,
                                           const uint32_t *prevIdx, uint32_t t)
{
    if (!prevKey)
        return true;
    const double pk = __longlong_as_double((long long)prevKey[t]);
    return key > pk || (key == pk && s > prevIdx[t]);
}

// final first-minimum over exactly re-scored candidates, three order-independent passes:
//   A: bestKey[t] = min key      B: bestIdx[t] = min s among key == bestKey      C: outputs
__global__ void dtw_final_key_kernel(const uint32_t *__restrict__ candHdr, const uint2 *__restrict__ pairs,
                                     const double *__restrict__ costs, uint32_t cap,
                                     const double *__restrict__ dist,
                                     const unsigned long long *__restrict__ prevKey,
                                     const uint32_t *__restrict__ prevIdx,
                                     unsigned long long *__restrict__ bestKey)
{
    const uint32_t n = min(candHdr[0], cap);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint2 p = pairs[k];
        const double key = fabs(costs[k] - (dist ? dist[p.y] : 0.0));
        if (key < __builtin_inf() && above_prev(key, p.x, prevKey, prevIdx, p.y))
            atomicMin(&bestKey[p.y], (unsigned long long)__double_as_longlong(key));
    }
}

__global__ void dtw_final_idx_kernel(const uint32_t *__restrict__ candHdr, const uint2 *__restrict__ pairs,
                                     const double *__restrict__ costs, uint32_t cap,
                                     const double *__restrict__ dist,
                                     const unsigned long long *__restrict__ prevKey,
                                     const uint32_t *__restrict__ prevIdx,
                                     const unsigned long long *__restrict__ bestKey,
                                     uint32_t *__restrict__ bestIdx)
{
    const uint32_t n = min(candHdr[0], cap);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint2 p = pairs[k];
        const double key = fabs(costs[k] - (dist ? dist[p.y] : 0.0));
        // (an infinite key equals the fold's start value but never wins: early abandoning appends its
        // candidate pairs whatever their cost, e.g. for an empty target)
        if (key < __builtin_inf() && (unsigned long long)__double_as_longlong(key) == bestKey[p.y] &&
            above_prev(key, p.x, prevKey, prevIdx, p.y))
            atomicMin(&bestIdx[p.y], p.x);
    }
}

__global__ void dtw_final_out_kernel(const uint32_t *__restrict__ candHdr, const uint2 *__restrict__ pairs,
                                     const double *__restrict__ costs, uint32_t cap,
                                     const uint32_t *__restrict__ bestIdx, uint32_t nTgt,
                                     uint32_t indexBase, uint32_t *__restrict__ outIdx,
                                     double *__restrict__ outCost)
{
    const uint32_t n = min(candHdr[0], cap);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    // targets with no finite candidate keep the fold start (index 0, +inf)
    for (uint32_t t = tid; t < nTgt; t += stride) {
        if (bestIdx[t] == 0xffffffffu) {
            outIdx[t] = indexBase;
            if (outCost)
                outCost[t] = __builtin_inf();
        }
    }
    for (uint32_t k = tid; k < n; k += stride) {
        const uint2 p = pairs[k];
        if (bestIdx[p.y] == p.x) {
            // duplicates of one (s,t) cannot occur: each pair is appended once
            outIdx[p.y] = p.x + indexBase;
            if (outCost)
                outCost[p.y] = costs[k];
        }
    }
}

// round r of top-k: entry r of every target's list, and the (key, index) the next round must exceed
__global__ void dtw_final_out_topk_kernel(const uint32_t *__restrict__ candHdr, const uint2 *__restrict__ pairs,
                                          const double *__restrict__ costs, uint32_t cap,
                                          const unsigned long long *__restrict__ bestKey,
                                          const uint32_t *__restrict__ bestIdx, uint32_t nTgt, uint32_t kTop,
                                          uint32_t r, uint32_t indexBase, uint32_t *__restrict__ outIdx,
                                          double *__restrict__ outCost, unsigned long long *__restrict__ prevKey,
                                          uint32_t *__restrict__ prevIdx)
{
    const uint32_t n = min(candHdr[0], cap);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = tid; t < nTgt; t += stride) {
        if (bestIdx[t] == 0xffffffffu) {
            outIdx[(size_t)t * kTop + r] = SSYM_NO_MATCH;
            if (outCost)
                outCost[(size_t)t * kTop + r] = __builtin_nan("");
            prevKey[t] = kInfBits;          // nothing is above +inf: later rounds stay empty
            prevIdx[t] = 0xffffffffu;
        } else {
            prevKey[t] = bestKey[t];
            prevIdx[t] = bestIdx[t];
        }
    }
    for (uint32_t k = tid; k < n; k += stride) {
        const uint2 p = pairs[k];
        if (bestIdx[p.y] == p.x) {
            outIdx[(size_t)p.y * kTop + r] = p.x + indexBase;
            if (outCost)
                outCost[(size_t)p.y * kTop + r] = costs[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic first-minimum over a full [nSrc][nTgt] f64 matrix (refcos similarities, or exact dtw
// costs when the filter is bypassed).  Chunks are folded in source order with a strict '<', so
// the result equals the reference's sequential fold.
// ---------------------------------------------------------------------------------------------
constexpr int kFoldChunk = 128;

__global__ __launch_bounds__(128) void fold_partial_kernel(const double *__restrict__ mat, uint32_t nSrc,
                                                           uint32_t nTgt, const double *__restrict__ dist,
                                                           double defaultDist, double init,
                                                           const double *__restrict__ prevVal,
                                                           const uint32_t *__restrict__ prevIdx,
                                                           uint32_t *__restrict__ partIdx,
                                                           double *__restrict__ partVal)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    const double delta = dist ? dist[t] : defaultDist;
    const uint32_t s0 = blockIdx.y * kFoldChunk;
    const uint32_t s1 = min(s0 + kFoldChunk, nSrc);
    uint32_t minIdx = 0xffffffffu;
    double minVal = init;
    // top-k rounds (prevVal != NULL): only entries above the previous round's (value, index)
    const double pv = prevVal ? prevVal[t] : -1.0;
    const uint32_t pi = prevVal ? prevIdx[t] : 0u;
    for (uint32_t s = s0; s < s1; ++s) {
        const double v = fabs(mat[(size_t)s * nTgt + t] - delta);   // src/sound.rs:359
        if (prevVal && !(v > pv || (v == pv && s > pi)))
            continue;
        if (v < minVal) {                                           // src/sound.rs:362 (NaN never wins)
            minIdx = s;
            minVal = v;
        }
    }
    partIdx[(size_t)blockIdx.y * nTgt + t] = minIdx;
    partVal[(size_t)blockIdx.y * nTgt + t] = minVal;
}

__global__ __launch_bounds__(128) void fold_final_kernel(const uint32_t *__restrict__ partIdx,
                                                         const double *__restrict__ partVal,
                                                         uint32_t nChunks, uint32_t nTgt, double init,
                                                         const double *__restrict__ gatherFrom,
                                                         uint32_t indexBase, uint32_t *__restrict__ outIdx,
                                                         double *__restrict__ outCost)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    uint32_t minIdx = 0;       // fold start (0usize, INIT), src/sound.rs:361
    double minVal = init;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const double v = partVal[(size_t)c * nTgt + t];
        if (v < minVal) {
            minVal = v;
            minIdx = partIdx[(size_t)c * nTgt + t];
        }
    }
    outIdx[t] = minIdx + indexBase;
    if (outCost) {
        // refcos reports the winning |sim - distance|; dtw reports the winner's cost itself
        if (gatherFrom && minVal < init)
            outCost[t] = gatherFrom[(size_t)minIdx * nTgt + t];
        else
            outCost[t] = minVal;
    }
}

__global__ __launch_bounds__(128) void fold_final_topk_kernel(const uint32_t *__restrict__ partIdx,
                                                              const double *__restrict__ partVal,
                                                              uint32_t nChunks, uint32_t nTgt, double init,
                                                              const double *__restrict__ gatherFrom,
                                                              uint32_t indexBase, uint32_t kTop, uint32_t r,
                                                              uint32_t *__restrict__ outIdx,
                                                              double *__restrict__ outCost,
                                                              double *__restrict__ prevVal,
                                                              uint32_t *__restrict__ prevIdx)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    uint32_t minIdx = 0xffffffffu;
    double minVal = init;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const double v = partVal[(size_t)c * nTgt + t];
        if (v < minVal) {
            minVal = v;
            minIdx = partIdx[(size_t)c * nTgt + t];
        }
    }
    const size_t o = (size_t)t * kTop + r;
    if (minIdx == 0xffffffffu) {            // nothing (left) below the fold start
        outIdx[o] = SSYM_NO_MATCH;
        if (outCost)
            outCost[o] = __builtin_nan("");
        prevVal[t] = __builtin_inf();
        prevIdx[t] = 0xffffffffu;
        return;
    }
    outIdx[o] = minIdx + indexBase;
    if (outCost)
        outCost[o] = gatherFrom ? gatherFrom[(size_t)minIdx * nTgt + t] : minVal;
    prevVal[t] = minVal;
    prevIdx[t] = minIdx;
}

// Source-sharded multi-GPU merge: smallest cost, lowest global index
// on equal cost.  Shards are ordered by index, so this is the same first-minimum rule.
__global__ void merge_shards_kernel(uint32_t nShards, uint32_t nTgt, const double *__restrict__ costs,
                                    const uint32_t *__restrict__ idx, const double *__restrict__ dist,
                                    uint32_t *__restrict__ outIdx, double *__restrict__ outCost,
                                    size_t costStride, size_t idxStride)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    // key = |cost - distance| (distance NULL = 0, where key = cost): the same key every shard folded on
    const double d = dist ? dist[t] : 0.0;
    double bc = costs[t], bk = fabs(bc - d);
    uint32_t bi = idx[t];
    for (uint32_t g = 1; g < nShards; ++g) {
        const double c = costs[g * costStride + t];
        const double k = fabs(c - d);
        const uint32_t i = idx[g * idxStride + t];
        if (k < bk || (k == bk && i < bi)) {
            bc = c;
            bk = k;
            bi = i;
        }
    }
    outIdx[t] = bi;
    if (outCost)
        outCost[t] = bc;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// scratch for top-k rounds: [M] u64 previous bound / key bits, [M] u32 previous index
static int32_t topk_scratch(ssym_ctx *ctx, uint32_t m, unsigned long long **prevKey, uint32_t **prevIdx)
{
    int32_t rc = ensure(ctx, ctx->topk, (sizeof(unsigned long long) + sizeof(uint32_t)) * (size_t)m);
    if (rc != SSYM_OK)
        return rc;
    *prevKey = (unsigned long long *)ctx->topk.ptr;
    *prevIdx = (uint32_t *)(*prevKey + m);
    return SSYM_OK;
}

int32_t launch_dtw_select2(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                           const float *xmin, const double *dist_dev, uint32_t cap, uint32_t k_top,
                           bool lower_bound_only, const uint32_t *known_src)
{
    hipStream_t st = ctx->stream;
    const MarginParams mp = margin_params(ctx, src, tgt);
    // (+ tgt.n: room for the early-abandoning candidates that join after the exact kernel)
    int32_t rc = ensure(ctx, ctx->cand2, sizeof(uint32_t) * 2 + sizeof(uint2) * ((size_t)cap + tgt.n));
    if (rc != SSYM_OK)
        return rc;
    rc = ensure(ctx, ctx->tmin2, sizeof(unsigned long long) * tgt.n);
    if (rc != SSYM_OK)
        return rc;
    unsigned long long *ub = (unsigned long long *)ctx->tmin2.ptr;
    const unsigned long long *ub1 = (const unsigned long long *)ctx->tmin.ptr;     // stage-1 threshold
    const uint32_t *hdr1 = (const uint32_t *)ctx->cand.ptr;
    const uint2 *pairs1 = (const uint2 *)(hdr1 + 2);
    uint32_t *hdr2 = (uint32_t *)ctx->cand2.ptr;
    uint2 *pairs2 = (uint2 *)(hdr2 + 2);
    const unsigned tb = (tgt.n + 255) / 256;
    init_best_kernel<<<tb, 256, 0, st>>>(ub, nullptr, tgt.n, hdr2);
    const unsigned blocks = std::max(1u, std::min((cap + 255) / 256, 2048u));
    if (lower_bound_only) {
        // wide frames: a filter cost bounds the pair from below only, so no new upper bound comes out of
        // list 1; the certificates still sharpen every pair's LOWER end against the stage-1 threshold
    } else if (k_top <= 1) {
        dtw_stage2_ub_kernel<<<blocks, 256, 0, st>>>(hdr1, pairs1, xmin, cap, cmat, tgt.n_pad, dist_dev, src.len,
                                                     src.max_sqnorm, tgt.len, tgt.max_sqnorm, mp, tgt.perm, nullptr, ub);
    } else {
        unsigned long long *prev;
        uint32_t *prevIdx;
        rc = topk_scratch(ctx, tgt.n, &prev, &prevIdx);
        if (rc != SSYM_OK)
            return rc;
        for (uint32_t r = 0; r < k_top; ++r) {
            dtw_stage2_ub_kernel<<<blocks, 256, 0, st>>>(hdr1, pairs1, xmin, cap, cmat, tgt.n_pad, dist_dev,
                                                         src.len, src.max_sqnorm, tgt.len, tgt.max_sqnorm, mp, tgt.perm,
                                                         r ? prev : nullptr, ub);
            topk_advance_kernel<<<tb, 256, 0, st>>>(ub, prev, tgt.n, (int)r, r + 1 == k_top);
        }
    }
    dtw_stage2_keep_kernel<<<blocks, 256, 0, st>>>(hdr1, pairs1, xmin, cap, cmat, tgt.n_pad, dist_dev, src.len,
                                                   src.max_sqnorm, tgt.len, tgt.max_sqnorm, mp, src.perm, tgt.perm, ub, ub1,
                                                   known_src, hdr2, pairs2);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

// stage-1 threshold per target -> ctx->tmin: the smallest (k-th smallest distinct) worst-case upper key bound
// ub[t] = the exact cost of a pair scored already (early abandoning: the target's candidate), a valid
// upper bound of the target's minimum that does not depend on the pair's filter value
__global__ void seed_bounds_kernel(const double *__restrict__ seed, uint32_t n, unsigned long long *__restrict__ ub)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    const double c = seed[t];
    ub[t] = (c >= 0.0 && c < __builtin_inf()) ? (unsigned long long)__double_as_longlong(c) : kInfBits;
}

int32_t launch_dtw_bounds(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                          const double *dist_dev, uint32_t k_top, const double *seed_by_slot)
{
    hipStream_t st = ctx->stream;
    const MarginParams mp = margin_params(ctx, src, tgt);
    int32_t rc = ensure(ctx, ctx->tmin, sizeof(unsigned long long) * tgt.n);
    if (rc != SSYM_OK)
        return rc;
    unsigned long long *ub = (unsigned long long *)ctx->tmin.ptr;
    const uint32_t nChunks = (src.n + kSelChunk - 1) / kSelChunk;
    if (seed_by_slot && k_top <= 1 && !dist_dev)
        seed_bounds_kernel<<<(tgt.n + 255) / 256, 256, 0, st>>>(seed_by_slot, tgt.n, ub);
    else
        fill_u64_kernel<<<(tgt.n + 255) / 256, 256, 0, st>>>(ub, kInfBits, tgt.n);
    dim3 grid((tgt.n + kSelTgt - 1) / kSelTgt, nChunks);
    const int pretest = select_pretest();
    if (k_top <= 1) {
        dtw_colmin_kernel<<<grid, kSelTgt * kSelSub, 0, st>>>(cmat, src.n, tgt.n, tgt.n_pad, dist_dev, src.len,
                                                src.max_sqnorm, tgt.len, tgt.max_sqnorm, mp, tgt.perm, nullptr, ub, pretest);
    } else {
        unsigned long long *prev;
        uint32_t *prevIdx;
        rc = topk_scratch(ctx, tgt.n, &prev, &prevIdx);
        if (rc != SSYM_OK)
            return rc;
        for (uint32_t r = 0; r < k_top; ++r) {
            dtw_colmin_kernel<<<grid, kSelTgt * kSelSub, 0, st>>>(cmat, src.n, tgt.n, tgt.n_pad, dist_dev, src.len, src.max_sqnorm,
                                                    tgt.len, tgt.max_sqnorm, mp, tgt.perm, r ? prev : nullptr, ub, pretest);
            topk_advance_kernel<<<(tgt.n + 255) / 256, 256, 0, st>>>(ub, prev, tgt.n, (int)r, r + 1 == k_top);
        }
    }
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

// ---- wide frames (more values per frame than the filter takes in) -------------------------------
// The filter then scores pairs on their first 42 values only: every local cost shrinks, DTW is
// monotone in the local costs, so its result is a LOWER bound of the pair's cost.  An upper bound per
// target comes from one exact evaluation: the pair the filter likes best.  Every pair whose lower
// bound (minus the filter's own error) does not exceed that exact cost may still win and is
// re-scored; the true first minimum is among them because its cost is <= the evaluated pair's.
__global__ __launch_bounds__(256) void dtw_partial_argmin_kernel(const float *__restrict__ cmat, uint32_t nSrc,
                                                                 uint32_t nTgt, uint32_t mPad,
                                                                 const uint32_t *__restrict__ permS,
                                                                 const uint32_t *__restrict__ permT,
                                                                 const double *__restrict__ dist,
                                                                 const int *__restrict__ srcLen,
                                                                 const int *__restrict__ tgtLen, int band,
                                                                 uint32_t *__restrict__ hdr,
                                                                 uint2 *__restrict__ pairs)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;      // target slot
    if (t == 0)
        hdr[0] = nTgt, hdr[1] = 0;
    if (t >= nTgt)
        return;
    // (with a per-target distance: the pair whose lower bound lies nearest to it -- any pair gives a valid bound,
    //  this one is the best guess)
    const float delta = dist ? (float)dist[permT[t]] : 0.0f;
    float best = __builtin_inff();
    uint32_t bs = 0;
    const int fb = tgtLen[t];
    for (uint32_t s = 0; s < nSrc; ++s) {
        if (band >= 0 && abs(srcLen[s] - fb) > band)
            continue;               // inside the band this pair has no path at all (the unbanded filter would not know)
        const float c0 = cmat[(size_t)s * mPad + t];
        const float c = c0 < __builtin_inff() ? __builtin_fabsf(c0 - delta) : c0;
        if (c < best) {
            best = c;
            bs = s;
        }
    }
    pairs[t] = make_uint2(permS[bs], permT[t]);                    // exact kernel: the caller's indices
}

// top-k on wide frames: the k pairs the filter likes best per target (their exact costs bound the k-th best from above)
constexpr int kPartialKMax = (int)SSYM_TOPK_MAX;
__global__ __launch_bounds__(256) void dtw_partial_topk_kernel(const float *__restrict__ cmat, uint32_t nSrc, uint32_t nTgt,
                                                               uint32_t mPad, const uint32_t *__restrict__ permS,
                                                               const uint32_t *__restrict__ permT,
                                                               const double *__restrict__ dist,
                                                               const int *__restrict__ srcLen, const int *__restrict__ tgtLen,
                                                               int band, uint32_t k, uint32_t *__restrict__ hdr,
                                                               uint2 *__restrict__ pairs, uint32_t *__restrict__ found)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;      // target slot
    if (t == 0)
        hdr[0] = nTgt * k, hdr[1] = 0;
    if (t >= nTgt)
        return;
    float bc[kPartialKMax];
    uint32_t bs[kPartialKMax];
    uint32_t cnt = 0;
    const float delta = dist ? (float)dist[permT[t]] : 0.0f;
    const int fb = tgtLen[t];
    for (uint32_t s = 0; s < nSrc; ++s) {
        const float c0 = cmat[(size_t)s * mPad + t];
        if (!(c0 < __builtin_inff()) || (band >= 0 && abs(srcLen[s] - fb) > band))
            continue;
        const float c = __builtin_fabsf(c0 - delta);
        if (cnt == k && !(c < bc[k - 1]))
            continue;
        uint32_t pos = cnt < k ? cnt : k - 1;                    // insertion into the ascending list of at most k
        while (pos > 0 && c < bc[pos - 1]) {
            bc[pos] = bc[pos - 1];
            bs[pos] = bs[pos - 1];
            --pos;
        }
        bc[pos] = c;
        bs[pos] = s;
        if (cnt < k)
            ++cnt;
    }
    for (uint32_t r = 0; r < k; ++r)                                // (short lists repeat a pair: the threshold is +inf then)
        pairs[(size_t)t * k + r] = make_uint2(permS[r < cnt ? bs[r] : (cnt ? bs[0] : 0)], permT[t]);
    found[t] = cnt;
}

__global__ void dtw_partial_topk_threshold_kernel(const double *__restrict__ exact, const uint32_t *__restrict__ found,
                                                  uint32_t nTgt, uint32_t k, const double *__restrict__ dist,
                                                  const uint32_t *__restrict__ permT, unsigned long long *__restrict__ ub)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    const double delta = dist ? dist[permT[t]] : 0.0;
    double worst = 0.0;
    bool ok = found[t] >= k;
    for (uint32_t r = 0; r < k && ok; ++r) {
        const double c = exact[(size_t)t * k + r];
        if (!(c < __builtin_inf()))
            ok = false;                                             // NaN / +inf: no bound from this pair
        else {
            const double key = fabs(c - delta);
            worst = key > worst ? key : worst;
        }
    }
    // fewer than k bounds: every finite pair stays (the largest finite double, as in the top-k rounds of the bounds
    // kernel -- +inf would mean "no pair at all")
    ub[t] = ok ? (unsigned long long)__double_as_longlong(worst) : kDblMaxBits;
}

__global__ void dtw_partial_threshold_kernel(const double *__restrict__ exact, const double *__restrict__ seed,
                                             uint32_t nTgt, const double *__restrict__ dist,
                                             const uint32_t *__restrict__ permT, unsigned long long *__restrict__ ub)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTgt)
        return;
    double c = exact[t];
    if (dist)
        c = fabs(c - dist[permT[t]]);           // the threshold lives in key space: |cost - distance|
    if (seed) {
        // early abandoning: the candidate's exact cost is an upper bound too.  (It does not replace the pair the
        // filter likes best: without close pairs the centroid candidate is a poor bound and the lower-bound
        // selection would keep 10^2 times more pairs -- measured 15 -> 110 ms at 64 values per frame.)
        const double s = seed[t];
        c = (s >= 0.0 && (s < c || c != c)) ? s : c;
    }
    // (no finite bound from the scored pair -- an empty side, NaN: every pair with a finite lower bound stays in)
    ub[t] = c < __builtin_inf() ? (unsigned long long)__double_as_longlong(c < 0.0 ? 0.0 : c) : kDblMaxBits;
}

int32_t launch_dtw_bounds_partial(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const float *cmat,
                                  const double *seed_by_slot, uint32_t k_top, const double *dist_dev)
{
    if (k_top > 1) {
        // the k-th best cost of a target is at most the largest exact cost among ANY k of its pairs: take the k the
        // filter likes best
        hipStream_t st = ctx->stream;
        const size_t nk = (size_t)tgt.n * k_top;
        int32_t rc = ensure(ctx, ctx->tmin, sizeof(unsigned long long) * tgt.n);
        if (rc == SSYM_OK)
            rc = ensure(ctx, ctx->cand2, sizeof(uint32_t) * 2 + sizeof(uint2) * nk);
        if (rc == SSYM_OK)
            rc = ensure(ctx, ctx->cand_cost, sizeof(double) * nk);
        if (rc == SSYM_OK)
            rc = ensure(ctx, ctx->selcnt, sizeof(uint32_t) * 2 * (size_t)tgt.n);
        if (rc != SSYM_OK)
            return rc;
        uint32_t *hdr = (uint32_t *)ctx->cand2.ptr;
        uint2 *pairs = (uint2 *)(hdr + 2);
        uint32_t *found = (uint32_t *)ctx->selcnt.ptr;
        const unsigned tb = (tgt.n + 255) / 256;
        dtw_partial_topk_kernel<<<tb, 256, 0, st>>>(cmat, src.n, tgt.n, tgt.n_pad, src.perm, tgt.perm, dist_dev, src.len, tgt.len,
                                                    ctx->band, k_top, hdr, pairs, found);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        rc = launch_dtw_exact(ctx, src, tgt, pairs, hdr, (uint32_t)nk, (double *)ctx->cand_cost.ptr);
        if (rc != SSYM_OK)
            return rc;
        dtw_partial_topk_threshold_kernel<<<tb, 256, 0, st>>>((const double *)ctx->cand_cost.ptr, found, tgt.n, k_top, dist_dev,
                                                              tgt.perm, (unsigned long long *)ctx->tmin.ptr);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        return SSYM_OK;
    }
    hipStream_t st = ctx->stream;
    int32_t rc = ensure(ctx, ctx->tmin, sizeof(unsigned long long) * tgt.n);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->cand2, sizeof(uint32_t) * 2 + sizeof(uint2) * (size_t)tgt.n);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->cand_cost, sizeof(double) * tgt.n);
    if (rc != SSYM_OK)
        return rc;
    uint32_t *hdr = (uint32_t *)ctx->cand2.ptr;
    uint2 *pairs = (uint2 *)(hdr + 2);
    const unsigned tb = (tgt.n + 255) / 256;
    dtw_partial_argmin_kernel<<<tb, 256, 0, st>>>(cmat, src.n, tgt.n, tgt.n_pad, src.perm, tgt.perm, dist_dev, src.len, tgt.len,
                                                  ctx->band, hdr, pairs);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    rc = launch_dtw_exact(ctx, src, tgt, pairs, hdr, tgt.n, (double *)ctx->cand_cost.ptr);
    if (rc != SSYM_OK)
        return rc;
    dtw_partial_threshold_kernel<<<tb, 256, 0, st>>>((const double *)ctx->cand_cost.ptr, seed_by_slot, tgt.n, dist_dev, tgt.perm,
                                                     (unsigned long long *)ctx->tmin.ptr);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

// stage 1: every pair whose worst-case lower key bound does not exceed ctx->tmin[target] -> list 1
int32_t launch_dtw_select(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                          const float *cmat, const double *dist_dev, uint32_t cap)
{
    hipStream_t st = ctx->stream;
    const MarginParams mp = margin_params(ctx, src, tgt);
    int32_t rc = ensure(ctx, ctx->cand, sizeof(uint32_t) * 2 + sizeof(uint2) * (size_t)cap);
    if (rc != SSYM_OK)
        return rc;
    const unsigned long long *ub = (const unsigned long long *)ctx->tmin.ptr;
    uint32_t *hdr = (uint32_t *)ctx->cand.ptr;
    uint2 *pairs = (uint2 *)(hdr + 2);
    const uint32_t nChunks = (src.n + kSelChunk - 1) / kSelChunk;
    rc = ensure(ctx, ctx->selmask, sizeof(unsigned long long) * (size_t)nChunks * tgt.n);
    if (rc != SSYM_OK)
        return rc;
    rc = ensure(ctx, ctx->selcnt, sizeof(uint32_t) * 2 * (size_t)tgt.n);
    if (rc != SSYM_OK)
        return rc;
    unsigned long long *mask = (unsigned long long *)ctx->selmask.ptr;
    uint32_t *cnt = (uint32_t *)ctx->selcnt.ptr, *fill = cnt + tgt.n;
    rc = zero_words(ctx, cnt, sizeof(uint32_t) * 2 * (size_t)tgt.n);
    if (rc != SSYM_OK)
        return rc;
    dim3 grid((tgt.n + 255) / 256, nChunks);
    dim3 markGrid((tgt.n + kSelTgt - 1) / kSelTgt, nChunks);
    dtw_mark_kernel<<<markGrid, kSelTgt * kSelSub, 0, st>>>(cmat, src.n, tgt.n, tgt.n_pad, dist_dev, src.len,
                                          src.max_sqnorm, tgt.len, tgt.max_sqnorm, mp, tgt.perm, ub, mask, cnt,
                                          src.max_sqnorm_all, (int)src.max_frames, select_pretest());
    dtw_scan_kernel<<<1, 1024, 0, st>>>(cnt, tgt.n, cap, hdr);
    dtw_scatter_kernel<<<grid, 256, 0, st>>>(mask, tgt.n, cnt, fill, hdr, pairs);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

int32_t launch_dtw_final(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                         const double *dist_dev, uint32_t cap, uint32_t index_base, uint32_t k_top,
                         uint32_t *out_idx_dev, double *out_cost_dev)
{
    (void)src;
    hipStream_t st = ctx->stream;
    int32_t rc = ensure(ctx, ctx->best, (sizeof(unsigned long long) + sizeof(uint32_t)) * (size_t)tgt.n);
    if (rc != SSYM_OK)
        return rc;
    unsigned long long *bestKey = (unsigned long long *)ctx->best.ptr;
    uint32_t *bestIdx = (uint32_t *)(bestKey + tgt.n);
    const uint32_t *hdr = (const uint32_t *)ctx->cand2.ptr;     // list 2: the exactly re-scored pairs
    const uint2 *pairs = (const uint2 *)(hdr + 2);
    const double *costs = (const double *)ctx->cand_cost.ptr;
    const unsigned tb = (tgt.n + 255) / 256;
    const unsigned blocks = std::max(1u, std::min((cap + 255) / 256, 1024u));
    const unsigned blocks2 = std::max(blocks, std::min(tb, 1024u));
    if (k_top <= 1) {
        init_best_kernel<<<tb, 256, 0, st>>>(bestKey, bestIdx, tgt.n, nullptr);
        dtw_final_key_kernel<<<blocks, 256, 0, st>>>(hdr, pairs, costs, cap, dist_dev, nullptr, nullptr, bestKey);
        dtw_final_idx_kernel<<<blocks, 256, 0, st>>>(hdr, pairs, costs, cap, dist_dev, nullptr, nullptr, bestKey,
                                                     bestIdx);
        dtw_final_out_kernel<<<blocks2, 256, 0, st>>>(hdr, pairs, costs, cap, bestIdx, tgt.n, index_base,
                                                      out_idx_dev, out_cost_dev);
    } else {
        unsigned long long *prevKey;
        uint32_t *prevIdx;
        rc = topk_scratch(ctx, tgt.n, &prevKey, &prevIdx);
        if (rc != SSYM_OK)
            return rc;
        for (uint32_t r = 0; r < k_top; ++r) {
            init_best_kernel<<<tb, 256, 0, st>>>(bestKey, bestIdx, tgt.n, nullptr);
            dtw_final_key_kernel<<<blocks, 256, 0, st>>>(hdr, pairs, costs, cap, dist_dev, r ? prevKey : nullptr,
                                                         prevIdx, bestKey);
            dtw_final_idx_kernel<<<blocks, 256, 0, st>>>(hdr, pairs, costs, cap, dist_dev, r ? prevKey : nullptr,
                                                         prevIdx, bestKey, bestIdx);
            dtw_final_out_topk_kernel<<<blocks2, 256, 0, st>>>(hdr, pairs, costs, cap, bestKey, bestIdx, tgt.n,
                                                               k_top, r, index_base, out_idx_dev, out_cost_dev,
                                                               prevKey, prevIdx);
        }
    }
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

static int32_t launch_fold(ssym_ctx *ctx, uint32_t n_src, uint32_t n_tgt, const double *mat,
                           const double *dist_dev, double default_dist, double init, bool gather,
                           uint32_t index_base, uint32_t k_top, uint32_t *out_idx_dev, double *out_cost_dev)
{
    hipStream_t st = ctx->stream;
    const uint32_t nChunks = (n_src + kFoldChunk - 1) / kFoldChunk;
    int32_t rc = ensure(ctx, ctx->part, (sizeof(double) + sizeof(uint32_t)) * (size_t)nChunks * n_tgt);
    if (rc != SSYM_OK)
        return rc;
    double *partVal = (double *)ctx->part.ptr;
    uint32_t *partIdx = (uint32_t *)(partVal + (size_t)nChunks * n_tgt);
    dim3 grid((n_tgt + 127) / 128, nChunks);
    if (k_top <= 1) {
        fold_partial_kernel<<<grid, 128, 0, st>>>(mat, n_src, n_tgt, dist_dev, default_dist, init, nullptr, nullptr,
                                                  partIdx, partVal);
        fold_final_kernel<<<(n_tgt + 127) / 128, 128, 0, st>>>(partIdx, partVal, nChunks, n_tgt, init,
                                                               gather ? mat : nullptr, index_base,
                                                               out_idx_dev, out_cost_dev);
    } else {
        unsigned long long *prevBits;
        uint32_t *prevIdx;
        rc = topk_scratch(ctx, n_tgt, &prevBits, &prevIdx);
        if (rc != SSYM_OK)
            return rc;
        double *prevVal = (double *)prevBits;
        for (uint32_t r = 0; r < k_top; ++r) {
            fold_partial_kernel<<<grid, 128, 0, st>>>(mat, n_src, n_tgt, dist_dev, default_dist, init,
                                                      r ? prevVal : nullptr, prevIdx, partIdx, partVal);
            fold_final_topk_kernel<<<(n_tgt + 127) / 128, 128, 0, st>>>(partIdx, partVal, nChunks, n_tgt, init,
                                                                        gather ? mat : nullptr, index_base, k_top,
                                                                        r, out_idx_dev, out_cost_dev, prevVal,
                                                                        prevIdx);
        }
    }
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

int32_t launch_dtw_final_allpairs(ssym_ctx *ctx, uint32_t n_src, uint32_t n_tgt, const double *costs,
                                  const double *dist_dev, uint32_t index_base, uint32_t k_top,
                                  uint32_t *out_idx_dev, double *out_cost_dev)
{
    return launch_fold(ctx, n_src, n_tgt, costs, dist_dev, 0.0, (double)INFINITY, true, index_base, k_top,
                       out_idx_dev, out_cost_dev);
}

int32_t launch_refcos_argmin(ssym_ctx *ctx, uint32_t n_src, uint32_t n_tgt, const double *sims,
                             const double *dist_dev, uint32_t index_base, uint32_t k_top,
                             uint32_t *out_idx_dev, double *out_cost_dev)
{
    // match_sound = at_distance(1.0, ..) (src/sound.rs:346-348); fold start 2.0 (src/sound.rs:361)
    return launch_fold(ctx, n_src, n_tgt, sims, dist_dev, 1.0, 2.0, false, index_base, k_top, out_idx_dev,
                       out_cost_dev);
}

int32_t launch_merge_shards(ssym_ctx *ctx, uint32_t n_shards, uint32_t n_targets, const double *costs,
                            const uint32_t *idx, const double *dist_dev, uint32_t *out_idx, double *out_cost,
                            size_t cost_stride, size_t idx_stride)
{
    if (n_targets == 0)
        return SSYM_OK;
    merge_shards_kernel<<<(n_targets + 255) / 256, 256, 0, ctx->stream>>>(
        n_shards, n_targets, costs, idx, dist_dev, out_idx, out_cost, cost_stride ? cost_stride : n_targets,
        idx_stride ? idx_stride : n_targets);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
