// refcos.hip -- the reference's own segment similarity, all pairs, bit for bit.
//
// Replaces cosine_sim (src/sound.rs:22-33) as called N x M times from at_distance
// (src/sound.rs:354):
//     len = min(|me|, |you|)                                   :24-28
//     nrm = norm(me) * norm(you)      (full vectors, SQUARED norms, src/sound.rs:35-38)   :30
//     dot = rulinalg::utils::dot(&me[..len], &you[..len])      :31   (rulinalg 0.4.2)
//     dot / nrm                                                :32
//
// Arithmetic order is the reference's, so results equal the CPU oracle's bit for bit:
//   * norm: sequential fold per segment, computed once in pack.hip (recomputing it per pair, as the
//     reference does, yields the same bits);
//   * dot: rulinalg's eight running sums p0..p7 over blocks of eight elements, combined in the association
//     include/ssym_rulinalg.h names (SSYM_RULINALG_COMBINE, shared with the oracle; default
//     ((((0+(p0+p4))+(p1+p5))+(p2+p6))+(p3+p7))), then the len%8 tail added one product at a time;
//     every product and sum is rounded separately (no FMA: -ffp-contract=off and __dmul_rn/__dadd_rn);
//   * the block structure depends on the PAIR's len, so each pair carries its own block count and
//     switches from the eight sums to the tail exactly where the reference does.
//
// Mapping: a 256-thread workgroup owns a 32 x 32 tile of pairs; 64-element chunks of the 32 source
// and 32 target segments are staged in LDS (element-major, so lanes read consecutive doubles);
// each thread keeps 2 x 2 pairs x 8 running sums in registers.
#include "ssym_internal.hpp"
#include "ssym_rulinalg.h"

#include <algorithm>

namespace ssym {

constexpr int kTS = 32;       // sources per tile
constexpr int kTT = 32;       // targets per tile
constexpr int kCH = 64;       // elements per staged chunk
constexpr int kLd = kTS + 1;  // padded leading dimension of the element-major LDS images

__global__ __launch_bounds__(256) void refcos_sims_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ srcNorm, const double *__restrict__ tgtRaw,
    const uint64_t *__restrict__ tgtOff, const double *__restrict__ tgtNorm, uint32_t nSrc,
    uint32_t nTgt, uint32_t dim, uint32_t maxLenVals, double *__restrict__ sims)
{
    __shared__ double sS[kCH * kLd];
    __shared__ double sT[kCH * kLd];
    __shared__ unsigned long long sBase[kTS + kTT];   // value offset of each staged segment
    __shared__ unsigned sLen[kTS + kTT];              // length in values

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const uint32_t sTile = blockIdx.y * kTS, tTile = blockIdx.x * kTT;

    if (tid < kTS + kTT) {
        const bool isS = tid < kTS;
        const uint32_t g = isS ? sTile + tid : tTile + (tid - kTS);
        const uint32_t n = isS ? nSrc : nTgt;
        const uint64_t *off = isS ? srcOff : tgtOff;
        unsigned long long base = 0;
        unsigned len = 0;
        if (g < n) {
            base = off[g] * dim;
            len = (unsigned)((off[g + 1] - off[g]) * dim);
        }
        sBase[tid] = base;
        sLen[tid] = len;
    }
    __syncthreads();

    // this thread's 2 x 2 pairs: sources {ty, ty+16}, targets {tx, tx+16}
    unsigned q[2][2], rem[2][2];
    bool done[2][2];
    double p[2][2][8];
    double dot[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned la = sLen[ty + 16 * a], lb = sLen[kTS + tx + 16 * b];
            const unsigned len = la < lb ? la : lb;            // src/sound.rs:24-28
            q[a][b] = len / 8;
            rem[a][b] = len % 8;
            done[a][b] = false;
            dot[a][b] = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                p[a][b][i] = 0.0;
        }

    const unsigned nChunks = (maxLenVals + kCH - 1) / kCH + 1;   // +1: a block index == q exists
    for (unsigned c = 0; c < nChunks; ++c) {
        // ---- stage chunk c of the 32 + 32 segments, element-major, zero beyond each length ----
        __syncthreads();
        {
            const int row = tid >> 3;          // 0..31
            const int e0 = (tid & 7) * 8;      // 8 consecutive elements
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const unsigned long long base = sBase[side * kTS + row];
                const unsigned len = sLen[side * kTS + row];
                const double *raw = side ? tgtRaw : srcRaw;
                double *dst = side ? sT : sS;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned e = c * kCH + e0 + i;
                    dst[(e0 + i) * kLd + row] = e < len ? raw[base + e] : 0.0;
                }
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int m = 0; m < kCH / 8; ++m) {
            const unsigned gm = c * (kCH / 8) + m;   // block-of-eight index within the segment
            double xs[2][8], yt[2][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xs[0][i] = sS[(8 * m + i) * kLd + ty];
                xs[1][i] = sS[(8 * m + i) * kLd + ty + 16];
                yt[0][i] = sT[(8 * m + i) * kLd + tx];
                yt[1][i] = sT[(8 * m + i) * kLd + tx + 16];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (gm < q[a][b]) {
                        // rulinalg dot main loop: p_i = p_i + xs[i] * ys[i]
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            p[a][b][i] = __dadd_rn(p[a][b][i], __dmul_rn(xs[a][i], yt[b][i]));
                    } else if (gm == q[a][b] && !done[a][b]) {
                        double s = 0.0;
                        s = SSYM_RULINALG_STEP(__dadd_rn, s, p[a][b][0], p[a][b][4]);
                        s = SSYM_RULINALG_STEP(__dadd_rn, s, p[a][b][1], p[a][b][5]);
                        s = SSYM_RULINALG_STEP(__dadd_rn, s, p[a][b][2], p[a][b][6]);
                        s = SSYM_RULINALG_STEP(__dadd_rn, s, p[a][b][3], p[a][b][7]);
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if ((unsigned)i < rem[a][b])
                                s = __dadd_rn(s, __dmul_rn(xs[a][i], yt[b][i]));
                        dot[a][b] = s;
                        done[a][b] = true;
                    }
                }
        }
    }

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t s = sTile + ty + 16 * a, t = tTile + tx + 16 * b;
            if (s < nSrc && t < nTgt) {
                const double nrm = __dmul_rn(srcNorm[s], tgtNorm[t]);   // src/sound.rs:30
                sims[(size_t)s * nTgt + t] = __ddiv_rn(dot[a][b], nrm); // src/sound.rs:32
            }
        }
}

// ---------------------------------------------------------------------------------------------
// ONE query against the dictionary in ONE launch: SoundDictionary::at_distance as the reference calls it, once per
// target (src/sound.rs:351-370, 453-454).  The batched path above costs a single query ~10 stream operations
// (pack copies, norm, similarity tile, two fold kernels, result copies) -- 125 us per call, almost all of it launch
// chain.  Here the kernel reads the query straight from the pinned staging buffer, recomputes its norm (the same
// sequential fold), gives every dictionary entry EIGHT lanes -- lane i keeps rulinalg's running sum p_i, so
// the eight sums and their combination are exactly those of the tile kernel --, folds |sim - distance| to the
// first minimum per workgroup, and the last workgroup to finish folds the workgroups in index order from the
// reference's start value (0, 2.0) and writes index and value back into pinned memory.
constexpr int kOneMaxVals = 4096;           // query values held in LDS
constexpr int kOneEntries = 32;             // dictionary entries per 256-thread workgroup
constexpr uint32_t kFewMaxQueries = 64;     // queries per call on this path

__global__ __launch_bounds__(256) void refcos_match_one_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, const double *__restrict__ srcNorm,
    uint32_t nSrc, uint32_t dim, const void *__restrict__ queries, const uint64_t *__restrict__ qOff /* frames */,
    int queryIsF32, const double *__restrict__ distances /* NULL: defaultDist */, double defaultDist,
    double *__restrict__ partValAll, uint32_t *__restrict__ partIdxAll, unsigned *__restrict__ tickets,
    uint32_t *__restrict__ outIdxAll, double *__restrict__ outValAll)
{
    // blockIdx.y = the query (a handful per call: ssym_match_one, small ssym_match_batch calls)
    const uint32_t y = blockIdx.y;
    const unsigned long long qBase = qOff[y] * dim;
    const uint32_t qLen = (uint32_t)((qOff[y + 1] - qOff[y]) * dim);
    const void *query = queryIsF32 ? (const void *)(static_cast<const float *>(queries) + qBase)
                                   : (const void *)(static_cast<const double *>(queries) + qBase);
    const double distance = distances ? distances[y] : defaultDist;
    double *partVal = partValAll + (size_t)y * gridDim.x;
    uint32_t *partIdx = partIdxAll + (size_t)y * gridDim.x;
    unsigned *ticket = tickets + y;
    uint32_t *outIdx = outIdxAll + y;
    double *outVal = outValAll + y;
    __shared__ double sq[kOneMaxVals];
    __shared__ double normQ;
    __shared__ double cv[kOneEntries];
    __shared__ uint32_t ci[kOneEntries];
    __shared__ bool last;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < qLen; i += 256)
        sq[i] = queryIsF32 ? (double)static_cast<const float *>(query)[i] : static_cast<const double *>(query)[i];
    __syncthreads();
    if (tid == 255) {                               // norm(you), src/sound.rs:35-38: memo = item * item + memo
        double memo = 0.0;
        for (uint32_t i = 0; i < qLen; ++i)
            memo = __dadd_rn(__dmul_rn(sq[i], sq[i]), memo);
        normQ = memo;
    }
    const int i8 = tid & 7;
    const uint32_t s = blockIdx.x * kOneEntries + (tid >> 3);
    double v = __builtin_inf();
    double dot = 0.0;
    uint32_t len = 0;
    unsigned long long base = 0;
    if (s < nSrc) {
        base = srcOff[s] * dim;
        const uint32_t la = (uint32_t)((srcOff[s + 1] - srcOff[s]) * dim);
        len = la < qLen ? la : qLen;                // src/sound.rs:24-28
    }
    const uint32_t qb = len / 8, rem = len % 8;
    double p = 0.0;                                 // running sum p_i of rulinalg's dot
    for (uint32_t k = 0; k < qb; ++k)
        p = __dadd_rn(p, __dmul_rn(srcRaw[base + 8 * k + i8], sq[8 * k + i8]));
    const int g0 = (tid & 63) & ~7;                 // first lane of this entry's group of eight
    const double p0 = __shfl(p, g0 + 0), p1 = __shfl(p, g0 + 1), p2 = __shfl(p, g0 + 2), p3 = __shfl(p, g0 + 3);
    const double p4 = __shfl(p, g0 + 4), p5 = __shfl(p, g0 + 5), p6 = __shfl(p, g0 + 6), p7 = __shfl(p, g0 + 7);
    if (i8 == 0 && s < nSrc) {
        double acc = 0.0;
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p0, p4);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p1, p5);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p2, p6);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p3, p7);
        for (uint32_t i = 0; i < rem; ++i)
            acc = __dadd_rn(acc, __dmul_rn(srcRaw[base + 8 * qb + i], sq[8 * qb + i]));
        dot = acc;
    }
    __syncthreads();                                // normQ
    if (i8 == 0) {
        if (s < nSrc) {
            const double nrm = __dmul_rn(srcNorm[s], normQ);        // src/sound.rs:30
            const double sim = __ddiv_rn(dot, nrm);                 // src/sound.rs:32
            v = fabs(__dsub_rn(sim, distance));                     // src/sound.rs:359
        }
        cv[tid >> 3] = v;
        ci[tid >> 3] = s;
    }
    __syncthreads();
    if (tid == 0) {
        // first minimum over this workgroup's entries in index order (NaN never wins, src/sound.rs:361-367)
        double best = __builtin_inf();
        uint32_t bi = 0xffffffffu;
        for (int e = 0; e < kOneEntries; ++e)
            if (cv[e] < best) {
                best = cv[e];
                bi = ci[e];
            }
        partVal[blockIdx.x] = best;
        partIdx[blockIdx.x] = bi;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        uint32_t minIdx = 0;                        // fold start (0usize, 2f64), src/sound.rs:361
        double minVal = 2.0;
        for (uint32_t b = 0; b < gridDim.x; ++b) {
            const double pv = static_cast<volatile double *>(partVal)[b];
            if (pv < minVal) {
                minVal = pv;
                minIdx = static_cast<volatile uint32_t *>(partIdx)[b];
            }
        }
        *outIdx = minIdx;
        *outVal = minVal;
        *ticket = 0;                                // ready for the next call
        __threadfence_system();
    }
}

bool refcos_few_supported(const ssym_ctx *ctx, const SegmentSet &src, const uint64_t *q_off, uint32_t n_queries)
{
    if (ctx->metric != SSYM_METRIC_REFCOS || src.n == 0 || n_queries == 0 || n_queries > kFewMaxQueries || !q_off)
        return false;
    for (uint32_t i = 0; i < n_queries; ++i)
        if (q_off[i + 1] < q_off[i] || (q_off[i + 1] - q_off[i]) * src.dim > (uint64_t)kOneMaxVals)
            return false;
    return true;
}

// queries / q_off / distances: device-accessible (pinned) copies -- features in the context's dtype, offsets in frames
// rebased to 0, distances or NULL; out_val / out_idx: pinned, n_queries entries each
int32_t launch_refcos_match_few(ssym_ctx *ctx, const SegmentSet &src, const void *queries, const uint64_t *q_off,
                                uint32_t n_queries, const double *distances, double default_dist, double *out_val,
                                uint32_t *out_idx)
{
    const uint32_t nb = (src.n + kOneEntries - 1) / kOneEntries;
    int32_t rc = ensure(ctx, ctx->part, (sizeof(double) + sizeof(uint32_t)) * (size_t)nb * n_queries + 256);
    if (rc != SSYM_OK)
        return rc;
    if (!ctx->one_ticket.ptr) {
        rc = ensure(ctx, ctx->one_ticket, sizeof(unsigned) * kFewMaxQueries);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(ctx->one_ticket.ptr, 0, sizeof(unsigned) * kFewMaxQueries, ctx->stream));
    }
    double *partVal = (double *)ctx->part.ptr;
    uint32_t *partIdx = (uint32_t *)(partVal + (size_t)nb * n_queries);
    refcos_match_one_kernel<<<dim3(nb, n_queries), 256, 0, ctx->stream>>>(
        src.raw, src.off, src.norm, src.n, src.dim, queries, q_off, ctx->dtype == SSYM_DTYPE_F32 ? 1 : 0, distances,
        default_dist, partVal, partIdx, (unsigned *)ctx->one_ticket.ptr, out_idx, out_val);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

int32_t launch_refcos_sims(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *sims)
{
    if (src.dim != tgt.dim) {
        ctx->err = "dim mismatch between dictionary and targets";
        return SSYM_E_INVALID;
    }
    if (src.n == 0 || tgt.n == 0)
        return SSYM_OK;
    const uint64_t maxLen = (uint64_t)std::min(src.max_frames, tgt.max_frames) * src.dim;
    if (maxLen > 0xfffffff0ull) {
        ctx->err = "segment too long";
        return SSYM_E_UNSUPPORTED;
    }
    dim3 grid((tgt.n + kTT - 1) / kTT, (src.n + kTS - 1) / kTS);
    refcos_sims_kernel<<<grid, 256, 0, ctx->stream>>>(src.raw, src.off, src.norm, tgt.raw, tgt.off,
                                                      tgt.norm, src.n, tgt.n, src.dim, (uint32_t)maxLen,
                                                      sims);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
