// certify.hip -- per-pair error certificates for the dtw filter, computed only where they matter.
//
// The filter's error on a pair is small unless some cell of the pair is nearly zero (the square
// root amplifies the cancellation error of the expanded form there).  Knowing
//     m(s,t) = min over all cells (i,j) of x~(i,j),   x~ = the f16-split MFMA value of |a_i - b_j|^2,
// bounds every cell's error by E / (2 sqrt(m - 2E)) (select.hip).  Tracking m inside the filter
// kernel costs half a v_min3_f32 per cell on all N x M pairs (measured +8 %); instead the first,
// worst-case-margin selection leaves a short list of pairs per target and this kernel computes m for
// those only: one wave per listed pair, plain 32 x 32 tiles (32 source frames x 32 target frames of
// ONE pair), three chained v_mfma_f32_32x32x16_f16 on the same operand records the filter uses,
// v_min3_f32 over the accumulators, one wave reduction at the end.
//
// The kernel is bound by fetching records, not by the matrix pipe (a 128 x 128 pair is 24 KB of
// records for 48 MFMAs), so the list arrives grouped by target (select.hip) and, when the targets
// have at most 128 frames, a wave walks a run of consecutive list entries with the current
// target's (up to) four column tiles held in registers: a pair then costs its source records only.
#include "ssym_internal.hpp"
#include "dtw_filter_kernel.hpp"

namespace ssym {

__global__ __launch_bounds__(64) void certify_kernel(const _Float16 *__restrict__ srcRec,
                                                     const _Float16 *__restrict__ tgtRec,
                                                     const int *__restrict__ srcLen, const int *__restrict__ tgtLen,
                                                     int srcSlots, int srcLead, int tgtSlots,
                                                     const uint32_t *__restrict__ candHdr,
                                                     const uint2 *__restrict__ pairs, uint32_t cap, float outScaleSq,
                                                     float *__restrict__ xmin)
{
    constexpr int REC = kFilterRecHalfs;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x;
    const int rc = lane & 31, kh = lane >> 5;
    const uint32_t n = candHdr[1] ? 0u : min(candHdr[0], cap);   // overflowed list: the host redoes stage 1
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint2 p = pairs[k];
        const int fa = srcLen[p.x], fb = tgtLen[p.y];
        float m = INF;
        if (fa > 0 && fb > 0) {
            // frame f of the source sits in slot first + f (end-aligned when srcLead < 0)
            const int first = srcLead < 0 ? srcSlots - fa : srcLead;
            const _Float16 *sBase = srcRec + ((size_t)p.x * srcSlots + first) * REC + kh * 24;
            const _Float16 *tBase = tgtRec + tgt_rec_offset(p.y, tgtSlots, 0, 0, kh);
            for (int i0 = 0; i0 < fa; i0 += 32) {
                half8 A[kFilterKM];
                load_rec(sBase + (size_t)min(i0 + rc, fa - 1) * REC, A);      // rows past the end repeat the last frame
                for (int j0 = 0; j0 < fb; j0 += 32) {
                    half8 B[kFilterKM];
                    load_tgt_rec(tBase, min(j0 + rc, fb - 1), B);
                    const f32x16 acc = mfma_tile<kFilterKM>(A, B);
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        m = __builtin_fminf(__builtin_fminf(m, acc[r]), acc[r + 1]);
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                m = __builtin_fminf(m, __shfl_xor(m, o));
        }
        if (lane == 0)
            xmin[k] = m * outScaleSq;
    }
}

constexpr int kCertRun = 8;   // consecutive list entries per wave visit (same target, mostly)

template <int NB>   // column tiles of a target kept in registers: targets have <= 32 NB frames
__global__ __launch_bounds__(64) void certify_run_kernel(const _Float16 *__restrict__ srcRec,
                                                         const _Float16 *__restrict__ tgtRec,
                                                         const int *__restrict__ srcLen,
                                                         const int *__restrict__ tgtLen, int srcSlots, int srcLead,
                                                         int tgtSlots, const uint32_t *__restrict__ candHdr,
                                                         const uint2 *__restrict__ pairs, uint32_t cap,
                                                         float outScaleSq, float *__restrict__ xmin)
{
    constexpr int REC = kFilterRecHalfs;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x;
    const int rc = lane & 31, kh = lane >> 5;
    const uint32_t n = candHdr[1] ? 0u : min(candHdr[0], cap);
    for (uint64_t g = (uint64_t)blockIdx.x * kCertRun; g < n; g += (uint64_t)gridDim.x * kCertRun) {
        const uint32_t gEnd = (uint32_t)min(g + kCertRun, (uint64_t)n);
        half8 B[NB][kFilterKM];
        uint32_t cachedT = 0xffffffffu;
        int fb = 0;
        for (uint32_t k = (uint32_t)g; k < gEnd; ++k) {
            const uint2 p = pairs[k];
            if (p.y != cachedT) {
                cachedT = p.y;
                fb = tgtLen[p.y];
                const _Float16 *tBase = tgtRec + tgt_rec_offset(p.y, tgtSlots, 0, 0, kh);
#pragma unroll
                for (int q = 0; q < NB; ++q)      // columns past the end repeat the last frame
                    load_tgt_rec(tBase, max(min(q * 32 + rc, fb - 1), 0), B[q]);
            }
            const int fa = srcLen[p.x];
            float m = INF;
            if (fa > 0 && fb > 0) {
                const int first = srcLead < 0 ? srcSlots - fa : srcLead;
                const _Float16 *sBase = srcRec + ((size_t)p.x * srcSlots + first) * REC + kh * 24;
                half8 A[kFilterKM], An[kFilterKM];
                load_rec(sBase + (size_t)min(rc, fa - 1) * REC, A);
                for (int i0 = 0; i0 < fa; i0 += 32) {
                    load_rec(sBase + (size_t)min(i0 + 32 + rc, fa - 1) * REC, An);   // next row tile in flight
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        if (q * 32 < fb) {
                            const f32x16 acc = mfma_tile<kFilterKM>(A, B[q]);
#pragma unroll
                            for (int r = 0; r < 16; r += 2)
                                m = __builtin_fminf(__builtin_fminf(m, acc[r]), acc[r + 1]);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < kFilterKM; ++v)
                        A[v] = An[v];
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1)
                    m = __builtin_fminf(m, __shfl_xor(m, o));
            }
            if (lane == 0)
                xmin[k] = m * outScaleSq;
        }
    }
}

int32_t launch_certify(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const uint32_t *candHdr,
                       const uint2 *pairs, uint32_t cap, float *xmin)
{
    if (cap == 0)
        return SSYM_OK;
    const double s = src.rec_scale > 0.0 ? src.rec_scale : 1.0;
    const float outScaleSq = (float)(1.0 / (s * s));
    if (tgt.max_frames <= 128) {
        const unsigned runs = (unsigned)std::min<uint64_t>(((uint64_t)cap + kCertRun - 1) / kCertRun,
                                                           (uint64_t)ctx->num_cus * 64);
#define SSYM_CERT_RUN(NB)                                                                                   \
    certify_run_kernel<NB><<<runs, 64, 0, ctx->stream>>>((const _Float16 *)src.rec, (const _Float16 *)tgt.rec, \
                                                         src.len, tgt.len, (int)src.rec_slots, src.rec_lead,  \
                                                         (int)tgt.rec_slots, candHdr, pairs, cap, outScaleSq, xmin)
        switch ((std::max<uint32_t>(tgt.max_frames, 1) + 31) / 32) {
        case 1: SSYM_CERT_RUN(1); break;
        case 2: SSYM_CERT_RUN(2); break;
        case 3: SSYM_CERT_RUN(3); break;
        default: SSYM_CERT_RUN(4); break;
        }
#undef SSYM_CERT_RUN
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        return SSYM_OK;
    }
    const unsigned grid = (unsigned)std::min<uint64_t>(cap, (uint64_t)ctx->num_cus * 64);
    certify_kernel<<<grid, 64, 0, ctx->stream>>>((const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len,
                                                  tgt.len, (int)src.rec_slots, src.rec_lead, (int)tgt.rec_slots,
                                                  candHdr, pairs, cap, outScaleSq, xmin);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
