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
// records for 48 MFMAs), so the list arrives grouped by target (select.hip) and a wave walks a run
// of consecutive list entries with (up to) four column tiles of the current target held in
// registers: a pair then costs its source records only.  Targets longer than 128 frames are covered
// in column groups of 128, the run's minima staying in registers between groups.
#include "ssym_internal.hpp"
#include "dtw_filter_kernel.hpp"

namespace ssym {

constexpr int kCertRun = 8;   // consecutive list entries per wave visit (same target, mostly)

// NB = column tiles of a target kept in registers at a time (32 NB frames); longer targets are
// covered by further column groups, with the running minimum of every entry of the run in registers.
template <int NB>
__global__ __launch_bounds__(64) void certify_run_kernel(const _Float16 *__restrict__ srcRec,
                                                         const _Float16 *__restrict__ tgtRec,
                                                         const int *__restrict__ srcLen,
                                                         const int *__restrict__ tgtLen, int srcSlots, int srcLead,
                                                         int tgtSlots, int tgtMaxFrames,
                                                         const uint32_t *__restrict__ candHdr,
                                                         const uint2 *__restrict__ pairs, uint32_t cap,
                                                         float outScaleSq, float *__restrict__ xmin)
{
    constexpr int REC = kFilterRecHalfs;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x;
    const int rc = lane & 31, kh = lane >> 5;
    const uint32_t n = candHdr[1] ? 0u : min(candHdr[0], cap);   // overflowed list: the host redoes stage 1
    for (uint64_t g = (uint64_t)blockIdx.x * kCertRun; g < n; g += (uint64_t)gridDim.x * kCertRun) {
        float mrun[kCertRun];
#pragma unroll
        for (int e = 0; e < kCertRun; ++e)
            mrun[e] = INF;
        for (int c0 = 0; c0 < tgtMaxFrames; c0 += 32 * NB) {
            half8 B[NB][kFilterKM];
            uint32_t cachedT = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < kCertRun; ++e) {
                const uint64_t k = g + e;
                if (k >= n)
                    continue;                                  // wave-uniform
                const uint2 p = pairs[k];
                const int fb = tgtLen[p.y], fa = srcLen[p.x];
                if (c0 >= fb || fa <= 0)
                    continue;                                  // wave-uniform
                // a target's ONLY candidate (the list is grouped by target) is kept by the second selection whatever
                // its certificate says: none is computed (0 = "no certificate", the worst-case error).  On data with
                // one close neighbour per target that is every entry of the list.
                if ((k == 0 || pairs[k - 1].y != p.y) && (k + 1 >= n || pairs[k + 1].y != p.y)) {
                    mrun[e] = 0.0f;
                    continue;                                  // wave-uniform
                }
                if (p.y != cachedT) {
                    cachedT = p.y;
                    const _Float16 *tBase = tgtRec + tgt_rec_offset(p.y, tgtSlots, 0, 0, kh);
#pragma unroll
                    for (int q = 0; q < NB; ++q)      // columns past the end repeat the last frame
                        load_tgt_rec(tBase, min(c0 + q * 32 + rc, fb - 1), B[q]);
                }
                // frame f of the source sits in slot first + f (end-aligned when srcLead < 0)
                const int first = srcLead < 0 ? srcSlots - fa : srcLead;
                const _Float16 *sBase = srcRec + ((size_t)p.x * srcSlots + first) * REC + kh * 24;
                half8 A[kFilterKM], An[kFilterKM];
                load_rec(sBase + (size_t)min(rc, fa - 1) * REC, A);   // rows past the end repeat the last frame
                float m = mrun[e];
                for (int i0 = 0; i0 < fa; i0 += 32) {
                    load_rec(sBase + (size_t)min(i0 + 32 + rc, fa - 1) * REC, An);   // next row tile in flight
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        if (c0 + q * 32 < fb) {
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
                mrun[e] = m;
            }
        }
#pragma unroll
        for (int e = 0; e < kCertRun; ++e) {
            float m = mrun[e];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                m = __builtin_fminf(m, __shfl_xor(m, o));
            if (lane == 0 && g + e < n)
                xmin[g + e] = m * outScaleSq;
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
    const int maxF = (int)std::max<uint32_t>(tgt.max_frames, 1);
    const unsigned runs = (unsigned)std::min<uint64_t>(((uint64_t)cap + kCertRun - 1) / kCertRun,
                                                       (uint64_t)ctx->num_cus * 64);
#define SSYM_CERT_RUN(NB)                                                                                   \
    certify_run_kernel<NB><<<runs, 64, 0, ctx->stream>>>((const _Float16 *)src.rec, (const _Float16 *)tgt.rec, \
                                                         src.len, tgt.len, (int)src.rec_slots, src.rec_lead,  \
                                                         (int)tgt.rec_slots, maxF, candHdr, pairs, cap,       \
                                                         outScaleSq, xmin)
    switch ((maxF + 31) / 32) {
    case 1: SSYM_CERT_RUN(1); break;
    case 2: SSYM_CERT_RUN(2); break;
    case 3: SSYM_CERT_RUN(3); break;
    default: SSYM_CERT_RUN(4); break;
    }
#undef SSYM_CERT_RUN
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
