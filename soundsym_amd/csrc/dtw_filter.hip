// dtw_filter.hip -- all-pairs DTW cost fill on the f32 MFMA pipe (gfx950), one pair per lane.
//
// Role on the path: replaces the N x M evaluations of the reference's inner loop
// (SoundDictionary::at_distance, src/sound.rs:352-359, called once per target by
// clone_from_dictionary, src/sound.rs:453-454) for the dtw metric.  Output is the f32 cost of every
// (source, target) pair; select.hip picks candidates from it and dtw_exact.hip re-scores them.
//
// Mapping (DESIGN.md "dtw filter kernel"):
//   * one wave = 2 sources x 32 targets = 64 pairs, ONE PAIR PER LANE;
//   * v_mfma_f32_32x32x2_f32: the 32 A-rows are 16 consecutive frames of source 0 interleaved (in
//     groups of four) with 16 frames of source 1, so that the accumulator rows a lane receives
//     ((reg&3) + 8*(reg>>2) + 4*(lane>>5)) are exactly frames 0..15 of ITS source; the 32 B-columns
//     are frame j of 32 DIFFERENT targets, so column (lane&31) is ITS target.  After KS k-steps
//     register r of a lane holds -2 a_r.b_j + |a_r|^2 for its own pair;
//   * the DP column D(., j) of the pair lives in NT*16 VGPRs of the lane; the min-of-three
//     recurrence is purely lane-local (no shuffles, no LDS), VALU work overlaps the MFMA pipe;
//   * |b_j|^2 rides in the pad slot of the B record and is added on the VALU; the local cost is
//     sqrt(|x|) (v_sqrt_f32 with the abs modifier) or |x| for squared-L2.
//
// Numerics: f32 expanded form, so costs of near-identical pairs carry cancellation error; the
// bound used by select.hip is derived there.  This kernel is a FILTER; returned costs and indices
// come from the exact f64 kernel.
#include "ssym_internal.hpp"

#include <algorithm>

namespace ssym {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N>
struct FloatVec {
    float v[N];
};

template <int KSP>
__device__ __forceinline__ void load_half(const float *__restrict__ p, float (&dst)[KSP])
{
    static_assert(KSP % 4 == 0, "half record must be whole float4s");
#pragma unroll
    for (int q = 0; q < KSP / 4; ++q) {
        float4 t = *reinterpret_cast<const float4 *>(p + 4 * q);
        dst[4 * q + 0] = t.x;
        dst[4 * q + 1] = t.y;
        dst[4 * q + 2] = t.z;
        dst[4 * q + 3] = t.w;
    }
}

// One 32x32 tile of the cost block: KS dependent k-steps into one accumulator.
template <int KS, int KSP>
__device__ __forceinline__ f32x16 mfma_tile(const float (&a)[KS], const float (&b)[KSP])
{
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

template <int NT, int KS, bool SQ>
__global__ __launch_bounds__(256, 2) void dtw_filter_kernel(
    const float *__restrict__ srcRec, const float *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int tgtFramesPad, int mPad,
    int nSrcBlocks, float *__restrict__ cmat)
{
    constexpr int KSP = ((KS + 1) + 3) / 4 * 4;
    constexpr int REC = 2 * KSP;
    constexpr int ROWS = NT * 16;
    const float INF = __builtin_inff();

    // XCD-aware task order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch); give each
    // XCD a contiguous range of the (target group, source block) space so that the 32 targets of a
    // group stay in that XCD's L2 while the source blocks stream past.
    const unsigned nBlocks = gridDim.x;
    const unsigned b = blockIdx.x;
    const unsigned xcd = b & 7u, qd = nBlocks >> 3, rm = nBlocks & 7u;
    const unsigned lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
    const int tg = (int)(lin / (unsigned)nSrcBlocks);
    const int sb = (int)(lin % (unsigned)nSrcBlocks);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sp = sb * 4 + wave;      // source pair handled by this wave
    const int col = lane & 31;         // B column / output column: target 32*tg + col
    const int half = lane >> 5;        // operand role: K half; output role: source 2*sp + half

    // ---- A operands: NT tiles x KS k-steps, resident for the whole task --------------------
    // Source frames are END-ALIGNED in their ROWS slots (pack.hip): a source of fa frames sits in
    // rows [ROWS-fa, ROWS); the rows above it carry |a|^2 = +inf, so their D stays +inf.
    float A[NT][KS];
    {
        const int arow = lane & 31;
        const int a_src = 2 * sp + ((arow >> 2) & 1);
        const int a_frm = (arow & 3) + 4 * (arow >> 3);
        const float *abase = srcRec + ((size_t)a_src * ROWS + a_frm) * REC + half * KSP;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            float tmp[KSP];
            load_half<KSP>(abase + (size_t)T * kRowsPerTile * REC, tmp);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                A[T][s] = tmp[s];
        }
    }

    const int fa = srcLen[2 * sp + half];
    const int fb_m1 = tgtLen[32 * tg + col] - 1;
    const int r0 = ROWS - fa;          // first real row of this lane's source

    // wave-uniform column bound: the longest target of the group
    int nCols = fb_m1 + 1;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1)
        nCols = max(nCols, __shfl_xor(nCols, o));
    nCols = __builtin_amdgcn_readfirstlane(nCols);

    // left[i] = D(i, j-1).  The virtual D(r0-1, -1) = 0 that starts the recurrence is planted in
    // the row above the first real row (or handed in as the column's first diagonal when r0 = 0).
    float left[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
        left[i] = (i == r0 - 1) ? 0.0f : INF;
    const float diag0 = (r0 == 0) ? 0.0f : INF;
    float res = INF;

    const float *bbase = tgtRec + ((size_t)(32 * tg + col) * tgtFramesPad) * REC + half * KSP;
    float Bc[KSP], Bn[KSP];
#pragma unroll
    for (int s = 0; s < KSP; ++s)
        Bc[s] = 0.0f;
    if (nCols > 0)
        load_half<KSP>(bbase, Bc);
    f32x16 acc = mfma_tile<KS, KSP>(A[0], Bc);

    for (int j = 0; j < nCols; ++j) {
        const int jn = min(j + 1, nCols - 1);
        load_half<KSP>(bbase + (size_t)jn * REC, Bn);

        const float nb = Bc[KSP - 1];
        float up = INF;
        float diag = (j == 0) ? diag0 : INF;

#pragma unroll
        for (int T = 0; T < NT; ++T) {
            // software pipeline: the MFMA chain of the NEXT tile (next column's first tile after
            // the last one) is issued ahead of this tile's DP, which only needs `acc`
            f32x16 accn;
            if (T + 1 < NT)
                accn = mfma_tile<KS, KSP>(A[T + 1], Bc);
            else
                accn = mfma_tile<KS, KSP>(A[0], Bn);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = T * 16 + r;
                const float x = acc[r] + nb;
                const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
                const float m = __builtin_fminf(__builtin_fminf(up, diag), left[idx]);
                diag = left[idx];
                const float cur = c + m;
                left[idx] = cur;
                up = cur;
            }
            acc = accn;
        }

        res = (j == fb_m1) ? up : res;   // D(fa-1, fb-1): the bottom row at the target's last frame
#pragma unroll
        for (int s = 0; s < KSP; ++s)
            Bc[s] = Bn[s];
    }

    cmat[(size_t)(2 * sp + half) * mPad + 32 * tg + col] = res;
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int pick_nt(int frames_pad)
{
    int nt = frames_pad / kRowsPerTile;
    const int avail[] = {1, 2, 3, 4, 6, 8};
    for (int a : avail)
        if (nt <= a)
            return a;
    return -1;
}

bool filter_supported(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    if (ctx->band >= 0)
        return false;   // banded shapes run on the exact kernel (DESIGN.md "limits")
    if (src.dim != tgt.dim || src.ks != tgt.ks)
        return false;
    if (src.ks > 7)
        return false;   // dim <= 13 on the MFMA path for now
    if (pick_nt((int)src.frames_pad) < 0)
        return false;   // more than 128 source frames
    return src.n > 0 && tgt.n > 0;
}

template <int NT, bool SQ>
static void launch_one(hipStream_t st, const SegmentSet &src, const SegmentSet &tgt, float *cmat)
{
    const int nSrcBlocks = (int)src.n_pad / 8;
    const int nTgtGroups = (int)tgt.n_pad / 32;
    dim3 grid((unsigned)nSrcBlocks * (unsigned)nTgtGroups);
    dtw_filter_kernel<NT, 7, SQ><<<grid, 256, 0, st>>>(src.rec, tgt.rec, src.len, tgt.len,
                                                       (int)tgt.frames_pad, (int)tgt.n_pad,
                                                       nSrcBlocks, cmat);
}

int32_t launch_dtw_filter(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, float *cmat)
{
    const int nt = pick_nt((int)src.frames_pad);
    if (nt < 0 || (int)src.frames_pad != nt * kRowsPerTile) {
        ctx->err = "dtw filter: source records not padded to a supported tile count";
        return SSYM_E_UNSUPPORTED;
    }
    hipStream_t st = ctx->stream;
    const bool sq = ctx->squared != 0;
#define SSYM_LAUNCH_NT(N)                                                      \
    case N:                                                                    \
        if (sq) launch_one<N, true>(st, src, tgt, cmat);                       \
        else launch_one<N, false>(st, src, tgt, cmat);                         \
        break;
    switch (nt) {
        SSYM_LAUNCH_NT(1)
        SSYM_LAUNCH_NT(2)
        SSYM_LAUNCH_NT(3)
        SSYM_LAUNCH_NT(4)
        SSYM_LAUNCH_NT(6)
        SSYM_LAUNCH_NT(8)
    default:
        ctx->err = "dtw filter: unsupported tile count";
        return SSYM_E_UNSUPPORTED;
    }
#undef SSYM_LAUNCH_NT
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
