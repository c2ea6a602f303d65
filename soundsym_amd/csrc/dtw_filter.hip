// dtw_filter.hip -- all-pairs DTW cost fill on the f16 matrix pipe (gfx950), one pair per lane.
//
// Role on the path: replaces the N x M evaluations of the reference's inner loop
// (SoundDictionary::at_distance, src/sound.rs:352-359, called once per target by
// clone_from_dictionary, src/sound.rs:453-454) for the dtw metric.  Output is the f32 cost of every
// (source, target) pair; select.hip picks candidates from it and dtw_exact.hip re-scores them.
// The kernel itself is in dtw_filter_kernel.hpp; this file builds its operand records and
// dispatches the (tiles per wave, row blocks) instantiation for the dictionary's longest segment.
//
// Numerics: operands are scaled by a common power of two s so that max |s v| < 64 and split into f16
// pieces (three record layouts, filter_pieces() in ssym_internal.hpp: up to 13 values per frame the
// source in two pieces and the target in one, K = 32; wider frames one piece on both sides, K = 48);
// the squared norms of the REPRESENTED frames ride along in two or three pieces, products are exact in
// the f32 accumulator.  The filter's error bound is derived in select.hip; returned costs and indices
// always come from the exact f64 kernel.
#include "ssym_internal.hpp"
#include "dtw_filter_kernel.hpp"
#include "dtw_filter_pk_kernel.hpp"
#include "dtw_filter_sp_kernel.hpp"
#include "dtw_band_kernel.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace ssym {

int filter_pieces(int dim)
{
    static const bool k48 = ssym_knob("SSYM_FILTER_K48") != nullptr;      // measurements: the symmetric K = 48 layout
    return dim > kFilterMaxDim2 ? 1 : (k48 ? 2 : 3);
}

// One thread per (segment, record slot).  Frame f of a segment with nf frames goes to slot
// lead + f (lead = -1: END-ALIGNED, slot frames_pad - nf + f, the unbanded kernel's source layout).
// Source slots that hold no frame (and all slots of padding segments) get |a|^2 = +inf so that
// their DP cells stay at +inf; empty target slots are all-zero records.
__global__ void build_filter_records_kernel(const double *__restrict__ raw, const uint64_t *__restrict__ off,
                                            const uint32_t *__restrict__ perm, uint32_t n, uint32_t dim,
                                            uint32_t dimUse, uint32_t frames_pad, int is_source,
                                            int lead, int pieces, double scale, _Float16 *__restrict__ rec,
                                            unsigned *__restrict__ resid)
{
    const uint32_t s = blockIdx.y;                              // record slot of the segment, < n_pad
    const uint32_t seg = perm[s];                               // the segment it holds (0xffffffff: none)
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= frames_pad)
        return;
    _Float16 out[kFilterRecHalfs];
#pragma unroll
    for (int i = 0; i < kFilterRecHalfs; ++i)
        out[i] = (_Float16)0.0f;
    const uint32_t nf = seg < n ? (uint32_t)(off[seg + 1] - off[seg]) : 0u;
    const uint32_t first = lead < 0 ? frames_pad - nf : (uint32_t)lead;
    const bool real = slot >= first && slot < first + nf;
    const uint32_t f = slot - first;
    // slot of the first |a|^2 piece: behind the product slots (layout 3: fixed at 28, K = 32)
    const int nbase = pieces == 3 ? 28 : (pieces == 2 ? 3 : 1) * (int)dimUse;
    const int npieces = pieces == 3 ? 2 : 3;                       // f16 pieces per squared norm
    if (!real && is_source)
        out[filter_slot_offset(nbase)] = (_Float16)__builtin_inff();   // |a|^2 = +inf
    if (real) {
        const double *p = raw + (off[seg] + f) * dim;
        double nrm = 0.0, res2 = 0.0;
        for (uint32_t e = 0; e < dimUse; ++e) {                  // frames wider than 42 values: the first 42
            const double v = p[e] * scale;                       // exact: scale is a power of two
            const _Float16 h1 = (_Float16)v;
            // second piece: both sides in layout 2; in layout 3 the source, and the target's first two values
            const bool two = pieces == 2 || (pieces == 3 && (is_source || e < 2));
            const _Float16 h2 = two ? (_Float16)(v - (double)h1) : (_Float16)0.0f;
            const double vh = (double)h1 + (double)h2;           // the value the MFMA will see
            nrm += vh * vh;                                      // norms of the REPRESENTED frame
            res2 += (v - vh) * (v - vh);                         // ... and how far it lies from the frame itself
            const _Float16 m1 = (_Float16)(-2.0f * (float)h1), m2 = (_Float16)(-2.0f * (float)h2);
            if (pieces == 2) {
                out[filter_slot_offset(3 * e + 0)] = is_source ? m1 : h1;
                out[filter_slot_offset(3 * e + 1)] = is_source ? m1 : h2;
                out[filter_slot_offset(3 * e + 2)] = is_source ? m2 : h1;
            } else if (pieces == 3) {
                // a . b~ = (a1 + a2) b1 for every value, + a1 b2 for the first two: b~ = b1 (+ b2), exactly the
                // frame whose norm rides along, so the accumulator is |a~ - b~|^2 with no cancellation error
                out[filter_slot_offset(2 * e + 0)] = is_source ? m1 : h1;
                out[filter_slot_offset(2 * e + 1)] = is_source ? m2 : h1;
                if (e < 2)
                    out[filter_slot_offset(26 + e)] = is_source ? m1 : h2;
            } else {
                out[filter_slot_offset(e)] = is_source ? m1 : h1;
            }
        }
        // |frame - represented frame|, unscaled, rounded up: a local cost of this frame moves by at most that much when the
        // record stands in for the frame (the margin's rounding term, dtw_margin.hpp); the slot keeps its frames' maximum
        const float r = __double2float_ru(sqrt(res2) / scale * 1.0000001);
        if (r > 0.0f)
            atomicMax(&resid[s], __float_as_uint(r));
        const _Float16 p1 = (_Float16)nrm;
        const _Float16 p2 = (_Float16)(nrm - (double)p1);
        const _Float16 p3 = (_Float16)(nrm - (double)p1 - (double)p2);
        const int mine = nbase + (is_source ? 0 : npieces), other = nbase + (is_source ? npieces : 0);
        out[filter_slot_offset(mine + 0)] = p1;
        out[filter_slot_offset(mine + 1)] = p2;
        out[filter_slot_offset(other + 0)] = (_Float16)1.0f;
        out[filter_slot_offset(other + 1)] = (_Float16)1.0f;
        if (npieces == 3) {
            out[filter_slot_offset(mine + 2)] = p3;
            out[filter_slot_offset(other + 2)] = (_Float16)1.0f;
        }
    }
    if (is_source) {
        _Float16 *dst = rec + ((size_t)s * frames_pad + slot) * kFilterRecHalfs;
#pragma unroll
        for (int i = 0; i < kFilterRecHalfs; ++i)
            dst[i] = out[i];
    } else {
        // group-major target layout (dtw_filter_kernel.hpp, tgt_rec_offset); n_pad is a multiple of 32
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < kFilterKM; ++m) {
                _Float16 *dst = rec + tgt_rec_offset(s, frames_pad, slot, m, h);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    dst[i] = out[h * 24 + m * 8 + i];
            }
    }
}

// banded layout: slot s of a source holds frame s - r; tile T of column j reads slots j + 16T .. +15
static uint32_t band_slots(int radius, const SegmentSet &src, const SegmentSet &tgt)
{
    const uint32_t kb = (uint32_t)(2 * radius + 1 + 15) / 16 * 16;
    return std::max<uint32_t>(radius + src.max_frames, std::max<uint32_t>(tgt.max_frames, 1) + kb) + 1;
}
static size_t band_lds_bytes(int radius, const SegmentSet &src, const SegmentSet &tgt)
{
    return ((size_t)2 * band_slots(radius, src, tgt) * kFilterRecHalfs + kBandImagePad) * sizeof(_Float16);
}

// A Sakoe-Chiba band the banded kernel cannot take -- more than 6 tiles of diagonals (r > 47), or a source pair that
// does not fit the LDS -- is served by the UNBANDED filter: a band only removes paths, so the unbanded cost bounds the
// banded one from below, and the lower-bound cascade (select.hip, "wide frames") with the banded exact kernel does
// the rest.
bool filter_band_as_bound(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    if (ctx->band < 0)
        return false;
    return 2 * ctx->band + 1 > 6 * 16 || band_lds_bytes(ctx->band, src, tgt) > 160 * 1024 - 64;   // (+ the kernel's few static bytes)
}

bool filter_supported(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    if (src.light || tgt.light)
        return false;   // packed for the exact kernels only
    if (src.dim != tgt.dim)
        return false;   // (frames wider than 42 values: capi.hip decides, the filter is then a lower bound only)
    if ((ctx->band < 0 || filter_band_as_bound(ctx, src, tgt)) && filter_shape((int)src.max_frames).nt == 0)
        return false;   // more than 4096 source frames
    if (!std::isfinite(src.max_abs) || !std::isfinite(tgt.max_abs))
        return false;   // inf / NaN features: exact kernel keeps IEEE semantics
    if (!std::isfinite(src.max_sqnorm_all) || !std::isfinite(tgt.max_sqnorm_all))
        return false;   // squared frame norms beyond f32: no common scale serves values and norms
    return src.n > 0 && tgt.n > 0;
}

// Common power-of-two scale s: max |s v| < 64 (in [32, 64) unless the norms below ask for less; 1 when
// everything is zero), AND every scaled squared frame norm below the f16 maximum -- the records carry
// |a|^2 as three f16 pieces, and from 17 values per frame on 64^2 * dim passes 65504: the first piece
// would be +inf, the third NaN, and the pair would silently drop out of the search.  A smaller s keeps
// every assumption of the error model (|s v| < 64 is an upper bound; its absolute terms are priced
// through 1 / s^2, dtw_margin.hpp).  max_sqnorm is rounded up and the records' norms are those of the
// f16-rounded frames (<= (1 + 2^-11)^2 larger), hence the slack.
static double common_scale(const SegmentSet &src, const SegmentSet &tgt)
{
    const double m = std::max(src.max_abs, tgt.max_abs);
    if (!(m > 0.0))
        return 1.0;
    int e = 0;
    (void)std::frexp(m, &e);          // m = f * 2^e, f in [0.5, 1)
    e = std::min(std::max(6 - e, -100), 100);
    const double sq = std::max(src.max_sqnorm_all, tgt.max_sqnorm_all);
    while (e > -200 && sq * std::ldexp(1.0, 2 * e) * 1.01 >= 65000.0)
        --e;
    const double ideal = std::ldexp(1.0, e);
    // The dictionary's records are cached on the scale they were built with.  A batch of quieter targets would ask
    // for a larger scale and the next louder one for the smaller again -- a full pass over the dictionary per call,
    // its time depending on the previous call.  A smaller scale than the ideal one is always admissible (|s v| < 64
    // and the norm bound only get easier; the error model prices the scale through 1 / s^2), so the cached scale is
    // kept while it is at most 16 times smaller than the ideal: four bits of the f16 pieces' headroom, not a rebuild.
    if (src.rec && src.rec_scale > 0.0 && src.rec_scale <= ideal && src.rec_scale * 16.0 >= ideal)
        return src.rec_scale;
    return ideal;
}

static int32_t ensure_records(ssym_ctx *ctx, const SegmentSet &set, double scale, uint32_t slots, int lead)
{
    const size_t bytes = (size_t)set.n_pad * slots * kFilterRecHalfs * sizeof(_Float16);
    if (set.rec && set.rec_scale == scale && set.rec_bytes == bytes && set.rec_slots == slots &&
        set.rec_lead == lead)
        return SSYM_OK;
    if (set.rec && set.rec_bytes != bytes) {
        dev_free(ctx, set.rec);
        set.rec = nullptr;
    }
    if (!set.rec) {
        int32_t rca = dev_alloc(ctx, &set.rec, bytes);
        if (rca != SSYM_OK)
            return rca;
    }
    set.rec_bytes = bytes;
    dim3 grid((slots + 63) / 64, set.n_pad);
    const int dimUse = filter_dim_used((int)set.dim);
    unsigned *resid = (unsigned *)(set.max_sqnorm + 2 * (size_t)set.n_pad);      // (non-negative floats: integer maximum)
    SSYM_HIP_CHECK(ctx, hipMemsetAsync(resid, 0, sizeof(unsigned) * set.n_pad, ctx->stream));
    build_filter_records_kernel<<<grid, 64, 0, ctx->stream>>>(set.raw, set.off, set.perm, set.n, set.dim,
                                                              (uint32_t)dimUse, slots, set.is_source ? 1 : 0, lead,
                                                              filter_pieces(dimUse), scale,
                                                              (_Float16 *)set.rec, resid);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    set.rec_scale = scale;
    set.rec_slots = slots;
    set.rec_lead = lead;
    return SSYM_OK;
}

// DP cells per lane (64 pairs of a task share them) a launch of the unbanded filter evaluates, padding included: per
// source pair the rows of the passes it runs (dtw_filter_kernel skips leading passes that hold only padding;
// dtw_filter_sp_kernel skips leading row blocks of `rowBlock` rows), per target group the columns of its longest member
// (at least `minCols`).  Rows depend on the pair only and columns on the group only, so the sum is a product.
static unsigned long long launch_cells(const SegmentSet &src, const SegmentSet &tgt, int spBase, int nSrcPairs,
                                       int rowOrigin, int passRows, int nPasses, int rowBlock, int minCols, int pairsPerTask = 1,
                                       bool skipTile = false)
{
    auto len = [](const SegmentSet &set, uint32_t slot) -> uint32_t {
        if (slot >= set.n)
            return 0u;
        const uint32_t s = set.h_perm[slot];
        return (uint32_t)(set.h_off[s + 1] - set.h_off[s]);
    };
    unsigned long long rows = 0, cols = 0;
    for (uint32_t g = 0; g < tgt.n_pad / 32; ++g) {
        uint32_t m = 0;
        for (uint32_t t = 0; t < 32; ++t)
            m = std::max(m, len(tgt, 32 * g + t));
        cols += rowBlock ? std::max<uint32_t>(m, (uint32_t)minCols) : m;
    }
    for (int sp = spBase; sp < spBase + nSrcPairs; ++sp) {
        int longer = (int)std::max(len(src, 2u * sp), len(src, 2u * sp + 1u));
        if (pairsPerTask > 1) {                  // multi-pair tasks: every pair of a task starts at the task's first row block
            const int first = spBase + (sp - spBase) / pairsPerTask * pairsPerTask;
            for (int k = first; k < std::min(first + pairsPerTask, spBase + nSrcPairs); ++k)
                longer = std::max(longer, (int)std::max(len(src, 2u * k), len(src, 2u * k + 1u)));
        }
        const int r0min = (int)src.frames_pad - longer;
        if (rowBlock) {
            const int sk = std::min(std::max(r0min - rowOrigin, 0), 15) / rowBlock;
            rows += (unsigned long long)(passRows - sk * rowBlock);
        } else {
            const int firstPass = std::min(std::max(r0min - rowOrigin, 0) / passRows, nPasses - 1);
            rows += (unsigned long long)(nPasses - firstPass) * passRows;
            if (skipTile && r0min - (rowOrigin + firstPass * passRows) >= 17)      // SKIP0: the first pass's empty first tile
                rows -= 16;
        }
    }
    return rows * cols;
}

template <int NT, bool SQ, int KU>
static void launch_one(hipStream_t st, const SegmentSet &src, const SegmentSet &tgt, int nPasses,
                       int gridBlocks, float outScale, float *handoff, unsigned *taskCtr, float *cmat,
                       const float *abandon, unsigned long long *colCtr, const uint32_t *candSlot,
                       int spBase, int nSrcPairs, int rowOrigin, unsigned long long *cellsOut, bool skipTile = false)
{
    // sources of at most 16 frames (one tile, one pass): three waves per SIMD, see filter_ring() -- 11 %
    // faster there; at 32 frames the gain was within 3 % and cost spills
    // NT == 8 (experiment, SSYM_FILTER_NT8=1): the whole 128-row column of a pair in ONE wave's registers, one wave per
    // SIMD with the whole register file -- no second pass, no hand-off rows
    constexpr int OCC = NT == 1 ? 3 : NT == 8 ? 1 : 2;
    gridBlocks = gridBlocks / 2 * OCC;
    const int nTgtGroups = (int)tgt.n_pad / 32;
    const int nTasks = nSrcPairs * nTgtGroups;            // one wave's 64 pairs each
    const int blocksWanted = (nTasks + kFilterWavesPerBlock - 1) / kFilterWavesPerBlock;
    const int grid = std::min(gridBlocks, (blocksWanted + 7) / 8 * 8);
    // tasks per grab: several SHORT tasks at a time so that the L2 atomics stay invisible (at 16 frames a
    // task is ~1 us of work), one at a time once a task is long enough to matter for the tail; always
    // at least 16 grabs per wave
    // (a task's size by the MEAN target length, not the longest: a ragged set's typical task is what the grab amortises)
    const long meanTgt = tgt.n ? (long)((tgt.total_frames + tgt.n - 1) / tgt.n) : 1;
    const long cellsPerTask = (long)(16 * NT * nPasses) * std::max<long>(meanTgt, 1);
    int taskChunk = (int)std::max(1L, std::min(8L, 8192 / std::max(1L, cellsPerTask)));
    taskChunk = std::max(1, std::min(taskChunk, nTasks / (grid * kFilterWavesPerBlock * 16)));
    // SSYM_FILTER_PK=1 (experiment, off in the product: +1...1.5 % measured): 64-row passes without early
    // abandoning on the two-block kernel whose additions are packed (dtw_filter_pk_kernel.hpp)
    static const bool pkOn = ssym_knob("SSYM_FILTER_PK") && atoi(ssym_knob("SSYM_FILTER_PK")) != 0;
    if (NT == 4 && !abandon && pkOn) {
        dtw_filter_pk_kernel<SQ, KU><<<dim3(grid), 64 * kFilterWavesPerBlock, 0, st>>>(
            (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad, nPasses,
            (int)tgt.frames_pad, (int)tgt.n_pad, nSrcPairs, nTasks, taskChunk, outScale, handoff, taskCtr, cmat,
            rowOrigin, spBase);
        return;
    }
    // single-pass launches without early abandoning: the kernel that pipelines across tasks and skips padding rows
    // (dtw_filter_sp_kernel.hpp; SSYM_FILTER_SP=0 keeps dtw_filter_kernel for A/B measurements, same bits either way)
    static const bool spOn = !(ssym_knob("SSYM_FILTER_SP") && atoi(ssym_knob("SSYM_FILTER_SP")) == 0);
    // (with three operand planes its LDS -- ring and staging block -- admits two workgroups per CU up to two tiles)
    // sources of at most 16 frames: three source pairs per wave (two with three operand planes: LDS), dtw_filter_sp_kernel MP
    // (SSYM_SP_MULTIPAIR, read per launch: 0 = never, 2 = whatever the size -- the tests' way to small multi-pair launches)
    const char *mpKnob = ssym_knob("SSYM_SP_MULTIPAIR");
    const bool mpOn = !(mpKnob && atoi(mpKnob) == 0), mpAlways = mpKnob && atoi(mpKnob) == 2;
    if constexpr (NT == 1) {
        // (... when the multi-pair tasks still give every wave slot of the chip one: a 284 x 55 search is 96 of them, each
        //  sweeping its group's longest target three pairs wide while nine tenths of the SIMDs idle -- one pair per wave there)
        constexpr int MPN = KU == 2 ? 3 : 2;
        const long mpTasksWanted = (long)((nSrcPairs + MPN - 1) / MPN) * nTgtGroups;
        if (!abandon && nPasses == 1 && spOn && mpOn && (mpAlways || mpTasksWanted >= (long)gridBlocks / OCC * 2 * kFilterWavesPerBlock)) {
            const int taskPairs = (nSrcPairs + MPN - 1) / MPN;
            const int mpTasks = taskPairs * nTgtGroups;
            const int gridMp = std::min(gridBlocks / OCC * 2, ((mpTasks + kFilterWavesPerBlock - 1) / kFilterWavesPerBlock + 7) / 8 * 8);
            const long cellsMp = (long)(16 * MPN) * std::max<long>(meanTgt, 1);
            int chunkMp = (int)std::max(1L, std::min(8L, 8192 / std::max(1L, cellsMp)));
            chunkMp = std::max(1, std::min(chunkMp, mpTasks / std::max(1, gridMp * kFilterWavesPerBlock * 16)));
            static const char *pbKnob = ssym_knob("SSYM_SP_PAIRBLOCK");
            int pairBlock = std::max(16, (1 << 20) / (MPN * 2 * 16 * kFilterRecHalfs * 2));
            if (pbKnob)
                pairBlock = atoi(pbKnob) > 0 ? atoi(pbKnob) : taskPairs;
            dtw_filter_sp_kernel<MPN, SQ, 2, KU, kSpRowBlock, true><<<dim3(gridMp), 64 * kFilterWavesPerBlock, 0, st>>>(
                (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad,
                (int)tgt.frames_pad, (int)tgt.n_pad, taskPairs, nSrcPairs, chunkMp, outScale, taskCtr, cmat, rowOrigin, spBase,
                std::min(pairBlock, std::max(taskPairs, 1)));
            if (cellsOut)
                *cellsOut += launch_cells(src, tgt, spBase, nSrcPairs, rowOrigin, 16, 1, kSpRowBlock, kSpRing, MPN);
            return;
        }
    }
    if constexpr (NT <= 3 && (KU == 2 || NT <= 2)) {
        if (!abandon && nPasses == 1 && spOn) {
#ifndef SSYM_SP_OCC_NT2
#define SSYM_SP_OCC_NT2 3        // two-tile tasks at three waves per SIMD (162 registers since the loop copies own their operand registers); 2: tools
#endif
            constexpr int OCCSP = (NT == 1 && KU == 2) ? 3 : (NT == 2 && KU == 2) ? SSYM_SP_OCC_NT2 : 2;
            const int gridSp = std::min(gridBlocks / OCC * OCCSP, (blocksWanted + 7) / 8 * 8);
            // source pairs per block of the task order: about 1 MB of their records (an XCD's L2 holds 4 MB: the block, the
            // XCD's target groups, the cost rows being written); SSYM_SP_PAIRBLOCK overrides (0: one block, the old order)
            static const char *pbKnob = ssym_knob("SSYM_SP_PAIRBLOCK");
            int pairBlock = std::max(16, (1 << 20) / (2 * 16 * NT * kFilterRecHalfs * 2));
            if (pbKnob)
                pairBlock = atoi(pbKnob) > 0 ? atoi(pbKnob) : nSrcPairs;
            dtw_filter_sp_kernel<NT, SQ, OCCSP, KU, kSpRowBlock><<<dim3(gridSp), 64 * kFilterWavesPerBlock, 0, st>>>(
                (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad,
                (int)tgt.frames_pad, (int)tgt.n_pad, nSrcPairs, nTasks, taskChunk, outScale, taskCtr, cmat, rowOrigin, spBase,
                std::min(pairBlock, std::max(nSrcPairs, 1)));
            if (cellsOut)
                *cellsOut += launch_cells(src, tgt, spBase, nSrcPairs, rowOrigin, 16 * NT, 1, kSpRowBlock, kSpRing);
            return;
        }
    }
    if constexpr (NT == 3 || NT == 4) {
        if (skipTile && !abandon) {       // a class whose shorter sources leave the first tile of their first pass empty
            if (cellsOut)
                *cellsOut += launch_cells(src, tgt, spBase, nSrcPairs, rowOrigin, 16 * NT, nPasses, 0, 0, 1, true);
            dtw_filter_kernel<NT, SQ, OCC, false, KU, true><<<dim3(grid), 64 * kFilterWavesPerBlock, 0, st>>>(
                (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad, nPasses,
                (int)tgt.frames_pad, (int)tgt.n_pad, nSrcPairs, nTasks, taskChunk, outScale, handoff, taskCtr, cmat,
                nullptr, nullptr, nullptr, rowOrigin, spBase);
            return;
        }
    }
    if (cellsOut && !abandon)
        *cellsOut += launch_cells(src, tgt, spBase, nSrcPairs, rowOrigin, 16 * NT, nPasses, 0, 0);
    if (abandon)
        dtw_filter_kernel<NT, SQ, OCC, true, KU><<<dim3(grid), 64 * kFilterWavesPerBlock, 0, st>>>(
            (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad, nPasses,
            (int)tgt.frames_pad, (int)tgt.n_pad, nSrcPairs, nTasks, taskChunk, outScale, handoff, taskCtr, cmat,
            abandon, colCtr, candSlot, rowOrigin, spBase);
    else
        dtw_filter_kernel<NT, SQ, OCC, false, KU><<<dim3(grid), 64 * kFilterWavesPerBlock, 0, st>>>(
            (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)src.frames_pad, nPasses,
            (int)tgt.frames_pad, (int)tgt.n_pad, nSrcPairs, nTasks, taskChunk, outScale, handoff, taskCtr, cmat,
            nullptr, nullptr, nullptr, rowOrigin, spBase);
}

template <int NTB, int WB, int OCC, bool SQ, int LASTN, bool PRUNE>
static int32_t launch_band_cfg2(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, uint32_t slots,
                                size_t lds, float outScale, float *cmat, const float *abandon,
                                unsigned long long *colCtr, const uint32_t *candSlot)
{
    const int nTgtBlocks = (int)tgt.n_pad / (32 * WB);
    const int nTasks = ((int)src.n_pad / 2) * nTgtBlocks;
    const int grid = std::max(1, std::min(ctx->num_cus, nTasks));
    int32_t rc = ensure(ctx, ctx->handoff, 8 * sizeof(unsigned));      // the banded kernel's task counter
    if (rc != SSYM_OK)
        return rc;
    unsigned *taskCtr = (unsigned *)ctx->handoff.ptr;
    rc = zero_words(ctx, taskCtr, sizeof(unsigned));
    if (rc != SSYM_OK)
        return rc;
    // operand planes the kernel multiplies: record layout 3 leaves the third one zero
    const bool two = filter_mfmas(filter_pieces(filter_dim_used((int)src.dim)), filter_dim_used((int)src.dim)) == 2;
    auto kern = two ? dtw_band_kernel<NTB, WB, OCC, SQ, LASTN, PRUNE, 2> : dtw_band_kernel<NTB, WB, OCC, SQ, LASTN, PRUNE, 3>;
    // two columns per source read (dtw_band_kernel.hpp, PC): an experiment that measured nothing (LAB.md R4.3), kept out of
    // the product library -- tools/band_paircols_ab.py builds its own with EXTRA=-DSSYM_BAND_PAIRCOLS_BUILD
#ifdef SSYM_BAND_PAIRCOLS_BUILD
    if constexpr (!PRUNE && LASTN == 1 && NTB >= 4 && OCC == 2) {
        const char *pc = ssym_knob("SSYM_BAND_PAIRCOLS");
        if (pc && atoi(pc) != 0)
            kern = two ? dtw_band_kernel<NTB, WB, OCC, SQ, LASTN, PRUNE, 2, true> : dtw_band_kernel<NTB, WB, OCC, SQ, LASTN, PRUNE, 3, true>;
    }
#endif
    if (lds > 64 * 1024)
        SSYM_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3(grid), 64 * WB, lds, ctx->stream>>>(
        (const _Float16 *)src.rec, (const _Float16 *)tgt.rec, src.len, tgt.len, (int)slots, ctx->band,
        (int)tgt.frames_pad, (int)tgt.n_pad, nTgtBlocks, nTasks, taskCtr, outScale, cmat, abandon, colCtr, candSlot);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

template <int NTB, int WB, int OCC, bool SQ, int LASTN>
static int32_t launch_band_cfg(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, uint32_t slots,
                               size_t lds, float outScale, float *cmat, const float *abandon,
                               unsigned long long *colCtr, const uint32_t *candSlot)
{
    return abandon ? launch_band_cfg2<NTB, WB, OCC, SQ, LASTN, true>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot)
                   : launch_band_cfg2<NTB, WB, OCC, SQ, LASTN, false>(ctx, src, tgt, slots, lds, outScale, cmat, nullptr, nullptr, nullptr);
}

// Up to 5 tiles of diagonals (r <= 39) run two waves per SIMD (8-wave workgroups); at 4 and 5 tiles
// that costs a few register spills but measured 10 % faster than one wave per SIMD at r = 32.
// 6 tiles run one wave per SIMD (4-wave workgroups) with the whole register file.  (Only the variant a tile count
// uses is instantiated: the other half of the kernels doubled the compile time for a tuning knob.)
template <int NTB, bool SQ>
static int32_t launch_band(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, uint32_t slots,
                           size_t lds, float outScale, float *cmat, const float *abandon, unsigned long long *colCtr, const uint32_t *candSlot)
{
    // radii that are multiples of 8 end exactly one diagonal into their last tile
    const bool last1 = 2 * ctx->band + 1 == 16 * (NTB - 1) + 1;
    if constexpr (NTB <= 5)
        return last1 ? launch_band_cfg<NTB, 8, 2, SQ, 1>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot)
                     : launch_band_cfg<NTB, 8, 2, SQ, 16>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot);
    else
        return last1 ? launch_band_cfg<NTB, 4, 1, SQ, 1>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot)
                     : launch_band_cfg<NTB, 4, 1, SQ, 16>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot);
}

static int32_t launch_dtw_filter_banded(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, float *cmat,
                                        const float *abandon, unsigned long long *colCtr, const uint32_t *candSlot)
{
    if (tgt.n_pad % kBandTgtQuantum != 0 || src.n_pad % 2 != 0) {
        ctx->err = "dtw band filter: segment set not padded for the banded kernel";
        return SSYM_E_UNSUPPORTED;
    }
    const uint32_t slots = band_slots(ctx->band, src, tgt);
    const size_t lds = band_lds_bytes(ctx->band, src, tgt);
    const double scale = common_scale(src, tgt);
    int32_t rc = ensure_records(ctx, src, scale, slots, ctx->band);
    if (rc != SSYM_OK)
        return rc;
    rc = ensure_records(ctx, tgt, scale, tgt.frames_pad, 0);
    if (rc != SSYM_OK)
        return rc;
    const bool sq = ctx->squared != 0;
    const float outScale = (float)(sq ? 1.0 / (scale * scale) : 1.0 / scale);
    const int ntb = (2 * ctx->band + 1 + 15) / 16;
#define SSYM_BCASE(N_)                                                                                  \
    case N_:                                                                                            \
        return sq ? launch_band<N_, true>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot)   \
                  : launch_band<N_, false>(ctx, src, tgt, slots, lds, outScale, cmat, abandon, colCtr, candSlot);
    switch (ntb) {
        SSYM_BCASE(1) SSYM_BCASE(2) SSYM_BCASE(3) SSYM_BCASE(4) SSYM_BCASE(5) SSYM_BCASE(6)
    default:
        ctx->err = "dtw band filter: band radius too large";
        return SSYM_E_UNSUPPORTED;
    }
#undef SSYM_BCASE
}

int32_t ensure_filter_records(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *scale_out)
{
    const double scale = common_scale(src, tgt);
    int32_t rc = ctx->band >= 0 && !filter_band_as_bound(ctx, src, tgt)
                     ? ensure_records(ctx, src, scale, band_slots(ctx->band, src, tgt), ctx->band)
                     : ensure_records(ctx, src, scale, src.frames_pad, -1);
    if (rc == SSYM_OK)
        rc = ensure_records(ctx, tgt, scale, tgt.frames_pad, 0);
    if (scale_out)
        *scale_out = scale;
    return rc;
}

int32_t launch_dtw_filter(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, float *cmat,
                          const float *abandon, unsigned long long *colCtr, const uint32_t *candSlot)
{
    ctx->filter_launches = 1;
    if (ctx->band >= 0 && !filter_band_as_bound(ctx, src, tgt))
        return launch_dtw_filter_banded(ctx, src, tgt, cmat, abandon, colCtr, candSlot);
    FilterShape shape = filter_shape((int)src.max_frames);
    if (shape.nt == 0 || (int)src.frames_pad != shape.rows() || src.n_pad % 8 != 0 || tgt.n_pad % 32 != 0) {
        ctx->err = "dtw filter: segment set not padded for the filter kernel";
        return SSYM_E_UNSUPPORTED;
    }
    // Early abandoning: the work of a pruned task is about the area where D <= threshold, counted in whole
    // row passes x columns, so lower passes waste less: with 32-row passes the headline grid takes 3.1 ms
    // instead of 4.6.  But a task that cannot be cut short pays two more hand-offs per 128 rows (+17 % on
    // data without close pairs).  Results do not depend on the pass height, so it follows the data: 32 rows
    // when the previous pruned call on this context swept less than a quarter of its cells, 64 otherwise
    // (and on the first call).  SSYM_PRUNE_NT=2|4 pins it for measurements.
    const char *pin = abandon ? ssym_knob("SSYM_PRUNE_NT") : nullptr;
    const int pinned = pin ? atoi(pin) : 0;
    const int pruneNt = pinned == 2 || pinned == 4 ? pinned : (ctx->prune_swept < 0.25f ? 2 : 4);
    const FilterShape setShape = shape;
    if (abandon && shape.nt == 4 && pruneNt == 2)
        shape = FilterShape{2, shape.rb * 2};
    const double scale = common_scale(src, tgt);
    int32_t rc = ensure_records(ctx, src, scale, src.frames_pad, -1);
    if (rc != SSYM_OK)
        return rc;
    rc = ensure_records(ctx, tgt, scale, tgt.frames_pad, 0);
    if (rc != SSYM_OK)
        return rc;
    // persistent grid: 2 workgroups of 4 waves per CU (2 waves per SIMD), a multiple of 8 so that
    // task & 7 is the XCD group; one hand-off row of [target frames][64] floats per wave
    const int gridBlocks = std::max(8, ctx->num_cus * 2 / 8 * 8);
    // + 4 x 8 task counters behind the hand-off rows (one set per launch below)
    // (sized for the three-workgroups-per-CU launch of the one-tile kernel too)
    const size_t handBytes = (size_t)(gridBlocks / 2 * 3) * kFilterWavesPerBlock * ((tgt.frames_pad + 3) / 4) * 256 * sizeof(float);
    const size_t ctrBytes = 8 * kTaskCtrStride * sizeof(unsigned);
    constexpr int kMaxClasses = 10;                 // counter sets: three single-pass classes + up to seven multi-pass ones
    rc = ensure(ctx, ctx->handoff, handBytes + kMaxClasses * ctrBytes);
    if (rc != SSYM_OK)
        return rc;
    unsigned *taskCtr = (unsigned *)((char *)ctx->handoff.ptr + handBytes);
    rc = zero_words(ctx, taskCtr, kMaxClasses * ctrBytes);
    if (rc != SSYM_OK)
        return rc;
    hipStream_t st = ctx->stream;
    const bool sq = ctx->squared != 0;
    const float outScale = (float)(sq ? 1.0 / (scale * scale) : 1.0 / scale);
    float *hand = (float *)ctx->handoff.ptr;
    ctx->launched_cells = 0;
    ctx->filter_launches = 0;

    // Record slots are ordered by segment length, so source pairs fall into contiguous CLASSES by the
    // 16-row tiles their longer member needs.  Pairs that fit one, two or three tiles run the single-pass
    // variant with exactly that many (on the last rows of their end-aligned slots) -- a dictionary of
    // 8...40-frame segments otherwise paid 48 rows for every pair --, the rest the set's own shape,
    // whose leading all-padding passes are skipped per task.  One launch per non-empty class.
    const int nPairs = (int)src.n_pad / 2;
    auto pairLen = [&](int sp) -> uint32_t {      // longer member of source pair sp; slots [0, n) are real, ascending
        auto len = [&](uint32_t p) -> uint32_t {
            if (p >= src.n)
                return 0u;
            const uint32_t s = src.h_perm[p];
            return (uint32_t)(src.h_off[s + 1] - src.h_off[s]);
        };
        return std::max(len(2u * sp), len(2u * sp + 1u));
    };
    const int nRealPairs = (int)((src.n + 1) / 2);  // pairs behind them hold padding only: they ride with the last class
    auto firstAbove = [&](uint32_t frames) {        // first real pair whose longer member exceeds `frames`
        int lo = 0, hi = nRealPairs;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (pairLen(mid) > frames)
                hi = mid;
            else
                lo = mid + 1;
        }
        return lo;
    };
    const int topTiles = setShape.nt;               // tiles of the set's own shape (4 = multi-pass)
    static const bool oneLaunch = ssym_knob("SSYM_FILTER_ONE_LAUNCH") != nullptr;     // measurements: the set's shape for every pair
    int bound[4] = {0, 0, 0, 0};                    // bound[c]: first pair that needs more than c tiles
    for (int c = 1; c < topTiles; ++c)
        bound[c] = oneLaunch ? 0 : firstAbove(16u * c);
    // operand planes the kernel multiplies: record layout 3 leaves the third one zero
    const bool two = filter_mfmas(filter_pieces(filter_dim_used((int)src.dim)), filter_dim_used((int)src.dim)) == 2;
    bool skipTile = false;                          // set for a multi-pass class below
#define SSYM_LAUNCH1(NT_, SQ_, KU_, PASSES_, ORIGIN_, LO_, HI_, K_)                                               \
    launch_one<NT_, SQ_, KU_>(st, src, tgt, PASSES_, gridBlocks, outScale, hand, taskCtr + (K_) * 8 * kTaskCtrStride, \
                              cmat, abandon, colCtr, candSlot, LO_, (HI_) - (LO_), ORIGIN_, &ctx->launched_cells, skipTile)
#define SSYM_LAUNCH(NT_, PASSES_, ORIGIN_, LO_, HI_, K_)                                                          \
    if ((HI_) > (LO_)) {                                                                                          \
        ++ctx->filter_launches;                                                                                   \
        if (sq && two) SSYM_LAUNCH1(NT_, true, 2, PASSES_, ORIGIN_, LO_, HI_, K_);                                \
        else if (sq) SSYM_LAUNCH1(NT_, true, 3, PASSES_, ORIGIN_, LO_, HI_, K_);                                  \
        else if (two) SSYM_LAUNCH1(NT_, false, 2, PASSES_, ORIGIN_, LO_, HI_, K_);                                \
        else SSYM_LAUNCH1(NT_, false, 3, PASSES_, ORIGIN_, LO_, HI_, K_);                                         \
    }
    const int rowsPad = (int)src.frames_pad;
    if (topTiles == 4) {
        SSYM_LAUNCH(1, 1, rowsPad - 16, bound[0], bound[1], 0)
        SSYM_LAUNCH(2, 1, rowsPad - 32, bound[1], bound[2], 1)
        SSYM_LAUNCH(3, 1, rowsPad - 48, bound[2], bound[3], 2)
        static const bool nt8 = ssym_knob("SSYM_FILTER_NT8") != nullptr;
        static const bool oneLong = ssym_knob("SSYM_FILTER_LONG_CLASSES") && atoi(ssym_knob("SSYM_FILTER_LONG_CLASSES")) == 0;
        if (shape.nt == 2) {
            SSYM_LAUNCH(2, shape.rb, 0, bound[3], nPairs, 3)
        } else if (nt8 && shape.rb == 2 && !abandon) {
            SSYM_LAUNCH(8, 1, 0, bound[3], nPairs, 3)
        } else if (abandon || oneLong || oneLaunch) {
            SSYM_LAUNCH(4, shape.rb, 0, bound[3], nPairs, 3)
        } else {
            // Sources beyond 48 frames, ragged: a pair pays whole row passes, so a 70-frame source swept in passes of 64
            // rows pays 128 -- but 96 in passes of 48.  The pairs (ascending by length) are cut into classes by the shape
            // that pads them least, passes of 64 rows (four tiles) or of 48 (three), the last rows of their end-aligned
            // slots as for the short classes; a class of fewer than 128 pairs rides with the next one (any larger shape
            // holds it), and so does everything beyond the counter sets there are.  Equal lengths: one class, the set's
            // own shape, as before.
            struct LongClass { int nt, passes, lo, hi; };
            LongClass cls[kMaxClasses];
            int nCls = 0;
            auto best = [&](uint32_t len, int &nt, int &passes) {
                const int p4 = (int)((std::max(len, 1u) + 63) / 64), p3 = (int)((std::max(len, 1u) + 47) / 48);
                if (48 * p3 < 64 * p4) { nt = 3; passes = p3; } else { nt = 4; passes = p4; }
            };
            for (int sp = bound[3]; sp < nPairs;) {
                int nt, passes;
                best(sp < nRealPairs ? pairLen(sp) : 0u, nt, passes);
                // the run of pairs this shape is the best for: up to the first pair longer than its rows
                const int end = std::max(sp + 1, std::min(nPairs, sp < nRealPairs ? firstAbove((uint32_t)(16 * nt * passes)) : nPairs));
                cls[nCls++] = LongClass{nt, passes, sp, end == nRealPairs ? nPairs : end};     // (padding pairs ride with the last real one)
                sp = cls[nCls - 1].hi;
                if (nCls == kMaxClasses - 3 && sp < nPairs) {                                   // no counter set left: the set's own shape takes the rest
                    cls[nCls - 1] = LongClass{4, shape.rb, cls[nCls - 1].lo, nPairs};
                    break;
                }
            }
            // small classes join their successor (whose rows are at least theirs); the last one keeps its own
            for (int c = 0; c + 1 < nCls;) {
                if (cls[c].hi - cls[c].lo < 128) {
                    cls[c + 1].lo = cls[c].lo;
                    for (int k = c; k + 1 < nCls; ++k)
                        cls[k] = cls[k + 1];
                    --nCls;
                } else {
                    ++c;
                }
            }
            const char *skipKnob = ssym_knob("SSYM_FILTER_SKIP0");        // (read per launch: the tests compare both settings)
            const bool skipOn = !(skipKnob && atoi(skipKnob) == 0);
            for (int c = 0; c < nCls; ++c) {
                const int origin = rowsPad - 16 * cls[c].nt * cls[c].passes;
                // at least an eighth of the class's pairs leave the first tile of their first pass empty: the variant that
                // skips such tiles' cells (equal lengths and classes that fill their rows keep the plain kernel)
                const int rowsCls = 16 * cls[c].nt * cls[c].passes;
                int skipping = 0;
                for (int sp = cls[c].lo; sp < std::min(cls[c].hi, nRealPairs); ++sp)
                    skipping += (rowsCls - (int)pairLen(sp)) % (16 * cls[c].nt) >= 17;
                skipTile = skipOn && 8 * skipping >= cls[c].hi - cls[c].lo && skipping > 0;
                if (cls[c].nt == 3) {
                    SSYM_LAUNCH(3, cls[c].passes, origin, cls[c].lo, cls[c].hi, 3 + c)
                } else {
                    SSYM_LAUNCH(4, cls[c].passes, origin, cls[c].lo, cls[c].hi, 3 + c)
                }
            }
            skipTile = false;
        }
    } else if (topTiles == 3) {
        SSYM_LAUNCH(1, 1, rowsPad - 16, bound[0], bound[1], 0)
        SSYM_LAUNCH(2, 1, rowsPad - 32, bound[1], bound[2], 1)
        SSYM_LAUNCH(3, 1, 0, bound[2], nPairs, 2)
    } else if (topTiles == 2) {
        SSYM_LAUNCH(1, 1, rowsPad - 16, bound[0], bound[1], 0)
        SSYM_LAUNCH(2, 1, 0, bound[1], nPairs, 1)
    } else {
        SSYM_LAUNCH(1, 1, 0, 0, nPairs, 0)
    }
#undef SSYM_LAUNCH
#undef SSYM_LAUNCH1
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym

#ifdef SSYM_SP_PROF
// tools only: read and clear the single-pass kernel's tick sums (dtw_filter_sp_kernel.hpp)
extern "C" __attribute__((visibility("default"))) int ssym_debug_sp_prof(unsigned long long *out)
{
    unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ssym::ssym_sp_prof), sizeof(zero)) != hipSuccess)
        return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ssym::ssym_sp_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
