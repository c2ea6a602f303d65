// dtw_filter_kernel.hpp -- device code of the dtw MFMA filter (included by dtw_filter.hip and by
// tools/filter_bench.hip, which times ablated variants of the same source).
//
// One (source, target) pair per lane; the DP column of the pair lives in the lane's registers.
//
//   * cost block on the MATRIX pipe: v_mfma_f32_32x32x16_f16.  (The f32-input MFMA was measured
//     first: on gfx950 it does not co-execute with VALU work -- kernel time was MFMA time PLUS DP
//     time -- whereas the f16 matrix pipe runs beside the VALU.)  f32 accuracy is recovered by
//     splitting every operand into two f16 pieces v = H1 + H2 (22 bits) and feeding the three
//     significant cross products through the K dimension; both squared norms ride along in three
//     f16 pieces each, so the accumulator IS |a - b|^2 (scaled), no VALU add:
//         K slots 3e+0..3e+2 : (-2a_e)1 * (b_e)1,  (-2a_e)1 * (b_e)2,  (-2a_e)2 * (b_e)1     e < 13
//         K slots 39..41     : |a|^2 pieces * 1          K slots 42..44 : 1 * |b|^2 pieces
//     = 45 of the 48 slots of three chained 32x32x16 MFMAs per 32x32 tile.  That was round 1 (layout 2, still
//     behind SSYM_FILTER_K48=1); the default for up to 13 values is now layout 3 in K = 32, TWO MFMAs per tile
//     (template parameter KU = 2), the target rounded to one piece and priced for it by select.hip:
//         K slots 2e, 2e+1   : (-2a_e)1 * (b_e)1,  (-2a_e)2 * (b_e)1                          e < 13
//         K slots 26, 27     : (-2a_e)1 * (b_e)2  for e = 0, 1 (the target's first two values keep both pieces)
//         K slots 28, 29     : |a~|^2 pieces * 1         K slots 30, 31 : 1 * |b~|^2 pieces
//     Frames wider than 13
//     values (up to 42) use ONE f16 piece per value (slot e: (-2a_e)1 * (b_e)1, norms after them):
//     the filter then sees data rounded to 11 bits, which select.hip prices per cell;
//   * tile = 16 frames of source 0 interleaved (groups of four) with 16 frames of source 1 as the
//     32 A-rows, frame j of 32 DIFFERENT targets as the 32 B-columns: accumulator register r of
//     lane (col = lane&31, half = lane>>5) is cell (row r, column j) of the pair
//     (source 2*sp+half, target 32*tg+col);
//   * DP: min-of-three recurrence, lane-local, 3 VALU instructions per cell (v_sqrt_f32 |x|,
//     v_min3_f32, v_add_f32); the previous column is read from one register array and the new
//     column written to the other (ping-pong over a 2-column unroll, no register moves).  The
//     per-pair error certificate (smallest cell) is NOT tracked here: certify.hip computes it
//     afterwards for the few pairs that survive a first, worst-case-margin selection, which is
//     8 % cheaper than half a v_min3 per cell on all N x M pairs;
//   * a wave keeps 16*NT rows of the column in registers; longer sources are processed in row-block
//     PASSES by the same wave: pass p sweeps all columns for rows [p*16*NT, (p+1)*16*NT) and leaves
//     the bottom row of its block, D(last row, j), in a per-wave hand-off row in global memory
//     (1 KB per four columns, L2 / Infinity-Cache resident because the grid is
//     persistent); pass p+1 reads it back as its top boundary.  Waves never synchronise with each
//     other -- a first version that pipelined row blocks across waves with one workgroup barrier
//     per column spent half of its wave-cycles waiting (profiles/);
//   * target records (group-major, 1 KB per MFMA operand plane per column) reach the wave through a
//     per-wave LDS ring filled by global_load_lds DMA three columns ahead of use, the hand-off row
//     four columns per 16-byte access (one store and one DMA per FOUR columns): loads in flight hold no registers (the kernel sits at the 256-VGPR limit of two
//     waves per SIMD) and s_waitcnt vmcnt(6) at the top of a column never waits for a young load;
//   * measured on MI355X (rocprofv3 PMC): plain VALU instructions occupy the SIMD for 4 cycles,
//     v_sqrt_f32 for 8, whatever the occupancy -- 16 cycles per cell is the floor of this
//     recurrence, and the kernel is VALU-bound, not MFMA- or HBM-bound.  Ablations (tools/
//     filter_bench.hip, -DSSYM_ABL_NOSTAGE / -DSSYM_ABL_NOHAND): the kernel runs 21.1 cycles per
//     cell at 2.07 GHz; without any per-column memory traffic 19.3 cycles at 2.22 GHz -- the chip
//     is power-limited, so moving data costs clock as well as cycles.  The same goes for the matrix pipe: with the
//     K = 32 records (two MFMAs per tile instead of three) the kernel runs 11.7 % faster (36.8 -> 32.5 ms on
//     configs[2]) although the pipe was 39 % busy before; WHERE the chain is issued does not matter -- one MFMA every
//     six cells instead of the compiler's back-to-back placement measured the same to 0.1 ms.
#pragma once
#include <hip/hip_runtime.h>

#ifndef SSYM_FILTER_MODE
#define SSYM_FILTER_MODE 0   // 0 = product; tools only: 1 = MFMA without DP, 2 = DP without MFMA
#endif

namespace ssym {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFilterRowsPerTile = 16;   // frames of one source per 32x32 tile
constexpr int kFilterKM = 3;             // chained K=16 MFMAs per tile
constexpr int kFilterRecHalfs = 48;      // f16 values per frame record (96 bytes)
constexpr int kFilterMaxDim2 = 13;       // two f16 pieces per value: 3*13 product slots + 6 norm slots <= 48
constexpr int kFilterMaxDim1 = 42;       // one f16 piece per value:  42 product slots + 6 norm slots <= 48

// Record layout (both sides): [khalf 0: 3 x 8 f16][khalf 1: 3 x 8 f16]; MFMA m of lane half h
// reads its 8 K-values at f16 offset h*24 + m*8, i.e. logical K slot 16*m + 8*h + j.
__host__ __device__ constexpr int filter_slot_offset(int k) { return ((k >> 3) & 1) * 24 + (k >> 4) * 8 + (k & 7); }

template <int KM>
__device__ __forceinline__ f32x16 mfma_tile(const half8 (&a)[KM], const half8 (&b)[KM])
{
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if SSYM_FILTER_MODE == 2
    float z = (float)a[0][0] + (float)b[0][0];
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = z + (float)r;
    return acc;
#else
#ifndef SSYM_ABL_DROP_MFMA
#define SSYM_ABL_DROP_MFMA 0      // tools only: issue this many MFMAs fewer per tile (wrong results; what a K = 32 record would save)
#endif
#pragma unroll
    for (int m = 0; m < KM - SSYM_ABL_DROP_MFMA; ++m)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[m], acc, 0, 0, 0);
    return acc;
#endif
}

template <int KU>
__device__ __forceinline__ void load_rec(const _Float16 *__restrict__ p, half8 (&dst)[KU])
{
#pragma unroll
    for (int m = 0; m < KU; ++m)
        dst[m] = *reinterpret_cast<const half8 *>(p + 8 * m);
}

// TARGET records are stored group-major so that a wave's column load is contiguous: for target
// group g = t / 32 and frame slot j, the 16 bytes that lane (h, c = t & 31) feeds to MFMA m sit at
//     f16 offset ((g * slots + j) * 3 + m) * 512 + (h * 32 + c) * 8,
// i.e. one global_load_dwordx4 of the wave covers 1 KB = 8 whole cache lines (with per-target
// records every lane touched a line of its own and the texture path, not the VALU, set the pace).
constexpr int kTgtFrameHalfs = kFilterKM * 512;      // f16 values per (target group, frame slot)
__host__ __device__ constexpr size_t tgt_rec_offset(uint32_t t, uint32_t slots, uint32_t j, int m, int h)
{
    return (((size_t)(t >> 5) * slots + j) * kFilterKM + m) * 512 + (size_t)(h * 32 + (t & 31u)) * 8;
}
// p = the lane's base for its target and K half (tgt_rec_offset(t, slots, 0, 0, h)); j = frame slot
template <int KU>
__device__ __forceinline__ void load_tgt_rec(const _Float16 *__restrict__ p, int j, half8 (&dst)[KU])
{
#pragma unroll
    for (int m = 0; m < KU; ++m)
        dst[m] = *reinterpret_cast<const half8 *>(p + (size_t)j * kTgtFrameHalfs + m * 512);
}

// One column of one row block: NT tiles, software-pipelined (the next tile's MFMA chain is in
// flight while this tile's 16 cells run on the VALU).  Lr = D(., j-1), Lw = D(., j).
// SKIP0 (skip0, wave-uniform): the block's first tile holds nothing but padding rows of both sources and not the row above
// a source's first either -- its 16 cells are +inf in every column (they are never written: both column arrays start at
// +inf) and so are the `up` and `diag` values it hands to the second tile.  The tile's MFMAs are still issued.
template <int NT, bool SQ, int KU, bool SKIP0 = false>
__device__ __forceinline__ float dp_column(const half8 (&A)[NT][KU], const half8 (&Bc)[KU],
                                           const half8 (&Bn)[KU], f32x16 &acc, float up, float diag,
                                           const float (&Lr)[NT * 16], float (&Lw)[NT * 16], const bool skip0 = false)
{
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x16 accn;
        if (T + 1 < NT)
            accn = mfma_tile<KU>(A[T + 1], Bc);
        else
            accn = mfma_tile<KU>(A[0], Bn);   // first tile of the next column
#if SSYM_FILTER_MODE == 1
        asm volatile("" ::"v"(acc[0]), "v"(acc[15]));
        up = acc[3];
        Lw[T * 16] = Lr[T * 16];
#else
        if (SKIP0 && T == 0 && skip0) {
            up = __builtin_inff();
            diag = __builtin_inff();
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = T * 16 + r;
                const float x = acc[r];
                const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
                const float m = __builtin_fminf(__builtin_fminf(up, diag), Lr[idx]);
                diag = Lr[idx];
                const float cur = c + m;
                Lw[idx] = cur;
                up = cur;
            }
        }
#endif
        acc = accn;
    }
    return up;   // D(last row of the block, j)
}

constexpr int kFilterWavesPerBlock = 4;
#ifndef SSYM_PRUNE_EVERY
#define SSYM_PRUNE_EVERY 8
#endif
#ifndef SSYM_PRUNE_FINE
#define SSYM_PRUNE_FINE 48
#endif
// pairs far above their threshold are dropped within the first few dozen columns: there the test runs every
// 4 columns (a later test costs a sixth of the columns such a task sweeps at all), afterwards every kPruneEvery
constexpr int kPruneFine = SSYM_PRUNE_FINE;
constexpr int kPruneEvery = SSYM_PRUNE_EVERY;   // PRUNE: columns between two abandon tests (a test is BR/2 v_min3 + a vote)
constexpr int kTaskCtrStride = 64;      // the 8 task counters sit in separate 256-byte lines (separate L2 channels)
// Target columns staged in LDS per wave: 4 at two waves per SIMD (three columns of lead); the single-pass
// kernels for short sources (NT <= 2) can run three waves per SIMD with a ring of 2 (one column of
// lead, latency covered by occupancy) so that three workgroups fit the LDS.
constexpr int filter_ring(int occ) { return occ >= 3 ? 2 : 4; }
constexpr int kFilterSlotBytes = kFilterKM * 1024;              // 64 lanes x 3 x 16 B operands of one column
constexpr int kFilterTopBytes = 2 * 1024;                       // hand-off values of 2 groups of 4 columns
constexpr int filter_wave_lds(int occ) { return filter_ring(occ) * kFilterSlotBytes + kFilterTopBytes; }
constexpr int wait_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n), nothing else
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])

// Persistent kernel: every wave keeps taking (source pair, target group) tasks of 64 pairs until
// none is left (taskCtr: 8 counters, zeroed by the host before the launch).  A launch covers the
// source pairs [spBase, spBase + nSrcPairs) and the rows [rowOrigin, rowOrigin + nPasses * 16 * NT) of
// their slots: record slots are ordered by segment length, so the host gives every class of source
// lengths the variant with just enough 16-row tiles (dtw_filter.hip).
//
// PRUNE (early abandoning, SSYM_DTW_PRUNE): abandon[t] is a per-target value, in the accumulator's
// units, that no pair of interest can exceed (prune.hip derives it from the exact cost of one
// candidate pair per target plus the filter's worst-case error).  Costs are non-negative, so the
// cost of a pair is at least the D value of any cell its optimal path visits.  A row pass STOPS at
// column j when, on every lane, the minimum over the pass's rows in column j exceeds abandon[t] (or the
// lane's target has ended, or the lane is dead: an empty side, or the target's candidate pair, candSlot,
// which needs no filter value at all) and the previous pass delivered nothing beyond column j: a path
// that is inside this pass's rows at column j, or enters them later, is above its threshold, so only
// paths that have left through the bottom row at a column <= j can still matter -- the next pass reads
// the bottoms up to j and +inf behind them.  The TASK is dropped after a pass whose bottoms are all
// above the thresholds (every path crosses that row; a lane whose source only begins in a later pass
// keeps the task alive): the remaining passes are skipped and the unfinished lanes report +inf, which
// selection treats as "never a candidate".  The work per task is then about the area where D <= threshold.
//
// KU = the operand planes (K = 16 each) a tile multiplies: 3, or 2 for record layout 3 of ssym_internal.hpp, whose
// third plane is zero -- neither loaded nor staged nor issued (the record strides stay those of three planes).
// SKIP0 (launches of ragged multi-pass classes whose shorter sources leave the first tile of their first pass empty, no
// PRUNE): that tile's cells are skipped, see dp_column -- a 70-frame source in two passes of 48 rows pays 80 rows of cells.
template <int NT, bool SQ, int OCC = 2, bool PRUNE = false, int KU = kFilterKM, bool SKIP0 = false>
__global__ __launch_bounds__(64 * kFilterWavesPerBlock, OCC) void dtw_filter_kernel(
    const _Float16 *__restrict__ srcRec, const _Float16 *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int srcRows, int nPasses,
    int tgtFramesPad, int mPad, int nSrcPairs, int nTasks, int taskChunk, float outScale,
    float *__restrict__ handoff, unsigned *__restrict__ taskCtr, float *__restrict__ cmat,
    const float *__restrict__ abandon = nullptr, unsigned long long *__restrict__ colCtr = nullptr,
    const uint32_t *__restrict__ candSlot = nullptr, int rowOrigin = 0, int spBase = 0)
{
    static_assert(!SKIP0 || (!PRUNE && NT >= 2), "SKIP0: unpruned launches of at least two tiles");
    constexpr int REC = kFilterRecHalfs;
    constexpr int BR = NT * 16;            // rows per pass
    const float INF = __builtin_inff();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;         // output column: target 32*tg + col
    const int half = lane >> 5;        // operand role: K half; output role: source 2*sp + half
    // this wave's hand-off row, [ceil(tgtFramesPad / 4)][64 lanes][4 columns] floats: one 16-byte
    // access per lane moves FOUR columns (a store and a DMA per column cost ~6 % of the kernel)
    const size_t handGroups = ((size_t)tgtFramesPad + 3) / 4;
    char *const handRow = reinterpret_cast<char *>(handoff + ((size_t)blockIdx.x * kFilterWavesPerBlock + wave) * handGroups * 256);
    const uint32_t laneOff16 = lane * 16;
    // OCC >= 3 is only instantiated for single-pass shapes: no hand-off traffic is in flight, so the waits
    // below count column groups of exactly KU DMAs
    constexpr int kFilterRing = filter_ring(OCC);
    constexpr int kWaitLead = wait_vmcnt(KU * (kFilterRing - 2)), kWaitFirst = wait_vmcnt(KU * (kFilterRing - 1));
    __shared__ __attribute__((aligned(16))) char ring[kFilterWavesPerBlock][filter_wave_lds(OCC)];
    char *const myRing = ring[wave];
    char *const myTop = myRing + kFilterRing * kFilterSlotBytes;

    // Work distribution: a task is one wave's (source pair, target group) = 64 pairs; waves take
    // tasks from 8 counters, one per XCD group (workgroups b, b+8, ... share an XCD and gridDim.x is a
    // multiple of 8), each counter covering the (target group, source pair) space of every eighth
    // target group, so that a group's targets stay in that XCD's L2.  Ranges are walked from their END:
    // record slots are ordered by segment length, so the longest tasks start first and the short
    // ones fill the tail (with ragged segment lengths a static assignment left most of the chip
    // waiting for the workgroup that held the longest sources).  A wave whose range is exhausted
    // helps the next XCD's range, so every wave leaves only when all counters are spent.
    unsigned colSteps = 0;      // PRUNE: columns this wave swept (wave-uniform), reported once at the end
    // (round 4: XCD x owns the target groups x, x + 8, ... instead of a contiguous eighth -- groups are ordered by length,
    //  and on ragged targets the last eighth held several times the first one's work; nothing changes on equal lengths)
    const unsigned nGroups = (unsigned)mPad >> 5;
    for (unsigned hop = 0; hop < 8; ++hop) {
      const unsigned xcd = (blockIdx.x + hop) & 7u;
      const unsigned rangeLen = xcd < nGroups ? ((nGroups - xcd + 7u) >> 3) * (unsigned)nSrcPairs : 0u;
      for (;;) {
        // tasks are taken kTaskChunk at a time: the counters are atomics in L2, and 2.6e5 single-task
        // grabs on one cache line were a 3 ms floor under every launch (visible for short segments)
        unsigned got = 0;
        if (lane == 0)
            got = atomicAdd(&taskCtr[xcd * kTaskCtrStride], (unsigned)taskChunk);
        got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
        if (got >= rangeLen)
            break;
        const unsigned gotEnd = min(got + (unsigned)taskChunk, rangeLen);
        // the lengths of the chunk's next task are fetched while the current one runs: for short tasks the
        // two dependent round trips of a task's start (lengths, then operands) were as long as its columns
        int preFa = 0, preFb = 0;
        bool havePre = false;                   // wave-uniform
       for (unsigned gi = got; gi < gotEnd; ++gi) {
        const unsigned lin = rangeLen - 1u - gi;                      // position in the XCD's range, walked from its end
        const int tg = (int)(xcd + 8u * (lin / (unsigned)nSrcPairs));
        const int sp = spBase + (int)(lin % (unsigned)nSrcPairs);    // source pair of this wave

        const int fa = havePre ? preFa : srcLen[2 * sp + half];
        const int fb_m1 = (havePre ? preFb : tgtLen[32 * tg + col]) - 1;
        havePre = gi + 1 < gotEnd;
        if (havePre) {
            const unsigned linN = lin - 1u;
            preFa = srcLen[2 * (spBase + (int)(linN % (unsigned)nSrcPairs)) + half];
            preFb = tgtLen[32 * (int)(xcd + 8u * (linN / (unsigned)nSrcPairs)) + col];
        }
        const int r0 = srcRows - fa;   // first real row: sources are END-ALIGNED in their row slots

        // wave-uniform bounds: columns up to the longest target of the group; passes that hold
        // nothing but padding rows of BOTH sources are skipped
        int nCols = fb_m1 + 1, r0min = r0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            nCols = max(nCols, __shfl_xor(nCols, o));
            r0min = min(r0min, __shfl_xor(r0min, o));
        }
        nCols = __builtin_amdgcn_readfirstlane(nCols);
        r0min = __builtin_amdgcn_readfirstlane(r0min);
        // (a source that begins exactly on a pass boundary needs nothing of the pass above: diagCol0 starts it)
        const int firstPass = min(max(r0min - rowOrigin, 0) / BR, nPasses - 1);

        float res = INF;
        float thr = INF;                       // PRUNE: threshold of the lane's target
        bool dropped = false;                  // wave-uniform
        int lastTop = -1;                      // PRUNE: last column the previous pass computed (its bottoms end there)
        if (PRUNE)
            thr = abandon[32 * tg + col];
        // PRUNE: an empty side (the result is +inf whatever happens), or the target's candidate pair, whose
        // exact cost is known and stands in for its filter value everywhere (prune.hip)
        bool dead = fa == 0 || fb_m1 < 0;
        if (PRUNE)
            dead = dead || candSlot[32 * tg + col] == (uint32_t)(2 * sp + half);
        const char *const tgtGroup = reinterpret_cast<const char *>(tgtRec) + (size_t)tg * tgtFramesPad * (kTgtFrameHalfs * 2);

        for (int pass = nCols > 0 ? firstPass : nPasses; pass < nPasses && !dropped; ++pass) {
            const int rowBase = rowOrigin + pass * BR;
#ifdef SSYM_ABL_NOHAND
            const bool haveTop = false;
#else
            const bool haveTop = pass > firstPass;      // wave-uniform
#endif
            const bool lastPass = pass == nPasses - 1;
            // SKIP0: the pass's first tile lies above both sources and above the row that starts them (wave-uniform; true in
            // a task's first pass at most)
            const bool skip0 = SKIP0 && r0min - rowBase >= 17;
            const bool started = r0 < rowBase + BR;     // PRUNE: the lane's source has rows in this pass or above

            // Target records (and the hand-off row above this row block) travel global -> LDS by
            // DMA, kFilterRing - 1 columns ahead of their use, and LDS -> registers one column
            // ahead: no registers are held by loads in flight.  Virtual column c (clamped to the
            // last real column) lives in ring slot c % kFilterRing.
            auto stage = [&](int c) {
                const int cc = min(c, nCols - 1);
                char *slot = myRing + (c & (kFilterRing - 1)) * kFilterSlotBytes;
                // one LDS base (M0) per column; the instruction offset moves the global and the LDS
                // address together, which the group-major record layout is made for
                const char *gb = tgtGroup + (size_t)cc * (kTgtFrameHalfs * 2);       // wave-uniform
                static_assert(KU == 2 || KU == 3, "two or three operand planes per column");
                const __attribute__((address_space(1))) void *gp =
                    (const __attribute__((address_space(1))) void *)(gb + laneOff16);
                __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)slot;
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
                if (KU == 3)
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
            };
            // hand-off values of column group g (4 columns) -> top buffer g & 1
            auto stageTop = [&](int g) {
                const int gg = min(g, (nCols - 1) >> 2);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(handRow + (size_t)gg * 1024 + laneOff16),
                    (__attribute__((address_space(3))) void *)(myTop + (g & 1) * 1024), 16, 0, 0);
            };
            auto fetch = [&](int c, half8 (&B)[KU], float &top) {
                const char *slot = myRing + (c & (kFilterRing - 1)) * kFilterSlotBytes;
#pragma unroll
                for (int m = 0; m < KU; ++m)
                    B[m] = *reinterpret_cast<const half8 *>(slot + m * 1024 + lane * 16);
                // read unconditionally (no branch, no wait at a block end); unused when !haveTop
                top = *reinterpret_cast<const float *>(myTop + ((c >> 2) & 1) * 1024 + lane * 16 + (c & 3) * 4);
            };
            // A operands of this pass: the pad rows above a source carry |a|^2 = +inf
            half8 A[NT][KU];
            {
                const int arow = lane & 31;
                const int a_src = 2 * sp + ((arow >> 2) & 1);
                const int a_frm = rowBase + (arow & 3) + 4 * (arow >> 3);
                const _Float16 *abase = srcRec + ((size_t)a_src * srcRows + a_frm) * REC + half * 24;
#pragma unroll
                for (int T = 0; T < NT; ++T)
                    load_rec<KU>(abase + (size_t)T * kFilterRowsPerTile * REC, A[T]);
            }

            asm volatile("" ::: "memory");      // A loads are issued (program order) before the staging DMAs
            if (haveTop)
                stageTop(0);
#pragma unroll
            for (int c = 0; c < kFilterRing; ++c)
                stage(c);

            // D(., -1): +inf, except the virtual D(r0-1, -1) = 0 that starts the recurrence
            float L0[BR], L1[BR];
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                L0[i] = (rowBase + i == r0 - 1) ? 0.0f : INF;
                L1[i] = L0[i];
            }
            const float diagCol0 = (rowBase == r0) ? 0.0f : INF;   // D(rowBase-1, -1)
            float prevTop = INF;                                    // D(rowBase-1, j-1)

            // B operands of the current and the next column swap roles every column (no copies)
            half8 B0[KU], B1[KU];
            float topN = INF;                                       // D(rowBase-1, j) for the coming column
            // everything issued so far except the last kFilterRing - 1 column groups has landed
            // (a group is KU or KU + 1 DMAs; the A loads are older): column 0 is in its slot
            __builtin_amdgcn_s_waitcnt(kWaitFirst);
            asm volatile("" ::: "memory");
            fetch(0, B0, topN);
            f32x16 acc = mfma_tile<KU>(A[0], B0);

            float bq[4] = {INF, INF, INF, INF};                     // bottoms of the current group of 4 columns
            float runBot = INF;                                     // PRUNE: min of this pass's bottoms so far
            bool passOver = false;                                  // PRUNE: nothing at or below a threshold is left in this pass
            int lastCol = nCols - 1;
            for (int j0 = 0; j0 < nCols && !passOver; j0 += 4) {
                if (PRUNE)
                    colSteps += (unsigned)min(4, nCols - j0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j0 + q;
                    if (j < nCols) {                                // wave-uniform
                        // (PRUNE: the previous pass may have stopped early; beyond its last column nothing enters from above)
                        const float up = (haveTop && (!PRUNE || j <= lastTop)) ? topN : INF;
                        const float diag = (j == 0) ? diagCol0 : prevTop;
                        prevTop = up;
                        // column j+1 was staged kFilterRing - 1 columns ago; at least the two
                        // groups after it (>= 2 KU DMAs) are younger, so vmcnt(2 KU) covers it -- and the
                        // hand-off group it may open, which was requested four columns ago
                        __builtin_amdgcn_s_waitcnt(kWaitLead);
                        asm volatile("" ::: "memory");
                        if ((q & 1) == 0)
                            fetch(j + 1, B1, topN);
                        else
                            fetch(j + 1, B0, topN);
#ifndef SSYM_ABL_NOSTAGE
                        stage(j + kFilterRing);                     // into the slot column j just left
#endif
                        if (q == 0 && haveTop)
                            stageTop((j >> 2) + 1);                 // the next group's top values
                        float bottom;
                        if ((q & 1) == 0)
                            bottom = dp_column<NT, SQ, KU, SKIP0>(A, B0, B1, acc, up, diag, L0, L1, skip0);
                        else
                            bottom = dp_column<NT, SQ, KU, SKIP0>(A, B1, B0, acc, up, diag, L1, L0, skip0);
                        bq[q] = bottom;
                        if (!lastPass) {
#ifndef SSYM_ABL_NOHAND
                            if (q == 3 || j == nCols - 1) {         // top boundary of the next pass, 4 columns at a time
                                typedef float f32x4 __attribute__((ext_vector_type(4)));
                                *reinterpret_cast<f32x4 *>(handRow + (size_t)(j >> 2) * 1024 + laneOff16) =
                                    f32x4{bq[0], bq[1], bq[2], bq[3]};
                            }
#else
                            res = (j == fb_m1 - 1) ? bottom : res;
#endif
                        } else {
                            res = (j == fb_m1) ? bottom : res;      // D(fa-1, fb-1)
                        }
                        if (PRUNE) {
                            runBot = __builtin_fminf(runBot, bottom);
                            if (q == 3 && (j < kPruneFine || (j & (kPruneEvery - 1)) == kPruneEvery - 1)) {
                                // q == 3: the column just written is L0
                                float cm = L0[0];
#pragma unroll
                                for (int i = 1; i < BR; i += 2)
                                    cm = __builtin_fminf(__builtin_fminf(cm, L0[i]), i + 1 < BR ? L0[i + 1] : L0[i]);
                                // every path still inside this pass's rows at column j is above the threshold (or the
                                // lane's target has ended, or the lane is dead), and nothing enters from above any
                                // more: whatever can still win has left through the bottom row already
                                const bool quiet = !(cm <= thr) || j >= fb_m1 || dead;
                                if (__all(quiet) && j >= lastTop) {
                                    passOver = true;
                                    lastCol = j;
                                }
                            }
                        }
                    }
                }
            }
            if (PRUNE && !lastPass) {
                // the paths that can still win crossed this pass's bottom row at a column <= lastCol (a source
                // that begins below this pass has no cell here yet: its lane keeps the task alive)
                lastTop = lastCol;
                dropped = __all((started && !(runBot <= thr)) || dead);
            }
        }
        cmat[(size_t)(2 * sp + half) * mPad + 32 * tg + col] = res * outScale;
       }
      }
    }
    if (PRUNE && colCtr && lane == 0)
        atomicAdd(colCtr, (unsigned long long)colSteps * BR);
}

}  // namespace ssym
