// dtw_filter_pk_kernel.hpp -- the unbanded dtw filter with TWO row blocks of a pass skewed by one column, so that
// the two `c + min3` additions of a step are ONE v_pk_add_f32 (device code; dispatched by dtw_filter.hip for sources
// of more than 48 frames when nothing is abandoned early).
//
// Same mapping as dtw_filter_kernel.hpp -- one (source, target) pair per lane, 64-row passes, cost block on the f16
// matrix pipe, operand ring, hand-off rows, task counters: all of that is shared -- but the 64 rows of a pass are
// two blocks of 32, and while block 0 (rows 0..31) works on column s, block 1 (rows 32..63) works on column s - 1:
//     cell A = (r, s)           needs D(r-1, s), D(r-1, s-1), D(r, s-1)            r in block 0
//     cell B = (32 + r, s - 1)  needs D(31+r, s-1), D(31+r, s-2), D(32+r, s-2)
// are independent of each other for every r, block 1's top row is block 0's bottom row one step earlier (two
// registers), and the pass's column state is ONE in-place array of 32 register pairs {D(r, .), D(32 + r, .)}:
//     {cA, cB}  = {v_sqrt_f32 |xA|, v_sqrt_f32 |xB|}
//     {mA, mB}  = {v_min3_f32(upA, diagA, L[r].x), v_min3_f32(upB, diagB, L[r].y)}
//     L[r]      = {cA, cB} + {mA, mB}                      one v_pk_add_f32, in place
// i.e. 2 x 8 + 2 x 4 + 4 = 28 issue cycles per two cells, 14 per cell instead of 16, and 64 registers of column
// state instead of the 128 of the ping-pong arrays.  (Round 1 tried the packed add twice: two PAIRS per lane needed
// twice the state, two skewed COLUMNS of one block left half of every register pair holding a temporary; two
// BLOCKS keep both halves of every pair live.)  A pass takes nCols + 1 steps: block 1 idles in step 0 (its state is
// restored after it) and block 0 computes one column too many at the end (never read).
//
// Measured on MI355X (configs[2], same box, same call; profiles/r02_filter_pk.md): correct (the whole GPU suite passes
// with it), the VALU work per cell falls as planned (15.5 busy cycles per wave-cell against 17.0; the listing is 2
// v_sqrt, 2 v_min3, 1 v_pk_add_f32 per two cells) -- and the launch is only 1...1.5 % faster (32.4 against 32.8 ms; 33.7
// against 34.0 on a slower box).  Without the MFMAs (accumulators opaque, SSYM_PK_ABL=2) the same kernel takes 26.5 ms,
// the 14-cycle floor: the packed addition does not overlap the matrix pipe the way the plain VALU instructions do.
// The compiler already replaces a v_pk_add_f32 in the shadow of the wave's own MFMA by two v_add_f32 (28 of 128 steps,
// +3 %); the issue port idles 8 % of the time against 3 % in the plain kernel, and the clock under load is 2.15
// against 2.22 GHz.  Kept as an experiment behind SSYM_FILTER_PK=1, off in the product.
#pragma once
#include "dtw_filter_kernel.hpp"

namespace ssym {

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SSYM_PK_ABL
#define SSYM_PK_ABL 0        // tools only: 1 = scalar additions, 2 = no MFMAs (opaque accumulators; wrong results)
#endif
template <int KU>
__device__ __forceinline__ f32x16 pk_mfma_tile(const half8 (&a)[KU], const half8 (&b)[KU])
{
#if SSYM_PK_ABL == 2
    f32x16 acc;
    asm volatile("" : "=v"(acc) : "v"(a[0]), "v"(b[0]));
    return acc;
#else
    return mfma_tile<KU>(a, b);
#endif
}

// 16 r-steps of both blocks: rows 16 T + r of block 0 against accA, of block 1 against accB
template <bool SQ, int OFF>
__device__ __forceinline__ void pk_cells(const f32x16 &accA, const f32x16 &accB, f32x2 (&L)[32],
                                         float &upA, float &diagA, float &upB, float &diagB)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        f32x2 c, m;
        c.x = SQ ? __builtin_fabsf(accA[r]) : __builtin_amdgcn_sqrtf(__builtin_fabsf(accA[r]));
        c.y = SQ ? __builtin_fabsf(accB[r]) : __builtin_amdgcn_sqrtf(__builtin_fabsf(accB[r]));
        const f32x2 old = L[OFF + r];
        m.x = __builtin_fminf(__builtin_fminf(upA, diagA), old.x);
        m.y = __builtin_fminf(__builtin_fminf(upB, diagB), old.y);
        diagA = old.x;
        diagB = old.y;
#if SSYM_PK_ABL == 1
        f32x2 cur;                               // tools: the two additions kept apart
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(cur.x) : "v"(c.x), "v"(m.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(cur.y) : "v"(c.y), "v"(m.y));
#else
        const f32x2 cur = c + m;                 // v_pk_add_f32
#endif
        L[OFF + r] = cur;
        upA = cur.x;
        upB = cur.y;
    }
}

template <bool SQ, int KU = kFilterKM>
__global__ __launch_bounds__(64 * kFilterWavesPerBlock, 2) void dtw_filter_pk_kernel(
    const _Float16 *__restrict__ srcRec, const _Float16 *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int srcRows, int nPasses,
    int tgtFramesPad, int mPad, int nSrcPairs, int nTasks, int taskChunk, float outScale,
    float *__restrict__ handoff, unsigned *__restrict__ taskCtr, float *__restrict__ cmat,
    int rowOrigin, int spBase)
{
    constexpr int REC = kFilterRecHalfs;
    constexpr int NT = 4, BR = 64, HB = 32;        // tiles and rows per pass, rows per block
    constexpr int OCC = 2;
    const float INF = __builtin_inff();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    const size_t handGroups = ((size_t)tgtFramesPad + 3) / 4;
    char *const handRow = reinterpret_cast<char *>(handoff + ((size_t)blockIdx.x * kFilterWavesPerBlock + wave) * handGroups * 256);
    const uint32_t laneOff16 = lane * 16;
    constexpr int kFilterRing = filter_ring(OCC);
    constexpr int kWaitLead = wait_vmcnt(KU * (kFilterRing - 2)), kWaitFirst = wait_vmcnt(KU * (kFilterRing - 1));
    __shared__ __attribute__((aligned(16))) char ring[kFilterWavesPerBlock][filter_wave_lds(OCC)];
    char *const myRing = ring[wave];
    char *const myTop = myRing + kFilterRing * kFilterSlotBytes;

    // work distribution: dtw_filter_kernel.hpp (8 counters, one per XCD group, ranges walked from their end)
    const unsigned qd = (unsigned)nTasks >> 3, rm = (unsigned)nTasks & 7u;
    for (unsigned hop = 0; hop < 8; ++hop) {
      const unsigned xcd = (blockIdx.x + hop) & 7u;
      const unsigned rangeLo = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd;
      const unsigned rangeLen = qd + (xcd < rm ? 1u : 0u);
      for (;;) {
        unsigned got = 0;
        if (lane == 0)
            got = atomicAdd(&taskCtr[xcd * kTaskCtrStride], (unsigned)taskChunk);
        got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
        if (got >= rangeLen)
            break;
        const unsigned gotEnd = min(got + (unsigned)taskChunk, rangeLen);
        int preFa = 0, preFb = 0;
        bool havePre = false;
       for (unsigned gi = got; gi < gotEnd; ++gi) {
        const unsigned lin = rangeLo + (rangeLen - 1u - gi);
        const int tg = (int)(lin / (unsigned)nSrcPairs);
        const int sp = spBase + (int)(lin % (unsigned)nSrcPairs);

        const int fa = havePre ? preFa : srcLen[2 * sp + half];
        const int fb_m1 = (havePre ? preFb : tgtLen[32 * tg + col]) - 1;
        havePre = gi + 1 < gotEnd;
        if (havePre) {
            const unsigned linN = lin - 1u;
            preFa = srcLen[2 * (spBase + (int)(linN % (unsigned)nSrcPairs)) + half];
            preFb = tgtLen[32 * (int)(linN / (unsigned)nSrcPairs) + col];
        }
        const int r0 = srcRows - fa;   // first real row: sources are END-ALIGNED in their row slots

        int nCols = fb_m1 + 1, r0min = r0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            nCols = max(nCols, __shfl_xor(nCols, o));
            r0min = min(r0min, __shfl_xor(r0min, o));
        }
        nCols = __builtin_amdgcn_readfirstlane(nCols);
        r0min = __builtin_amdgcn_readfirstlane(r0min);
        const int firstPass = min(max(r0min - rowOrigin, 0) / BR, nPasses - 1);

        float res = INF;
        const char *const tgtGroup = reinterpret_cast<const char *>(tgtRec) + (size_t)tg * tgtFramesPad * (kTgtFrameHalfs * 2);

        for (int pass = nCols > 0 ? firstPass : nPasses; pass < nPasses; ++pass) {
            const int rowBase = rowOrigin + pass * BR;
            const bool haveTop = pass > firstPass;      // wave-uniform
            const bool lastPass = pass == nPasses - 1;

            // operand ring and hand-off tops: as in dtw_filter_kernel.hpp (virtual column c, clamped to the last
            // real one, lives in ring slot c % kFilterRing)
            auto stage = [&](int c) {
                const int cc = min(c, nCols - 1);
                char *slot = myRing + (c & (kFilterRing - 1)) * kFilterSlotBytes;
                const char *gb = tgtGroup + (size_t)cc * (kTgtFrameHalfs * 2);
                static_assert(KU == 2 || KU == 3, "two or three operand planes per column");
                const __attribute__((address_space(1))) void *gp =
                    (const __attribute__((address_space(1))) void *)(gb + laneOff16);
                __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)slot;
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
                if (KU == 3)
                    __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
            };
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
                top = *reinterpret_cast<const float *>(myTop + ((c >> 2) & 1) * 1024 + lane * 16 + (c & 3) * 4);
            };
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
            f32x2 L[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                L[i].x = (rowBase + i == r0 - 1) ? 0.0f : INF;
                L[i].y = (rowBase + HB + i == r0 - 1) ? 0.0f : INF;
            }
            const float diagCol0 = (rowBase == r0) ? 0.0f : INF;   // D(rowBase-1, -1)
            float prevTop = INF;                                    // D(rowBase-1, s-1)
            float botPrev = L[HB - 1].x;                            // D(rowBase+31, s-1): block 1's top row ...
            float botPrev2 = INF;                                   // ... and D(rowBase+31, s-2)

            // B operands of columns s (block 0) and s - 1 (block 1) swap roles every step
            half8 B0[KU], B1[KU];
            float topN = INF;
            __builtin_amdgcn_s_waitcnt(kWaitFirst);
            asm volatile("" ::: "memory");
            fetch(0, B0, topN);
            fetch(0, B1, topN);                 // (step 0 multiplies block 1's rows by SOMETHING: the result is discarded)
            f32x16 accA0 = pk_mfma_tile<KU>(A[0], B0);
            f32x16 accB0 = pk_mfma_tile<KU>(A[2], B0);
            float bq[4] = {INF, INF, INF, INF};                     // bottoms of the current group of 4 columns

            // One step: block 0 on column s (operands Bs), block 1 on column s - 1 (operands Bp).  The body is
            // straight-line: the state L is not touched inside any conditional, so that no copies are needed where
            // control flow meets (with the step under an `if` the allocator moved all 32 pairs at every merge).
            auto step = [&](const int s, half8 (&Bs)[KU], half8 (&Bp)[KU], const int q, const bool first) {
                const float topS = (haveTop && s < nCols) ? topN : INF;
                float upA = topS;
                float diagA = (s == 0) ? diagCol0 : prevTop;
                prevTop = topS;
                float upB = botPrev, diagB = botPrev2;
                // the second tiles of both blocks in this step ...
                f32x16 accA1 = pk_mfma_tile<KU>(A[1], Bs);
                f32x16 accB1 = pk_mfma_tile<KU>(A[3], Bp);
                // ... after which column s - 1's operands are dead: column s + 1 takes their registers
                __builtin_amdgcn_s_waitcnt(kWaitLead);
                asm volatile("" ::: "memory");
                fetch(s + 1, Bp, topN);
                stage(s + kFilterRing);
                if ((s & 3) == 0 && haveTop)
                    stageTop((s >> 2) + 1);
                pk_cells<SQ, 0>(accA0, accB0, L, upA, diagA, upB, diagB);
                // the first tiles of the next step: block 0 on column s + 1, block 1 on column s
                accA0 = pk_mfma_tile<KU>(A[0], Bp);
                accB0 = pk_mfma_tile<KU>(A[2], Bs);
                pk_cells<SQ, 16>(accA1, accB1, L, upA, diagA, upB, diagB);
                botPrev2 = botPrev;
                botPrev = upA;                                      // D(rowBase+31, s)
                if (first)                                          // (a constant at both call sites)
                    return;
                const int j = s - 1;                                // the column block 1 has just finished
                const float bottom = upB;                           // D(rowBase+63, j)
                bq[q] = bottom;
                if (!lastPass) {
                    if ((q == 3 && j < nCols) || j == nCols - 1) {  // (steps past the last column compute nothing that is kept)
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<f32x4 *>(handRow + (size_t)(j >> 2) * 1024 + laneOff16) =
                            f32x4{bq[0], bq[1], bq[2], bq[3]};
                    }
                } else {
                    res = (j == fb_m1) ? bottom : res;              // D(fa-1, fb-1)
                }
            };
            // step 0: block 0 on column 0; block 1 has no column yet -- it runs on column 0's operands and its half of
            // the state, and the value it hands on, are put back afterwards
            step(0, B0, B1, 0, true);
#pragma unroll
            for (int i = 0; i < HB; ++i)
                L[i].y = (rowBase + HB + i == r0 - 1) ? 0.0f : INF;
            botPrev2 = (rowBase + HB - 1 == r0 - 1) ? 0.0f : INF;   // D(rowBase+31, -1)
            // steps 1 ... nCols in groups of four (column j = s - 1 of block 1 is q modulo 4); up to three steps past the
            // end run on clamped operands and leave nothing behind
            for (int s0 = 1; s0 <= nCols; s0 += 4) {
                step(s0 + 0, B1, B0, 0, false);
                step(s0 + 1, B0, B1, 1, false);
                step(s0 + 2, B1, B0, 2, false);
                step(s0 + 3, B0, B1, 3, false);
            }
        }
        cmat[(size_t)(2 * sp + half) * mPad + 32 * tg + col] = res * outScale;
       }
      }
    }
}

}  // namespace ssym
