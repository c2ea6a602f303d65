// dtw_filter_sp_kernel.hpp -- the dtw MFMA filter for SHORT sources: one row pass (at most 16 * NT frames per source),
// software-pipelined ACROSS tasks, DP rows that hold only padding skipped.
//
// Why a second kernel.  The reference's segments are short and ragged -- add_segments gives a segment seg / HOP * NCOEFFS
// values (src/sound.rs:330-343), seg = letters x 256 samples (src/lib.rs:137): 5...40 frames -- and dtw_filter_kernel
// (dtw_filter_kernel.hpp), shaped for 128-frame segments, lost more than half of its rate there, in two ways measured on
// MI355X (tools/shape_timing.py, round 4):
//   * a task's start costs two dependent memory round trips (source operands, then the first target columns through the
//     LDS ring): 4.9 us of a wave's time per task at 32 rows, 5.9 at 48, whatever the number of columns -- as much as ten
//     columns of work; the partner wave of the SIMD does not fill it (a wave alone on a SIMD runs the dependent
//     min3 -> add chain at about half the paired rate);
//   * every source paid whole 16-row tiles: a 17-frame source 32 DP rows.
// Here a wave knows its NEXT task while it works on the current one:
//   * the next task's source operands and lengths are requested right after the current task's first column has been
//     fetched -- by DMA into a per-wave LDS staging block, so that no register waits for them and the compiler sees no
//     pending load it would have to wait for -- and the ring never drains: the stage slot a column of the current task
//     vacates during its last four columns takes the first four columns of the NEXT task's target group.  A task's start
//     is then a handful of LDS reads and one MFMA chain.  (Operands prefetched into a second REGISTER set were tried
//     first: at three tiles the allocator spilled them, each spill behind an s_waitcnt vmcnt(0).)
//   * the DP of the first tile starts at the first row (rounded down to G) at which either source of the wave has a frame:
//     sources are END-aligned in their slots and ordered by length, so the rows above are padding for both (their cells
//     would be +inf).  The matrix pipe still multiplies the whole tile; the VALU, which sets the pace, skips the cells.
//     The first row is a COMPILE-TIME constant of the column loop, which exists once per row block (16 / G copies, chosen
//     per task by a scalar switch): a first version tested the row block inside the column (a branch per block) and the
//     joins cost register copies of the accumulator tile and 22 instead of ~10 overhead instructions per column.
// Everything else is dtw_filter_kernel's: one pair per lane, accumulator register r = row r of the lane's own pair,
// min-of-three lane-local, 3 VALU instructions per cell, target records group-major through a per-wave LDS ring by
// global_load_lds DMA, tasks from eight XCD-local counters walked from their end.  Results are bit-identical to
// dtw_filter_kernel's (tests/test_gpu_numerics.py): the same MFMAs in the same order, the same cells in the same order.
//
// s_waitcnt bookkeeping (vmcnt counts loads, DMAs and stores in issue order; KU DMAs per staged column, ring of 4).
// S(j) = the stage issued during column j (data of column j + 4, or of the next task's column j + 4 - nCols).  Column j
// reads column j + 1 from the ring at its top, i.e. S(j - 3): younger are S(j - 2), S(j - 1) -> vmcnt(2 KU).  The P =
// 2 + NT * KU prefetch DMAs of the next task are issued between the task's first wait and S(0): columns 1 and 2 need
// S(-2), S(-1) of the previous task, which are OLDER than those DMAs, so their waits allow P more (vmcnt(2 KU + P));
// column 3's wait (every task has at least four columns) is younger than S(0), S(1): everything prefetched has landed
// long before the task's end reads it.  The result store of the previous task (and an occasional task-counter atomic)
// sit in the same queue: they are never counted as allowed, which can only make a wait stricter than needed by an
// operation that was issued more than a column ago.
#pragma once
#include "dtw_filter_kernel.hpp"
#include <type_traits>

namespace ssym {

// tools only (-DSSYM_SP_PROF): shader-clock ticks per wave summed into ssym_sp_prof -- [0] waves, [1] whole life of a wave,
// [2] column loops, [3] task starts (staging reads ... first MFMA), [4] waiting for task-counter grabs, [5] tasks, [6] columns,
// [7] the first task's operands (nothing to hide them behind), [8] from a wave's start to its first task, [9] from a task's
// last column to the next task (result store, task bookkeeping), [10] after the last task
#ifdef SSYM_SP_PROF
__device__ unsigned long long ssym_sp_prof[12];
#define SSYM_SP_T(v_) const unsigned long long v_ = __builtin_readcyclecounter()
#define SSYM_SP_ACC(k_, d_) prof[k_] += (d_)
#else
#define SSYM_SP_T(v_)
#define SSYM_SP_ACC(k_, d_)
#endif

#ifndef SSYM_SP_ROWBLOCK
#define SSYM_SP_ROWBLOCK 4
#endif
constexpr int kSpRowBlock = SSYM_SP_ROWBLOCK;                    // G: DP rows of the first tile are skipped in blocks of G
constexpr int kSpRing = 4;                                     // staged columns per wave
constexpr int sp_slot_bytes(int ku) { return ku * 1024; }      // 64 lanes x KU x 16 B operands of one column

// One column: NT tiles, the first from row R0 (rows above hold padding for both sources of the wave); the next tile's
// MFMA chain is in flight while a tile's cells run.  Lr = D(., j-1), Lw = D(., j).  Straight-line code.
// FIRST: column 0, where D(i, 0) = c(i, 0) + (i == the source's first row ? 0 : D(i - 1, 0)) -- nothing is read from the
// column before it, so neither column array needs initialising when a task starts (r0rel: the lane's first real row).
// MP ("multi-pair", one-tile sources): the NT tiles of a column are NT DIFFERENT source pairs of at most 16 frames each --
// NT independent recurrences per lane against the same target column.  One-tile columns fetched 2 KB of target operands for
// 8...16 cells per lane and ran into the DMA path's 36 bytes per clock and CU (the one-tile class lost 13-21 % to its own
// staging, tools: -DSSYM_SP_NOSTAGE); three pairs per wave take a third of the bytes per cell, amortise the column's
// overhead over three times the cells and give the wave three independent chains to issue from.
template <int NT, bool SQ, int KU, int R0, bool FIRST, bool MP>
__device__ __forceinline__ void dp_column_sp(const half8 (&A)[NT][KU], const half8 (&Bc)[KU], const half8 (&Bn)[KU],
                                             f32x16 &acc, const int (&r0rel)[MP ? NT : 1], const float (&Lr)[NT * 16],
                                             float (&Lw)[NT * 16], float (&bottom)[MP ? NT : 1])
{
    float up = __builtin_inff(), diag = __builtin_inff();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x16 accn;
        if (T + 1 < NT)
            accn = mfma_tile<KU>(A[T + 1], Bc);
        else
            accn = mfma_tile<KU>(A[0], Bn);   // first tile of the next column
        if (MP) {                             // another pair: its recurrence starts afresh
            up = __builtin_inff();
            diag = __builtin_inff();
        }
#pragma unroll
        for (int r = ((MP || T == 0) ? R0 : 0); r < 16; ++r) {
            const int idx = T * 16 + r;
            const float x = acc[r];
            const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
            float m;
            if (FIRST) {
                m = ((MP ? r : idx) == r0rel[MP ? T : 0]) ? 0.0f : up;
            } else {
                m = __builtin_fminf(__builtin_fminf(up, diag), Lr[idx]);
                diag = Lr[idx];
            }
            const float cur = c + m;
            Lw[idx] = cur;
            up = cur;
        }
        if (MP)
            bottom[T] = up;                   // D(last row of this pair, j)
        acc = accn;
    }
    if (!MP)
        bottom[0] = up;                       // D(last row, j)
}

// BYTES per lane: (scalar base + the lane's constant offset) -> LDS block l (+ OFF on both sides).  The base is made opaque
// so that it stays in scalar registers (the optimiser would otherwise fold it into 64-bit vector additions per access).
// (A macro: as a __device__ function template its inline-asm constraints fail the HOST pass silently, and the kernel's
// later instantiations lose their launch stubs.)
#define SSYM_SP_DMA(BASE_, LANEOFF_, LDS_, BYTES_, OFF_)                                                          \
    do {                                                                                                          \
        const char *dmaBase_ = (BASE_);                                                                           \
        asm volatile("" : "+s"(dmaBase_));                                                                        \
        asm volatile("" : "+v"(LANEOFF_));                                                                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dmaBase_ + (LANEOFF_)), \
                                         (__attribute__((address_space(3))) void *)(LDS_), BYTES_, OFF_, 0);      \
    } while (0)

// max over the wave's lanes 0..31 of a non-negative value (the group's longest target): five DPP steps instead of the
// LDS round trips of a shuffle reduction (gfx9 row_shr / row_bcast), result as a scalar
__device__ __forceinline__ int wave_max_lo32(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of a row: the row's max
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> lane 31: rows 0 and 1
    return __builtin_amdgcn_readlane(v, 31);
}

// LDS per wave: the ring, then the next task's source operands [NT][KU][64 lanes][16 B] and lengths [2][64 lanes][4 B]
constexpr int sp_wave_lds(int nt, int ku) { return kSpRing * sp_slot_bytes(ku) + nt * ku * 1024 + 512; }

// nSrcPairs: the launch's source pairs -- MP: its TASK pairs, NT source pairs each, of which the launch holds pairLimit
template <int NT, bool SQ, int OCC, int KU, int G, bool MP = false>
__global__ __launch_bounds__(64 * kFilterWavesPerBlock, OCC) void dtw_filter_sp_kernel(
    const _Float16 *__restrict__ srcRec, const _Float16 *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int srcRows, int tgtFramesPad, int mPad,
    int nSrcPairs, int pairLimit, int taskChunk, float outScale, unsigned *__restrict__ taskCtr,
    float *__restrict__ cmat, int rowOrigin, int spBase, int pairBlock)
{
    constexpr int NP = MP ? NT : 1;                              // source pairs per task
    constexpr int REC = kFilterRecHalfs;
    constexpr int BR = NT * 16;
    constexpr int SLOT = sp_slot_bytes(KU);
    constexpr int P = 2 + NT * KU;                               // prefetch DMAs per task
    constexpr int kWaitStrict = wait_vmcnt(2 * KU), kWaitRelaxed = wait_vmcnt(2 * KU + P);
    constexpr unsigned kColBytes = kTgtFrameHalfs * 2;           // one frame slot of a target group
    const float INF = __builtin_inff();
#ifdef SSYM_SP_PROF
    unsigned long long prof[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long profMark = 0;
    auto prof_flush = [&]() {
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 12; ++k)
                atomicAdd(&ssym_sp_prof[k], prof[k]);
    };
#endif
    SSYM_SP_T(tLife0);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    __shared__ __attribute__((aligned(16))) char lds[kFilterWavesPerBlock][sp_wave_lds(NT, KU)];
    char *const myRing = lds[wave];
    char *const myA = myRing + kSpRing * SLOT;
    char *const myLen = myA + NT * KU * 1024;
    const unsigned long long groupBytes = (unsigned long long)tgtFramesPad * kColBytes;
    const unsigned long long pairBytes = (unsigned long long)srcRows * (2 * REC * 2);        // the two sources of a pair
    const int lastSlot = tgtFramesPad - 1;
    // per-lane constants of the task's addresses: every access below is (scalar base of the task) + (one of these)
    unsigned laneOff16 = lane * 16;                                                          // a column's operands
    unsigned laneA;                                                                          // the lane's A rows in its pair
    {
        const int arow = lane & 31;
        laneA = (unsigned)((((arow >> 2) & 1) * srcRows + rowOrigin + (arow & 3) + 4 * (arow >> 3)) * (REC * 2) + half * 48);
    }
    unsigned laneSrcLen = (MP ? min(lane, 2 * NP - 1) : half) * 4, laneTgtLen = col * 4;     // (MP: the 2 NP lengths of the task's pairs)
    unsigned laneOut = ((unsigned)half * (unsigned)mPad + (unsigned)col) * 4u;               // (mPad < 2^29: n_pad is 32 bits)

    // ---- task sequence: eight XCD-local counters (workgroups b, b + 8, ... share an XCD), handed out one task ahead.
    // XCD x owns the target groups x, x + 8, x + 16, ... -- groups are ordered by length, so every XCD gets the same mix
    // of short and long targets (contiguous ranges gave the last XCD eight times the first one's work on 5...40-frame
    // targets) -- times all source pairs.  Its range is ordered [block of `pairBlock` source pairs][its groups][pairs of the
    // block]: a block's source records (about 1 MB) and the XCD's target groups stay in its L2 while every (group, pair) of
    // the block is worked off -- with all pairs under one group the sources did not fit the L2 and were fetched from the
    // Infinity Cache once per target group (1.0 GB of fabric traffic per 4096 x 4096 search of 5...40-frame segments).  The
    // short first block takes the remainder; a range is walked from its END (longest sources and targets first); a wave
    // whose range is spent helps the next XCD's.  Inside a chunk the walk is a decrement: divisions per grab, none per task.
    const unsigned nGroups = (unsigned)mPad >> 5;
    const unsigned PB = (unsigned)max(pairBlock, 1), firstBlock = (unsigned)nSrcPairs % PB;     // pairs [0, firstBlock) form block 0
    unsigned hop = 0, xcd = 0, rangeLen = 0, chunkLeft = 0, nG = 0;
    int ctg = 0, csp = 0, cLo = 0, cHi = 0;      // the next task of the chunk: target group, source pair; its block's pairs [cLo, cHi)
    auto set_range = [&]() {
        xcd = (blockIdx.x + hop) & 7u;
        nG = xcd < nGroups ? (nGroups - xcd + 7u) >> 3 : 0u;
        rangeLen = nG * (unsigned)nSrcPairs;
    };
    set_range();
    auto next_task = [&](int &tg, int &sp) -> bool {      // false when every range is spent
        for (;;) {
            if (chunkLeft) {
                tg = ctg;
                sp = csp;
                --chunkLeft;
                if (--csp < cLo) {                        // the block's pairs under the next (shorter) group ...
                    csp = cHi - 1;
                    ctg -= 8;
                    if (ctg < 0) {                        // ... or, all its groups done, the block before it
                        cHi = cLo;
                        cLo = max(cLo - (int)PB, 0);
                        csp = cHi - 1;
                        ctg = (int)(xcd + 8u * (nG - 1u));
                    }
                }
                return true;
            }
            if (hop >= 8)
                return false;
            SSYM_SP_T(tg0);
            unsigned got = 0;
            if (lane == 0)
                got = atomicAdd(&taskCtr[xcd * kTaskCtrStride], (unsigned)taskChunk);
            got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
            SSYM_SP_T(tg1);
            SSYM_SP_ACC(4, tg1 - tg0);
            if (got < rangeLen) {
                const unsigned i = rangeLen - 1u - got, head = nG * firstBlock;      // position in the range; tasks of block 0
                unsigned gl, within;
                if (i < head) {
                    cLo = 0;
                    cHi = (int)firstBlock;
                    gl = i / firstBlock;
                    within = i - gl * firstBlock;
                } else {
                    const unsigned r = i - head, perBlock = nG * PB, b = r / perBlock, rb = r - b * perBlock;
                    cLo = (int)(firstBlock + b * PB);
                    cHi = cLo + (int)PB;
                    gl = rb / PB;
                    within = rb - gl * PB;
                }
                chunkLeft = min((unsigned)taskChunk, rangeLen - got);
                csp = cLo + (int)within;
                ctg = (int)(xcd + 8u * gl);
            } else {
                ++hop;
                if (hop < 8)
                    set_range();
            }
        }
    };
    // frame slot c of the target group at `base` -> ring slot of virtual column vcol (a group's slots span less than 2^32 bytes)
    auto group_base = [&](int tg) { return reinterpret_cast<const char *>(tgtRec) + (unsigned long long)tg * groupBytes; };
    auto stage = [&](const char *base, int c, unsigned vcol) {
        const char *gb = base + (unsigned)min(c, lastSlot) * kColBytes;
        char *slot = myRing + (vcol & (kSpRing - 1)) * SLOT;
        SSYM_SP_DMA(gb, laneOff16, slot, 16, 0);
        SSYM_SP_DMA(gb, laneOff16, slot, 16, 1024);
        if (KU == 3)
            SSYM_SP_DMA(gb, laneOff16, slot, 16, 2048);
    };
    auto fetch = [&](unsigned vcol, half8 (&B)[KU]) {
        const char *slot = myRing + (vcol & (kSpRing - 1)) * SLOT;
#pragma unroll
        for (int m = 0; m < KU; ++m)
            B[m] = *reinterpret_cast<const half8 *>(slot + m * 1024 + lane * 16);
    };
    // a task's lengths and source operands -> the staging block (P DMAs)
    // (MP: task pair sp holds the launch's source pairs NP sp ... NP sp + NP - 1; those beyond pairLimit belong to the next
    //  class: their operands are read from the launch's last pair instead, their lengths count as 0, nothing is stored)
    auto first_pair = [&](int sp) { return spBase + (MP ? NP * sp : sp); };
    auto prefetch = [&](int tg, int sp) {
        const int fp = first_pair(sp), lastPair = spBase + pairLimit - 1;
        SSYM_SP_DMA(reinterpret_cast<const char *>(srcLen + 2 * fp), laneSrcLen, myLen, 4, 0);
        SSYM_SP_DMA(reinterpret_cast<const char *>(tgtLen + 32 * tg), laneTgtLen, myLen + 256, 4, 0);
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const char *pb = reinterpret_cast<const char *>(srcRec) +
                             (unsigned long long)(MP ? min(fp + T, lastPair) : fp) * pairBytes + (MP ? 0 : T * (kFilterRowsPerTile * REC * 2));
#pragma unroll
            for (int m = 0; m < KU; ++m)
                SSYM_SP_DMA(pb + m * 16, laneA, myA + (T * KU + m) * 1024, 16, 0);
        }
    };

    int tg, sp, ntg = 0, nsp = 0;
    if (!next_task(tg, sp)) {
#ifdef SSYM_SP_PROF
        prof_flush();
#endif
        return;
    }
    unsigned vc = 0;                              // virtual column of the current task's column 0 (ring slot = vc & 3)
    prefetch(tg, sp);                             // the first task: nothing to hide its operands behind
#pragma unroll
    for (int c = 0; c < kSpRing; ++c)
        stage(group_base(tg), c, (unsigned)c);
    bool haveNext = next_task(ntg, nsp);
    {
        SSYM_SP_T(tf0);
        __builtin_amdgcn_s_waitcnt(wait_vmcnt(0));
        SSYM_SP_T(tf1);
        SSYM_SP_ACC(7, tf1 - tf0);
    }

    for (;;) {
        if (!haveNext) {                          // (no next task: the prefetches repeat this one, unused)
            ntg = tg;
            nsp = sp;
        }
        // this task's lengths and source operands: prefetched during the previous task (landed before its column 3)
        SSYM_SP_T(tTask0);
#ifdef SSYM_SP_PROF
        prof[profMark ? 9 : 8] += tTask0 - (profMark ? profMark : tLife0);
#endif
        asm volatile("" ::: "memory");
        int fa[NP];
#pragma unroll
        for (int T = 0; T < NP; ++T) {
            fa[T] = *reinterpret_cast<const int *>(myLen + (MP ? 2 * T + half : lane) * 4);
            if (MP && NP * sp + T >= pairLimit)               // (wave-uniform)
                fa[T] = 0;
        }
        const int fb = *reinterpret_cast<const int *>(myLen + 256 + lane * 4);
        half8 A[NT][KU];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int m = 0; m < KU; ++m)
                A[T][m] = *reinterpret_cast<const half8 *>(myA + (T * KU + m) * 1024 + lane * 16);

        const int fb_m1 = fb - 1;
        int r0rel[NP];                                        // the lane's first real row(s): sources are END-ALIGNED in their slots
        int r0lo = 16 * NT;
#pragma unroll
        for (int T = 0; T < NP; ++T) {
            r0rel[T] = srcRows - fa[T] - rowOrigin;
            r0lo = min(r0lo, min(__builtin_amdgcn_readlane(r0rel[T], 0), __builtin_amdgcn_readlane(r0rel[T], 32)));
        }
        // at least a ring's worth of columns: the ring stays in step (columns beyond a target's end read zero records)
        const int nCols = max(wave_max_lo32(fb), kSpRing);
        const int sk = min(max(r0lo, 0), 15) / G;             // first row block of the first tile with a frame

        // columns 0 and 1 have landed (S(-4), S(-3) of the previous task; younger: S(-2), S(-1), its result store)
        __builtin_amdgcn_s_waitcnt(kWaitStrict);
        asm volatile("" ::: "memory");
        // every read of the staging block has returned before the next task's operands overwrite it
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        prefetch(ntg, nsp);
        asm volatile("" ::: "memory");

        float res[NP];
#pragma unroll
        for (int T = 0; T < NP; ++T)
            res[T] = INF;
        const char *const ownBase = group_base(tg), *const nextBase = group_base(ntg);
        SSYM_SP_T(tCols0);
        SSYM_SP_ACC(3, tCols0 - tTask0);
        SSYM_SP_ACC(5, 1);
        SSYM_SP_ACC(6, (unsigned long long)nCols);
        // The columns, once per first row R0 of the first tile: column 0, then two columns per trip (B0 / B1 and the column
        // arrays swap roles).  Column j reads column j + 1 from the ring and stages into the slot column j just left: this
        // task's column j + 4, or the next task's first columns.
        auto run_columns = [&](auto r0tag) {
            constexpr int R0 = decltype(r0tag)::value;
            float L0[BR], L1[BR];                             // D(., j) for odd / even j; column 0 needs no column before it
            // (operand registers and the accumulator tile belong to the loop copy: defined in front of the switch they had
            //  to sit in the same registers at every copy's entry, which cost ~30 registers)
            half8 B0[KU], B1[KU];
            fetch(vc, B0);
            f32x16 acc = mfma_tile<KU>(A[0], B0);
            auto stage_next = [&](int j) {
#ifndef SSYM_SP_NOSTAGE      // (tools only: timing without the columns' DMAs -- the MFMAs then run on stale bytes)
                const bool own = j + kSpRing < nCols;
                stage(own ? ownBase : nextBase, own ? j + kSpRing : j + kSpRing - nCols, vc + j);
#else
                (void)j;
#endif
            };
            {
                fetch(vc + 1, B1);
                stage_next(0);
                float bottom[NP];
                dp_column_sp<NT, SQ, KU, R0, true, MP>(A, B0, B1, acc, r0rel, L0, L1, bottom);
#pragma unroll
                for (int T = 0; T < NP; ++T)
                    res[T] = (0 == fb_m1) ? bottom[T] : res[T];            // D(fa-1, fb-1)
            }
            for (int j = 1; j < nCols; j += 2) {
                {   // odd column j: operands in B1, D(., j-1) in L1
                    if (j == 1)
                        __builtin_amdgcn_s_waitcnt(kWaitRelaxed);
                    else
                        __builtin_amdgcn_s_waitcnt(kWaitStrict);
                    asm volatile("" ::: "memory");
                    fetch(vc + j + 1, B0);
                    stage_next(j);
                    float bottom[NP];
                    dp_column_sp<NT, SQ, KU, R0, false, MP>(A, B1, B0, acc, r0rel, L1, L0, bottom);
#pragma unroll
                    for (int T = 0; T < NP; ++T)
                        res[T] = (j == fb_m1) ? bottom[T] : res[T];
                }
                if (j + 1 < nCols) {                          // even column j + 1 (wave-uniform)
                    if (j == 1)
                        __builtin_amdgcn_s_waitcnt(kWaitRelaxed);
                    else
                        __builtin_amdgcn_s_waitcnt(kWaitStrict);
                    asm volatile("" ::: "memory");
                    fetch(vc + j + 2, B1);
                    stage_next(j + 1);
                    float bottom[NP];
                    dp_column_sp<NT, SQ, KU, R0, false, MP>(A, B0, B1, acc, r0rel, L0, L1, bottom);
#pragma unroll
                    for (int T = 0; T < NP; ++T)
                        res[T] = (j + 1 == fb_m1) ? bottom[T] : res[T];
                }
            }
        };
        static_assert(G == 2 || G == 4 || G == 8 || G == 16, "row blocks of 2, 4, 8 or 16");
#define SSYM_SP_CASE(K_)                                                                              \
        case K_:                                                                                      \
            if constexpr ((K_) * G < 16)                                                              \
                run_columns(std::integral_constant<int, ((K_) * G < 16 ? (K_) * G : 0)>{});           \
            break;
        switch (sk) {
            SSYM_SP_CASE(0) SSYM_SP_CASE(1) SSYM_SP_CASE(2) SSYM_SP_CASE(3)
            SSYM_SP_CASE(4) SSYM_SP_CASE(5) SSYM_SP_CASE(6) SSYM_SP_CASE(7)
        }
#undef SSYM_SP_CASE
        {
            SSYM_SP_T(tCols1);
            SSYM_SP_ACC(2, tCols1 - tCols0);
#ifdef SSYM_SP_PROF
            profMark = tCols1;
#endif
        }
#pragma unroll
        for (int T = 0; T < NP; ++T) {   // the result(s): (scalar base of the task's pair) + (the lane's constant offset)
            if (MP && NP * sp + T >= pairLimit)
                continue;
            char *ob = reinterpret_cast<char *>(cmat) + ((unsigned long long)(2 * (first_pair(sp) + T)) * (unsigned)mPad + 32u * (unsigned)tg) * 4ull;
            asm volatile("" : "+s"(ob));
            asm volatile("" : "+v"(laneOut));
            *reinterpret_cast<float *>(ob + laneOut) = res[T] * outScale;
        }
        if (!haveNext)
            break;
        // the next task becomes the current one
        vc += (unsigned)nCols;
        tg = ntg;
        sp = nsp;
        haveNext = next_task(ntg, nsp);
    }
    // (the last task's stages of "next" columns are still in flight: they must have landed before the LDS is released)
    __builtin_amdgcn_s_waitcnt(wait_vmcnt(0));
#ifdef SSYM_SP_PROF
    {
        SSYM_SP_T(tLife1);
        SSYM_SP_ACC(1, tLife1 - tLife0);
        SSYM_SP_ACC(10, tLife1 - profMark);
        prof_flush();
    }
#endif
}

}  // namespace ssym

#undef SSYM_SP_DMA
