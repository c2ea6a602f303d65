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

#ifndef SSYM_SP_ROWBLOCK
#define SSYM_SP_ROWBLOCK 4
#endif
constexpr int kSpRowBlock = SSYM_SP_ROWBLOCK;                    // G: DP rows of the first tile are skipped in blocks of G
constexpr int kSpRing = 4;                                     // staged columns per wave
constexpr int sp_slot_bytes(int ku) { return ku * 1024; }      // 64 lanes x KU x 16 B operands of one column

// One column: NT tiles, the first from row R0 (rows above hold padding for both sources of the wave); the next tile's
// MFMA chain is in flight while a tile's cells run.  Lr = D(., j-1), Lw = D(., j).  Straight-line code.
template <int NT, bool SQ, int KU, int R0>
__device__ __forceinline__ float dp_column_sp(const half8 (&A)[NT][KU], const half8 (&Bc)[KU], const half8 (&Bn)[KU],
                                              f32x16 &acc, float diagEnter, const float (&Lr)[NT * 16],
                                              float (&Lw)[NT * 16])
{
    float up = __builtin_inff(), diag = diagEnter;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x16 accn;
        if (T + 1 < NT)
            accn = mfma_tile<KU>(A[T + 1], Bc);
        else
            accn = mfma_tile<KU>(A[0], Bn);   // first tile of the next column
#pragma unroll
        for (int r = (T == 0 ? R0 : 0); r < 16; ++r) {
            const int idx = T * 16 + r;
            const float x = acc[r];
            const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
            const float m = __builtin_fminf(__builtin_fminf(up, diag), Lr[idx]);
            diag = Lr[idx];
            const float cur = c + m;
            Lw[idx] = cur;
            up = cur;
        }
        acc = accn;
    }
    return up;   // D(last row, j)
}

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

template <int NT, bool SQ, int OCC, int KU, int G>
__global__ __launch_bounds__(64 * kFilterWavesPerBlock, OCC) void dtw_filter_sp_kernel(
    const _Float16 *__restrict__ srcRec, const _Float16 *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int srcRows, int tgtFramesPad, int mPad,
    int nSrcPairs, int nTasks, int taskChunk, float outScale, unsigned *__restrict__ taskCtr,
    float *__restrict__ cmat, int rowOrigin, int spBase)
{
    constexpr int REC = kFilterRecHalfs;
    constexpr int BR = NT * 16;
    constexpr int SLOT = sp_slot_bytes(KU);
    constexpr int P = 2 + NT * KU;                               // prefetch DMAs per task
    constexpr int kWaitStrict = wait_vmcnt(2 * KU), kWaitRelaxed = wait_vmcnt(2 * KU + P);
    constexpr unsigned NONE = 0xffffffffu;
    constexpr unsigned kColBytes = kTgtFrameHalfs * 2;           // one frame slot of a target group
    const float INF = __builtin_inff();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    unsigned laneOff16 = lane * 16;
    __shared__ __attribute__((aligned(16))) char lds[kFilterWavesPerBlock][sp_wave_lds(NT, KU)];
    char *const myRing = lds[wave];
    char *const myA = myRing + kSpRing * SLOT;
    char *const myLen = myA + NT * KU * 1024;
    const unsigned long long groupBytes = (unsigned long long)tgtFramesPad * kColBytes;
    const int lastSlot = tgtFramesPad - 1;

    // ---- task sequence: the eight XCD-local counters of dtw_filter_kernel, handed out one task ahead -----------------
    const unsigned qd = (unsigned)nTasks >> 3, rm = (unsigned)nTasks & 7u;
    unsigned hop = 0, xcd = 0, rangeLo = 0, rangeLen = 0, gi = 0, giEnd = 0;
    auto set_range = [&]() {
        xcd = (blockIdx.x + hop) & 7u;
        rangeLo = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd;
        rangeLen = qd + (xcd < rm ? 1u : 0u);
    };
    set_range();
    auto next_task = [&]() -> unsigned {         // linear task id, NONE when every range is spent
        for (;;) {
            if (gi < giEnd)
                return rangeLo + (rangeLen - 1u - gi++);
            if (hop >= 8)
                return NONE;
            unsigned got = 0;
            if (lane == 0)
                got = atomicAdd(&taskCtr[xcd * kTaskCtrStride], (unsigned)taskChunk);
            got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
            if (got < rangeLen) {
                gi = got;
                giEnd = min(got + (unsigned)taskChunk, rangeLen);
            } else {
                ++hop;
                if (hop < 8)
                    set_range();
            }
        }
    };
    // frame slot c of the target group at byte offset groupOff -> ring slot of virtual column vcol.  The address is a
    // scalar base plus the lane's constant 32-bit offset: no vector instruction per column
    auto stage = [&](unsigned long long groupOff, int c, unsigned vcol) {
        const char *gb = reinterpret_cast<const char *>(tgtRec) + groupOff + (unsigned long long)min(c, lastSlot) * kColBytes;
        asm volatile("" : "+s"(gb));              // (opaque: keeps the base out of 64-bit vector additions)
        asm volatile("" : "+v"(laneOff16));
        char *slot = myRing + (vcol & (kSpRing - 1)) * SLOT;
        const __attribute__((address_space(1))) void *gp = (const __attribute__((address_space(1))) void *)(gb + laneOff16);
        __attribute__((address_space(3))) void *lp = (__attribute__((address_space(3))) void *)slot;
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
        if (KU == 3)
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
    };
    auto fetch = [&](unsigned vcol, half8 (&B)[KU]) {
        const char *slot = myRing + (vcol & (kSpRing - 1)) * SLOT;
#pragma unroll
        for (int m = 0; m < KU; ++m)
            B[m] = *reinterpret_cast<const half8 *>(slot + m * 1024 + lane * 16);
    };
    auto task_tg = [&](unsigned lin) { return (int)(lin / (unsigned)nSrcPairs); };
    auto task_sp = [&](unsigned lin) { return spBase + (int)(lin % (unsigned)nSrcPairs); };
    // a task's lengths and source operands -> the staging block (P DMAs)
    auto prefetch = [&](int tg, int sp) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcLen + 2 * sp + half),
                                         (__attribute__((address_space(3))) void *)myLen, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tgtLen + 32 * tg + col),
                                         (__attribute__((address_space(3))) void *)(myLen + 256), 4, 0, 0);
        const int arow = lane & 31;
        const int a_src = 2 * sp + ((arow >> 2) & 1);
        const int a_frm = rowOrigin + (arow & 3) + 4 * (arow >> 3);
        const char *abase = reinterpret_cast<const char *>(srcRec + ((size_t)a_src * srcRows + a_frm) * REC + half * 24);
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int m = 0; m < KU; ++m)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(abase + (size_t)T * kFilterRowsPerTile * REC * 2 + m * 16),
                    (__attribute__((address_space(3))) void *)(myA + (T * KU + m) * 1024), 16, 0, 0);
    };

    unsigned cur = next_task();
    if (cur == NONE)
        return;
    unsigned vc = 0;                              // virtual column of the current task's column 0 (ring slot = vc & 3)
    {   // the first task: nothing to hide its operands behind
        const int tg = task_tg(cur), sp = task_sp(cur);
        prefetch(tg, sp);
#pragma unroll
        for (int c = 0; c < kSpRing; ++c)
            stage((unsigned long long)tg * groupBytes, c, (unsigned)c);
    }
    unsigned nxt = next_task();
    __builtin_amdgcn_s_waitcnt(wait_vmcnt(0));

    for (;;) {
        const int tg = task_tg(cur), sp = task_sp(cur);
        const unsigned nl = nxt == NONE ? cur : nxt;          // (no next task: the prefetches repeat this one, unused)
        const int ntg = task_tg(nl), nsp = task_sp(nl);
        const unsigned long long tgtOff = (unsigned long long)tg * groupBytes, nextOff = (unsigned long long)ntg * groupBytes;

        // this task's lengths and source operands: prefetched during the previous task (landed before its column 3)
        asm volatile("" ::: "memory");
        const int fa = *reinterpret_cast<const int *>(myLen + lane * 4);
        const int fb = *reinterpret_cast<const int *>(myLen + 256 + lane * 4);
        half8 A[NT][KU];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int m = 0; m < KU; ++m)
                A[T][m] = *reinterpret_cast<const half8 *>(myA + (T * KU + m) * 1024 + lane * 16);

        const int fb_m1 = fb - 1;
        const int r0 = srcRows - fa;                          // first real row: sources are END-ALIGNED in their slots
        // at least a ring's worth of columns: the ring stays in step (columns beyond a target's end read zero records)
        const int nCols = max(wave_max_lo32(fb), kSpRing);
        const int mk = r0 - 1 - rowOrigin;                    // the lane's marker row: D(r0 - 1, -1) = 0 starts the recurrence
        const int mk0 = __builtin_amdgcn_readlane(mk, 0), mk1 = __builtin_amdgcn_readlane(mk, 32);
        const int sk = min(max(min(mk0, mk1) + 1, 0), 15) / G;           // first row block of the first tile with a frame
        const int rowEnter = rowOrigin + sk * G;

        // D(., -1): +inf, except the marker -- at most two rows of the wave (one per source), found with scalar compares;
        // a source whose first row is the entry row starts from diagCol0 instead
        float L0[BR], L1[BR];
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            L0[i] = INF;
            if (i == mk0 || i == mk1)                         // wave-uniform
                L0[i] = (i == mk) ? 0.0f : INF;
            L1[i] = INF;
        }
        const float diagCol0 = (r0 == rowEnter) ? 0.0f : INF;

        // columns 0 and 1 have landed (S(-4), S(-3) of the previous task; younger: S(-2), S(-1), its result store)
        __builtin_amdgcn_s_waitcnt(kWaitStrict);
        asm volatile("" ::: "memory");
        half8 B0[KU], B1[KU];
        fetch(vc, B0);
        f32x16 acc = mfma_tile<KU>(A[0], B0);
        // every read of the staging block has returned before the next task's operands overwrite it
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0)
        asm volatile("" ::: "memory");
        prefetch(ntg, nsp);
        asm volatile("" ::: "memory");

        float res = INF;
        // the column loop, two columns per trip (B0 / B1 and L0 / L1 swap roles), once per first row R0 of the first tile
        auto run_columns = [&](auto r0tag) {
            constexpr int R0 = decltype(r0tag)::value;
            for (int j = 0; j < nCols; j += 2) {
                {   // even column j: operands in B0, D(., j-1) in L0
                    if (j == 2)
                        __builtin_amdgcn_s_waitcnt(kWaitRelaxed);
                    else if (j != 0)
                        __builtin_amdgcn_s_waitcnt(kWaitStrict);
                    asm volatile("" ::: "memory");
                    fetch(vc + j + 1, B1);
                    // into the slot column j just left: this task's column j + 4, or the next task's first columns
                    const bool own = j + kSpRing < nCols;
                    stage(own ? tgtOff : nextOff, own ? j + kSpRing : j + kSpRing - nCols, vc + j);
                    const float bottom = dp_column_sp<NT, SQ, KU, R0>(A, B0, B1, acc, (j == 0) ? diagCol0 : INF, L0, L1);
                    res = (j == fb_m1) ? bottom : res;        // D(fa-1, fb-1)
                }
                if (j + 1 < nCols) {                          // odd column j + 1 (wave-uniform)
                    if (j == 0)
                        __builtin_amdgcn_s_waitcnt(kWaitRelaxed);
                    else
                        __builtin_amdgcn_s_waitcnt(kWaitStrict);
                    asm volatile("" ::: "memory");
                    fetch(vc + j + 2, B0);
                    const bool own = j + 1 + kSpRing < nCols;
                    stage(own ? tgtOff : nextOff, own ? j + 1 + kSpRing : j + 1 + kSpRing - nCols, vc + j + 1);
                    const float bottom = dp_column_sp<NT, SQ, KU, R0>(A, B1, B0, acc, INF, L1, L0);
                    res = (j + 1 == fb_m1) ? bottom : res;
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
        cmat[(size_t)(2 * sp + half) * mPad + 32 * tg + col] = res * outScale;
        if (nxt == NONE)
            break;
        // the next task becomes the current one
        vc += (unsigned)nCols;
        cur = nxt;
        nxt = next_task();
    }
    // (the last task's stages of "next" columns are still in flight: they must have landed before the LDS is released)
    __builtin_amdgcn_s_waitcnt(wait_vmcnt(0));
}

}  // namespace ssym
