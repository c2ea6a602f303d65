// dtw_band_kernel.hpp -- Sakoe-Chiba banded variant of the dtw MFMA filter (|i - j| <= r).
//
// Same idea as dtw_filter_kernel.hpp -- one (source, target) pair per lane, cost block on the f16
// matrix pipe, lane-local recurrence on the VALU -- but in DIAGONAL coordinates: column j keeps
// the 2r+1 in-band cells k = i - j + r in registers L[0..2r].  Then
//     D(j,k) = c + min3( D(j,k-1) [cell (i-1,j)],  Dprev(k+1) [cell (i,j-1)],  Dprev(k) [cell (i-1,j-1)] )
// updates L in place for ascending k (L[k] is consumed as the diagonal of cell k after it served as
// the left neighbour of cell k-1), the band edges need no masks (k = 0 has no upper neighbour,
// L[2r+1] stays +inf), and the work per pair is F*(2r+1) cells instead of F^2.
// The price: the 16 source frames an MFMA tile needs move down by one frame per column.  A
// workgroup therefore keeps ITS source pair (all frames, padded with |a|^2 = +inf records before
// frame 0 and after the last frame) in LDS; its 8 waves work on 8 different target groups against
// that pair, never synchronising except when the workgroup moves to the next source pair.
#pragma once
#include "dtw_filter_kernel.hpp"

#include <type_traits>

namespace ssym {

constexpr int kBandTgtQuantum = 256;   // targets are padded to this (8 groups of 32)
// The waves of a workgroup take the target groups of the staged source pair from an LDS counter instead of owning
// one group per block: the two waves of a SIMD do not advance at the same pace (the older one wins the VALU
// arbitration), and with a fixed share the faster wave of every SIMD sat at the workgroup barrier while the slower
// one finished alone.
#ifndef SSYM_BAND_DYNAMIC_GROUPS
#define SSYM_BAND_DYNAMIC_GROUPS 1
#endif
constexpr bool kBandDynamicGroups = SSYM_BAND_DYNAMIC_GROUPS != 0;
#ifndef SSYM_BAND_ABL
#define SSYM_BAND_ABL 0   // tools only (wrong results, valid timing): 1 = no MFMAs, 2 = no LDS operand reads, 3 = no target loads
#endif
#ifndef SSYM_BAND_PHASES
#define SSYM_BAND_PHASES 1      // 0: tools only, every tile of every column (the round-1 loop)
#endif
constexpr bool kBandPhases = SSYM_BAND_PHASES != 0;
constexpr int kBandImagePad = 8;       // halfs between the two sources' LDS images (16 bytes = 4 banks, see the kernel)

// Records of one source in the banded layout: slot s holds frame s - lead, lead = r.
// WB = waves per workgroup = target groups (of 32) per task; OCC = waves per SIMD the register
// budget is held to.  Up to 3 tiles of diagonals fit two waves per SIMD; wider bands (r = 32 is
// 5 tiles: 81 column registers) run one wave per SIMD with the whole 512-register file.
// LASTN = diagonals of the last tile that get a DP cell: 16, or 1 when 2r+1 = 16(NTB-1) + 1 (every
// radius that is a multiple of 8, r = 32 included) -- the matrix pipe still produces the whole tile,
// but the VALU, which sets the pace, skips the 15 cells that lie outside the band.
// PRUNE (early abandoning, SSYM_DTW_PRUNE, see dtw_filter_kernel.hpp): the whole in-band column of a
// pair is in the lane's registers, every path crosses every column inside the band, so the smallest
// L[k] after column j bounds the pair's cost from below; a wave drops its task when that exceeds the
// target's threshold on all 64 lanes.  Tasks then differ in length by an order of magnitude, so the
// waves of a workgroup take target groups from an LDS counter instead of owning one group per block.
// KU = operand planes a tile multiplies (2 for record layout 3, whose third plane is zero: dtw_filter_kernel.hpp).
// PC (pair columns, LASTN == 1 and no PRUNE only): the 16 source frames tile T reads for column j are the diagonals
// 16T - 1 .. 16T + 14 of column j + 1, so one LDS read of the source operands serves both columns: the steps of a column
// PAIR run (j, T), (j + 1, T), (j, T + 1), ... on the same in-place column registers (column j + 1 trails column j by
// one diagonal, which is all the recurrence needs), with two `up` values.  Half of the source reads go; both columns'
// target operands stay live for the whole pair while the next pair's are in flight (four operand sets instead of two).
// Bit-identical and NOT faster (configs[4]'s shape: 39.78 ms either way; 13 values +0.6 %): the source reads were not
// what the columns waited for.  Instantiated by tools builds only (-DSSYM_BAND_PAIRCOLS_BUILD); LAB.md R4.3.
template <int NTB, int WB, int OCC, bool SQ, int LASTN, bool PRUNE = false, int KU = kFilterKM, bool PC = false>
__global__ __launch_bounds__(64 * WB, OCC) void dtw_band_kernel(
    const _Float16 *__restrict__ srcRec, const _Float16 *__restrict__ tgtRec,
    const int *__restrict__ srcLen, const int *__restrict__ tgtLen, int srcSlots, int radius,
    int tgtFramesPad, int mPad, int nTgtBlocks, int nTasks, unsigned *__restrict__ taskCtr, float outScale,
    float *__restrict__ cmat, const float *__restrict__ abandon = nullptr,
    unsigned long long *__restrict__ colCtr = nullptr, const uint32_t *__restrict__ candSlot = nullptr)
{
    constexpr int REC = kFilterRecHalfs;
    constexpr int KB = (NTB - 1) * 16 + LASTN;   // diagonals held in registers (>= 2r+1)
    const float INF = __builtin_inff();
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsSrc[];   // [srcSlots][48], 16 bytes, [srcSlots][48]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31;
    const int half = lane >> 5;
    const int twoR = 2 * radius;

    // lane's A-operand position inside a tile: source (arow>>2)&1, local frame (arow&3)+4(arow>>3)
    const int arow = lane & 31;
    const int a_h = (arow >> 2) & 1;
    const int a_local = (arow & 3) + 4 * (arow >> 3);
    // The second source's image starts 16 bytes late: records are 96 bytes, so every record starts on a bank that is a
    // multiple of 8 and the eight records per source that one 16-lane group of a ds_read_b128 touches fill the same
    // eight 4-bank slots in both images -- a two-way conflict on every read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    // = 0.50 measured); shifted by 4 banks the two images interleave.
    const _Float16 *const aLane = ldsSrc + ((size_t)a_h * srcSlots + a_local) * REC + half * 24 + a_h * kBandImagePad;

    // Workgroups take SOURCE PAIRS from one counter, longest first (record slots are ordered by segment
    // length, so the list is walked from its end), and sweep all target blocks against the pair while
    // it sits in LDS.  (Handing out single (pair, target block) tasks instead cost 8 % at r = 32: the
    // pair was re-staged for every task.)
    __shared__ unsigned sTask, sGroup;
    const int nPairs = nTasks / nTgtBlocks;
    unsigned colSteps = 0;                             // PRUNE: columns this wave swept
    for (;;) {
        __syncthreads();                               // everyone has read the previous sTask and left LDS
        if (threadIdx.x == 0) {
            sTask = atomicAdd(taskCtr, 1u);
            sGroup = 0;
        }
        __syncthreads();
        const unsigned got = sTask;
        if (got >= (unsigned)nPairs)
            break;
        const int sp = nPairs - 1 - (int)got;
        {
            const uint4 *g = reinterpret_cast<const uint4 *>(srcRec + (size_t)(2 * sp) * srcSlots * REC);
            uint4 *l = reinterpret_cast<uint4 *>(ldsSrc);
            const int n16 = 2 * srcSlots * REC * 2 / 16, per = n16 / 2;
            for (int i = threadIdx.x; i < n16; i += 64 * WB)
                l[i + (i >= per ? kBandImagePad * 2 / 16 : 0)] = g[i];
            __syncthreads();
        }
      for (int tb = nTgtBlocks - 1;; --tb) {
        int tg;
        if (PRUNE || kBandDynamicGroups) {             // next target group of this source pair, longest first
            unsigned g = 0;
            if (lane == 0)
                g = atomicAdd(&sGroup, 1u);
            g = (unsigned)__builtin_amdgcn_readfirstlane((int)g);
            if (g >= (unsigned)(nTgtBlocks * WB))
                break;
            tg = nTgtBlocks * WB - 1 - (int)g;
        } else {
            if (tb < 0)
                break;
            tg = tb * WB + wave;
        }

        const int fa = srcLen[2 * sp + half];
        const int fb_m1 = tgtLen[32 * tg + col] - 1;
        int nCols = fb_m1 + 1;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1)
            nCols = max(nCols, __shfl_xor(nCols, o));
        nCols = __builtin_amdgcn_readfirstlane(nCols);

        // column -1: everything +inf except the virtual D(-1,-1) = 0 on diagonal k = r
        float L[KB + 1];
#pragma unroll
        for (int k = 0; k <= KB; ++k)
            L[k] = (k == radius) ? 0.0f : INF;
        float res = INF;
        const int kstar = fa - 1 - fb_m1 + radius;     // diagonal of the end cell (fa-1, fb-1)
        float thr = INF;
        if (PRUNE)
            thr = abandon[32 * tg + col];
        bool dead = fa == 0 || fb_m1 < 0;              // an empty side: +inf whatever happens
        if (PRUNE)                                     // ... or the target's candidate pair (exact cost known)
            dead = dead || candSlot[32 * tg + col] == (uint32_t)(2 * sp + half);
        bool dropped = false;                          // wave-uniform

        const _Float16 *bbase = tgtRec + tgt_rec_offset(32 * tg + col, tgtFramesPad, 0, 0, half);
        half8 B0[KU], B1[KU];
#pragma unroll
        for (int m = 0; m < KU; ++m)
            B0[m] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (nCols > 0)
            load_tgt_rec(bbase, 0, B0);
        // Software pipeline: while the cells of tile T run on the VALU, the chained MFMAs of the NEXT
        // step (next tile; after the last tile, the first tile of the next column) are issued BETWEEN the cells --
        // issued back to back at the top of a tile each waited for its predecessor and kept the wave off
        // the VALU for ~80 cycles per tile -- and the operands of the step AFTER that are on their way
        // from LDS (An).  Steps run column-major: (j, TLO), ..., (j, THI), (j+1, TLO), ...
        //
        // Tiles TLO..THI of a column: the band's corners lie outside the matrix -- above row 0 in the first columns
        // (tile 0 while j < r - 15), below the longer source's last row in the last ones (the last tile from
        // j >= faMax + r - 16 (NTB-1), the one before it 16 columns later) -- r (r + 1) cells per pair, 6 % at r = 32.
        // They cost nothing to leave out (their diagonals hold +inf either way) as long as no test stands in front of
        // every tile (that variant lost 4.7 %): the column loop runs in up to four PHASES with the tile range a
        // compile-time constant each.  A phase starts on an even column (the operand registers swap by parity), at
        // least one column after its tiles have fallen outside (so that their diagonals already hold the +inf the
        // full kernel computed there), and re-primes the pipeline.
        f32x16 acc;
        half8 An[KU];
        auto run_phase = [&](auto tlo_c, auto thi_c, const int jBegin, const int jEnd) {
            constexpr int TLO = decltype(tlo_c)::value, THI = decltype(thi_c)::value;
            constexpr int NTP = THI - TLO + 1;                 // tiles per column in this phase
            if (jBegin >= jEnd || dropped)
                return;
            {
                half8 A[KU];
                load_rec(aLane + (size_t)jBegin * REC + (size_t)TLO * 16 * REC, A);      // (jBegin, TLO); jBegin is even: B0
                acc = mfma_tile<KU>(A, B0);
                load_rec(aLane + (size_t)(jBegin + 1 / NTP) * REC + (size_t)(TLO + 1 % NTP) * 16 * REC, An);   // the step after it
            }
            for (int j0 = jBegin; j0 < jEnd && !dropped; j0 += 2) {
                if (PRUNE)
                    colSteps += (unsigned)min(2, jEnd - j0);
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int j = j0 + par;
                    if (j < jEnd) {                        // wave-uniform
                        const int jn = min(j + 1, nCols - 1);
#if SSYM_BAND_ABL != 3
                        if (par == 0)
                            load_tgt_rec(bbase, jn, B1);
                        else
                            load_tgt_rec(bbase, jn, B0);
#else
                        (void)jn;
                        if (par == 0)
                            for (int m = 0; m < KU; ++m) B1[m] = B0[m];
#endif
                        // tile T of column j needs source frames j - r + 16T + local = slots j + 16T + local
                        const _Float16 *aCol = aLane + (size_t)j * REC;
                        float up = INF;
#pragma unroll
                        for (int T = TLO; T <= THI; ++T) {
                            // this step's MFMA operands (loaded during the previous tile) ...
                            half8 Ac[KU];
#pragma unroll
                            for (int m = 0; m < KU; ++m)
                                Ac[m] = An[m];
                            // ... and the next one's (slots stay inside the staged window)
#if SSYM_BAND_ABL != 2
                            load_rec(aCol + (size_t)((T - TLO + 2) / NTP) * REC + (size_t)(TLO + (T - TLO + 2) % NTP) * 16 * REC, An);
#else
                            for (int m = 0; m < KU; ++m)
                                asm volatile("" : "+v"(An[m]));                               // opaque: no MFMA is merged
#endif
                            const bool sameCol = T < THI;          // the step being issued belongs to column j
                            f32x16 accn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                            const int nCells = (T == NTB - 1) ? LASTN : 16;
#pragma unroll
                            for (int r = 0; r < (T == NTB - 1 ? LASTN : 16); ++r) {
                                // one MFMA of the chain every few cells (all at once when the tile has one cell)
#pragma unroll
                                for (int m = 0; m < KU; ++m)
                                    if (r == (nCells >= 11 ? 5 * m : 0)) {
                                        const half8 &b = ((par == 0) == sameCol) ? B0[m] : B1[m];
#if SSYM_BAND_ABL == 1
                                        asm volatile("" : "+v"(accn) : "v"(Ac[m]), "v"(b));   // opaque: nothing folds
#else
                                        accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac[m], b, accn, 0, 0, 0);
#endif
                                    }
                                const int k = T * 16 + r;
                                const float x = acc[r];
                                float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
                                if (T == NTB - 1 && LASTN > 1)
                                    c = (k <= twoR) ? c : INF;       // diagonals beyond the band (wave-uniform)
                                const float m3 = __builtin_fminf(__builtin_fminf(up, L[k]), L[k + 1]);
                                const float cur = c + m3;
                                L[k] = cur;
                                up = cur;
                            }
                            acc = accn;
                            // keep the scheduler from hoisting every tile's LDS reads and MFMA chains to
                            // the top of the column (5 accumulators + 15 operand quads live = spills)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (__any(j == fb_m1)) {           // the end cell of some lane's pair is in this column
                            const bool mine = (j == fb_m1);
#pragma unroll
                            for (int k = 0; k < KB; ++k)
                                res = (mine && k == kstar) ? L[k] : res;
                        }
                        if (PRUNE && par == 1 && (j & (kPruneEvery - 1)) == kPruneEvery - 1) {
                            float lb = L[0];
#pragma unroll
                            for (int k = 1; k + 1 < KB; k += 2)
                                lb = __builtin_fminf(__builtin_fminf(lb, L[k]), L[k + 1]);
                            if ((KB & 1) == 0)
                                lb = __builtin_fminf(lb, L[KB - 1]);
                            dropped = __all(!(lb <= thr) || j >= fb_m1 || dead);
                        }
                    }
                }
            }
        };
        // PC: one pass of the same phases, two columns per source read
        auto run_pairs = [&](auto tlo_c, auto thi_c, const int jBegin, const int jEnd) {
            constexpr int TLO = decltype(tlo_c)::value, THI = decltype(thi_c)::value;
            if (jBegin >= jEnd)
                return;
            half8 Ac[KU];                                      // the source operands of the tile in hand
            load_rec(aLane + (size_t)jBegin * REC + (size_t)TLO * 16 * REC, Ac);
            acc = mfma_tile<KU>(Ac, B0);                       // (jBegin, TLO); B0 / B1 hold columns jBegin, jBegin + 1
            half8 Bx[KU], By[KU];
            // columns j (operands P0) and j + 1 (P1); the next pair's operands travel to N0 / N1 meanwhile
            auto pair = [&](const int j, const half8 (&P0)[KU], const half8 (&P1)[KU], half8 (&N0)[KU], half8 (&N1)[KU]) {
                load_tgt_rec(bbase, min(j + 2, nCols - 1), N0);
                load_tgt_rec(bbase, min(j + 3, nCols - 1), N1);
                const _Float16 *aCol = aLane + (size_t)j * REC;
                float up0 = INF, up1 = INF;
                const bool mine0 = j == fb_m1, mine1 = j + 1 == fb_m1;
                const bool any0 = __any(mine0), any1 = __any(mine1);       // some lane's end cell lies in the column
#pragma unroll
                for (int T = TLO; T <= THI; ++T) {
                    // the next tile's source operands (the first tile of the next pair after the last one)
                    half8 An[KU];
                    load_rec(T < THI ? aCol + (size_t)(T + 1) * 16 * REC : aCol + 2 * REC + (size_t)TLO * 16 * REC, An);
                    // step (j, T): cells of column j, the chain of (j + 1, T) between them
                    {
                        f32x16 accn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        const int nCells = (T == NTB - 1) ? LASTN : 16;
#pragma unroll
                        for (int r = 0; r < nCells; ++r) {
#pragma unroll
                            for (int m = 0; m < KU; ++m)
                                if (r == (nCells >= 11 ? 5 * m : 0))
                                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac[m], P1[m], accn, 0, 0, 0);
                            const int k = T * 16 + r;
                            const float x = acc[r];
                            const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
                            const float m3 = __builtin_fminf(__builtin_fminf(up0, L[k]), L[k + 1]);
                            const float cur = c + m3;
                            L[k] = cur;
                            up0 = cur;
                        }
                        if (any0) {                            // (before column j + 1 overwrites these diagonals)
#pragma unroll
                            for (int r = 0; r < nCells; ++r)
                                res = (mine0 && T * 16 + r == kstar) ? L[T * 16 + r] : res;
                        }
                        acc = accn;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // step (j + 1, T): the same frames are diagonals 16T - 1 .. 16T + 14 of column j + 1; the chain of
                    // (j, T + 1) -- of (j + 2, TLO) after the last tile -- between them
                    {
                        f32x16 accn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        const int rLo = (T == 0) ? 1 : 0, rHi = (T == NTB - 1) ? LASTN + 1 : 16, nCells = rHi - rLo;
#pragma unroll
                        for (int r = rLo; r < rHi; ++r) {
#pragma unroll
                            for (int m = 0; m < KU; ++m)
                                if (r - rLo == (nCells >= 11 ? 5 * m : 0))
                                    accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(An[m], T < THI ? P0[m] : N0[m], accn, 0, 0, 0);
                            const int k = T * 16 + r - 1;
                            const float x = acc[r];
                            const float c = SQ ? __builtin_fabsf(x) : __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
                            const float m3 = __builtin_fminf(__builtin_fminf(up1, L[k]), L[k + 1]);
                            const float cur = c + m3;
                            L[k] = cur;
                            up1 = cur;
                        }
                        // a phase without the last tile(s): the next diagonal of column j + 1 is the first frame of the
                        // tile left out, beyond the longer source's end
                        if constexpr (THI < NTB - 1) {
                            if (T == THI)
                                L[THI * 16 + 15] = INF;
                        }
                        if (any1) {
#pragma unroll
                            for (int r = rLo; r < rHi; ++r)
                                res = (mine1 && T * 16 + r - 1 == kstar) ? L[T * 16 + r - 1] : res;
                        }
#pragma unroll
                        for (int m = 0; m < KU; ++m)
                            Ac[m] = An[m];
                        acc = accn;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            for (int j0 = jBegin; j0 < jEnd; j0 += 4) {
                pair(j0, B0, B1, Bx, By);
                if (j0 + 2 < jEnd) {
                    pair(j0 + 2, Bx, By, B0, B1);
                } else {
#pragma unroll
                    for (int m = 0; m < KU; ++m) {
                        B0[m] = Bx[m];
                        B1[m] = By[m];
                    }
                }
            }
        };
        if constexpr (PC) {
            static_assert(!PC || (LASTN == 1 && !PRUNE && NTB >= 4 && kBandPhases), "pair columns: unpruned, 2r + 1 = 16 (NTB - 1) + 1");
            if (nCols > 1)
                load_tgt_rec(bbase, 1, B1);
            else
                for (int m = 0; m < KU; ++m)
                    B1[m] = B0[m];
            // an odd last column computes one column too many (in place, after every end cell has been read): it ends
            // the task, because every phase boundary but nCols is even
            const int faMax = max(__shfl(fa, 0), __shfl(fa, 32));
            const int head = min(max(radius - 15, 0) & ~1, nCols & ~1);
            const int out1 = faMax + radius - 16 * (NTB - 1), out2 = out1 + 16;
            const int tail1 = min(max((max(out1, 0) + 2) & ~1, head), (nCols + 1) & ~1);
            const int tail2 = min(max((max(out2, 0) + 2) & ~1, tail1), (nCols + 1) & ~1);
            run_pairs(std::integral_constant<int, 1>{}, std::integral_constant<int, NTB - 1>{}, 0, min(head, nCols));
            run_pairs(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 1>{}, head, min(tail1, nCols));
            run_pairs(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 2>{}, tail1, min(tail2, nCols));
            run_pairs(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 3>{}, tail2, nCols);
        } else if constexpr (kBandPhases && NTB >= 4 && !PRUNE) {     // (pruned tasks rarely reach their last columns; half the compile time)
            // (wave-uniform bounds; every phase boundary is even and inside [0, nCols])
            const int faMax = max(__shfl(fa, 0), __shfl(fa, 32));
            const int head = min(max(radius - 15, 0) & ~1, nCols & ~1);
            const int out1 = faMax + radius - 16 * (NTB - 1), out2 = out1 + 16;     // first columns with the last tile(s) outside
            const int tail1 = min(max((max(out1, 0) + 2) & ~1, head), (nCols + 1) & ~1);
            const int tail2 = min(max((max(out2, 0) + 2) & ~1, tail1), (nCols + 1) & ~1);
            run_phase(std::integral_constant<int, 1>{}, std::integral_constant<int, NTB - 1>{}, 0, min(head, nCols));
            run_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 1>{}, head, min(tail1, nCols));
            run_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 2>{}, tail1, min(tail2, nCols));
            run_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 3>{}, tail2, nCols);
        } else {
            run_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, NTB - 1>{}, 0, nCols);
        }
        cmat[(size_t)(2 * sp + half) * mPad + 32 * tg + col] = res * outScale;
      }
    }
    if (PRUNE && colCtr && lane == 0)
        atomicAdd(colCtr, (unsigned long long)colSteps * KB);
}

}  // namespace ssym
