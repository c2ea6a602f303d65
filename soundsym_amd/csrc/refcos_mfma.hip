// refcos_mfma.hip -- the reference's own search (cosine_sim + at_distance) through the f64 matrix pipe,
// results still bit for bit the reference's.
//
// Role on the path: SoundDictionary::at_distance (src/sound.rs:351-370) maps every dictionary entry to
// |cosine_sim(entry, other) - distance| (src/sound.rs:22-33, 359) and folds to the first minimum.  N x M
// prefix dot products are one zero-padded GEMM: a segment's values beyond its length are
// staged as zeros, so the product over the padded length equals rulinalg's dot over the common prefix.
//
//   1. refcos_mfma_kernel    every pair's dot on v_mfma_f64_16x16x4_f64 (FMA chains, another summation
//                            order than the reference's -- so only a FILTER), turned at once into a rigorous
//                            interval [key_lo, key_hi] for the reference's key |dot/nrm - distance|; the
//                            smallest key_hi per target is the target's threshold, pairs whose key_lo does
//                            not exceed the threshold seen so far are listed (list 1)
//   2. refcos_keep_kernel    list 1 against the FINAL thresholds -> candidates
//   3. refcos_pairs_kernel   the candidates' keys in the reference's own operation order (eight running sums,
//                            separately rounded products and sums, the same code shape as refcos.hip)
//   4. refcos_fold_*         first minimum among the candidates in index order, strict '<' from (0, 2.0)
//
// Interval (u = 2^-53, L = common prefix length, P = sum |a_i b_i| <= sqrt(norm(me) norm(you)) by
// Cauchy-Schwarz, norm = the reference's sum of squares):
//   any floating-point evaluation of the L-term dot, products rounded or fused, sums in any order, stays
//   within gamma_n P of the exact dot with n = number of roundings on a path <= 2 L + 8; the reference's
//   value d_ref and the matrix pipe's d_m therefore differ by at most E = (3 L + 16) u * 1.02 * sqrt(na nb).
//   The reference's key is k = |fl(fl(d_ref / nrm) - dist)| with nrm = fl(na nb) (the norms are computed in its
//   order at pack time); each of its two roundings moves the value by at most u times its magnitude, so with
//   x = d_ref / nrm:  | k - |x - dist| | <= (2 |x| + |dist|) u (1 + u).  The kernel evaluates
//   s = fl(d_m * fl(ia * ib)) with ia = fl(1 / na), ib = fl(1 / nb), which is d_m / nrm up to 5.2 u, and
//   z = fl(s - dist); collecting the terms,
//       | k - |z| |  <=  R := 1.0001 E ia ib + 9 u (|s| + |dist|) + 1e-290
//   (the last term covers the subnormal range, where relative bounds stop holding), so
//       key_lo = (|z| - R)(1 - 4u) or 0,   key_hi = (|z| + R)(1 + 4u).
//   nrm = 0 or NaN: the reference's key is NaN or +inf and never wins (src/sound.rs:362) -- the pair is dropped;
//   anything else that is not finite, or norms whose product leaves 1e-280 .. 1e280 (1 / nrm or single products
//   would leave the normal range, where these relative bounds hold): the pair is kept with [0, +inf].
// The true first minimum w has key(w) <= key(s) <= key_hi(s) for all s, so key_lo(w) <= threshold: it is
// among the candidates, and the fold over exact keys in index order returns what the reference returns.
#include "ssym_internal.hpp"
#include "refcos_filter.hpp"
#include "ssym_rulinalg.h"

#include <algorithm>
#include <cmath>
#include <type_traits>

namespace ssym {

namespace {

constexpr int kMT = 128;            // sources per workgroup tile
constexpr int kNT = 128;            // targets per workgroup tile
constexpr int kKC = 16;             // elements per staged chunk (one 128-byte line per segment)
constexpr int kLdk = kKC;           // LDS row stride in doubles: rows are 128 bytes, their 16-byte pieces XOR-swizzled (below)
typedef double double4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

// Epilogue shared by the two main loops: dots -> key intervals -> thresholds and list 1.  `info` is LDS no wave reads
// any more (the caller has passed its last barrier); sLen holds the tile's segment lengths.
// TOPK (ssym_match_topk): the threshold a column offers is not its smallest key_hi but the kTop-th smallest DISTINCT
// one among the wave's 64 rows (kTop rounds of "smallest value above the previous round's", +inf when the rows hold
// fewer): at least kTop pairs then have their exact key below it, so the kTop-th smallest exact key of the target does
// too, and every member of the exact top kTop has key_lo <= key <= threshold -- the same argument as for dtw
// (select.hip), with the smallest such bound over all waves and tiles as the target's threshold.
template <bool WRITE_SIMS, bool TOPK>
__device__ __forceinline__ void refcos_epilogue(double4v (&acc)[4][4], RowInfo *info, const unsigned *sLen, int tid, int lane,
                                                int wm, int wn, int lr, int lg, uint32_t sTile, uint32_t tTile, uint32_t nSrc,
                                                uint32_t nTgt, const double *__restrict__ srcNorm,
                                                const double *__restrict__ tgtNorm, const double *__restrict__ dist,
                                                double defaultDist, unsigned long long *__restrict__ thr,
                                                uint32_t *__restrict__ hdr, PairEntry *__restrict__ list, uint32_t cap,
                                                double *__restrict__ dotOut, uint32_t kTop, bool plain)
{
    // ---- epilogue: dots -> key intervals -> thresholds and list 1 ------------------------------------------
    // D[4 i + lane / 16][lane % 16]: this lane holds, per block pair (a, b), source rows 4 i + lg and target column lr.
    // `info` (per-segment values of the tile) was written before the main loop: no load and no barrier stands between
    // the last MFMA and the first interval.  The four column blocks' thresholds travel as four atomics issued back to
    // back (their return values are the freshest thresholds there are) and the room in list 1 is reserved once per wave:
    // two memory round trips per epilogue instead of nine (round 3).
    const double INF = __builtin_inf();
    const double cUnit = 1.1102230246251565e-16 * 1.02;
    // (segments beyond the sets' ends carry norm 0 in `info`: their pairs come out as [+inf, +inf] without a test)
    // Pass 1, row-major: a row's values are fetched once for its four columns (the columns' sit in registers), every dot
    // becomes its key interval, key_lo takes the dot's place in the accumulator registers (the dot is not needed
    // again) and key_hi goes into the column's running minimum -- no second set of 64 values, no spills.
    RowInfo ci[4];
    unsigned lb[4];
    double colMin[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        ci[b] = info[kMT + wn * 64 + b * 16 + lr];
        lb[b] = sLen[kMT + wn * 64 + b * 16 + lr];
        colMin[b] = INF;
    }
    double khis[TOPK ? 4 : 1][TOPK ? 4 : 1][TOPK ? 4 : 1];
    if (!WRITE_SIMS && plain) {
        // Every segment of this wave's 64 rows and 64 columns is PLAIN (inside the sets, norm in [1e-139, 1e139], hence
        // all its values finite; a finite distance): none of refcos_key_interval's tests can fire -- nrm and inv are
        // products of two factors within 1e+-139, every dot is finite -- and what is left is the arithmetic, with the
        // factors that belong to a row or a column alone taken out of the pair (rq = sq * inv, the length factor:
        // (3 min(la, lb) + 16) = min over the two; the roundings this moves are paid from the 1.0001 * 1.02 of slack
        // the bound carries for exactly that).  16 operations per pair instead of ~45 with their selects.
        const double u = 1.1102230246251565e-16;
        double cq[4], cLb[4], dabs9[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            cq[b] = ci[b].sq * ci[b].inv;
            cLb[b] = 1.0001 * ((3.0 * (double)lb[b] + 16.0) * cUnit);
            dabs9[b] = fabs(ci[b].dist);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + a * 16 + 4 * i + lg;
                const RowInfo ri = info[row];
                const double rq = ri.sq * ri.inv;
                const double cLa = 1.0001 * ((3.0 * (double)sLen[row] + 16.0) * cUnit);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double sv = acc[a][b][i] * (ri.inv * ci[b].inv);
                    const double z = fabs(sv - ci[b].dist);
                    const double R = __fma_rn(9.0 * u, fabs(sv) + dabs9[b], fmin(cLa, cLb[b]) * (rq * cq[b])) + 1e-290;
                    acc[a][b][i] = fmax((z - R) * (1.0 - 4.0 * u), 0.0);
                    const double khi = (z + R) * (1.0 + 4.0 * u);
                    if (TOPK)
                        khis[b][a][i] = khi;
                    else
                        colMin[b] = fmin(colMin[b], khi);
                }
            }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + a * 16 + 4 * i + lg;
                const RowInfo ri = info[row];
                const unsigned la = sLen[row];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned len = la < lb[b] ? la : lb[b];               // src/sound.rs:24-28
                    const double nrm = __dmul_rn(ri.norm, ci[b].norm);          // src/sound.rs:30
                    double klo, khi;
                    refcos_key_interval(acc[a][b][i], ri.sq * ci[b].sq, ri.inv * ci[b].inv, nrm,
                                        (3.0 * (double)len + 16.0) * cUnit, ci[b].dist, klo, khi);
                    if (WRITE_SIMS) {
                        const uint32_t s = sTile + row, t = tTile + wn * 64 + b * 16 + lr;
                        if (s < nSrc && t < nTgt)
                            dotOut[(size_t)s * nTgt + t] = __ddiv_rn(acc[a][b][i], nrm);
                    }
                    acc[a][b][i] = klo;
                    if (TOPK)
                        khis[b][a][i] = khi;
                    else
                        colMin[b] = fmin(colMin[b], khi);
                }
            }
    }
    double cur[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double cmin = colMin[b];
        // smallest key_hi of the wave's 64 rows in this column (TOPK: the kTop-th smallest distinct one), then against
        // the threshold every tile works on
        if (TOPK) {
            double prev = -1.0;                                           // (keys are >= 0)
            for (uint32_t r = 0; r < kTop; ++r) {
                double m = INF;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        m = (khis[b][a][i] > prev && khis[b][a][i] < m) ? khis[b][a][i] : m;
                m = fmin(m, __shfl_xor(m, 16));
                m = fmin(m, __shfl_xor(m, 32));
                prev = m;                                                 // +inf once the rows are used up: it stays
            }
            cmin = prev;
        } else {
            cmin = fmin(cmin, __shfl_xor(cmin, 16));
            cmin = fmin(cmin, __shfl_xor(cmin, 32));
        }
        cur[b] = cmin;
    }
    unsigned long long seenBits[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t t = tTile + wn * 64 + b * 16 + lr;
        seenBits[b] = kInfBitsU;
        if (lg == 0 && t < nTgt)
            seenBits[b] = atomicMin(&thr[t], (unsigned long long)__double_as_longlong(cur[b]));
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        seenBits[b] = __shfl(seenBits[b], lr);
        // list 1: pairs the threshold known so far does not exclude (the final threshold can only be smaller); capped at
        // the largest finite value, so that "key_lo <= cur" also says "key_lo is finite" (+inf: a pair that cannot win)
        cur[b] = fmin(fmin(cur[b], __longlong_as_double((long long)seenBits[b])), 1.7976931348623157e308);
    }
    // Count, reserve, write: the count is a sum of wave-wide ballots (scalar), ONE atomic reserves the room, and the write
    // pass skips a register slot none of the 64 lanes has an entry in with a scalar branch; a lane's position inside a
    // slot is the number of entries below its lane (mbcnt) -- no prefix sum, no per-entry branch on the common path.
    unsigned total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                total += (unsigned)__popcll(__ballot(acc[a][b][i] <= cur[b]));
    if (total) {                                                          // wave-uniform
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(&hdr[0], total);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = acc[a][b][i] <= cur[b];
                    const unsigned long long m = __ballot(in);
                    if (m) {                                              // wave-uniform
                        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (in) {
                            if (pos < cap) {
                                if constexpr (TOPK) {
                                    PairEntryK e;
                                    e.s = sTile + wm * 64 + a * 16 + 4 * i + lg;
                                    e.t = tTile + wn * 64 + b * 16 + lr;
                                    e.key_lo = acc[a][b][i];
                                    e.key_hi = khis[b][a][i];
                                    reinterpret_cast<PairEntryK *>(list)[pos] = e;
                                } else {
                                    PairEntry e;
                                    e.s = sTile + wm * 64 + a * 16 + 4 * i + lg;
                                    e.t = tTile + wn * 64 + b * 16 + lr;
                                    e.key_lo = acc[a][b][i];
                                    list[pos] = e;
                                }
                            } else {
                                hdr[1] = 1;
                            }
                        }
                        base += (unsigned)__popcll(m);
                    }
                }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 128 x 128 pairs per workgroup, four waves of 64 x 64, K in chunks of 16 elements staged through LDS
// (zero beyond each segment's length).  MFMA operand layout (gfx950 v_mfma_f64_16x16x4_f64): A[m][k] in lane
// m + 16 k, B[k][n] in lane n + 16 k, D[4 i + lane / 16][lane % 16] in register pair i (measured: not the f32 forms' 4 (lane / 16) + i).  The summation index
// is free to permute: lane group g = lane / 16 takes elements 4 g .. 4 g + 3 of a chunk, one per MFMA step, so
// a lane's four A (or B) values of a chunk are 32 contiguous bytes of LDS -- two ds_read_b128.
template <bool WRITE_SIMS, bool TOPK>
__global__ __launch_bounds__(256, 2) void refcos_mfma_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, const double *__restrict__ srcNorm,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, const double *__restrict__ tgtNorm,
    uint32_t nSrc, uint32_t nTgt, uint32_t dim, unsigned long long srcVals, unsigned long long tgtVals,
    const double *__restrict__ dist, double defaultDist, unsigned long long *__restrict__ thr /* [nTgt] smallest key_hi so far (bits) */,
    uint32_t *__restrict__ hdr /* {count, overflow} */, PairEntry *__restrict__ list, uint32_t cap,
    double *__restrict__ dotOut /* nullable: [nSrc][nTgt] sims from the matrix pipe's dots (ssym_pair_matrix, exact = 2) */,
    uint32_t kTop /* TOPK: entries wanted per target */, const double *__restrict__ zeros /* >= 16 bytes of 0.0 */)
{
    // four DISTINCT objects, addressed with compile-time buffer numbers: the compiler orders a DMA into LDS against every
    // LDS read it cannot prove disjoint (s_waitcnt vmcnt(0) in front of the operand reads -- with one two-buffer array
    // the DMA of chunk c + 1 was waited for before chunk c's first MFMA)
    __shared__ __attribute__((aligned(16))) double sA0[kMT * kLdk], sA1[kMT * kLdk];
    __shared__ __attribute__((aligned(16))) double sB0[kNT * kLdk], sB1[kNT * kLdk];
    __shared__ unsigned long long sBase[kMT + kNT];
    __shared__ unsigned sLen[kMT + kNT];
    __shared__ unsigned sMaxLen[2], sMinLen[2], sOdd;
    __shared__ unsigned long long sEnd[2];                 // one past the last value of the tile's segments, per side
    __shared__ unsigned sPlain[4];                         // rows 0..63, 64..127, columns 0..63, 64..127: all plain?
    __shared__ RowInfo sInfo[kMT + kNT];                   // the epilogue's per-segment values, fetched up front

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: LDS bases of the DMAs)
    const int wm = wave >> 1, wn = wave & 1;              // this wave's 64 x 64 quadrant
    // Tile of this workgroup.  Workgroups go to the eight XCDs round robin (id % 8), each XCD with its own L2: XCD k takes
    // the k-th eighth of the tiles in an order that walks panels of eight tile columns row by row, so the 64 workgroups
    // an XCD runs at a time cover 8 x 8 tiles -- 16 operand stripes for 64 tiles (identity order: 4 + 16 for 64).
    uint32_t bx = blockIdx.x, by = blockIdx.y;
#ifndef SSYM_RM_NOXCD
    {
        const uint32_t nx = gridDim.x, ny = gridDim.y, total = nx * ny;
        if ((nx & 7u) == 0 && (total & 7u) == 0) {
            const uint32_t lin = bx + nx * by;
            const uint32_t q = (lin & 7u) * (total >> 3) + (lin >> 3);
            const uint32_t r = q % (8u * ny);
            bx = 8u * (q / (8u * ny)) + (r & 7u);
            by = r >> 3;
        }
    }
#endif
    const uint32_t sTile = by * kMT, tTile = bx * kNT;

    if (tid < 2) {
        sMaxLen[tid] = 0;
        sMinLen[tid] = 0xffffffffu;
        sEnd[tid] = 0;
    }
    if (tid == 2)
        sOdd = 0;
    if (tid < 4)
        sPlain[tid] = 1;
    __syncthreads();
    {
        const bool isS = tid < kMT;
        const uint32_t g = isS ? sTile + tid : tTile + (tid - kMT);
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
        atomicMax(&sMaxLen[isS ? 0 : 1], len);
        atomicMin(&sMinLen[isS ? 0 : 1], len);
        atomicMax(&sEnd[isS ? 0 : 1], base + len);
        if (len & 1)
            atomicOr(&sOdd, 1u);
        // (segments beyond the sets' ends carry norm 0: their pairs come out as [+inf, +inf] without a test)
        const double *nr = isS ? srcNorm : tgtNorm;
        RowInfo r;
        r.norm = g < n ? nr[g] : 0.0;
        r.sq = g < n ? nr[n + g] : 0.0;
        r.inv = g < n ? nr[2 * (size_t)n + g] : 0.0;
        r.dist = (!isS && dist && g < n) ? dist[g] : defaultDist;
        sInfo[tid] = r;
        // plain: what the epilogue's short form may assume of a segment (refcos_epilogue)
        const bool ok = g < n && r.norm >= 1e-139 && r.norm <= 1e139 && fabs(r.dist) <= 1e300;
        if (!ok)
            atomicAnd(&sPlain[tid >> 6], 0u);
    }
    __syncthreads();
    const unsigned kMax = min(sMaxLen[0], sMaxLen[1]);    // beyond it every product of the tile is zero
    const unsigned nChunks = (kMax + kKC - 1) / kKC;

    // staging: thread -> (row = tid / 8 + 32 p, one 16-byte piece of the row's 128-byte chunk) for p = 0..3, both sides.
    // The rows of a tile are consecutive segments, i.e. one contiguous stretch of the value buffer: a row's start is
    // a 32-bit offset from the tile's first value, kept in registers with the row's length, so that the eight loads
    // of a chunk issue back to back (with the lengths and starts looked up in LDS in front of every load, and a branch
    // around it, the fetch phase cost ~40 % of a chunk's MFMA time).
    // Global -> LDS by DMA (global_load_lds, 16 bytes per lane): no staging registers, no selects, no LDS stores (round 2
    // staged through registers and zeroed there: 1.5 % slower).
    // A wave's DMA lands as 1 KB of consecutive lanes, i.e. 8 rows x 8 pieces of 16 bytes with a row stride of 128
    // bytes -- which would be an 8-way bank conflict for the MFMA operand reads (16 lanes read one piece of 16
    // consecutive rows).  So the 8 pieces of a row are stored XOR-swizzled: position q of row r holds piece
    // q ^ ((r >> 1) & 7), chosen on the LOAD side (the lane at position q fetches that piece), undone on the read side;
    // rows 2k and 2k + 1 share a swizzle and differ in the upper half of the 64 banks, rows of different k differ in the
    // position: the 16 rows of an operand read cover all 64 banks once.  What lies beyond a segment's end is read from
    // 16 bytes of zeros instead; the one piece that straddles the end of a segment of odd length is corrected in LDS
    // after it has landed.
    const int sr = tid >> 3;                               // row of sweep 0 (32 rows per sweep, 4 sweeps)
    const int se = 2 * ((tid & 7) ^ ((sr >> 1) & 7));      // the two elements of the piece this lane's position holds
    auto uniform64 = [](unsigned long long v) {            // a value every lane holds, moved to scalar registers
        return (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)v) |
               ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32);
    };
    const unsigned long long tbA = uniform64(min(sBase[0], srcVals - 1)), tbB = uniform64(min(sBase[kMT], tgtVals - 1));
    const double *const tileA = srcRaw + tbA, *const tileB = tgtRaw + tbB;
    // The SHORT form of a chunk's fetch (round 3): while a chunk lies inside EVERY segment of the tile -- all of them on
    // equal-length sets, all but the tail on ragged ones -- nothing has to be selected per lane, and a row's address is
    // (tile base + 128 bytes per chunk) in scalar registers plus a 32-bit byte offset per lane that never changes: the
    // eight DMAs of a chunk cost no vector instruction at all.  (Vector instructions issued beside the f64 MFMAs are
    // what the staging cost: ~130 per two chunks took 14 % of the kernel with the barrier and the misses free, 5.5.)
    const unsigned long long spanA = uniform64(sEnd[0]) - tbA, spanB = uniform64(sEnd[1]) - tbB;
    const bool spanOk = sEnd[0] >= tbA && sEnd[1] >= tbB && spanA < (1ull << 29) && spanB < (1ull << 29);
    const unsigned fastLen = spanOk ? (unsigned)__builtin_amdgcn_readfirstlane(min(sMinLen[0], sMinLen[1])) : 0u;
    const bool tileOdd = __builtin_amdgcn_readfirstlane(sOdd) != 0;
    unsigned relA[4], relB[4], lenA[4], lenB[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = sr + 32 * p;
        lenA[p] = sLen[row];
        lenB[p] = sLen[kMT + row];
        relA[p] = lenA[p] ? (unsigned)(sBase[row] - tbA) : 0u;
        relB[p] = lenB[p] ? (unsigned)(sBase[kMT + row] - tbB) : 0u;
    }
    unsigned offA[4], offB[4];                             // byte offsets of this lane's pieces from the tile's chunk base
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        offA[p] = (relA[p] + (unsigned)se) * 8u;
        offB[p] = (relB[p] + (unsigned)se) * 8u;
    }
    auto fetch = [&](unsigned c, auto BUF) {               // DMAs only: nothing here waits for them
        double *const dA = decltype(BUF)::value ? sA1 : sA0, *const dB = decltype(BUF)::value ? sB1 : sB0;
        if ((c + 1) * kKC <= fastLen) {                    // (scalar)
            const char *ua = (const char *)(tileA + (size_t)c * kKC), *ub = (const char *)(tileB + (size_t)c * kKC);
            // (opaque to the optimiser: it would otherwise fold the chunk's base into 64-bit vector additions per row
            // instead of the scalar-base + 32-bit-offset form of the instruction)
            asm volatile("" : "+s"(ua), "+s"(ub));
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                asm volatile("" : "+v"(offA[p]), "+v"(offB[p]));      // (the zero-extension stays here, inside the instruction)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ua + offA[p]),
                                                 (__attribute__((address_space(3))) void *)&dA[(32 * p + 8 * wave) * kLdk], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + offB[p]),
                                                 (__attribute__((address_space(3))) void *)&dB[(32 * p + 8 * wave) * kLdk], 16, 0, 0);
            }
            return;
        }
        const unsigned stE = c * kKC + se;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double *ga = stE < lenA[p] ? tileA + relA[p] + stE : zeros;
            const double *gb = stE < lenB[p] ? tileB + relB[p] + stE : zeros;
            // (LDS side: the wave's 8 rows of this sweep, lanes in order; M0 carries the wave-uniform base)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ga,
                                             (__attribute__((address_space(3))) void *)&dA[(32 * p + 8 * wave) * kLdk], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gb,
                                             (__attribute__((address_space(3))) void *)&dB[(32 * p + 8 * wave) * kLdk], 16, 0, 0);
        }
    };
    auto stash = [&](unsigned c, auto BUF) {               // once the DMAs have landed: the straddling pieces
        double *const dA = decltype(BUF)::value ? sA1 : sA0, *const dB = decltype(BUF)::value ? sB1 : sB0;
#ifndef SSYM_RM_NOWAIT    // (tools only: what does waiting for the DMAs cost?)
        __builtin_amdgcn_s_waitcnt(rm_wait_vmcnt(0));
#endif
        asm volatile("" ::: "memory");
        if (!tileOdd)                                      // (scalar: no segment of the tile has an odd length)
            return;
        const unsigned stE = c * kKC + se;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = sr + 32 * p;
            if (stE + 1 == lenA[p])
                dA[row * kLdk + 2 * (tid & 7) + 1] = 0.0;
            if (stE + 1 == lenB[p])
                dB[row * kLdk + 2 * (tid & 7) + 1] = 0.0;
        }
    };

    double4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[a][b] = double4v{0.0, 0.0, 0.0, 0.0};

    const int lr = lane & 15, lg = lane >> 4;
    const int swz = (lr >> 1) & 7;                         // the swizzle of this lane's operand rows (16 blk + lr, any blk)
    // one chunk: the other buffer is filled while this one is multiplied
    auto chunk = [&](unsigned c, auto BUF) {
        constexpr bool kOdd = decltype(BUF)::value;
        using Other = std::integral_constant<bool, !kOdd>;
        const double *const rA = kOdd ? sA1 : sA0, *const rB = kOdd ? sB1 : sB0;
#if !defined(SSYM_RM_NOFETCH) && !defined(SSYM_RM_NODMA)   // (tools only: without it the MFMAs run on the first chunk over and over)
        if (c + 1 < nChunks)
            fetch(c + 1, Other{});                         // in flight under the MFMAs (DMA: straight into the other buffer,
                                                           // which nobody has read since the barrier that ended chunk c - 1)
#endif
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                   // two MFMA steps per 16-byte LDS read
            double av[4][2], bv[4][2];                     // [block][step]
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int pc = 2 * ((2 * lg + hf) ^ swz);
                const double2v a2 = *reinterpret_cast<const double2v *>(&rA[(wm * 64 + blk * 16 + lr) * kLdk + pc]);
                const double2v b2 = *reinterpret_cast<const double2v *>(&rB[(wn * 64 + blk * 16 + lr) * kLdk + pc]);
                av[blk][0] = a2[0]; av[blk][1] = a2[1];
                bv[blk][0] = b2[0]; bv[blk][1] = b2[1];
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][st], bv[b][st], acc[a][b], 0, 0, 0);
#ifndef SSYM_RM_NOFETCH
#endif
        }
#if !defined(SSYM_RM_NOFETCH) && !defined(SSYM_RM_NODMA)
        if (c + 1 < nChunks)
            stash(c + 1, Other{});                         // the DMAs were issued 64 MFMAs ago: this wait is a formality
#endif
#if !defined(SSYM_RM_NOFETCH) && !defined(SSYM_RM_NOBAR)   // (tools only: timing without the barrier / without the DMAs)
        __syncthreads();
#endif
    };
    if (nChunks > 0) {
        fetch(0, std::false_type{});
        stash(0, std::false_type{});
    }
    __syncthreads();
    for (unsigned c = 0; c < nChunks; c += 2) {
        chunk(c, std::false_type{});
        if (c + 1 < nChunks)
            chunk(c + 1, std::true_type{});
    }

#ifdef SSYM_RM_NOEPI      // tools only: the main loop alone, every accumulator kept alive (results meaningless)
    {
        double4v t = acc[0][0];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                t += acc[a][b];
        if (t[0] + t[1] + t[2] + t[3] == 12345.678 && hdr[0] == 77)
            thr[0] = 1;
        return;
    }
#endif
    refcos_epilogue<WRITE_SIMS, TOPK>(acc, sInfo, sLen, tid, lane, wm, wn, lr, lg, sTile, tTile,
                                      nSrc, nTgt, srcNorm, tgtNorm, dist, defaultDist, thr, hdr, list, cap, dotOut, kTop,
                                      (sPlain[wm] & sPlain[2 + wn]) != 0);
}

// list 1 against the final thresholds -> list 2 (pairs only)
__global__ void refcos_keep_kernel(const uint32_t *__restrict__ hdr1, const PairEntry *__restrict__ list1, uint32_t cap,
                                   const unsigned long long *__restrict__ thr, uint32_t *__restrict__ hdr2,
                                   uint2 *__restrict__ pairs, unsigned long long *stamps)
{
    const uint32_t n = min(hdr1[0], cap);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && stamps)
        stamps[1] = (unsigned long long)wall_clock64();    // the main kernel is done
    bool keep = false;
    PairEntry e{};
    if (i < n) {
        e = list1[i];
        keep = e.key_lo <= __longlong_as_double((long long)thr[e.t]);
    }
    const unsigned long long mask = __ballot(keep);
    if (!mask)
        return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (int)__builtin_ctzll(mask))
        base = atomicAdd(&hdr2[0], (uint32_t)__popcll(mask));
    base = __shfl(base, (int)__builtin_ctzll(mask));
    if (keep)
        pairs[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = make_uint2(e.s, e.t);
}

// ---- top-k searches: list 1 -> per-target thresholds from ALL of a target's listed pairs -> candidates ----------------------
// The threshold a wave offers a column is the k-th smallest upper bound among ITS 64 rows; every pair that can be among a
// target's k smallest keys is in list 1 under the smallest such offer, and so are at least k pairs whose upper bound does
// not exceed it.  Hence the k-th smallest DISTINCT key_hi over a target's listed pairs is the k-th smallest over all its
// pairs -- a threshold as good as if every row had been looked at (the wave-local one keeps ~25 k pairs per target, this
// one k and what ties with it).  The entries are grouped by target (count, scan, scatter), then one wave per target
// runs the k rounds on its group and hands the pairs with key_lo under the threshold to the exact keys.
__global__ void refcos_topk_count_kernel(const uint32_t *__restrict__ hdr1, const PairEntryK *__restrict__ list, uint32_t cap,
                                         uint32_t *__restrict__ cnt, unsigned long long *stamps)
{
    const uint32_t n = min(hdr1[0], cap);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && stamps)
        stamps[1] = (unsigned long long)wall_clock64();    // the main kernel is done
    for (uint32_t k = i; k < n; k += gridDim.x * blockDim.x)
        atomicAdd(&cnt[list[k].t], 1u);
}

// exclusive scan of cnt[0..m) into start[0..m] by ONE workgroup (m = targets of a call); cnt becomes the groups' fill cursors
__global__ __launch_bounds__(1024) void refcos_topk_scan_kernel(uint32_t *__restrict__ cnt, uint32_t m, uint32_t *__restrict__ start)
{
    __shared__ uint32_t sTot[1024];
    const uint32_t per = (m + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, m);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i)
        sum += cnt[i];
    sTot[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)o ? sTot[threadIdx.x - o] : 0u;
        __syncthreads();
        sTot[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? sTot[threadIdx.x - 1] : 0u;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = cnt[i];
        start[i] = run;
        cnt[i] = 0;
        run += c;
    }
    if (threadIdx.x == 1023)
        start[m] = sTot[1023];
}

__global__ void refcos_topk_scatter_kernel(const uint32_t *__restrict__ hdr1, const PairEntryK *__restrict__ list, uint32_t cap,
                                           const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                           PairEntryK *__restrict__ sorted)
{
    const uint32_t n = min(hdr1[0], cap);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const PairEntryK e = list[k];
        sorted[start[e.t] + atomicAdd(&cursor[e.t], 1u)] = e;
    }
}

// one wave per target: the k-th smallest distinct key_hi of its group, then its pairs with key_lo under it -> list 2
__global__ __launch_bounds__(256) void refcos_topk_select_kernel(const uint32_t *__restrict__ start, const PairEntryK *__restrict__ sorted,
                                                                 uint32_t m, uint32_t kTop, uint32_t cap2, uint32_t *__restrict__ hdr2,
                                                                 uint2 *__restrict__ pairs, uint2 *__restrict__ region /* [m]: {first, count} in list 2 */)
{
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= m)
        return;
    const uint32_t lo = start[t], hi = start[t + 1];
    const double INF = __builtin_inf();
    double prev = -1.0;                                               // (keys are >= 0)
    if (hi - lo <= 64u * 64u) {
        // the group's upper bounds in registers, read once (k rounds over global memory were 5.5 of k = 64's 9 ms)
        double v[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const uint32_t k = lo + (uint32_t)lane + 64u * j;
            v[j] = k < hi ? sorted[k].key_hi : INF;
        }
        for (uint32_t r = 0; r < kTop; ++r) {
            double best = INF;
#pragma unroll
            for (int j = 0; j < 64; ++j)
                best = (v[j] > prev && v[j] < best) ? v[j] : best;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                best = fmin(best, __shfl_xor(best, o));
            prev = best;                                              // +inf once the group is used up: it stays
            if (!(prev < INF))
                break;
        }
    } else {
        for (uint32_t r = 0; r < kTop; ++r) {
            double best = INF;
            for (uint32_t k = lo + lane; k < hi; k += 64) {
                const double v = sorted[k].key_hi;
                best = (v > prev && v < best) ? v : best;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                best = fmin(best, __shfl_xor(best, o));
            prev = best;
            if (!(prev < INF))
                break;
        }
    }
    const double thr = prev;
    // the target's candidates as ONE stretch of list 2 (counted first, reserved once): the final fold reads them back per target
    uint32_t total = 0;
    for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool keep = k < hi && sorted[k].key_lo <= thr && sorted[k].key_lo < INF;
        total += (uint32_t)__popcll(__ballot(keep));
    }
    uint32_t base = 0;
    if (lane == 0 && total)
        base = atomicAdd(&hdr2[0], total);
    base = __shfl(base, 0);
    if (lane == 0)
        region[t] = make_uint2(base, total);
    for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
        const uint32_t k = k0 + lane;
        PairEntryK e{};
        bool keep = false;
        if (k < hi) {
            e = sorted[k];
            keep = e.key_lo <= thr && e.key_lo < INF;
        }
        const unsigned long long mask = __ballot(keep);
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (keep) {
            if (pos < cap2)
                pairs[pos] = make_uint2(e.s, e.t);
            else
                hdr2[1] = 1;
        }
        base += (uint32_t)__popcll(mask);
    }
}

// The k rounds of the first-minimum fold over a target's exactly keyed candidates (one wave per target, its candidates one
// stretch of list 2): round r takes the smallest (key, index) above round r - 1's -- what k successive at_distance calls
// would return if each winner left the dictionary (src/sound.rs:361-367; the rules of select.hip's dtw_final_*_kernel, which
// needed four launches per round).  Keys of +inf (the reference's key was >= 2.0 or NaN) never enter; rows run out into
// SSYM_NO_MATCH / NaN.
__global__ __launch_bounds__(256) void refcos_topk_final_kernel(const uint2 *__restrict__ region, const uint2 *__restrict__ pairs,
                                                                const double *__restrict__ keys, uint32_t m, uint32_t kTop,
                                                                uint32_t indexBase, uint32_t *__restrict__ outIdx,
                                                                double *__restrict__ outCost)
{
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= m)
        return;
    const uint2 rg = region[t];
    const double INF = __builtin_inf();
    double prevKey = -1.0;
    uint32_t prevIdx = 0;
    bool any = true;
    for (uint32_t r = 0; r < kTop; ++r) {
        double bk = INF;
        uint32_t bi = 0xffffffffu;
        if (any)
            for (uint32_t j = lane; j < rg.y; j += 64) {
                const double key = keys[rg.x + j];
                const uint32_t idx = pairs[rg.x + j].x;
                const bool above = r == 0 || key > prevKey || (key == prevKey && idx > prevIdx);
                if (key < INF && above && (key < bk || (key == bk && idx < bi))) {
                    bk = key;
                    bi = idx;
                }
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ok = __shfl_xor(bk, o);
            const uint32_t oi = __shfl_xor(bi, o);
            if (ok < bk || (ok == bk && oi < bi)) {
                bk = ok;
                bi = oi;
            }
        }
        any = bi != 0xffffffffu;
        if (lane == 0) {
            outIdx[(size_t)t * kTop + r] = any ? bi + indexBase : SSYM_NO_MATCH;
            if (outCost)
                outCost[(size_t)t * kTop + r] = any ? bk : __builtin_nan("");
        }
        prevKey = bk;
        prevIdx = bi;
    }
}

// The candidates' keys exactly as the reference computes them: eight lanes per pair, lane i keeps rulinalg's
// running sum p_i (the code shape of refcos_match_one_kernel), combination, tail, division, |sim - distance|;
// the smallest key below the fold start 2.0 per target is kept with an integer atomic (keys >= 0).
__global__ __launch_bounds__(256) void refcos_pairs_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, const double *__restrict__ srcNorm,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, const double *__restrict__ tgtNorm,
    uint32_t dim, const double *__restrict__ dist, double defaultDist, const uint32_t *__restrict__ hdr2,
    const uint2 *__restrict__ pairs, double *__restrict__ keys, unsigned long long *__restrict__ bestKey /* NULL: top-k */)
{
    const uint32_t n = hdr2[0];
    const int i8 = threadIdx.x & 7;
    const int g0 = (threadIdx.x & 63) & ~7;
    for (uint32_t k = blockIdx.x * 32 + (threadIdx.x >> 3); k < ((n + 31) & ~31u); k += gridDim.x * 32) {
        const bool live = k < n;
        const uint2 pr = live ? pairs[k] : make_uint2(0, 0);
        unsigned long long ba = 0, bb = 0;
        uint32_t len = 0;
        if (live) {
            ba = srcOff[pr.x] * dim;
            bb = tgtOff[pr.y] * dim;
            const uint32_t la = (uint32_t)((srcOff[pr.x + 1] - srcOff[pr.x]) * dim);
            const uint32_t lb = (uint32_t)((tgtOff[pr.y + 1] - tgtOff[pr.y]) * dim);
            len = la < lb ? la : lb;                                   // src/sound.rs:24-28
        }
        const uint32_t qb = len / 8, rem = len % 8;
        // (the sum is a chain, the loads are not: sixteen blocks' operands are requested before the first is added --
        // one load pair per trip made the kernel wait out a memory round trip per block of eight elements)
        double p = 0.0;
        uint32_t j = 0;
        for (; j + 16 <= qb; j += 16) {
            double av[16], bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                av[u] = srcRaw[ba + 8 * (j + u) + i8];
                bv[u] = tgtRaw[bb + 8 * (j + u) + i8];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                p = __dadd_rn(p, __dmul_rn(av[u], bv[u]));
        }
        for (; j < qb; ++j)
            p = __dadd_rn(p, __dmul_rn(srcRaw[ba + 8 * j + i8], tgtRaw[bb + 8 * j + i8]));
        const double p0 = __shfl(p, g0 + 0), p1 = __shfl(p, g0 + 1), p2 = __shfl(p, g0 + 2), p3 = __shfl(p, g0 + 3);
        const double p4 = __shfl(p, g0 + 4), p5 = __shfl(p, g0 + 5), p6 = __shfl(p, g0 + 6), p7 = __shfl(p, g0 + 7);
        if (i8 == 0 && live) {
            double acc = 0.0;
            acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p0, p4);      // (the association: include/ssym_rulinalg.h)
            acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p1, p5);
            acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p2, p6);
            acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p3, p7);
            for (uint32_t i = 0; i < rem; ++i)
                acc = __dadd_rn(acc, __dmul_rn(srcRaw[ba + 8 * qb + i], tgtRaw[bb + 8 * qb + i]));
            const double nrm = __dmul_rn(srcNorm[pr.x], tgtNorm[pr.y]);   // src/sound.rs:30
            const double sim = __ddiv_rn(acc, nrm);                       // src/sound.rs:32
            const double d = dist ? dist[pr.y] : defaultDist;
            const double key = fabs(__dsub_rn(sim, d));                   // src/sound.rs:359
            if (bestKey) {
                keys[k] = key;
                if (key < 2.0)                                            // the fold's start value (NaN: false)
                    atomicMin(&bestKey[pr.y], (unsigned long long)__double_as_longlong(key));
            } else {
                // top-k: the rounds of select.hip's fold take every finite key; only keys below the fold start 2.0 qualify
                // (src/sound.rs:361-362; NaN never does)
                keys[k] = key < 2.0 ? key : __builtin_inf();
            }
        }
    }
}

// lowest index among a target's candidates that reach its smallest key
__global__ void refcos_fold_idx_kernel(const uint32_t *__restrict__ hdr2, const uint2 *__restrict__ pairs,
                                       const double *__restrict__ keys, const unsigned long long *__restrict__ bestKey,
                                       uint32_t *__restrict__ bestIdx)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= hdr2[0])
        return;
    const uint2 pr = pairs[k];
    if ((unsigned long long)__double_as_longlong(keys[k]) == bestKey[pr.y])
        atomicMin(&bestIdx[pr.y], pr.x);
}

// Behind the search's results, for the caller's one copy back: four header words of the two lists and three device
// timestamps (start, main kernel done, now).  tail: [h1 x 2][h2 x 2][stamp x 3 (8-byte aligned)].
__device__ __forceinline__ void refcos_pack_tail(int lane, const uint32_t *__restrict__ h1, const uint32_t *__restrict__ h2,
                                                 uint32_t *__restrict__ tail, const unsigned long long *__restrict__ stamps)
{
    if (lane < 2)
        tail[lane] = h1[lane];
    else if (lane < 4)
        tail[lane] = h2[lane - 2];
    else if (lane == 4 && stamps) {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(tail + 4);
        out[0] = stamps[0];
        out[1] = stamps[1];
        out[2] = (unsigned long long)wall_clock64();
    }
}

__global__ void refcos_pack_tail_kernel(const uint32_t *__restrict__ h1, const uint32_t *__restrict__ h2, uint32_t *__restrict__ tail,
                                        const unsigned long long *__restrict__ stamps)
{
    refcos_pack_tail((int)threadIdx.x, h1, h2, tail, stamps);
}

__global__ void refcos_fold_out_kernel(const unsigned long long *__restrict__ bestKey, const uint32_t *__restrict__ bestIdx,
                                       uint32_t nTgt, uint32_t indexBase, uint32_t *__restrict__ outIdx,
                                       double *__restrict__ outCost, const uint32_t *__restrict__ h1,
                                       const uint32_t *__restrict__ h2, uint32_t *__restrict__ tail,
                                       const unsigned long long *__restrict__ stamps)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x >= blockDim.x - 8)      // (the search's last kernel packs the tail itself)
        refcos_pack_tail((int)(threadIdx.x - (blockDim.x - 8)), h1, h2, tail, stamps);
    if (t >= nTgt)
        return;
    const bool won = bestIdx[t] != 0xffffffffu;          // something beat the fold start (0, 2.0), src/sound.rs:361-367
    outIdx[t] = (won ? bestIdx[t] : 0u) + indexBase;
    if (outCost)
        outCost[t] = won ? __longlong_as_double((long long)bestKey[t]) : 2.0;
}

__global__ void refcos_init_kernel(unsigned long long *thr, unsigned long long *bestKey, uint32_t *bestIdx, uint32_t nTgt,
                                   uint32_t *hdr1, uint32_t *hdr2, unsigned long long *stamps)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && stamps)
        stamps[0] = (unsigned long long)wall_clock64();    // the search starts (device clock: no event between the kernels)
    if (t < nTgt) {
        thr[t] = kInfBitsU;
        bestKey[t] = 0x4000000000000000ull;              // 2.0: only smaller keys enter
        bestIdx[t] = 0xffffffffu;
    }
    if (t == 0) {
        hdr1[0] = hdr1[1] = 0;
        hdr2[0] = hdr2[1] = 0;
    }
}

}  // namespace

bool refcos_mfma_supported(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    static const bool off = ssym_knob("SSYM_REFCOS_MFMA") && atoi(ssym_knob("SSYM_REFCOS_MFMA")) == 0;
    if (off || ctx->metric != SSYM_METRIC_REFCOS || src.dim != tgt.dim)
        return false;
    // small problems: the exact tile kernel is one launch and already fast; long segments: the error constant
    // (3 L + 16) u must stay far below 1
    const uint64_t maxLen = (uint64_t)std::min(src.max_frames, tgt.max_frames) * src.dim;
    // (its staging reads a tile's first value in place of the values beyond a segment's end: there must be one; a tile's
    //  128 segments are addressed by 32-bit offsets)
    // Each SIDE separately: a tile's 128 consecutive segments are addressed relative to the tile's first value with
    // 32-bit offsets and their lengths are kept as unsigned, so 128 of a side's longest segments must stay below 2^32
    // values whatever the other side's lengths are (long segments against short ones would otherwise pass the test
    // on the common prefix and stage wrong data).
    const uint64_t tileSrc = (uint64_t)src.max_frames * src.dim * 128, tileTgt = (uint64_t)tgt.max_frames * tgt.dim * 128;
    return (uint64_t)src.n * tgt.n >= 65536 && maxLen <= (1u << 20) && tileSrc < (1ull << 32) && tileTgt < (1ull << 32) &&
           src.total_frames > 0 && tgt.total_frames > 0;
}

// 256 bytes of zeros on the device, made once per context (stream-ordered before their first use)
static int32_t refcos_zeros(ssym_ctx *ctx, const double **out)
{
    if (!ctx->zeros.ptr) {
        int32_t rc = ensure(ctx, ctx->zeros, 256);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(ctx->zeros.ptr, 0, 256, ctx->stream));
    }
    *out = (const double *)ctx->zeros.ptr;
    return SSYM_OK;
}

size_t refcos_list_capacity(uint32_t n_src, uint32_t n_tgt)
{
    const uint64_t tiles = (n_src + kMT - 1) / kMT;
    // a tile lists about one pair per target column while thresholds are still loose
    return (size_t)std::min<uint64_t>((uint64_t)n_src * n_tgt, std::max<uint64_t>(4ull * tiles * n_tgt + 65536, 1u << 20));
}

// The filtered search.  Everything is enqueued; *overflow_dev_hdr receives the device header of list 1 ({wanted,
// overflow}) so that the caller can look at it after its synchronisation and fall back to the exact tile kernel.
int32_t launch_refcos_match_mfma(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const double *dist_dev,
                                 uint32_t index_base, uint32_t *out_idx_dev, double *out_cost_dev,
                                 const uint32_t **list1_hdr, const uint32_t **list2_hdr, uint32_t k_top, bool integer_filter,
                                 unsigned long long *stamps, uint32_t *tail)
{
    const uint32_t N = src.n, M = tgt.n;
    hipStream_t st = ctx->stream;
    // (entries are counted in 32 bits and a top-k list is held twice at 24 bytes an entry: 2^28 entries at the most -- a list
    //  that would need more overflows, and the call takes the next filter or the tile kernel)
    // (... and the kernels that sweep a list with one thread per entry are launched with at most 65535 x 16 workgroups)
    const size_t cap = std::min<uint64_t>(std::min<uint64_t>((uint64_t)N * M, std::min<uint64_t>(1ull << 28, 65535ull * 16 * 256)),
                                          (uint64_t)refcos_list_capacity(N, M) * std::min<uint32_t>(k_top, 64));
    const size_t entryBytes = k_top > 1 ? sizeof(PairEntryK) : sizeof(PairEntry);
    int32_t rc = ensure(ctx, ctx->cand, 4 * sizeof(uint32_t) + entryBytes * cap);
    if (rc == SSYM_OK && k_top > 1)      // top-k: the entries grouped by target, the groups' counts / cursors and starts
        rc = ensure(ctx, ctx->cmat, sizeof(PairEntryK) * cap);
    if (rc == SSYM_OK && k_top > 1)
        rc = ensure(ctx, ctx->selcnt, sizeof(uint32_t) * (4 * (size_t)M + 4));
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->cand2, 4 * sizeof(uint32_t) + sizeof(uint2) * cap);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->cand_cost, sizeof(double) * cap);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->tmin, sizeof(unsigned long long) * M);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->best, (sizeof(unsigned long long) + sizeof(uint32_t)) * (size_t)M + 16);
    const double *zeros = nullptr;
    if (rc == SSYM_OK)
        rc = refcos_zeros(ctx, &zeros);
    if (rc != SSYM_OK)
        return rc;
    uint32_t *hdr1 = (uint32_t *)ctx->cand.ptr;
    PairEntry *list1 = (PairEntry *)(hdr1 + 4);
    uint32_t *hdr2 = (uint32_t *)ctx->cand2.ptr;
    uint2 *pairs = (uint2 *)(hdr2 + 2);              // (the layout of the dtw lists: launch_dtw_final folds it for top-k)
    unsigned long long *thr = (unsigned long long *)ctx->tmin.ptr;
    unsigned long long *bestKey = (unsigned long long *)ctx->best.ptr;
    uint32_t *bestIdx = (uint32_t *)(bestKey + M);
    double *keys = (double *)ctx->cand_cost.ptr;

    refcos_init_kernel<<<(M + 255) / 256, 256, 0, st>>>(thr, bestKey, bestIdx, M, hdr1, hdr2, stamps);
    dim3 grid((M + kNT - 1) / kNT, (N + kMT - 1) / kMT);
    // the integer filter (refcos_q8.hip) where both sets have its records, the f64 matrix pipe otherwise: both leave
    // thresholds and list 1 in the same form
    if (integer_filter) {                // (the caller has asked refcos_q8_ready)
        rc = launch_refcos_q8_kernel(ctx, src, tgt, dist_dev, thr, hdr1, list1, (uint32_t)cap, k_top, nullptr);
        if (rc != SSYM_OK)
            return rc;
    } else if (k_top > 1)
        refcos_mfma_kernel<false, true><<<grid, 256, 0, st>>>(src.raw, src.off, src.norm, tgt.raw, tgt.off, tgt.norm, N, M, src.dim,
                                                 (unsigned long long)src.total_frames * src.dim,
                                                 (unsigned long long)tgt.total_frames * tgt.dim, dist_dev, 1.0, thr, hdr1,
                                                 list1, (uint32_t)cap, nullptr, k_top, zeros);
    else
        refcos_mfma_kernel<false, false><<<grid, 256, 0, st>>>(src.raw, src.off, src.norm, tgt.raw, tgt.off, tgt.norm, N, M, src.dim,
                                                 (unsigned long long)src.total_frames * src.dim,
                                                 (unsigned long long)tgt.total_frames * tgt.dim, dist_dev, 1.0, thr, hdr1,
                                                 list1, (uint32_t)cap, nullptr, 1, zeros);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    if (!stamps)                         // (with device timestamps no event sits between the kernels: each costs ~7 us of gap)
        SSYM_HIP_CHECK(ctx, hipEventRecord(ctx->ev[1], st));      // main kernel | selection, exact keys, fold
    const unsigned keepBlocks = (unsigned)std::min<size_t>((cap + 255) / 256, 65535u * 16u);
    if (k_top > 1) {
        uint32_t *cnt = (uint32_t *)ctx->selcnt.ptr, *start = cnt + M;
        PairEntryK *sorted = (PairEntryK *)ctx->cmat.ptr;
        const PairEntryK *listK = reinterpret_cast<const PairEntryK *>(list1);
        rc = zero_words(ctx, cnt, sizeof(uint32_t) * M);
        if (rc != SSYM_OK)
            return rc;
        const unsigned sweepBlocks = std::min<unsigned>(keepBlocks, (unsigned)ctx->num_cus * 8);
        refcos_topk_count_kernel<<<sweepBlocks, 256, 0, st>>>(hdr1, listK, (uint32_t)cap, cnt, stamps);
        refcos_topk_scan_kernel<<<1, 1024, 0, st>>>(cnt, M, start);
        refcos_topk_scatter_kernel<<<sweepBlocks, 256, 0, st>>>(hdr1, listK, (uint32_t)cap, start, cnt, sorted);
        refcos_topk_select_kernel<<<(M + 3) / 4, 256, 0, st>>>(start, sorted, M, k_top, (uint32_t)cap, hdr2, pairs,
                                                               (uint2 *)(start + M + 2));
    } else {
        refcos_keep_kernel<<<keepBlocks, 256, 0, st>>>(hdr1, list1, (uint32_t)cap, thr, hdr2, pairs, stamps);
    }
    refcos_pairs_kernel<<<std::min<unsigned>((unsigned)((cap + 31) / 32), (unsigned)ctx->num_cus * 16), 256, 0, st>>>(
        src.raw, src.off, src.norm, tgt.raw, tgt.off, tgt.norm, src.dim, dist_dev, 1.0, hdr2, pairs, keys,
        k_top > 1 ? nullptr : bestKey);
    if (k_top > 1) {
        // the k rounds of the first-minimum fold over the exactly keyed candidates, per target (the keys are the values it
        // folds and reports: no distance left to subtract)
        refcos_topk_final_kernel<<<(M + 3) / 4, 256, 0, st>>>((const uint2 *)((uint32_t *)ctx->selcnt.ptr + 2 * (size_t)M + 2), pairs, keys, M,
                                                              k_top, index_base, out_idx_dev, out_cost_dev);
        if (tail)
            refcos_pack_tail_kernel<<<1, 64, 0, st>>>(hdr1, hdr2, tail, stamps);
    } else {
        refcos_fold_idx_kernel<<<keepBlocks, 256, 0, st>>>(hdr2, pairs, keys, bestKey, bestIdx);
        refcos_fold_out_kernel<<<(M + 255) / 256, 256, 0, st>>>(bestKey, bestIdx, M, index_base, out_idx_dev, out_cost_dev, hdr1, hdr2,
                                                              tail, stamps);
    }
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    *list1_hdr = hdr1;
    *list2_hdr = hdr2;
    return SSYM_OK;
}

// ssym_pair_matrix(exact = 2): the similarities the matrix pipe's dots give (FMA chains; within the bound above of the
// reference's), for tests and for looking at the filter
int32_t launch_refcos_mfma_sims(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *sims, bool integer_filter)
{
    const uint32_t N = src.n, M = tgt.n;
    hipStream_t st = ctx->stream;
    const size_t cap = 1024;
    int32_t rc = ensure(ctx, ctx->cand, 4 * sizeof(uint32_t) + sizeof(PairEntry) * cap);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->cand2, 4 * sizeof(uint32_t));
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->tmin, sizeof(unsigned long long) * M);
    if (rc == SSYM_OK)
        rc = ensure(ctx, ctx->best, (sizeof(unsigned long long) + sizeof(uint32_t)) * (size_t)M + 16);
    const double *zeros = nullptr;
    if (rc == SSYM_OK)
        rc = refcos_zeros(ctx, &zeros);
    if (rc != SSYM_OK)
        return rc;
    uint32_t *hdr1 = (uint32_t *)ctx->cand.ptr;
    unsigned long long *bestKey = (unsigned long long *)ctx->best.ptr;
    refcos_init_kernel<<<(M + 255) / 256, 256, 0, st>>>((unsigned long long *)ctx->tmin.ptr, bestKey, (uint32_t *)(bestKey + M), M,
                                                       hdr1, (uint32_t *)ctx->cand2.ptr, nullptr);
    if (integer_filter)              // exact = 3: what refcos_q8.hip's integer dots give (the caller has checked refcos_q8_ready)
        return launch_refcos_q8_kernel(ctx, src, tgt, nullptr, (unsigned long long *)ctx->tmin.ptr, hdr1, hdr1 + 4, (uint32_t)cap,
                                       1, sims);
    dim3 grid((M + kNT - 1) / kNT, (N + kMT - 1) / kMT);
    refcos_mfma_kernel<true, false><<<grid, 256, 0, st>>>(src.raw, src.off, src.norm, tgt.raw, tgt.off, tgt.norm, N, M, src.dim,
                                             (unsigned long long)src.total_frames * src.dim,
                                             (unsigned long long)tgt.total_frames * tgt.dim, nullptr,
                                             1.0, (unsigned long long *)ctx->tmin.ptr, hdr1, (PairEntry *)(hdr1 + 4),
                                             (uint32_t)cap, sims, 1, zeros);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
