// refcos.hip -- the reference's own segment similarity, all pairs, bit for bit.
//
// Replaces cosine_sim (src/sound.rs:22-33) as called N x M times from at_distance
// (src/sound.rs:354):
//     len = min(|me|, |you|)                                   :24-28
//     nrm = norm(me) * norm(you)      (full vectors, SQUARED norms, src/sound.rs:35-38)   :30
//     dot = rulinalg::utils::dot(&me[..len], &you[..len])      :31   (rulinalg 0.4.2)
//     dot / nrm                                                :32
//
// Arithmetic order is the reference's, so results equal the CPU oracle's bit for bit:
//   * norm: sequential fold per segment, computed once in pack.hip (recomputing it per pair, as the
//     reference does, yields the same bits);
//   * dot: rulinalg's eight running sums p0..p7 over blocks of eight elements, combined in the association
//     include/ssym_rulinalg.h names (SSYM_RULINALG_COMBINE, shared with the oracle; default
//     ((((0+(p0+p4))+(p1+p5))+(p2+p6))+(p3+p7))), then the len%8 tail added one product at a time;
//     every product and sum is rounded separately (no FMA: -ffp-contract=off and __dmul_rn/__dadd_rn);
//   * the block structure depends on the PAIR's len, so each pair carries its own block count and
//     switches from the eight sums to the tail exactly where the reference does.
//
// Mapping: a 256-thread workgroup owns a 32 x 32 tile of pairs; 64-element chunks of the 32 source
// and 32 target segments reach LDS by DMA; eight lanes share a pair, one running sum each
// (refcos_sims8_kernel below).
#include "ssym_internal.hpp"
#include "ssym_rulinalg.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace ssym {

// ---------------------------------------------------------------------------------------------
// EIGHT LANES PER PAIR -- lane i of a group keeps rulinalg's running sum p_i, as refcos_match_one_kernel and
// refcos_pairs_kernel do -- and a 4 x 8 block of pairs per group.
//
// Rounds 1-2 gave a thread 2 x 2 pairs x 8 sums: per block of eight elements it read 32 values from LDS for 64
// multiply + add instructions, and with four waves of a workgroup on four SIMDs the LDS (one per CU) delivered half of
// what the VALUs could take: 0.30 of the f64 issue rate, VALU busy 0.38, a seventh of the LDS cycles bank conflicts
// (profiles/r03_refcos_tile.md keeps its counters).  With a lane per running sum the state of a pair is ONE register
// pair per lane, a group holds 4 sources x 8 targets = 32 pairs, and a block of eight elements costs 12 reads for 64
// instructions: 4096 x 4096 x 128f x 12d 4.36 -> 3.07 ms, ragged 4...160 frames 1.42 -> 0.62 ms, VALU busy 0.58.
//
// A pair accumulates while the block index m is below ITS q = min(|me|, |you|) / 8 and is finished at m == q (combine
// step, then the len % 8 tail, one product at a time: exactly where and how the reference does it).  The rows of a tile
// are taken in the order of their length (SegmentSet::len_order; results go back to the caller's positions), so the q of
// a wave's 8 x 32 pairs lie close together: blocks below the wave's smallest q run without a test per pair, the few
// between its smallest and largest q with one.  Nothing is multiplied that the reference does not multiply (no zero
// padding enters a sum), so non-finite values behave as in the reference.
constexpr int kS8Rows = 32;              // sources (and targets) per workgroup tile
constexpr int kS8Chunk = 64;             // elements per staged chunk: 8 blocks of eight, one 512-byte row per segment

__global__ __launch_bounds__(256, 2) void refcos_sims8_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, const double *__restrict__ srcNorm,
    const uint32_t *__restrict__ srcOrder, const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff,
    const double *__restrict__ tgtNorm, const uint32_t *__restrict__ tgtOrder, uint32_t nSrc, uint32_t nTgt, uint32_t dim,
    double *__restrict__ sims)
{
    // Two buffers per side, four DISTINCT objects (the compiler orders a DMA into LDS against every LDS read it cannot
    // prove disjoint, see refcos_mfma.hip).  A row's chunk is 512 bytes, row after row as the DMA lands them; its eight
    // 64-byte blocks are stored XOR-swizzled by bit 3 of the row, so that the two groups of a 16-lane read phase (same
    // source rows: a broadcast; target rows 8 apart) fall into different quarters of the banks.
    __shared__ __attribute__((aligned(16))) double sS0[kS8Rows * kS8Chunk], sS1[kS8Rows * kS8Chunk];
    __shared__ __attribute__((aligned(16))) double sT0[kS8Rows * kS8Chunk], sT1[kS8Rows * kS8Chunk];
    __shared__ unsigned long long sBase[2 * kS8Rows];
    __shared__ unsigned sLen[2 * kS8Rows];
    __shared__ uint32_t sSeg[2 * kS8Rows];
    __shared__ unsigned sMaxQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i8 = lane & 7, gl = lane >> 3;              // element lane, group of the wave
    const int quad = 2 * wave + (gl >> 2), oct = gl & 3;  // the group's 4 sources and 8 targets inside the tile
    const int g0 = lane & ~7;
    const uint32_t sTile = blockIdx.y * kS8Rows, tTile = blockIdx.x * kS8Rows;

    if (tid == 0)
        sMaxQ = 0;
    __syncthreads();
    if (tid < 2 * kS8Rows) {
        const bool isS = tid < kS8Rows;
        const uint32_t pos = isS ? sTile + tid : tTile + (tid - kS8Rows);
        const uint32_t n = isS ? nSrc : nTgt;
        unsigned long long base = 0;
        unsigned len = 0;
        uint32_t seg = 0xffffffffu;
        if (pos < n) {
            seg = (isS ? srcOrder : tgtOrder)[pos];
            const uint64_t *off = isS ? srcOff : tgtOff;
            base = off[seg] * dim;
            len = (unsigned)((off[seg + 1] - off[seg]) * dim);
        }
        sBase[tid] = base;
        sLen[tid] = len;
        sSeg[tid] = seg;
    }
    __syncthreads();

    // this group's pairs: q (whole blocks) and the tail per pair follow from the two lengths
    unsigned la[4], lb[8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        la[a] = sLen[4 * quad + a];
#pragma unroll
    for (int b = 0; b < 8; ++b)
        lb[b] = sLen[kS8Rows + 8 * oct + b];
    unsigned qLo = 0xffffffffu, qHi = 0;                  // over the wave's pairs that exist
    bool liveA[4], liveB[8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        liveA[a] = sSeg[4 * quad + a] != 0xffffffffu;
#pragma unroll
    for (int b = 0; b < 8; ++b)
        liveB[b] = sSeg[kS8Rows + 8 * oct + b] != 0xffffffffu;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool live = liveA[a] && liveB[b];
            const unsigned q = min(la[a], lb[b]) / 8;
            qLo = live ? min(qLo, q) : qLo;
            qHi = live ? max(qHi, q) : qHi;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        qLo = min(qLo, (unsigned)__shfl_xor((int)qLo, o));
        qHi = max(qHi, (unsigned)__shfl_xor((int)qHi, o));
    }
    qLo = (unsigned)__builtin_amdgcn_readfirstlane((int)qLo);
    qHi = (unsigned)__builtin_amdgcn_readfirstlane((int)qHi);
    const bool waveLive = qLo != 0xffffffffu;             // (a wave whose pairs all lie beyond the sets has nothing to do)
    if (lane == 0 && waveLive)
        atomicMax(&sMaxQ, qHi);
    __syncthreads();
    const unsigned nChunks = (sMaxQ + kS8Chunk / 8 - 1) / (kS8Chunk / 8);     // whole blocks only: the tails are fetched at the end

    double p[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
            p[a][b] = 0.0;

    // staging by DMA (global_load_lds, 16 bytes per lane, no registers): in round j the workgroup's 256 lanes land 4 KB =
    // rows 8 j .. 8 j + 7 of one side, lane t the 16-byte piece t % 32 of row 8 j + t / 32.  What a lane fetches is the
    // piece that belongs at its position after the swizzle; beyond a row's end it fetches the row's first piece instead
    // (whatever lies there is never added to a sum).
    const int stRowL = tid >> 5, stPiece = tid & 31;      // row within a round, physical piece
    // (bit 3 of row 8 (j & 3) + stRowL is j & 1: the logical element of this position, for even and odd rounds)
    const unsigned stElem0 = 8 * (stPiece >> 2) + 2 * (stPiece & 3), stElem1 = 8 * ((stPiece >> 2) ^ 1) + 2 * (stPiece & 3);
    auto fetch = [&](unsigned c, auto BUF) {
        double *const dS = decltype(BUF)::value ? sS1 : sS0, *const dT = decltype(BUF)::value ? sT1 : sT0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                     // rounds 0..3: sources, 4..7: targets
            // (the rows' starts and lengths are looked up in LDS every time: kept in registers they were spilled, and a
            //  scratch reload in front of every DMA serialised the eight of them)
            const int idx = (j < 4 ? 0 : kS8Rows) + 8 * (j & 3) + stRowL;
            const unsigned e = c * kS8Chunk + ((j & 1) ? stElem1 : stElem0);
            const double *g = (j < 4 ? srcRaw : tgtRaw) + sBase[idx] + (e < sLen[idx] ? e : 0u);
            double *const d = (j < 4 ? dS : dT) + (8 * (j & 3) + 2 * wave) * kS8Chunk;      // the wave's 1 KB of this round
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        }
    };

    // operand addresses: row base + the swizzled block + the lane's element
    const int swzB = oct & 1;                              // (rows 8 oct + b: bit 3 of the row = oct & 1; source rows 4 quad + a: quad >> 1)
    const int swzA = (quad >> 1) & 1;
    auto chunk = [&](unsigned c, auto BUF) {
        constexpr bool kOdd = decltype(BUF)::value;
        using Other = std::integral_constant<bool, !kOdd>;
        const double *const rS = (kOdd ? sS1 : sS0) + (4 * quad) * kS8Chunk + i8;
        const double *const rT = (kOdd ? sT1 : sT0) + (8 * oct) * kS8Chunk + i8;
        if (c + 1 < nChunks)
            fetch(c + 1, Other{});                         // lands in the other buffer while this one is multiplied
        if (waveLive) {
#pragma unroll 2
            for (int mp = 0; mp < kS8Chunk / 16; ++mp) {
                const unsigned m0 = c * (kS8Chunk / 8) + 2 * mp;      // first block of the pair of blocks
                if (m0 >= qHi)                                         // wave-uniform: no pair of the wave has a whole block left
                    break;
                double xa[4][2], yb[8][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        xa[a][h] = rS[a * kS8Chunk + 8 * ((2 * mp + h) ^ swzA)];
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        yb[b][h] = rT[b * kS8Chunk + 8 * ((2 * mp + h) ^ swzB)];
                }
                if (m0 + 1 < qLo) {
                    // both blocks lie below every pair's q: p_i = p_i + xs[i] * ys[i], block m0 then block m0 + 1
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 8; ++b)
                                p[a][b] = __dadd_rn(p[a][b], __dmul_rn(xa[a][h], yb[b][h]));
                } else {
                    // some pair of the wave ends here: a pair takes a block only while it is below its own q
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 8; ++b) {
                                const unsigned q = min(la[a], lb[b]) / 8;          // src/sound.rs:24-28
                                const double sum = __dadd_rn(p[a][b], __dmul_rn(xa[a][h], yb[b][h]));
                                p[a][b] = m0 + h < q ? sum : p[a][b];
                            }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's pieces of the next chunk have landed
        asm volatile("" ::: "memory");
        __syncthreads();                                   // ... everybody's have, and everybody is done with this buffer
    };
    if (nChunks > 0)
        fetch(0, std::false_type{});
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
    for (unsigned c = 0; c < nChunks; c += 2) {
        chunk(c, std::false_type{});
        if (c + 1 < nChunks)
            chunk(c + 1, std::true_type{});
    }

    // every pair's eight sums are complete: rulinalg's combine step over the group's lanes, then the len % 8 tail one
    // product at a time (lane i fetches element 8 q + i of both segments itself), the division, the store
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const uint32_t sg = sSeg[4 * quad + a];
        const unsigned long long baseA = sBase[4 * quad + a];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t tg = sSeg[kS8Rows + 8 * oct + b];
            const bool live = sg != 0xffffffffu && tg != 0xffffffffu;      // (group-uniform)
            const unsigned len = min(la[a], lb[b]);
            const unsigned q8 = len & ~7u, rem = len & 7u;
            double t = 0.0;
            if (live && (unsigned)i8 < rem)
                t = __dmul_rn(srcRaw[baseA + q8 + i8], tgtRaw[sBase[kS8Rows + 8 * oct + b] + q8 + i8]);
            const double pv = p[a][b];
            const double p0 = __shfl(pv, g0 + 0), p1 = __shfl(pv, g0 + 1), p2 = __shfl(pv, g0 + 2), p3 = __shfl(pv, g0 + 3);
            const double p4 = __shfl(pv, g0 + 4), p5 = __shfl(pv, g0 + 5), p6 = __shfl(pv, g0 + 6), p7 = __shfl(pv, g0 + 7);
            double s = 0.0;
            s = SSYM_RULINALG_STEP(__dadd_rn, s, p0, p4);
            s = SSYM_RULINALG_STEP(__dadd_rn, s, p1, p5);
            s = SSYM_RULINALG_STEP(__dadd_rn, s, p2, p6);
            s = SSYM_RULINALG_STEP(__dadd_rn, s, p3, p7);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const double ti = __shfl(t, g0 + i);               // x[8 q + i] * y[8 q + i]
                if ((unsigned)i < rem)
                    s = __dadd_rn(s, ti);
            }
            if (i8 == 0 && live) {
                const double nrm = __dmul_rn(srcNorm[sg], tgtNorm[tg]);     // src/sound.rs:30
                sims[(size_t)sg * nTgt + tg] = __ddiv_rn(s, nrm);           // src/sound.rs:32
            }
            __builtin_amdgcn_sched_barrier(0);      // one pair at a time: 32 pairs' loads and shuffles hoisted together spill
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ONE query against the dictionary in ONE launch: SoundDictionary::at_distance as the reference calls it, once per
// target (src/sound.rs:351-370, 453-454).  The batched path above costs a single query ~10 stream operations
// (pack copies, norm, similarity tile, two fold kernels, result copies) -- 125 us per call, almost all of it launch
// chain.  Here the kernel reads the query straight from the pinned staging buffer, recomputes its norm (the same
// sequential fold), gives every dictionary entry EIGHT lanes -- lane i keeps rulinalg's running sum p_i, so
// the eight sums and their combination are exactly those of the tile kernel --, folds |sim - distance| to the
// first minimum per workgroup, and the last workgroup to finish folds the workgroups in index order from the
// reference's start value (0, 2.0) and writes index and value back into pinned memory.
constexpr int kOneMaxVals = 4096;           // query values held in LDS
constexpr int kOneEntries = 32;             // dictionary entries per 256-thread workgroup
constexpr uint32_t kFewMaxQueries = 64;     // queries per call on this path

__global__ __launch_bounds__(256) void refcos_match_one_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, const double *__restrict__ srcNorm,
    uint32_t nSrc, uint32_t dim, const void *__restrict__ queries, const uint64_t *__restrict__ qOff /* frames */,
    int queryIsF32, const double *__restrict__ distances /* NULL: defaultDist */, double defaultDist,
    double *__restrict__ partValAll, uint32_t *__restrict__ partIdxAll, unsigned *__restrict__ tickets,
    uint32_t *__restrict__ outIdxAll, double *__restrict__ outValAll,
    unsigned long long *__restrict__ stamps /* nullable, host-visible: [0] a first workgroup starts, [1 + y] query y is done */)
{
    if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        stamps[0] = (unsigned long long)wall_clock64();
    // blockIdx.y = the query (a handful per call: ssym_match_one, small ssym_match_batch calls)
    const uint32_t y = blockIdx.y;
    const unsigned long long qBase = qOff[y] * dim;
    const uint32_t qLen = (uint32_t)((qOff[y + 1] - qOff[y]) * dim);
    const void *query = queryIsF32 ? (const void *)(static_cast<const float *>(queries) + qBase)
                                   : (const void *)(static_cast<const double *>(queries) + qBase);
    const double distance = distances ? distances[y] : defaultDist;
    double *partVal = partValAll + (size_t)y * gridDim.x;
    uint32_t *partIdx = partIdxAll + (size_t)y * gridDim.x;
    unsigned *ticket = tickets + y;
    uint32_t *outIdx = outIdxAll + y;
    double *outVal = outValAll + y;
    __shared__ double sq[kOneMaxVals];
    __shared__ double normQ;
    __shared__ double cv[kOneEntries];
    __shared__ uint32_t ci[kOneEntries];
    __shared__ bool last;
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < qLen; i += 256)
        sq[i] = queryIsF32 ? (double)static_cast<const float *>(query)[i] : static_cast<const double *>(query)[i];
    __syncthreads();
    if (tid == 255) {                               // norm(you), src/sound.rs:35-38: memo = item * item + memo
        double memo = 0.0;
        for (uint32_t i = 0; i < qLen; ++i)
            memo = __dadd_rn(__dmul_rn(sq[i], sq[i]), memo);
        normQ = memo;
    }
    const int i8 = tid & 7;
    const uint32_t s = blockIdx.x * kOneEntries + (tid >> 3);
    double v = __builtin_inf();
    double dot = 0.0;
    uint32_t len = 0;
    unsigned long long base = 0;
    if (s < nSrc) {
        base = srcOff[s] * dim;
        const uint32_t la = (uint32_t)((srcOff[s + 1] - srcOff[s]) * dim);
        len = la < qLen ? la : qLen;                // src/sound.rs:24-28
    }
    const uint32_t qb = len / 8, rem = len % 8;
    double p = 0.0;                                 // running sum p_i of rulinalg's dot
    uint32_t k = 0;
    for (; k + 16 <= qb; k += 16) {                          // (sixteen blocks' loads in flight, then the chain of sums)
        double av[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            av[u] = srcRaw[base + 8 * (k + u) + i8];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            p = __dadd_rn(p, __dmul_rn(av[u], sq[8 * (k + u) + i8]));
    }
    for (; k < qb; ++k)
        p = __dadd_rn(p, __dmul_rn(srcRaw[base + 8 * k + i8], sq[8 * k + i8]));
    const int g0 = (tid & 63) & ~7;                 // first lane of this entry's group of eight
    const double p0 = __shfl(p, g0 + 0), p1 = __shfl(p, g0 + 1), p2 = __shfl(p, g0 + 2), p3 = __shfl(p, g0 + 3);
    const double p4 = __shfl(p, g0 + 4), p5 = __shfl(p, g0 + 5), p6 = __shfl(p, g0 + 6), p7 = __shfl(p, g0 + 7);
    if (i8 == 0 && s < nSrc) {
        double acc = 0.0;
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p0, p4);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p1, p5);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p2, p6);
        acc = SSYM_RULINALG_STEP(__dadd_rn, acc, p3, p7);
        for (uint32_t i = 0; i < rem; ++i)
            acc = __dadd_rn(acc, __dmul_rn(srcRaw[base + 8 * qb + i], sq[8 * qb + i]));
        dot = acc;
    }
    __syncthreads();                                // normQ
    if (i8 == 0) {
        if (s < nSrc) {
            const double nrm = __dmul_rn(srcNorm[s], normQ);        // src/sound.rs:30
            const double sim = __ddiv_rn(dot, nrm);                 // src/sound.rs:32
            v = fabs(__dsub_rn(sim, distance));                     // src/sound.rs:359
        }
        cv[tid >> 3] = v;
        ci[tid >> 3] = s;
    }
    __syncthreads();
    if (tid == 0) {
        // first minimum over this workgroup's entries in index order (NaN never wins, src/sound.rs:361-367)
        double best = __builtin_inf();
        uint32_t bi = 0xffffffffu;
        for (int e = 0; e < kOneEntries; ++e)
            if (cv[e] < best) {
                best = cv[e];
                bi = ci[e];
            }
        partVal[blockIdx.x] = best;
        partIdx[blockIdx.x] = bi;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        uint32_t minIdx = 0;                        // fold start (0usize, 2f64), src/sound.rs:361
        double minVal = 2.0;
        for (uint32_t b = 0; b < gridDim.x; ++b) {
            const double pv = static_cast<volatile double *>(partVal)[b];
            if (pv < minVal) {
                minVal = pv;
                minIdx = static_cast<volatile uint32_t *>(partIdx)[b];
            }
        }
        *outIdx = minIdx;
        *outVal = minVal;
        *ticket = 0;                                // ready for the next call
        if (stamps)
            stamps[1 + y] = (unsigned long long)wall_clock64();
        __threadfence_system();
    }
}

bool refcos_few_supported(const ssym_ctx *ctx, const SegmentSet &src, const uint64_t *q_off, uint32_t n_queries)
{
    if (ctx->metric != SSYM_METRIC_REFCOS || src.n == 0 || n_queries == 0 || n_queries > kFewMaxQueries || !q_off)
        return false;
    for (uint32_t i = 0; i < n_queries; ++i)
        if (q_off[i + 1] < q_off[i] || (q_off[i + 1] - q_off[i]) * src.dim > (uint64_t)kOneMaxVals)
            return false;
    return true;
}

// queries / q_off / distances: device-accessible (pinned) copies -- features in the context's dtype, offsets in frames
// rebased to 0, distances or NULL; out_val / out_idx: pinned, n_queries entries each
int32_t launch_refcos_match_few(ssym_ctx *ctx, const SegmentSet &src, const void *queries, const uint64_t *q_off,
                                uint32_t n_queries, const double *distances, double default_dist, double *out_val,
                                uint32_t *out_idx, unsigned long long *stamps)
{
    const uint32_t nb = (src.n + kOneEntries - 1) / kOneEntries;
    int32_t rc = ensure(ctx, ctx->part, (sizeof(double) + sizeof(uint32_t)) * (size_t)nb * n_queries + 256);
    if (rc != SSYM_OK)
        return rc;
    if (!ctx->one_ticket.ptr) {
        rc = ensure(ctx, ctx->one_ticket, sizeof(unsigned) * kFewMaxQueries);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(ctx->one_ticket.ptr, 0, sizeof(unsigned) * kFewMaxQueries, ctx->stream));
    }
    double *partVal = (double *)ctx->part.ptr;
    uint32_t *partIdx = (uint32_t *)(partVal + (size_t)nb * n_queries);
    refcos_match_one_kernel<<<dim3(nb, n_queries), 256, 0, ctx->stream>>>(
        src.raw, src.off, src.norm, src.n, src.dim, queries, q_off, ctx->dtype == SSYM_DTYPE_F32 ? 1 : 0, distances,
        default_dist, partVal, partIdx, (unsigned *)ctx->one_ticket.ptr, out_idx, out_val, stamps);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

// the set's segments ordered by length (stable), on the device; built once per content (pack.hip drops it with the rest)
static int32_t ensure_len_order(ssym_ctx *ctx, const SegmentSet &set)
{
    if (set.len_order && set.len_order_n == set.n)
        return SSYM_OK;
    dev_free(ctx, set.len_order);
    set.len_order = nullptr;
    set.len_order_n = 0;
    std::vector<uint32_t> order(set.n);
    for (uint32_t i = 0; i < set.n; ++i)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return set.h_off[x + 1] - set.h_off[x] < set.h_off[y + 1] - set.h_off[y];
    });
    int32_t rc = dev_alloc(ctx, (void **)&set.len_order, sizeof(uint32_t) * std::max<uint32_t>(set.n, 1));
    if (rc != SSYM_OK)
        return rc;
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(set.len_order, order.data(), sizeof(uint32_t) * set.n, hipMemcpyHostToDevice, ctx->stream));
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // (`order` goes out of scope)
    set.len_order_n = set.n;
    return SSYM_OK;
}

int32_t launch_refcos_sims(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, double *sims)
{
    if (src.dim != tgt.dim) {
        ctx->err = "dim mismatch between dictionary and targets";
        return SSYM_E_INVALID;
    }
    if (src.n == 0 || tgt.n == 0)
        return SSYM_OK;
    const uint64_t maxLen = (uint64_t)std::min(src.max_frames, tgt.max_frames) * src.dim;
    if (maxLen > 0xfffffff0ull) {
        ctx->err = "segment too long";
        return SSYM_E_UNSUPPORTED;
    }
    int32_t rc = ensure_len_order(ctx, src);
    if (rc == SSYM_OK)
        rc = ensure_len_order(ctx, tgt);
    if (rc != SSYM_OK)
        return rc;
    dim3 grid8((tgt.n + kS8Rows - 1) / kS8Rows, (src.n + kS8Rows - 1) / kS8Rows);
    refcos_sims8_kernel<<<grid8, 256, 0, ctx->stream>>>(src.raw, src.off, src.norm, src.len_order, tgt.raw, tgt.off,
                                                       tgt.norm, tgt.len_order, src.n, tgt.n, src.dim, sims);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
