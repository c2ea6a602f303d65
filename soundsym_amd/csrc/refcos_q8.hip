// refcos_q8.hip -- the refcos search's filter on the i8 matrix pipe: every pair's dot as EXACT integer arithmetic on
// 8-bit digits, v_mfma_i32_32x32x32_i8 (64 x the multiply-adds per clock of the f64 form refcos_mfma.hip uses).
//
// Role on the path: the same as refcos_mfma_kernel's -- SoundDictionary::at_distance (src/sound.rs:351-370) needs, per
// target, the dictionary entry with the smallest |cosine_sim - distance| (src/sound.rs:22-33, 359); this kernel gives
// every pair a rigorous interval for that key, and only the pairs whose interval reaches down to a target's smallest
// upper bound are keyed in the reference's own arithmetic (refcos_pairs_kernel).  Same bits out.
//
// Fixed point.  Segment a (length La) gets E_a with max |a_i| 2^E_a in [2^21, 2^22); n_i = rint(a_i 2^E_a) is an integer
// of at most 23 bits, written in balanced base 256: n = 2^16 q1 + 2^8 q2 + q3, q2, q3 in [-128, 127], |q1| <= 65.
// With m_i, r1..r3, E_b the same for segment b and L = min(La, Lb) (src/sound.rs:24-28):
//     a_i 2^E_a = n_i + alpha_i, |alpha_i| <= 1/2       b_i 2^E_b = m_i + beta_i, |beta_i| <= 1/2
//     D 2^(E_a + E_b) = sum (n_i + alpha_i)(m_i + beta_i)          (D: the exact dot over the common prefix)
//     | D 2^(E_a+E_b) - G | <= S_m / 2 + S_n / 2 + L / 4,          G = sum n_i m_i,  S_n = sum |n_i|, S_m = sum |m_i|
//     G = 2^32 T11 + 2^24 (T12 + T21) + 2^16 (T13 + T22 + T31) + 2^8 (T23 + T32) + T33,      Tkl = sum q_k,i r_l,i
// The kernel forms K0 = T11, K1 = T12 + T21, K2 = T13 + T22 + T31 -- six integer GEMMs, exact in 32 bits for
// L <= 32768 -- and leaves out 2^8 (T23 + T32) + T33, which is at most 2^15 (sum |q2_i| + sum |r2_i|) + 2^14 L in
// magnitude (|q3|, |r3| <= 128).  So with Gk = 2^32 K0 + 2^24 K1 + 2^16 K2 (exact in a double: a multiple of 2^16 below
// 2^60) and dq = Gk 2^-(E_a + E_b):
//     | D - dq | <= ( Eseg_a + Eseg_b + 16384.25 L ) 2^-(E_a + E_b),        Eseg = S / 2 + 2^15 sum |q2_i|   (per segment)
// and the reference's own floating-point dot is within gamma_(2L+8) sum |a_i b_i| of D (top of refcos_mfma.hip): the
// f64 filter's bound with this one added.  In the units of a similarity (times ia ib, the reciprocal norms):
//     extra = A1 B2 + A2 B1 + A3 B3,    A1 = Eseg_a A2,  A2 = 2^-E_a ia,  A3 = sqrt(16384.25 La) A2   (L <= sqrt(La Lb))
// -- three multiply-adds per pair on per-segment numbers.  For Gaussian-like data the bound is ~1e-5 of a similarity's
// scale: a target keeps its winner and what lies within that of it.
//
// Sets this filter does not take (the f64 matrix pipe does): a value that is not finite, max |a_i| outside
// [2^-120, 2^120], a segment longer than 32768 values, record memory beyond the budget below.  Segments of norm 0
// (empty or all zeros) are rows of zeros here and are dropped by the reference's rule (nrm = 0: src/sound.rs:362).
#include "ssym_internal.hpp"
#include "refcos_filter.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ssym {

namespace {

constexpr int kQT = 128;             // segments per side of a workgroup tile
constexpr int kQG = 32;              // elements per group (one 128-byte line of a row: three digit planes + 32 bytes of zeros)
constexpr uint32_t kQMaxLen = 32768;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- records ------------------------------------------------------------------------------------------------------
// One workgroup per row.  info[row] = { A1' = Eseg 2^-E (times ia in the kernel), 2^-E, sqrt(16384.25 len) 2^-E, 0 }.
__global__ __launch_bounds__(256) void refcos_q8_records_kernel(const double *__restrict__ raw, const uint64_t *__restrict__ off,
                                                                uint32_t n, uint32_t dim, uint32_t groups,
                                                                int8_t *__restrict__ q8, double *__restrict__ info,
                                                                unsigned *__restrict__ bad)
{
    __shared__ double sMax[256];
    __shared__ unsigned long long sSum[256], sSum2[256];
    __shared__ int sNonFinite;
    const uint32_t g = blockIdx.x;
    const int tid = threadIdx.x;
    int8_t *row = q8 + (size_t)g * groups * 128;
    if (g >= n) {                                    // (rows beyond the set: the buffer was zeroed)
        if (tid < 4)
            info[4 * (size_t)g + tid] = 0.0;
        return;
    }
    const unsigned long long base = off[g] * dim;
    const unsigned long long len64 = (off[g + 1] - off[g]) * dim;
    const uint32_t len = (uint32_t)std::min<unsigned long long>(len64, 0xffffffffull);
    if (tid == 0)
        sNonFinite = 0;
    __syncthreads();
    double amax = 0.0;
    for (uint32_t i = tid; i < len; i += 256) {
        const double a = fabs(raw[base + i]);
        if (!(a < __builtin_inf()))
            sNonFinite = 1;
        amax = fmax(amax, a);
    }
    sMax[tid] = amax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
            sMax[tid] = fmax(sMax[tid], sMax[tid + o]);
        __syncthreads();
    }
    amax = sMax[0];
    const bool outside = sNonFinite || len64 > kQMaxLen || (amax != 0.0 && !(amax >= 0x1p-120 && amax <= 0x1p120));
    if (outside || amax == 0.0) {                    // zeros stay zeros; `outside` makes the whole set take the f64 filter
        if (outside && tid == 0)
            atomicOr(bad, 1u);
        if (tid < 4)
            info[4 * (size_t)g + tid] = tid == 1 ? 1.0 : 0.0;
        return;
    }
    int e;
    (void)frexp(amax, &e);                           // amax = f 2^e, f in [0.5, 1)
    const int E = 22 - e;                            // amax 2^E in [2^21, 2^22)
    unsigned long long sN = 0, s2 = 0;
    for (uint32_t i = tid; i < len; i += 256) {
        const long long nI = (long long)rint(ldexp(raw[base + i], E));      // |nI| <= 2^22; the scaling is exact
        const int q3 = (int)(((nI + 128) & 255) - 128);
        const long long n1 = (nI - q3) >> 8;
        const int q2 = (int)(((n1 + 128) & 255) - 128);
        const int q1 = (int)((n1 - q2) >> 8);
        int8_t *grp = row + (size_t)(i / kQG) * 128 + (i % kQG);
        grp[0] = (int8_t)q1;
        grp[32] = (int8_t)q2;
        grp[64] = (int8_t)q3;
        sN += (unsigned long long)(nI < 0 ? -nI : nI);
        s2 += (unsigned)(q2 < 0 ? -q2 : q2);
    }
    sSum[tid] = sN;
    sSum2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            sSum[tid] += sSum[tid + o];
            sSum2[tid] += sSum2[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double scl = ldexp(1.0, -E);
        const double eseg = 0.5 * (double)sSum[0] + 32768.0 * (double)sSum2[0];      // exact: integers below 2^53
        info[4 * (size_t)g + 0] = eseg * scl;                                         // (a power of two: exact)
        info[4 * (size_t)g + 1] = scl;
        info[4 * (size_t)g + 2] = sqrt(16384.25 * (double)len) * (1.0 + 0x1p-50) * scl;
        info[4 * (size_t)g + 3] = 0.0;
    }
}

// x * 2^w as a double without a conversion instruction: the bits {hi, x ^ 0x80000000} are 2^(52+w) + (x + 2^31) 2^w
__device__ __forceinline__ double q8_scaled(int x, unsigned hi, double bias)
{
    const unsigned long long bits = ((unsigned long long)hi << 32) | (unsigned)(x ^ 0x80000000);
    return __longlong_as_double((long long)bits) - bias;
}

struct QInfo {                       // per segment of a tile, in LDS
    double sq, inv, norm, dist;      // as RowInfo
    double a1, a2, a3, a4;           // error mass, 2^-E ia, length term (all times ia), the reference-rounding term's factor
    double scl;                      // 2^-E
    double cl;                       // (3 len + 16) u 1.02
};

// ---- main kernel ----------------------------------------------------------------------------------------------------
// 128 x 128 pairs per workgroup, four waves of 64 x 64 (2 x 2 MFMA blocks of 32 x 32, three accumulators each: the digit
// products of weight 2^32, 2^24, 2^16), one workgroup per CU (192 accumulator registers per lane).  A chunk is one group
// of 32 elements of every row: 128 bytes per row (a full line), global -> LDS by DMA, the 16-byte pieces of a row
// XOR-swizzled as in refcos_mfma.hip (piece q of row r sits at position q ^ ((r >> 1) & 7): the operand reads of 32 rows
// at one piece are conflict-free), FOUR chunks in flight: every row is padded with zeros to the sets' common length, so
// a DMA's address is a scalar base plus a constant per lane and nothing is selected.
template <bool WRITE_SIMS, bool TOPK, int CB>
__global__ __launch_bounds__(CB == 2 ? 256 : 512, 1) void refcos_q8_kernel(
    const int8_t *__restrict__ srcQ, const double *__restrict__ srcQInfo, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ srcNorm, const int8_t *__restrict__ tgtQ, const double *__restrict__ tgtQInfo,
    const uint64_t *__restrict__ tgtOff, const double *__restrict__ tgtNorm, uint32_t nSrc, uint32_t nTgt, uint32_t dim,
    uint32_t srcGroups, uint32_t tgtGroups, const double *__restrict__ dist, double defaultDist,
    unsigned long long *__restrict__ thr, uint32_t *__restrict__ hdr, PairEntry *__restrict__ list, uint32_t cap,
    double *__restrict__ simOut, uint32_t kTop, uint32_t tilesX, uint32_t tilesY)
{
    __shared__ __attribute__((aligned(16))) unsigned char sA0[kQT * 128], sA1[kQT * 128], sA2[kQT * 128], sA3[kQT * 128];
    __shared__ __attribute__((aligned(16))) unsigned char sB0[kQT * 128], sB1[kQT * 128], sB2[kQT * 128], sB3[kQT * 128];
    __shared__ QInfo sInfo[2 * kQT];
    __shared__ unsigned sLen[2 * kQT];
    __shared__ unsigned sMaxLen[2];
    __shared__ unsigned sPlain[6];                         // rows 0..63, 64..127; columns in four groups of 32: all plain?
    __shared__ unsigned long long sRowMax[2][4];           // per half of the tile's rows: max of a1..a4 (bits: they are >= 0)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // CB = 2: four waves of 64 x 64 (one per SIMD).  CB = 1: EIGHT waves of 64 x 32, two per SIMD -- half the accumulators
    // per wave (96 registers), so that a second wave runs on every SIMD while the first one waits (LDS, barrier, the
    // epilogue's memory round trips), at 1.5 x the operand reads per MFMA.
    constexpr int kWN = CB == 2 ? 2 : 4, kNT = 64 * 2 * kWN, kSweep = kNT / 8, kP = kQT / kSweep;
    const int wm = wave / kWN, wn = wave % kWN;
    // PERSISTENT workgroups (round 4): one per CU, tile after tile (linear tile numbers blockIdx.x, + gridDim.x, ...), so
    // that the next tile's first four chunks can be requested BEFORE the current tile's epilogue -- with one workgroup per
    // CU nothing else hides a tile's first memory round trip.  (XCD-aware tile order as in refcos_mfma.hip: workgroup
    // number & 7 is the XCD, and gridDim.x is a multiple of 8 whenever the order applies.)
    const uint32_t tilesTotal = tilesX * tilesY;
    auto tile_of = [&](uint32_t lin, uint32_t &bx, uint32_t &by) {
        bx = lin % tilesX;
        by = lin / tilesX;
        if ((tilesX & 7u) == 0 && (tilesTotal & 7u) == 0 && (gridDim.x & 7u) == 0) {
            const uint32_t q = (lin & 7u) * (tilesTotal >> 3) + (lin >> 3);
            const uint32_t r = q % (8u * tilesY);
            bx = 8u * (q / (8u * tilesY)) + (r & 7u);
            by = r >> 3;
        }
    };
    bool prefetched = false;                               // the tile's first four chunks were requested by the tile before
  for (uint32_t tileLin = blockIdx.x; tileLin < tilesTotal; tileLin += gridDim.x) {
    uint32_t bx, by;
    tile_of(tileLin, bx, by);
    const uint32_t sTile = by * kQT, tTile = bx * kQT;

    // staging: thread -> (row = tid / 8 + 32 p, position tid & 7) for p = 0..3 on both sides; the piece it fetches is
    // position ^ swizzle(row).  Per-lane byte offsets are constants, the chunk's base is scalar.
    const int sr = tid >> 3;
    unsigned offA[kP], offB[kP];
#pragma unroll
    for (int p = 0; p < kP; ++p) {
        const int row = sr + kSweep * p;
        const unsigned piece = (unsigned)((tid & 7) ^ ((row >> 1) & 7));
        offA[p] = (unsigned)row * srcGroups * 128u + piece * 16u;
        offB[p] = (unsigned)row * tgtGroups * 128u + piece * 16u;
    }
    const unsigned char *const tileA = (const unsigned char *)srcQ + (size_t)sTile * srcGroups * 128;
    const unsigned char *const tileB = (const unsigned char *)tgtQ + (size_t)tTile * tgtGroups * 128;
    auto stageA = [&](auto S) -> unsigned char * {
        constexpr int s = decltype(S)::value;
        return s == 0 ? sA0 : s == 1 ? sA1 : s == 2 ? sA2 : sA3;
    };
    auto stageB = [&](auto S) -> unsigned char * {
        constexpr int s = decltype(S)::value;
        return s == 0 ? sB0 : s == 1 ? sB1 : s == 2 ? sB2 : sB3;
    };
    auto fetch_from = [&](const unsigned char *baseA, const unsigned char *baseB, unsigned c, auto S) {   // eight DMAs, nothing waits here
        unsigned char *const dA = stageA(S), *const dB = stageB(S);
        const unsigned char *ua = baseA + (size_t)c * 128, *ub = baseB + (size_t)c * 128;
        asm volatile("" : "+s"(ua), "+s"(ub));            // (scalar base + 32-bit lane offset: refcos_mfma.hip)
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            asm volatile("" : "+v"(offA[p]), "+v"(offB[p]));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ua + offA[p]),
                                             (__attribute__((address_space(3))) void *)&dA[(kSweep * p + 8 * wave) * 128], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ub + offB[p]),
                                             (__attribute__((address_space(3))) void *)&dB[(kSweep * p + 8 * wave) * 128], 16, 0, 0);
        }
    };

    auto fetch = [&](unsigned c, auto S) { fetch_from(tileA, tileB, c, S); };
    // The first four chunks are requested before anything is known about the tile's segments (their rows exist and are
    // zero beyond a segment's end whatever its length): the per-segment values below are fetched under them.
    const unsigned nGroupsMin = min(srcGroups, tgtGroups);
    // (always four groups of eight DMAs, a chunk index beyond the rows' end clamped to their last group: the waits below
    //  count groups, and a group that is not needed lands in a stage nobody reads)
    auto first_four = [&](const unsigned char *baseA, const unsigned char *baseB) {
        fetch_from(baseA, baseB, 0, std::integral_constant<int, 0>{});
        fetch_from(baseA, baseB, min(1u, nGroupsMin - 1), std::integral_constant<int, 1>{});
        fetch_from(baseA, baseB, min(2u, nGroupsMin - 1), std::integral_constant<int, 2>{});
        fetch_from(baseA, baseB, min(3u, nGroupsMin - 1), std::integral_constant<int, 3>{});
    };
    if (nGroupsMin > 0 && !prefetched)
        first_four(tileA, tileB);
    if (tid < 2)
        sMaxLen[tid] = 0;
    if (tid < 6)
        sPlain[tid] = 1;
    if (tid < 8)
        sRowMax[tid >> 2][tid & 3] = 0;
    __syncthreads();
    if (tid < 2 * kQT) {
        const bool isS = tid < kQT;
        const uint32_t g = isS ? sTile + tid : tTile + (tid - kQT);
        const uint32_t n = isS ? nSrc : nTgt;
        const uint64_t *off = isS ? srcOff : tgtOff;
        const double *nr = isS ? srcNorm : tgtNorm;
        const double *qi = isS ? srcQInfo : tgtQInfo;
        unsigned len = 0;
        QInfo r;
        r.norm = r.sq = r.inv = 0.0;                 // (segments beyond the sets' ends: norm 0, dropped like an empty segment)
        r.a1 = r.a3 = 0.0;
        r.scl = 1.0;
        if (g < n) {
            len = (unsigned)((off[g + 1] - off[g]) * dim);
            r.norm = nr[g];
            r.sq = nr[n + g];
            r.inv = nr[2 * (size_t)n + g];
            r.a1 = qi[4 * (size_t)g + 0];
            r.scl = qi[4 * (size_t)g + 1];
            r.a3 = qi[4 * (size_t)g + 2];
        }
        r.dist = (!isS && dist && g < n) ? dist[g] : defaultDist;
        r.a2 = r.scl * r.inv;
        r.a1 = r.a1 * r.inv;
        r.a3 = r.a3 * r.inv;
        r.cl = (3.0 * (double)len + 16.0) * (1.1102230246251565e-16 * 1.02);
        r.a4 = sqrt(r.cl) * (1.0 + 0x1p-50) * (r.sq * r.inv);   // cL(min) sa sb ia ib <= a4(a) a4(b): min(x, y) <= sqrt(x y)
        sInfo[tid] = r;
        sLen[tid] = len;
        atomicMax(&sMaxLen[isS ? 0 : 1], len);
        const bool ok = g < n && r.norm >= 1e-139 && r.norm <= 1e139 && fabs(r.dist) <= 1e300;
        if (!ok)
            atomicAnd(&sPlain[isS ? tid >> 6 : 2 + ((tid - kQT) >> 5)], 0u);
        if (isS && ok) {                                   // (a half with a row that is not plain takes the general form)
            atomicMax(&sRowMax[tid >> 6][0], (unsigned long long)__double_as_longlong(r.a1));
            atomicMax(&sRowMax[tid >> 6][1], (unsigned long long)__double_as_longlong(r.a2));
            atomicMax(&sRowMax[tid >> 6][2], (unsigned long long)__double_as_longlong(r.a3));
            atomicMax(&sRowMax[tid >> 6][3], (unsigned long long)__double_as_longlong(r.a4));
        }
    }
    __syncthreads();
    const unsigned kMax = __builtin_amdgcn_readfirstlane(min(sMaxLen[0], sMaxLen[1]));
    const unsigned nChunks = (kMax + kQG - 1) / kQG;

    v16i acc[3][2][CB];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < CB; ++b)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    acc[l][a][b][g] = 0;

    // operand reads: lane (r = lane & 31, h = lane >> 5) takes, of row (block row 32 blk + r) and digit plane p, the 16
    // bytes k = 16 h .. 16 h + 15 of the group: piece 2 p + h, at position (2 p + h) ^ swizzle(row)
    const int lr = lane & 31, lh = lane >> 5;
    // Software pipeline with ONE wave per SIMD: nothing else hides what a wave waits for, so a chunk's operands are read
    // from LDS into a second register set while the PREVIOUS chunk's MFMAs run.  Iteration c: [chunk c + 1 has landed
    // (own DMAs) | own reads of chunk c are complete | barrier] -> chunk c's stage is free and chunk c + 1 is complete
    // for everybody -> DMA of chunk c + 4 into the freed stage, operand reads of chunk c + 1 into the other register
    // set, MFMAs of chunk c.  (Reads, DMA issue and barrier skew sat in front of every chunk's MFMAs before: 1600
    // cycles per chunk for 768 of MFMAs.)  The barrier is the bare instruction, not __syncthreads(): that one is a fence
    // too, and a fence after DMAs into LDS makes the compiler wait for ALL of them.
    v4i av[2][3][2], bv[2][3][CB];                         // [register set][digit plane][block]
    auto readops = [&](auto S, auto SET) {
        constexpr int set = decltype(SET)::value;
        const unsigned char *const rA = stageA(S), *const rB = stageB(S);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int rowA = wm * 64 + blk * 32 + lr;
                av[set][p][blk] = *reinterpret_cast<const v4i *>(&rA[rowA * 128 + (((2 * p + lh) ^ ((rowA >> 1) & 7)) << 4)]);
                if (blk < CB) {
                    const int rowB = wn * 32 * CB + blk * 32 + lr;
                    bv[set][p][blk] = *reinterpret_cast<const v4i *>(&rB[rowB * 128 + (((2 * p + lh) ^ ((rowB >> 1) & 7)) << 4)]);
                }
            }
    };
    auto mfmas = [&](auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb + pa < 3; ++pb)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < CB; ++b)
                        acc[pa + pb][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[set][pa][a], bv[set][pb][b], acc[pa + pb][a][b], 0, 0, 0);
    };
    auto step = [&](unsigned c, auto S) {
        constexpr int s = decltype(S)::value;
        using Set = std::integral_constant<int, s & 1>;
        using OtherSet = std::integral_constant<int, (s & 1) ^ 1>;
        using NextStage = std::integral_constant<int, (s + 1) & 3>;
        // chunk c + 1 has landed (the groups of c + 2 and c + 3 may still be in flight: every step issues one group)
        __builtin_amdgcn_s_waitcnt(rm_wait_vmcnt(4 * kP));
        __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): this wave's reads of chunk c are in its registers
        asm volatile("s_barrier" ::: "memory");
        // One basic block from here: 8 DMAs (chunk c + 4 into the stage just vacated), 12 operand reads (chunk c + 1 into
        // the other register set), 24 MFMAs (chunk c) -- none of them conditional, so that the scheduler can be told
        // to spread the memory instructions BETWEEN the MFMAs instead of in front of them (20 issue slots of 16 cycles
        // and more with the matrix pipe idle, per chunk).  A chunk beyond the tile's last is a clamped, unused copy.
#ifndef SSYM_Q8_NODMA        // (tools only: timing without the loop's DMAs -- the MFMAs then run on stale bytes)
        fetch(min(c + 4, nGroupsMin - 1), S);
#endif
        readops(NextStage{}, OtherSet{});
        mfmas(Set{});
#pragma unroll
        for (int i = 0; i < 4 * CB; ++i) {                 // (2 kP DMAs, 6 + 3 CB reads, 12 CB MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read (the DMA)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    if (nChunks > 0) {
        __builtin_amdgcn_s_waitcnt(rm_wait_vmcnt(6 * kP));
        asm volatile("s_barrier" ::: "memory");
        readops(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    for (unsigned c = 0; c < nChunks; c += 4) {
        step(c, std::integral_constant<int, 0>{});
        if (c + 1 < nChunks)
            step(c + 1, std::integral_constant<int, 1>{});
        if (c + 2 < nChunks)
            step(c + 2, std::integral_constant<int, 2>{});
        if (c + 3 < nChunks)
            step(c + 3, std::integral_constant<int, 3>{});
    }

    // The next tile's first four chunks, requested before this tile's epilogue (every wave's reads of the stages are in its
    // registers -- lgkmcnt(0) -- and every wave has got there -- the barrier -- so the stages are free; the clamped copies
    // the last steps requested are older and land first).  Its epilogue's own memory operations queue up behind them.
    prefetched = false;
#ifndef SSYM_Q8_NOPREFETCH
    if (nGroupsMin > 0 && tileLin + gridDim.x < tilesTotal) {
        uint32_t nbx, nby;
        tile_of(tileLin + gridDim.x, nbx, nby);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("s_barrier" ::: "memory");
        first_four((const unsigned char *)srcQ + (size_t)(nby * kQT) * srcGroups * 128,
                   (const unsigned char *)tgtQ + (size_t)(nbx * kQT) * tgtGroups * 128);
        prefetched = true;
    }
#endif
#ifdef SSYM_Q8_NOEPI         // (tools only: the main loop alone, every accumulator kept alive)
    {
        v16i t = acc[0][0][0];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < CB; ++b2)
                    t += acc[l][a][b2];
        int x = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g)
            x ^= t[g];
        if (x == 0x12345678 && hdr[0] == 77)
            thr[0] = 1;
        __syncthreads();
        continue;
    }
#endif
    // ---- epilogue: integer dots -> key intervals -> thresholds and list 1 (the steps of refcos_mfma.hip's) -----------
    // D layout of the 32 x 32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    const double INF = __builtin_inf();
    const double u = 1.1102230246251565e-16;
    const bool plain = !WRITE_SIMS && (sPlain[wm] & sPlain[2 + wn * CB] & sPlain[2 + wn * CB + CB - 1]) != 0;
    QInfo ci[CB];
    double colMin[CB];
#pragma unroll
    for (int b = 0; b < CB; ++b) {
        ci[b] = sInfo[kQT + wn * 32 * CB + b * 32 + lr];
        colMin[b] = INF;
    }
    double klo[CB][2][16];                                 // [column block][row block][register]
    double khis[TOPK ? CB : 1][TOPK ? 2 : 1][TOPK ? 16 : 1];
    // PLAIN waves (all 64 + 64 segments inside the sets, norms in [1e-139, 1e139], finite distances -- and, the records
    // existing, all values finite): per pair only z = |s - dist| is formed, 10 operations; the half-width of the interval
    // comes from per-COLUMN constants -- the column's own numbers against the largest a1..a4 among the wave's 64 rows,
    // and |s| <= z + |dist| -- so R <= Rub(z) = c9 (z + 2 |dist|) + C, increasing in z: the column's threshold is the
    // bound of its smallest z, and key_lo >= z (1 - c9) - K with K = 2 c9 |dist| + C, one multiply-add per pair after the
    // thresholds are known.  (The general form below keeps every row's own numbers: ~27 operations per pair.)
    const double c9 = 9.0 * u * 1.0000001;
    double Cb[CB], Kb[CB];
    if (plain) {
        const double m1 = __longlong_as_double((long long)sRowMax[wm][0]), m2 = __longlong_as_double((long long)sRowMax[wm][1]);
        const double m3 = __longlong_as_double((long long)sRowMax[wm][2]), m4 = __longlong_as_double((long long)sRowMax[wm][3]);
        double zmin[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            Cb[b] = 1.0001 * __fma_rn(m4, ci[b].a4, __fma_rn(m1, ci[b].a2, __fma_rn(m2, ci[b].a1, m3 * ci[b].a3))) + 1e-290;
            Kb[b] = __fma_rn(2.0 * c9, fabs(ci[b].dist), Cb[b]) * (1.0 + 4.0 * u);
            zmin[b] = INF;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                const double ra2 = sInfo[row].a2;
#pragma unroll
                for (int b = 0; b < CB; ++b) {
                    // the kept part of the integer dot, exactly: a multiple of 2^16 below 2^60
                    // (through the exponent trick, not v_cvt_f64_i32: three conversions per pair were 2.5 % of the kernel)
                    const double gk = q8_scaled(acc[0][a][b][g], 0x45300000u, 0x1p84 + 0x1p63) +
                                      (q8_scaled(acc[1][a][b][g], 0x44b00000u, 0x1p76 + 0x1p55) +
                                       q8_scaled(acc[2][a][b][g], 0x44300000u, 0x1p68 + 0x1p47));
                    const double z = fabs(gk * (ra2 * ci[b].a2) - ci[b].dist);
                    klo[b][a][g] = z;                      // (becomes key_lo once the thresholds are known)
                    zmin[b] = fmin(zmin[b], z);
                }
            }
#pragma unroll
        for (int b = 0; b < CB; ++b)
            colMin[b] = (zmin[b] + __fma_rn(c9, zmin[b] + 2.0 * fabs(ci[b].dist), Cb[b])) * (1.0 + 4.0 * u);
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                const QInfo ri = sInfo[row];
#pragma unroll
                for (int b = 0; b < CB; ++b) {
                    const double gk = __fma_rn(0x1p32, (double)acc[0][a][b][g],
                                               __fma_rn(0x1p24, (double)acc[1][a][b][g], 0x1p16 * (double)acc[2][a][b][g]));
                    const double extra = __fma_rn(ri.a1, ci[b].a2, __fma_rn(ri.a2, ci[b].a1, ri.a3 * ci[b].a3));
                    const unsigned la = sLen[row], lb = sLen[kQT + wn * 32 * CB + b * 32 + lr];
                    const unsigned len = la < lb ? la : lb;
                    const double nrm = __dmul_rn(ri.norm, ci[b].norm);
                    const double dotm = gk * (ri.scl * ci[b].scl);
                    double lo, hi;
                    refcos_key_interval(dotm, ri.sq * ci[b].sq, ri.inv * ci[b].inv, nrm,
                                        (3.0 * (double)len + 16.0) * (u * 1.02), ci[b].dist, lo, hi, extra);
                    if (WRITE_SIMS) {
                        const uint32_t s = sTile + row, t = tTile + wn * 32 * CB + b * 32 + lr;
                        if (s < nSrc && t < nTgt)
                            simOut[(size_t)s * nTgt + t] = __ddiv_rn(dotm, nrm);
                    }
                    klo[b][a][g] = lo;
                    if (TOPK)
                        khis[b][a][g] = hi;
                    else
                        colMin[b] = fmin(colMin[b], hi);
                }
            }
    }
    double cur[CB];
#pragma unroll
    for (int b = 0; b < CB; ++b) {
        double cmin = colMin[b];
        if (TOPK) {
            // the kTop-th smallest DISTINCT upper bound among the wave's 64 rows (refcos_mfma.hip); in plain waves the
            // upper bound increases with z, so the rounds run on the z themselves and the bound is formed once
            double prev = -1.0;
            for (uint32_t r = 0; r < kTop; ++r) {
                double m = INF;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const double v = plain ? klo[b][a][g] : khis[b][a][g];
                        m = (v > prev && v < m) ? v : m;
                    }
                m = fmin(m, __shfl_xor(m, 32));
                prev = m;
            }
            cmin = plain ? (prev + __fma_rn(c9, prev + 2.0 * fabs(ci[b].dist), Cb[b])) * (1.0 + 4.0 * u) : prev;
        } else {
            cmin = fmin(cmin, __shfl_xor(cmin, 32));
        }
        cur[b] = cmin;
    }
    unsigned long long seenBits[CB];
#pragma unroll
    for (int b = 0; b < CB; ++b) {
        const uint32_t t = tTile + wn * 32 * CB + b * 32 + lr;
        seenBits[b] = kInfBitsU;
        if (lh == 0 && t < nTgt)
            seenBits[b] = atomicMin(&thr[t], (unsigned long long)__double_as_longlong(cur[b]));
    }
#pragma unroll
    for (int b = 0; b < CB; ++b) {
        seenBits[b] = __shfl(seenBits[b], lr);
        cur[b] = fmin(fmin(cur[b], __longlong_as_double((long long)seenBits[b])), 1.7976931348623157e308);
    }
    // (top-k searches list key_hi beside key_lo: in plain waves both are functions of the z still sitting in klo[], formed
    //  where they are needed; otherwise z becomes key_lo in place)
    auto lo_of = [&](int b, int a, int g) -> double {
        if (TOPK && plain)
            return fmax(__fma_rn(klo[b][a][g], 1.0 - c9, -Kb[b]) * (1.0 - 4.0 * u), 0.0);
        return klo[b][a][g];
    };
    auto hi_of = [&](int b, int a, int g) -> double {
        if (plain)
            return (klo[b][a][g] + __fma_rn(c9, klo[b][a][g] + 2.0 * fabs(ci[b].dist), Cb[b])) * (1.0 + 4.0 * u);
        return khis[TOPK ? b : 0][TOPK ? a : 0][TOPK ? g : 0];
    };
    if (plain && !TOPK) {
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    klo[b][a][g] = fmax(__fma_rn(klo[b][a][g], 1.0 - c9, -Kb[b]) * (1.0 - 4.0 * u), 0.0);
    }
    unsigned total = 0;
#pragma unroll
    for (int b = 0; b < CB; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                total += (unsigned)__popcll(__ballot(lo_of(b, a, g) <= cur[b]));
    if (total) {
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(&hdr[0], total);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const double lo = lo_of(b, a, g);
                    const bool in = lo <= cur[b];
                    const unsigned long long m = __ballot(in);
                    if (m) {
                        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (in) {
                            if (pos < cap) {
                                if constexpr (TOPK) {
                                    PairEntryK e;
                                    e.s = sTile + wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                                    e.t = tTile + wn * 32 * CB + b * 32 + lr;
                                    e.key_lo = lo;
                                    e.key_hi = hi_of(b, a, g);
                                    reinterpret_cast<PairEntryK *>(list)[pos] = e;
                                } else {
                                    PairEntry e;
                                    e.s = sTile + wm * 64 + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                                    e.t = tTile + wn * 32 * CB + b * 32 + lr;
                                    e.key_lo = lo;
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
    __syncthreads();       // every wave is through the epilogue before the next tile rewrites the per-segment values in LDS
  }
    __builtin_amdgcn_s_waitcnt(rm_wait_vmcnt(0));          // (the last, unused DMA groups have long landed: the LDS they write
                                                           //  into must still be this workgroup's when they do)
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------
void refcos_q8_release(ssym_ctx *ctx, const SegmentSet &set)
{
    dev_free(ctx, set.q8);
    dev_free(ctx, set.q8_info);
    set.q8 = nullptr;
    set.q8_info = nullptr;
    set.q8_rows = set.q8_groups = 0;
    set.q8_state = 0;
}

int32_t refcos_q8_ensure(ssym_ctx *ctx, const SegmentSet &set)
{
    if (set.q8_state != 0)
        return SSYM_OK;
    set.q8_state = -1;
    const uint64_t maxLen = (uint64_t)set.max_frames * set.dim;
    if (set.n == 0 || maxLen == 0 || maxLen > kQMaxLen || !(set.max_abs < __builtin_inf()))
        return SSYM_OK;
    const uint32_t rows = (set.n + kQT - 1) / kQT * kQT;
    const uint32_t groups = (uint32_t)((maxLen + kQG - 1) / kQG);
    const uint64_t bytes = (uint64_t)rows * groups * 128;
    // every row is as long as the longest: a set of many short segments and one long one would pay for it
    const uint64_t rawBytes = set.total_frames * set.dim * sizeof(double);
    if (bytes > 2 * rawBytes + (64ull << 20) || (uint64_t)kQT * groups * 128 >= (1ull << 32))
        return SSYM_OK;
    int32_t rc = dev_alloc(ctx, (void **)&set.q8, bytes);
    if (rc == SSYM_OK)
        rc = dev_alloc(ctx, (void **)&set.q8_info, sizeof(double) * (4 * (size_t)rows + 1));      // (+ one word: "outside")
    if (rc != SSYM_OK) {
        refcos_q8_release(ctx, set);
        set.q8_state = -1;
        return rc == SSYM_E_NOMEM ? SSYM_OK : rc;      // (no room for the records: the f64 filter)
    }
    unsigned *bad = (unsigned *)(set.q8_info + 4 * (size_t)rows);
    SSYM_HIP_CHECK(ctx, hipMemsetAsync(set.q8, 0, bytes, ctx->stream));
    SSYM_HIP_CHECK(ctx, hipMemsetAsync(bad, 0, sizeof(unsigned), ctx->stream));
    refcos_q8_records_kernel<<<rows, 256, 0, ctx->stream>>>(set.raw, set.off, set.n, set.dim, groups, set.q8, set.q8_info, bad);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    unsigned hbad = 0;
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(&hbad, bad, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    set.q8_rows = rows;
    set.q8_groups = groups;
    if (hbad) {
        refcos_q8_release(ctx, set);
        set.q8_state = -1;
        return SSYM_OK;
    }
    set.q8_state = 1;
    return SSYM_OK;
}

bool refcos_q8_ready(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    // (read per call, not once: bench.py and the tests time and check both filters in one process)
    const char *knob = ssym_knob("SSYM_REFCOS_Q8");
    const bool off = knob && atoi(knob) == 0;
    if (off || ctx->metric != SSYM_METRIC_REFCOS || src.dim != tgt.dim)
        return false;
    if (ctx->stream_only && (src.q8_state == 0 || tgt.q8_state == 0))
        return false;                                  // (building the records synchronises: not inside a stream-only step)
    if (refcos_q8_ensure(ctx, src) != SSYM_OK || refcos_q8_ensure(ctx, tgt) != SSYM_OK)
        return false;
    return src.q8_state == 1 && tgt.q8_state == 1;
}

int32_t launch_refcos_q8_kernel(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt, const double *dist_dev,
                                unsigned long long *thr, uint32_t *hdr1, void *list1, uint32_t cap, uint32_t k_top,
                                double *sims)
{
    const uint32_t N = src.n, M = tgt.n;
    const uint32_t tilesX = (M + kQT - 1) / kQT, tilesY = (N + kQT - 1) / kQT;
    // persistent: one workgroup per CU (a multiple of 8 so that workgroup number & 7 stays the XCD), tile after tile
    const uint32_t tiles = tilesX * tilesY;
    static const bool persist = !(ssym_knob("SSYM_REFCOS_Q8_PERSIST") && atoi(ssym_knob("SSYM_REFCOS_Q8_PERSIST")) == 0);
    const uint32_t cus = (uint32_t)std::max(8, ctx->num_cus / 8 * 8);
    dim3 grid(persist ? std::min(tiles, cus) : tiles);
    hipStream_t st = ctx->stream;
    // four waves of 64 x 64 per workgroup (one per SIMD); SSYM_REFCOS_Q8_WAVES=8 takes eight of 64 x 32, two per SIMD, at
    // 1.5 x the operand reads per MFMA -- measured 2 % SLOWER (0.191 against 0.187 ms: what the waves wait for is not
    // hidden by a second wave), kept as a measurement switch
    const char *knob = ssym_knob("SSYM_REFCOS_Q8_WAVES");
    const bool four = !(knob && atoi(knob) == 8);
#define SSYM_Q8_LAUNCH(WS, TK, CBV, SIMS, KT)                                                                                  \
    refcos_q8_kernel<WS, TK, CBV><<<grid, CBV == 2 ? 256 : 512, 0, st>>>(src.q8, src.q8_info, src.off, src.norm, tgt.q8,       \
                                                                       tgt.q8_info, tgt.off, tgt.norm, N, M, src.dim,          \
                                                                       src.q8_groups, tgt.q8_groups, dist_dev, 1.0, thr, hdr1, \
                                                                       (PairEntry *)list1, cap, SIMS, KT, tilesX, tilesY)
    if (sims) {
        SSYM_Q8_LAUNCH(true, false, 2, sims, 1);
    } else if (k_top > 1) {
        if (four)
            SSYM_Q8_LAUNCH(false, true, 2, nullptr, k_top);
        else
            SSYM_Q8_LAUNCH(false, true, 1, nullptr, k_top);
    } else {
        if (four)
            SSYM_Q8_LAUNCH(false, false, 2, nullptr, 1);
        else
            SSYM_Q8_LAUNCH(false, false, 1, nullptr, 1);
    }
#undef SSYM_Q8_LAUNCH
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
