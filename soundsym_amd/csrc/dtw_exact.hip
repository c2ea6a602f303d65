// dtw_exact.hip -- exact f64 DTW for a list of (source, target) pairs, or for every pair.
//
// Role on the path: (1) re-scores the candidates the f32 MFMA filter leaves per target, so that
// the returned indices and costs are those of an f64 evaluation; (2) is the whole dtw path for
// shapes the filter does not cover (banded, > 128 source frames, dim > 13; DESIGN.md "limits").
//
// Arithmetic follows the definition in DESIGN.md operation by operation (the same order the CPU
// oracle uses): c(i,j) = sqrt(sum_k (a_ik - b_jk)^2), k ascending, sub / mul / add rounded
// separately in f64 (library is built with -ffp-contract=off); D(i,j) = c + min3.
//
// Mapping: one wave per pair.  Lane l owns row c0+l of a 64-row chunk and walks the
// anti-diagonals: at step tau it evaluates column j = tau - l.  D(i-1, j) arrives from lane l-1
// by a wave shuffle, D(i-1, j-1) is the value shuffled in one step earlier, D(i, j-1) is the
// lane's own previous value.  The bottom row of a chunk is handed to the next chunk through LDS.
// Both segments' frames are staged in LDS when they fit, else read through L1/L2.
#include "ssym_internal.hpp"

#include <algorithm>
#include <type_traits>

namespace ssym {

// value of lane - 1 (lane 0 keeps its own): one DPP move per half (wave_shr:1, gfx9 encoding 0x138)
// instead of a ds_bpermute round trip through the LDS crossbar -- the shuffle sits on the critical
// path of every anti-diagonal step
__device__ __forceinline__ double shfl_up1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <bool LDS_FRAMES>
__global__ __launch_bounds__(64) void dtw_exact_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, uint32_t nSrc,
    uint32_t nTgt, uint32_t dim, int band, int squared, const uint2 *__restrict__ pairs,
    const uint32_t *__restrict__ countDev, uint32_t maxPairs, uint32_t fbCap,
    double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // LDS frame rows are padded to an ODD number of doubles: lane l reads frame (tau - l), i.e. a
    // stride of `ld` doubles between lanes, and ds_read_b64 is conflict-free iff ld is odd
    // (ld = 40 would be a 16-way conflict)
    const uint32_t ld = LDS_FRAMES ? (dim | 1u) : dim;
    double *bound0 = smem;                  // [fbCap]
    double *bound1 = smem + fbCap;          // [fbCap]
    double *ldsA = smem + 2 * (size_t)fbCap;        // [64][ld]      (LDS_FRAMES only)
    double *ldsB = ldsA + 64 * (size_t)ld;           // [fbCap][ld]   (LDS_FRAMES only)

    const double INF = __builtin_inf();
    const int lane = threadIdx.x;

    uint64_t total;
    if (pairs) {
        uint32_t c = *countDev;
        total = c < maxPairs ? c : maxPairs;
    } else {
        total = (uint64_t)nSrc * nTgt;
    }

    for (uint64_t k = blockIdx.x; k < total; k += gridDim.x) {
        uint32_t s, t;
        if (pairs) {
            uint2 p = pairs[k];
            s = p.x;
            t = p.y;
        } else {
            s = (uint32_t)(k / nTgt);
            t = (uint32_t)(k % nTgt);
        }
        const int Fa = (int)(srcOff[s + 1] - srcOff[s]);
        const int Fb = (int)(tgtOff[t + 1] - tgtOff[t]);
        const double *a0 = srcRaw + srcOff[s] * dim;
        const double *b0 = tgtRaw + tgtOff[t] * dim;
        if (Fa == 0 || Fb == 0) {
            if (lane == 0)
                out[k] = INF;
            continue;
        }
        __syncthreads();   // previous pair's LDS reads are done
        const double *bsrc = b0;
        if (LDS_FRAMES) {
            for (int i = lane; i < Fb * (int)dim; i += 64)
                ldsB[(size_t)(i / (int)dim) * ld + (i % (int)dim)] = b0[i];
            bsrc = ldsB;
        }
        double result = INF;
        int chunk = 0;
        for (int c0 = 0; c0 < Fa; c0 += 64, ++chunk) {
            const int r = c0 + lane;
            const bool rowValid = r < Fa;
            const int rowsHere = min(64, Fa - c0);
            const double *arow = a0 + (size_t)(rowValid ? r : c0) * dim;
            if (LDS_FRAMES) {
                __syncthreads();   // previous chunk's A rows no longer read
                for (int i = lane; i < rowsHere * (int)dim; i += 64)
                    ldsA[(size_t)(i / (int)dim) * ld + (i % (int)dim)] = a0[(size_t)c0 * dim + i];
                arow = ldsA + (size_t)(rowValid ? lane : 0) * ld;
            }
            const double *boundPrev = (chunk & 1) ? bound0 : bound1;
            double *boundCur = (chunk & 1) ? bound1 : bound0;
            // columns this chunk can reach: all of them, or the band around its rows; the boundary
            // row it leaves behind is +inf wherever it does not compute
            int jlo = 0, jhi = Fb - 1;
            if (band >= 0) {
                jlo = max(0, c0 - band);
                jhi = min(Fb - 1, c0 + rowsHere - 1 + band);
            }
            for (int j = lane; j < Fb; j += 64)
                boundCur[j] = INF;
            __syncthreads();       // staged frames, cleared boundary, previous chunk's boundary row visible

            double mine = INF;      // D(r, j-1)
            double diagReg = INF;   // D(r-1, j-1)
            const int tauEnd = jhi + rowsHere;     // exclusive: lane l works on column tau - l
            for (int tau = jlo; tau < tauEnd; ++tau) {
                const int j = tau - lane;
                double fromAbove = shfl_up1(mine);        // D(r-1, j) for lanes >= 1
                double diagv = diagReg;
                if (lane == 0) {
                    if (c0 == 0) {
                        fromAbove = INF;
                        diagv = (j == 0) ? 0.0 : INF;     // virtual D(-1,-1) = 0
                    } else {
                        fromAbove = (j >= 0 && j < Fb) ? boundPrev[j] : INF;
                        diagv = (j >= 1 && j <= Fb) ? boundPrev[j - 1] : INF;
                    }
                }
                const bool active = rowValid && j >= 0 && j < Fb;
                if (active) {
                    double cur = INF;
                    const int dij = r - j;
                    if (band < 0 || (dij <= band && -dij <= band)) {
                        const double *bj = bsrc + (size_t)j * ld;
                        // sum_k (a_k - b_k)^2 with k ascending (the oracle's order); the loads and the
                        // sub / mul of a block of eight are independent, only the adds are a chain
                        double acc = 0.0;
                        uint32_t e = 0;
                        for (; e + 8 <= dim; e += 8) {
                            // all sixteen loads of the block first, then the arithmetic: left alone the
                            // compiler waits on LDS after every two values
                            double av[8], bw[8];
#pragma unroll
                            for (int v = 0; v < 8; ++v) {
                                av[v] = arow[e + v];
                                bw[v] = bj[e + v];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int v = 0; v < 8; ++v) {
                                const double df = __dsub_rn(av[v], bw[v]);
                                acc = __dadd_rn(acc, __dmul_rn(df, df));
                            }
                        }
                        for (; e < dim; ++e) {
                            const double df = __dsub_rn(arow[e], bj[e]);
                            acc = __dadd_rn(acc, __dmul_rn(df, df));
                        }
                        const double c = squared ? acc : sqrt(acc);
                        double best = fromAbove;              // D(i-1, j)
                        if (mine < best) best = mine;         // D(i,   j-1)
                        if (diagv < best) best = diagv;       // D(i-1, j-1)
                        cur = __dadd_rn(c, best);
                    }
                    if (lane == 63)
                        boundCur[j] = cur;
                    if (r == Fa - 1 && j == Fb - 1)
                        result = cur;
                    mine = cur;
                }
                diagReg = fromAbove;
            }
        }
        if ((Fa - 1) % 64 == lane)
            out[k] = result;
    }
}

// Variant for frames of at most DIMR values (instantiated for 12, 14, 16, 40 and 48: the 12 / 13
// coefficients of the reference and the north star, and the 40-dimensional configuration).  Same wavefront,
// same operation order, but
//   * the lane's own source frame sits in REGISTERS for the whole chunk (it was re-read from LDS at
//     every step), padded with zeros up to DIMR;
//   * target frames are staged in LDS zero-padded to DIMR values, rows of LD = 2 (mod 4) doubles
//     (16-byte aligned and, for 128-bit reads by consecutive lanes, bank-conflict free), so a step issues
//     DIMR / 2 unconditional ds_read_b128 back to back -- the generic kernel waited for LDS after
//     every two values -- and the loads of step tau + 1 are in flight while step tau computes;
//   * (0 - 0)^2 = +0.0 added to a non-negative sum leaves it unchanged bit for bit, so the padding
//     does not alter the result.
//   * BT = float when the context's features were f32 (they were widened exactly, so narrowing the staged
//     target frames back is exact too): half the LDS per pair means twice the resident waves, and the
//     short candidate lists this kernel usually sees are bound by latency, not by issue.
template <int DIMR>
constexpr int exact_ld(bool f32)
{
    // row stride in elements: 16-byte aligned rows whose 128-bit reads by consecutive lanes tile the banks
    return f32 ? ((DIMR + 3) / 4 * 4) + ((((DIMR + 3) / 4 * 4) % 8 == 4) ? 0 : 4)      // = 4 (mod 8) floats
               : ((DIMR % 4 == 2) ? DIMR : DIMR + 2);                                     // = 2 (mod 4) doubles
}

template <int DIMR, int PARTS = 1, typename BT = double>
__global__ __launch_bounds__(64) void dtw_exact_reg_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, uint32_t nSrc,
    uint32_t nTgt, uint32_t dim, int band, int squared, const uint2 *__restrict__ pairs,
    const uint32_t *__restrict__ countDev, uint32_t maxPairs, uint32_t fbCap,
    double *__restrict__ out, uint64_t totalLo = 0, uint64_t totalHi = ~0ull,
    const unsigned *__restrict__ redoFlag = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool BF32 = sizeof(BT) == 4;
    constexpr int LD = exact_ld<DIMR>(BF32);
    constexpr int VPR = 16 / (int)sizeof(BT);    // values per 128-bit LDS read
    double *bound0 = smem;                       // [fbCap]
    double *bound1 = smem + fbCap;               // [fbCap]
    BT *ldsB = reinterpret_cast<BT *>(smem + 2 * (size_t)fbCap);     // [rows][LD], fbCap is even so rows stay 16-byte aligned
    const double INF = __builtin_inf();
    const int lane = threadIdx.x;

    uint64_t total;
    if (pairs) {
        uint32_t c = *countDev;
        total = c < maxPairs ? c : maxPairs;
    } else {
        total = (uint64_t)nSrc * nTgt;
    }
    // the list length decides between this kernel and its pipelined sibling (see the launcher) -- unless a wave of
    // the sibling gave up waiting (its failure count is not zero): then this kernel scores the whole list again
    const bool redo = redoFlag && *redoFlag != 0;
    if (!redo && (total < totalLo || total > totalHi))
        return;
    for (uint64_t k = blockIdx.x; k < total; k += gridDim.x) {
        uint32_t s, t;
        if (pairs) {
            uint2 p = pairs[k];
            s = p.x;
            t = p.y;
        } else {
            s = (uint32_t)(k / nTgt);
            t = (uint32_t)(k % nTgt);
        }
        const int Fa = (int)(srcOff[s + 1] - srcOff[s]);
        const int Fb = (int)(tgtOff[t + 1] - tgtOff[t]);
        const double *a0 = srcRaw + srcOff[s] * dim;
        const double *b0 = tgtRaw + tgtOff[t] * dim;
        if (Fa == 0 || Fb == 0) {
            if (lane == 0)
                out[k] = INF;
            continue;
        }
        __syncthreads();   // previous pair's LDS reads are done
        if (band < 0)
            for (int i = lane; i < Fb * DIMR; i += 64) {
                const int fr = i / DIMR, e = i % DIMR;
                ldsB[(size_t)fr * LD + e] = (BT)(e < (int)dim ? b0[(size_t)fr * dim + e] : 0.0);
            }
        double result = INF;
        int chunk = 0;
        for (int c0 = 0; c0 < Fa; c0 += 64, ++chunk) {
            const int r = c0 + lane;
            const bool rowValid = r < Fa;
            const int rowsHere = min(64, Fa - c0);
            double ar[DIMR];
            {
                const double *arow = a0 + (size_t)(rowValid ? r : c0) * dim;
#pragma unroll
                for (int e = 0; e < DIMR; ++e)
                    ar[e] = e < (int)dim ? arow[e] : 0.0;
            }
            const double *boundPrev = (chunk & 1) ? bound0 : bound1;
            double *boundCur = (chunk & 1) ? bound1 : bound0;
            int jlo = 0, jhi = Fb - 1;
            if (band >= 0) {
                jlo = max(0, c0 - band);
                jhi = min(Fb - 1, c0 + rowsHere - 1 + band);
            }
            // banded: only the columns this chunk can reach are staged (frame j at row j - wlo), so long
            // targets with wide frames still fit a few workgroups per CU
            const int wlo = band >= 0 ? jlo : 0, whi = band >= 0 ? jhi : Fb - 1;
            if (band >= 0) {
                __syncthreads();   // the previous chunk's window is no longer read
                for (int i = lane; i < (whi - wlo + 1) * DIMR; i += 64) {
                    const int fr = i / DIMR, e = i % DIMR;
                    ldsB[(size_t)fr * LD + e] = (BT)(e < (int)dim ? b0[(size_t)(wlo + fr) * dim + e] : 0.0);
                }
            }
            for (int j = lane; j < Fb; j += 64)
                boundCur[j] = INF;
            __syncthreads();       // staged frames, cleared boundary, previous chunk's boundary row visible

            double mine = INF;      // D(r, j-1)
            double diagReg = INF;   // D(r-1, j-1)
            const int tauEnd = jhi + rowsHere;     // exclusive: lane l works on column tau - l
            typedef BT d2 __attribute__((ext_vector_type(VPR)));        // one 128-bit read
            constexpr int PW = DIMR / PARTS;              // values fetched at a time
            constexpr int NV = (PW + VPR - 1) / VPR;      // reads per fetch (a row's padding covers the overhang)
            // (PARTS == 1: the frames of this step and of the next one live in two register sets that swap roles from
            // step to step -- copying the prefetched frame over cost NV x 4 v_mov per step, 18 % of the VALU work at 40 values)
            d2 bA[NV], bB[PARTS == 1 ? NV : 1];
            if (PARTS == 1) {
                const int jc = min(max(jlo - lane, wlo), whi) - wlo;
                const d2 *bp = reinterpret_cast<const d2 *>(ldsB + (size_t)jc * LD);
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    bA[e] = bp[e];
            }
            auto step = [&](const int tau, d2 (&bv)[NV], d2 (&bn)[PARTS == 1 ? NV : 1]) {
                const int j = tau - lane;
                if (PARTS == 1) {
                    // next step's frame (clamped to a valid row; unused when out of range)
                    const int jc = min(max(j + 1, wlo), whi) - wlo;
                    const d2 *bp = reinterpret_cast<const d2 *>(ldsB + (size_t)jc * LD);
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        bn[e] = bp[e];
                }
                double fromAbove = shfl_up1(mine);        // D(r-1, j) for lanes >= 1
                double diagv = diagReg;
                if (lane == 0) {
                    if (c0 == 0) {
                        fromAbove = INF;
                        diagv = (j == 0) ? 0.0 : INF;     // virtual D(-1,-1) = 0
                    } else {
                        fromAbove = (j >= 0 && j < Fb) ? boundPrev[j] : INF;
                        diagv = (j >= 1 && j <= Fb) ? boundPrev[j - 1] : INF;
                    }
                }
                // sum_k (a_k - b_k)^2, k ascending, sub / mul / add rounded separately (the oracle's order)
                double acc = 0.0;
                if (PARTS == 1) {
#pragma unroll
                    for (int e = 0; e < DIMR; ++e) {
                        const double df = __dsub_rn(ar[e], (double)bv[e / VPR][e % VPR]);
                        acc = __dadd_rn(acc, __dmul_rn(df, df));
                    }
                } else {
                    // frames too wide to double-buffer in registers: this step's frame in PARTS fetches
                    const int jc = min(max(j, wlo), whi) - wlo;
#pragma unroll
                    for (int h = 0; h < PARTS; ++h) {
                        const d2 *bp = reinterpret_cast<const d2 *>(ldsB + (size_t)jc * LD + h * PW);
#pragma unroll
                        for (int e = 0; e < NV; ++e)
                            bv[e] = bp[e];
#pragma unroll
                        for (int e = 0; e < PW; ++e) {
                            const double df = __dsub_rn(ar[h * PW + e], (double)bv[e / VPR][e % VPR]);
                            acc = __dadd_rn(acc, __dmul_rn(df, df));
                        }
                    }
                }
                const double c = squared ? acc : sqrt(acc);
                const bool active = rowValid && j >= 0 && j < Fb;
                if (active) {
                    double cur = INF;
                    const int dij = r - j;
                    if (band < 0 || (dij <= band && -dij <= band)) {
                        double best = fromAbove;              // D(i-1, j)
                        if (mine < best) best = mine;         // D(i,   j-1)
                        if (diagv < best) best = diagv;       // D(i-1, j-1)
                        cur = __dadd_rn(c, best);
                    }
                    if (lane == 63)
                        boundCur[j] = cur;
                    if (r == Fa - 1 && j == Fb - 1)
                        result = cur;
                    mine = cur;
                }
                diagReg = fromAbove;
            };
            if constexpr (PARTS == 1 && DIMR > 16) {
                int tau = jlo;
                for (; tau + 1 < tauEnd; tau += 2) {
                    step(tau, bA, bB);
                    step(tau + 1, bB, bA);
                }
                if (tau < tauEnd)
                    step(tau, bA, bB);
            } else if constexpr (PARTS == 1) {
                // narrow frames: the unrolled pair of steps costs more registers (a wave per SIMD) than the copy costs cycles
                for (int tau = jlo; tau < tauEnd; ++tau) {
                    step(tau, bA, bB);
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        bA[e] = bB[e];
                }
            } else {
                for (int tau = jlo; tau < tauEnd; ++tau)
                    step(tau, bA, bB);
            }
        }
        if ((Fa - 1) % 64 == lane)
            out[k] = result;
    }
}

// Row chunks PIPELINED over the waves of a workgroup (sources of 65...512 frames): wave w owns chunk w
// and starts as soon as the wave above has finished the first columns of its bottom row, instead of one
// wave walking the chunks one after the other -- the steps in sequence drop from chunks x (Fb + 64) to
// about Fb + 80 x chunks, which is what a short candidate list (early abandoning's M pairs, list 2) is bound
// by.  Same arithmetic, same order per cell.  Hand-over through LDS: the producer's lane 63 writes
// D(bottom row, j) and then publishes "columns < p are final" (prog[w]); the consumer reads prog[w - 1]
// only when it has caught up and then waits for 16 columns more than it needs, so it polls once per 16
// steps.  Nobody waits for a wave BELOW it, wave 0 waits for nobody, and every wave publishes "all
// final" when it leaves its chunk, so the chain always drains; the spin is bounded all the same, and a
// wave that gives up counts itself in `failCount` instead of hanging the GPU: the one-wave-per-pair kernel,
// launched right behind this one, reads the count and, when it is not zero, scores the whole list again
// (no host in between; ssym_timings.exact_redone reports it).  SSYM_EXACT_PIPE_FORCE_GIVEUP=1 makes wave 1
// of the first pair give up at once (tests).
constexpr int kPipeLag = 16;
constexpr int kPipeDone = 0x7fffffff;

template <int DIMR, typename BT>
__global__ __launch_bounds__(512) void dtw_exact_pipe_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, uint32_t nSrc,
    uint32_t nTgt, uint32_t dim, int band, int squared, const uint2 *__restrict__ pairs,
    const uint32_t *__restrict__ countDev, uint32_t maxPairs, uint32_t fbCap,
    double *__restrict__ out, uint64_t totalLo, uint64_t totalHi, unsigned *__restrict__ failCount, int forceGiveUp)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool BF32 = sizeof(BT) == 4;
    constexpr int LD = exact_ld<DIMR>(BF32);
    constexpr int VPR = 16 / (int)sizeof(BT);
    constexpr int NV = (DIMR + VPR - 1) / VPR;
    typedef BT bvec __attribute__((ext_vector_type(VPR)));
    const int W = (int)(blockDim.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    double *bound = smem;                                                  // [W][fbCap]
    BT *ldsB = reinterpret_cast<BT *>(smem + (size_t)W * fbCap);           // [fbCap][LD]  (fbCap even: 16-byte aligned)
    volatile int *prog = reinterpret_cast<volatile int *>(ldsB + (size_t)fbCap * LD);      // [W], then the pair's failure flag
    volatile int *failed = prog + W;
    const double INF = __builtin_inf();

    uint64_t total;
    if (pairs) {
        uint32_t c = *countDev;
        total = c < maxPairs ? c : maxPairs;
    } else {
        total = (uint64_t)nSrc * nTgt;
    }
    if (total < totalLo || total > totalHi)      // the list length decides between this kernel and its sibling (see the launcher)
        return;
    for (uint64_t k = blockIdx.x; k < total; k += gridDim.x) {
        uint32_t s, t;
        if (pairs) {
            uint2 p = pairs[k];
            s = p.x;
            t = p.y;
        } else {
            s = (uint32_t)(k / nTgt);
            t = (uint32_t)(k % nTgt);
        }
        const int Fa = (int)(srcOff[s + 1] - srcOff[s]);
        const int Fb = (int)(tgtOff[t + 1] - tgtOff[t]);
        const double *a0 = srcRaw + srcOff[s] * dim;
        const double *b0 = tgtRaw + tgtOff[t] * dim;
        if (Fa == 0 || Fb == 0) {                    // block-uniform
            if (threadIdx.x == 0)
                out[k] = INF;
            continue;
        }
        __syncthreads();                             // the previous pair's frames and rows are no longer read
        for (int i = threadIdx.x; i < Fb * DIMR; i += blockDim.x) {
            const int fr = i / DIMR, e = i % DIMR;
            ldsB[(size_t)fr * LD + e] = (BT)(e < (int)dim ? b0[(size_t)fr * dim + e] : 0.0);
        }
        for (int i = threadIdx.x; i < W * (int)fbCap; i += blockDim.x)
            bound[i] = INF;
        if (lane == 0)
            prog[wave] = 0;
        if (threadIdx.x == 0)
            *failed = 0;
        __syncthreads();

        const int c0 = wave * 64;
        if (c0 < Fa) {                               // this wave has a chunk (the host sized W for the longest source)
            const int r = c0 + lane;
            const bool rowValid = r < Fa;
            const int rowsHere = min(64, Fa - c0);
            double ar[DIMR];
            {
                const double *arow = a0 + (size_t)(rowValid ? r : c0) * dim;
#pragma unroll
                for (int e = 0; e < DIMR; ++e)
                    ar[e] = e < (int)dim ? arow[e] : 0.0;
            }
            const double *boundPrev = bound + (size_t)(wave > 0 ? wave - 1 : 0) * fbCap;
            double *boundCur = bound + (size_t)wave * fbCap;
            int jlo = 0, jhi = Fb - 1;
            if (band >= 0) {
                jlo = max(0, c0 - band);
                jhi = min(Fb - 1, c0 + rowsHere - 1 + band);
            }
            double mine = INF, diagReg = INF, result = INF;
            const int tauEnd = jhi + rowsHere;       // exclusive: lane l works on column tau - l
            bvec bv[NV], bn[NV];
            {
                const int jc = min(max(jlo - lane, 0), Fb - 1);
                const bvec *bp = reinterpret_cast<const bvec *>(ldsB + (size_t)jc * LD);
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    bn[e] = bp[e];
            }
            int known = wave > 0 ? 0 : kPipeDone;    // columns of the row above known to be final
            bool gaveUp = false;
            for (int tau = jlo; tau < tauEnd; ++tau) {
                const int j = tau - lane;
                const int need = min(tau + 1, Fb);   // lane 0 reads boundPrev[tau] and boundPrev[tau - 1]
                if (forceGiveUp && k == 0 && wave == 1 && tau == jlo) {      // test knob: the give-up path, once
                    gaveUp = true;
                    break;
                }
                if (known < need) {                  // wave-uniform
                    const int want = min(need + kPipeLag, Fb);
                    int spins = 0;
                    for (;;) {
                        known = __builtin_amdgcn_readfirstlane(prog[wave - 1]);
                        if (known >= want)
                            break;
                        if (++spins > (1 << 22)) {
                            gaveUp = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (gaveUp)
                        break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // the row above is read after its progress
                }
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    bv[e] = bn[e];
                {
                    const int jc = min(max(j + 1, 0), Fb - 1);
                    const bvec *bp = reinterpret_cast<const bvec *>(ldsB + (size_t)jc * LD);
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        bn[e] = bp[e];
                }
                double fromAbove = shfl_up1(mine);        // D(r-1, j) for lanes >= 1
                double diagv = diagReg;
                if (lane == 0) {
                    if (c0 == 0) {
                        fromAbove = INF;
                        diagv = (j == 0) ? 0.0 : INF;     // virtual D(-1,-1) = 0
                    } else {
                        fromAbove = (j >= 0 && j < Fb) ? boundPrev[j] : INF;
                        diagv = (j >= 1 && j <= Fb) ? boundPrev[j - 1] : INF;
                    }
                }
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < DIMR; ++e) {
                    const double df = __dsub_rn(ar[e], (double)bv[e / VPR][e % VPR]);
                    acc = __dadd_rn(acc, __dmul_rn(df, df));
                }
                const double c = squared ? acc : sqrt(acc);
                const bool active = rowValid && j >= 0 && j < Fb;
                if (active) {
                    double cur = INF;
                    const int dij = r - j;
                    if (band < 0 || (dij <= band && -dij <= band)) {
                        double best = fromAbove;              // D(i-1, j)
                        if (mine < best) best = mine;         // D(i,   j-1)
                        if (diagv < best) best = diagv;       // D(i-1, j-1)
                        cur = __dadd_rn(c, best);
                    }
                    if (lane == 63)
                        boundCur[j] = cur;
                    if (r == Fa - 1 && j == Fb - 1)
                        result = cur;
                    mine = cur;
                }
                diagReg = fromAbove;
                // lane 63 has just finished column tau - 63 (columns left of the band stay +inf and count as final)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // ... and its value is stored before that is said
                if (lane == 63)
                    prog[wave] = tau - 62;
            }
            if (gaveUp && lane == 0) {
                *failed = 1;
                atomicAdd(failCount, 1u);            // the launcher's next kernel sees it and scores the list again
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 63)
                prog[wave] = kPipeDone;              // also when this wave gave up: the waves below must not wait for it
            if ((Fa - 1) / 64 == wave && (Fa - 1) % 64 == lane)      // (the last chunk ends after every chunk above it)
                out[k] = (gaveUp || *failed) ? __builtin_nan("") : result;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-phase variant (round 3): the local costs of a 64-row chunk FIRST, by all four waves of a workgroup and without
// any dependency between cells, the recurrence AFTERWARDS, by one wave over costs that are already there.
//
// The kernels above evaluate c(i,j) inside the anti-diagonal step: ~220 instructions per step for 13 values (3 f64
// operations per value, the f64 square root, conversions, LDS reads), of which the recurrence itself is ~15 -- and they
// run on a wavefront that is two thirds full without a band (a 64-row chunk needs Fb + 63 steps for Fb columns) and one
// third full inside one (lane l is busy for 2r + 1 of the chunk's 2r + 127 steps).  Splitting the two
//   * takes the local costs off the dependent chain: a short candidate list (what a rank of a source-sharded step,
//     or early abandoning's M candidates, hands over) is bound by the LENGTH of a pair's chain, and the chain is now
//     ~15 instructions per step instead of ~220;
//   * computes only the cells that exist: 64 x Fb per chunk without a band, 64 x (2r + 1) inside one (in diagonal
//     coordinates x = j - i + r), every lane busy -- 1.5x / 3x fewer instructions than on the wavefront.
// Same operations in the same order per cell as the kernels above and as the oracle (k ascending, sub / mul / add and
// the square root rounded separately, then c + min3 with the same comparisons), so the bits are the same.
//
// LDS: cells[x][lane] f64 for one PANEL of at most 128 x-values (64 KB; x = column inside the panel, or diagonal), the
// staged target frames (zero-padded to DIMR, f32 when the context's features are), two boundary rows.  Targets
// longer than a panel are swept panel by panel: the wavefront of a panel ends with every lane on the panel's last
// column, `mine` (D(i, j-1)) simply stays in its register, and the first step of the next panel finds D(i-1, j-1) in
// the shuffle of the step before exactly as inside a panel -- no extra state.
template <int DIMR, typename BT>
__global__ __launch_bounds__(256) void dtw_exact_cells_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff,
    const double *__restrict__ tgtRaw, const uint64_t *__restrict__ tgtOff, uint32_t nSrc,
    uint32_t nTgt, uint32_t dim, int band, int squared, const uint2 *__restrict__ pairs,
    const uint32_t *__restrict__ countDev, uint32_t maxPairs, uint32_t fbCap, uint32_t panelX,
    double *__restrict__ out, uint64_t totalHi)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool BF32 = sizeof(BT) == 4;
    constexpr int LD = exact_ld<DIMR>(BF32);
    constexpr int VPR = 16 / (int)sizeof(BT);    // values per 128-bit LDS read
    constexpr int NV = (DIMR + VPR - 1) / VPR;
    typedef BT bvec __attribute__((ext_vector_type(VPR)));
    double *bound0 = smem;                       // [fbCap]
    double *bound1 = smem + fbCap;               // [fbCap]
    double *cells = smem + 2 * (size_t)fbCap;    // [panelX][64]
    BT *ldsB = reinterpret_cast<BT *>(cells + (size_t)panelX * 64);     // [rows][LD] (16-byte aligned: fbCap is even)
    const double INF = __builtin_inf();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool banded = band >= 0;               // block-uniform

    uint64_t total;
    if (pairs) {
        uint32_t c = *countDev;
        total = c < maxPairs ? c : maxPairs;
    } else {
        total = (uint64_t)nSrc * nTgt;
    }
    if (total > totalHi)        // the list length (it lives on the device) decides between this kernel and its siblings
        return;
    for (uint64_t k = blockIdx.x; k < total; k += gridDim.x) {
        uint32_t s, t;
        if (pairs) {
            uint2 p = pairs[k];
            s = p.x;
            t = p.y;
        } else {
            s = (uint32_t)(k / nTgt);
            t = (uint32_t)(k % nTgt);
        }
        const int Fa = (int)(srcOff[s + 1] - srcOff[s]);
        const int Fb = (int)(tgtOff[t + 1] - tgtOff[t]);
        const double *a0 = srcRaw + srcOff[s] * dim;
        const double *b0 = tgtRaw + tgtOff[t] * dim;
        if (Fa == 0 || Fb == 0) {                // block-uniform
            if (threadIdx.x == 0)
                out[k] = INF;
            continue;
        }
        __syncthreads();                         // the previous pair's LDS is no longer read
        if (!banded)
            for (int i = threadIdx.x; i < Fb * DIMR; i += 256) {
                const int fr = i / DIMR, e = i % DIMR;
                ldsB[(size_t)fr * LD + e] = (BT)(e < (int)dim ? b0[(size_t)fr * dim + e] : 0.0);
            }
        double result = INF;
        int chunk = 0;
        for (int c0 = 0; c0 < Fa; c0 += 64, ++chunk) {
            const int r = c0 + lane;
            const bool rowValid = r < Fa;
            const int rowsHere = min(64, Fa - c0);
            double ar[DIMR];                     // the lane's own source frame (every wave holds the chunk's rows)
            {
                const double *arow = a0 + (size_t)(rowValid ? r : c0) * dim;
#pragma unroll
                for (int e = 0; e < DIMR; ++e)
                    ar[e] = e < (int)dim ? arow[e] : 0.0;
            }
            const double *boundPrev = (chunk & 1) ? bound0 : bound1;
            double *boundCur = (chunk & 1) ? bound1 : bound0;
            int jlo = 0, jhi = Fb - 1;
            if (banded) {
                jlo = max(0, c0 - band);
                jhi = min(Fb - 1, c0 + rowsHere - 1 + band);
            }
            const int wlo = banded ? jlo : 0, whi = banded ? jhi : Fb - 1;     // staged frames: rows wlo..whi
            // wave 0 may still be walking the previous chunk's last panel: the window it reads from (banded) and the
            // boundary row it reads (this chunk's boundCur is that chunk's boundPrev) change only after it is done
            __syncthreads();
            if (banded) {
                for (int i = threadIdx.x; i < (whi - wlo + 1) * DIMR; i += 256) {
                    const int fr = i / DIMR, e = i % DIMR;
                    ldsB[(size_t)fr * LD + e] = (BT)(e < (int)dim ? b0[(size_t)(wlo + fr) * dim + e] : 0.0);
                }
            }
            for (int j = threadIdx.x; j < Fb; j += 256)
                boundCur[j] = INF;
            // x of a cell: its column inside the panel, or (banded) its diagonal j - i + band in 0 .. 2 band
            const int nPanels = banded ? 1 : (Fb + (int)panelX - 1) / (int)panelX;
            double mine = INF;                   // D(r, j-1)      } the recurrence's state, in wave 0
            double diagReg = INF;                // D(r-1, j-1)    }
            for (int p = 0; p < nPanels; ++p) {
                const int j0 = p * (int)panelX;
                const int Xp = banded ? 2 * band + 1 : min((int)panelX, Fb - j0);
                const int xorg = banded ? r - band : j0;          // j = xorg + x
                __syncthreads();                 // frames staged; the cells of the previous panel have been consumed
                // ---- phase 1: the panel's local costs, x = wave, wave + 4, ... (no cell waits for another) ----
                // (two cells per trip, their sums interleaved by the scheduler: one cell alone is a chain of DIMR dependent
                //  additions and the square root's refinement steps)
                auto frame_of = [&](int x, bvec (&bv)[NV]) {
                    const int jc = min(max(xorg + x, wlo), whi) - wlo;        // (a clamped cell is never used)
                    const bvec *bp = reinterpret_cast<const bvec *>(ldsB + (size_t)jc * LD);
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        bv[e] = bp[e];
                };
                for (int x = wave; x < Xp; x += 8) {
                    const bool two = x + 4 < Xp;                              // wave-uniform
                    bvec bv0[NV], bv1[NV];
                    frame_of(x, bv0);
                    frame_of(two ? x + 4 : x, bv1);
                    // sum_k (a_k - b_k)^2, k ascending, sub / mul / add rounded separately (the oracle's order);
                    // (0 - 0)^2 = +0.0 added to a non-negative sum changes nothing: the padding is invisible
                    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                    for (int e = 0; e < DIMR; ++e) {
                        const double d0 = __dsub_rn(ar[e], (double)bv0[e / VPR][e % VPR]);
                        const double d1 = __dsub_rn(ar[e], (double)bv1[e / VPR][e % VPR]);
                        acc0 = __dadd_rn(acc0, __dmul_rn(d0, d0));
                        acc1 = __dadd_rn(acc1, __dmul_rn(d1, d1));
                    }
                    cells[(size_t)x * 64 + lane] = squared ? acc0 : sqrt(acc0);
                    if (two)
                        cells[(size_t)(x + 4) * 64 + lane] = squared ? acc1 : sqrt(acc1);
                }
                __syncthreads();
                // ---- phase 2: the recurrence over the panel, one wave, lane l on row c0 + l, column tau - l ----
                // What a step depends on is the chain  D(r-1, j) of the lane above (one DPP move) -> min3 -> add, so
                // everything else is taken off it: the step's cost and lane 0's row-above value are read from LDS one
                // step ahead, lane 0's diagonal is its row-above value of the step before like every other lane's, and
                // min(D(i-1, j), D(i-1, j-1)) is formed before D(i, j-1) joins.  (Order of the comparisons: the oracle
                // starts from D(i-1, j) and takes D(i, j-1), then D(i-1, j-1), each if smaller.  Taking D(i-1, j-1)
                // first gives the same value in every case: a NaN in the first place stays, a NaN elsewhere is skipped,
                // and equal numbers are the same number.)
                if (wave == 0) {
                    const int tauLo = banded ? jlo : j0;
                    const int tauHi = banded ? jhi + rowsHere : j0 + Xp + rowsHere - 1;      // exclusive
                    const bool isLane0 = lane == 0, isLane63 = lane == 63;
                    const double *cl = cells + lane;
                    // FIRST: the chunk starts at row 0 (nothing above it); BANDED: compile-time copies of the two
                    // block-uniform flags, so that the loop body is straight-line code
                    auto walk = [&](auto FIRST, auto BANDED) {
                        constexpr bool kFirst = decltype(FIRST)::value, kBanded = decltype(BANDED)::value;
                        int x = tauLo - lane - xorg;
                        double cv = cl[(size_t)min(max(x, 0), Xp - 1) * 64];
                        // lane 0's D(r-1, tau): the boundary row, read one step ahead and masked when it is used
                        double bRaw = kFirst ? INF : boundPrev[min(tauLo, Fb - 1)];
                        if (isLane0)                                                            // lane 0's D(r-1, tau-1)
                            diagReg = kFirst ? (tauLo == 0 ? 0.0 : INF)
                                             : ((tauLo >= 1 && tauLo <= Fb) ? boundPrev[tauLo - 1] : INF);
#pragma unroll 2
                        for (int tau = tauLo; tau < tauHi; ++tau) {
                            const double c = cv;
                            cv = cl[(size_t)min(max(x + 1, 0), Xp - 1) * 64];               // the next step's cost
                            const double bA = (kFirst || tau >= Fb) ? INF : bRaw;
                            if (!kFirst)
                                bRaw = boundPrev[min(tau + 1, Fb - 1)];
                            double fromAbove = shfl_up1(mine);        // D(r-1, j) for lanes >= 1
                            fromAbove = isLane0 ? bA : fromAbove;
                            double best = fromAbove;                  // D(i-1, j)
                            if (diagReg < best) best = diagReg;       // D(i-1, j-1)
                            if (mine < best) best = mine;             // D(i,   j-1)
                            const double cur = __dadd_rn(c, best);
                            const bool inX = (unsigned)x < (unsigned)Xp;      // in the panel / inside the band
                            const int j = x + xorg;
                            // without a band a lane outside the panel holds still (its D(r, j-1) is the previous panel's
                            // last column); inside a band the cells of the row beyond the band ARE +inf, as in the
                            // kernels above
                            const bool active = rowValid && (kBanded ? (unsigned)j < (unsigned)Fb : inX);
                            const double next = (kBanded && !inX) ? INF : cur;
                            mine = active ? next : mine;
                            if (isLane63 && active)
                                boundCur[j] = next;
                            diagReg = fromAbove;
                            ++x;
                        }
                    };
                    if (c0 == 0) {
                        if (banded)
                            walk(std::true_type{}, std::true_type{});
                        else
                            walk(std::true_type{}, std::false_type{});
                    } else {
                        if (banded)
                            walk(std::false_type{}, std::true_type{});
                        else
                            walk(std::false_type{}, std::false_type{});
                    }
                    // (after its last column a lane holds still, so the last row's lane ends on D(Fa-1, Fb-1); when that
                    //  cell lies outside the band the chunk never reaches its column, and the pair's cost is +inf)
                    result = (!banded || abs(Fa - Fb) <= band) ? mine : INF;
                }
            }
        }
        if (wave == 0 && (Fa - 1) % 64 == lane)
            out[k] = result;
    }
}

// ---------------------------------------------------------------------------------------------
// A handful of short queries against a dictionary of short entries in ONE launch (ssym_match_one / small
// ssym_match_batch calls with the dtw metric: the reference's one-query-at-a-time pattern).  One wave per
// (entry, query) pair: lane = entry frame (entries of at most 64 frames: one chunk, no boundary rows), the
// entry's frame in registers, the query's frames zero-padded to 16 values in LDS, the same wavefront and the
// same operations per cell as the kernels above; then |cost - distance| is folded to the first minimum per
// workgroup, and the last workgroup of a query folds the workgroups in index order and writes index and cost
// into pinned memory.  The queries are read from the pinned staging window: no pack, no copies.
constexpr int kFewDim = 16;                 // values per frame (zero-padded)
constexpr int kFewQVals = 2048;             // query frames x 16 held in LDS
constexpr int kFewWaves = 16;               // entries per workgroup (every workgroup pulls the query over PCIe: few, large ones)

__global__ __launch_bounds__(64 * kFewWaves) void dtw_match_few_kernel(
    const double *__restrict__ srcRaw, const uint64_t *__restrict__ srcOff, uint32_t nSrc, uint32_t dim,
    const void *__restrict__ queries, const uint64_t *__restrict__ qOff, int queryIsF32,
    const double *__restrict__ distances, int band, int squared, double *__restrict__ partKeyAll,
    double *__restrict__ partCostAll, uint32_t *__restrict__ partIdxAll, unsigned *__restrict__ tickets,
    uint32_t *__restrict__ outIdxAll, double *__restrict__ outCostAll)
{
    __shared__ __attribute__((aligned(16))) double sq[kFewQVals];
    __shared__ double ck[kFewWaves], cc[kFewWaves];
    __shared__ bool last;
    const double INF = __builtin_inf();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t y = blockIdx.y;
    const int Fb = (int)(qOff[y + 1] - qOff[y]);
    const unsigned long long qBase = qOff[y] * dim;
    for (int i = tid; i < Fb * kFewDim; i += 64 * kFewWaves) {
        const int fr = i / kFewDim, e = i % kFewDim;
        double v = 0.0;
        if (e < (int)dim)
            v = queryIsF32 ? (double)static_cast<const float *>(queries)[qBase + (size_t)fr * dim + e]
                           : static_cast<const double *>(queries)[qBase + (size_t)fr * dim + e];
        sq[i] = v;
    }
    __syncthreads();

    const uint32_t s = blockIdx.x * kFewWaves + wave;
    const int Fa = s < nSrc ? (int)(srcOff[s + 1] - srcOff[s]) : 0;
    double result = INF;
    if (Fa > 0 && Fb > 0) {
        const bool rowValid = lane < Fa;
        double ar[kFewDim];
        {
            const double *arow = srcRaw + (srcOff[s] + (rowValid ? lane : 0)) * dim;
#pragma unroll
            for (int e = 0; e < kFewDim; ++e)
                ar[e] = e < (int)dim ? arow[e] : 0.0;
        }
        int jlo = 0, jhi = Fb - 1;
        if (band >= 0)
            jhi = min(Fb - 1, Fa - 1 + band);
        double mine = INF, diagReg = INF;
        const int tauEnd = jhi + Fa;
        for (int tau = jlo; tau < tauEnd; ++tau) {
            const int j = tau - lane;
            double fromAbove = shfl_up1(mine);
            double diagv = diagReg;
            if (lane == 0) {
                fromAbove = INF;
                diagv = (j == 0) ? 0.0 : INF;             // virtual D(-1,-1) = 0
            }
            const bool active = rowValid && j >= 0 && j < Fb;
            if (active) {
                double cur = INF;
                const int dij = lane - j;
                if (band < 0 || (dij <= band && -dij <= band)) {
                    const double *bj = sq + (size_t)j * kFewDim;
                    double acc = 0.0;                     // sum_k (a_k - b_k)^2, k ascending; (0 - 0)^2 = +0 changes nothing
#pragma unroll
                    for (int e = 0; e < kFewDim; ++e) {
                        const double df = __dsub_rn(ar[e], bj[e]);
                        acc = __dadd_rn(acc, __dmul_rn(df, df));
                    }
                    const double c = squared ? acc : sqrt(acc);
                    double best = fromAbove;
                    if (mine < best) best = mine;
                    if (diagv < best) best = diagv;
                    cur = __dadd_rn(c, best);
                }
                if (lane == Fa - 1 && j == Fb - 1)
                    result = cur;
                mine = cur;
            }
            diagReg = fromAbove;
        }
    }
    // the pair's cost sits in lane Fa - 1 (or nowhere: +inf)
    const int owner = Fa > 0 ? Fa - 1 : 0;
    const double cost = __shfl(result, owner);
    const double distance = distances ? distances[y] : 0.0;
    if (lane == 0) {
        cc[wave] = s < nSrc ? cost : INF;
        ck[wave] = s < nSrc ? fabs(__dsub_rn(cost, distance)) : INF;
    }
    __syncthreads();
    const uint32_t nb = gridDim.x;
    double *partKey = partKeyAll + (size_t)y * nb, *partCost = partCostAll + (size_t)y * nb;
    uint32_t *partIdx = partIdxAll + (size_t)y * nb;
    if (tid == 0) {
        double best = INF, bc = INF;
        uint32_t bi = 0xffffffffu;
        for (int w = 0; w < kFewWaves; ++w)
            if (ck[w] < best) {                           // first minimum in index order, NaN never wins
                best = ck[w];
                bc = cc[w];
                bi = blockIdx.x * kFewWaves + w;
            }
        partKey[blockIdx.x] = best;
        partCost[blockIdx.x] = bc;
        partIdx[blockIdx.x] = bi;
        __threadfence();
        last = atomicAdd(&tickets[y], 1u) == nb - 1;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        uint32_t minIdx = 0;                              // the fold's start for dtw: (0, +inf)
        double minKey = INF, minCost = INF;
        for (uint32_t b = 0; b < nb; ++b) {
            const double pk = static_cast<volatile double *>(partKey)[b];
            if (pk < minKey) {
                minKey = pk;
                minCost = static_cast<volatile double *>(partCost)[b];
                minIdx = static_cast<volatile uint32_t *>(partIdx)[b];
            }
        }
        outIdxAll[y] = minIdx;
        outCostAll[y] = minCost;                          // dtw reports the winner's cost itself (+inf: nothing finite)
        tickets[y] = 0;
        __threadfence_system();
    }
}

bool dtw_few_supported(const ssym_ctx *ctx, const SegmentSet &src, const uint64_t *q_off, uint32_t n_queries)
{
    if (ctx->metric != SSYM_METRIC_DTW || src.n == 0 || n_queries == 0 || n_queries > 4 || !q_off || src.dim > (uint32_t)kFewDim ||
        src.max_frames > 64 || (uint64_t)src.n * n_queries > 8192)
        return false;
    for (uint32_t i = 0; i < n_queries; ++i)
        if (q_off[i + 1] < q_off[i] || (q_off[i + 1] - q_off[i]) * kFewDim > (uint64_t)kFewQVals)
            return false;
    return true;
}

// queries / q_off / distances / outputs: pinned, as for launch_refcos_match_few
int32_t launch_dtw_match_few(ssym_ctx *ctx, const SegmentSet &src, const void *queries, const uint64_t *q_off,
                             uint32_t n_queries, const double *distances, double *out_cost, uint32_t *out_idx)
{
    const uint32_t nb = (src.n + kFewWaves - 1) / kFewWaves;
    int32_t rc = ensure(ctx, ctx->part, (2 * sizeof(double) + sizeof(uint32_t)) * (size_t)nb * n_queries + 256);
    if (rc != SSYM_OK)
        return rc;
    if (!ctx->one_ticket.ptr) {
        rc = ensure(ctx, ctx->one_ticket, 256);
        if (rc != SSYM_OK)
            return rc;
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(ctx->one_ticket.ptr, 0, 256, ctx->stream));
    }
    double *partKey = (double *)ctx->part.ptr;
    double *partCost = partKey + (size_t)nb * n_queries;
    uint32_t *partIdx = (uint32_t *)(partCost + (size_t)nb * n_queries);
    dtw_match_few_kernel<<<dim3(nb, n_queries), 64 * kFewWaves, 0, ctx->stream>>>(
        src.raw, src.off, src.n, src.dim, queries, q_off, ctx->dtype == SSYM_DTYPE_F32 ? 1 : 0, distances, ctx->band,
        ctx->squared, partKey, partCost, partIdx, (unsigned *)ctx->one_ticket.ptr, out_idx, out_cost);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

int32_t launch_dtw_exact(ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt,
                         const uint2 *pairs, const uint32_t *count_dev, uint32_t max_pairs,
                         double *out)
{
    if (src.dim != tgt.dim) {
        ctx->err = "dim mismatch between dictionary and targets";
        return SSYM_E_INVALID;
    }
    const uint32_t dim = src.dim;
    const uint32_t fbCap = std::max<uint32_t>(tgt.max_frames, 1);
    const size_t boundBytes = 2 * (size_t)fbCap * sizeof(double);
    const size_t frameBytes = (64 * (size_t)(dim | 1u) + (size_t)fbCap * (dim | 1u)) * sizeof(double);
    if (boundBytes > 120 * 1024) {
        ctx->err = "dtw exact: target segment too long (boundary row does not fit LDS)";
        return SSYM_E_UNSUPPORTED;
    }
    uint64_t total = pairs ? max_pairs : (uint64_t)src.n * tgt.n;
    if (total == 0)
        return SSYM_OK;
    // enough single-wave blocks to fill the chip several times over; grid-stride covers the rest
    unsigned grid = (unsigned)std::min<uint64_t>(total, (uint64_t)ctx->num_cus * 64);
    hipStream_t st = ctx->stream;
    // frames of up to 48 values: source frame in registers, target frames zero-padded in LDS
    const uint32_t fbEven = (fbCap + 1) & ~1u;
    const int dimr = dim <= 12 ? 12 : dim <= 14 ? 14 : dim <= 16 ? 16 : dim <= 40 ? 40 : dim <= 48 ? 48 : dim <= 64 ? 64
                                                                                                   : dim <= 96 ? 96 : 0;
    const bool bf32 = ctx->dtype == SSYM_DTYPE_F32;     // every feature buffer of the context was f32: exact in float
    const int up4 = (dimr + 3) / 4 * 4;
    const int ldr = bf32 ? up4 + (up4 % 8 == 4 ? 0 : 4) : ((dimr % 4 == 2) ? dimr : dimr + 2);   // exact_ld<>
    // staged target rows: all of them, or (banded) the widest window a 64-row chunk can reach
    // (a chunk of 64 rows reaches 64 + 2 r columns; an even count keeps the rows 16-byte aligned.  Two rows of slack here
    // cost configs[4] a workgroup per CU: 6 x 26 KB fit the 160 KB, 5 x 27 KB did)
    const uint32_t winRows = ctx->band >= 0 ? std::min<uint32_t>(fbEven, (64 + 2 * (uint32_t)ctx->band + 1) & ~1u) : fbEven;
    const size_t regLds = 2 * (size_t)fbEven * sizeof(double) + (size_t)winRows * ldr * (bf32 ? sizeof(float) : sizeof(double));
    // frames of up to 48 values, bands of up to r = 63: local costs first, recurrence afterwards (dtw_exact_cells_kernel).
    // Measured on MI355X (tools/exact_timing.py, against the kernels below): inside a band it wins at every list length
    // (configs[4]: 4096 pairs 2.12 -> 1.49 ms, 512 pairs 0.51 -> 0.19 ms, every pair of 128 x 128 segments 6.5 -> 5.2 ms);
    // without a band it wins while the list is short enough to be bound by the length of a pair's chain (512 pairs of
    // 128 x 128 frames 0.187 -> 0.080 ms: what a rank of a source-sharded step re-scores after the bound exchange) and
    // loses once the chip is full (4096 pairs 0.48 -> 0.57 ms: its 64 KB of cells admit two workgroups per CU), so there
    // it takes lists of up to 4 pairs per CU and sources of up to 256 frames, and the kernels below take the rest.
    static const bool cellsOff = ssym_knob("SSYM_EXACT_CELLS") && atoi(ssym_knob("SSYM_EXACT_CELLS")) == 0;
    uint64_t lowBound = 0;              // lists at least this long are the business of the kernels below
    {
        const bool banded = ctx->band >= 0;
        static const int panelEnv = ssym_knob("SSYM_CELLS_PANEL") ? atoi(ssym_knob("SSYM_CELLS_PANEL")) : 0;         // experiments
        static const long long maxEnv = ssym_knob("SSYM_CELLS_MAX") ? atoll(ssym_knob("SSYM_CELLS_MAX")) : 0;
        const uint32_t panelX = banded ? 2 * (uint32_t)ctx->band + 1 : std::min<uint32_t>(fbEven, panelEnv > 0 ? panelEnv : 128);
        const size_t cellsLds = 2 * (size_t)fbEven * sizeof(double) + (size_t)panelX * 64 * sizeof(double) +
                                (size_t)winRows * ldr * (bf32 ? sizeof(float) : sizeof(double));
        const uint64_t cellsMax = banded ? ~0ull : (maxEnv > 0 ? (uint64_t)maxEnv : (uint64_t)ctx->num_cus * 4);
        const bool cellsOk = !cellsOff && dimr && dimr <= 48 && panelX <= 128 && cellsLds <= 150 * 1024 &&
                             (banded || (src.max_frames <= 256 && regLds <= 64 * 1024)) &&      // (its siblings below are gated by length)
                             (pairs != nullptr || total <= cellsMax);
        if (cellsOk) {
            const unsigned cgrid = (unsigned)std::min<uint64_t>(std::min<uint64_t>(total, cellsMax), (uint64_t)ctx->num_cus * 8);
#define SSYM_EXACT_CELLS2(D_, T_)                                                                              \
    do {                                                                                                       \
        auto kern = dtw_exact_cells_kernel<D_, T_>;                                                            \
        if (cellsLds > 64 * 1024)                                                                              \
            SSYM_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                    (int)cellsLds));                                           \
        kern<<<cgrid, 256, cellsLds, st>>>(src.raw, src.off, tgt.raw, tgt.off, src.n, tgt.n, dim, ctx->band,  \
                                           ctx->squared, pairs, count_dev, max_pairs, fbEven, panelX, out,     \
                                           cellsMax);                                                          \
    } while (0)
#define SSYM_EXACT_CELLS(D_)                                                                                   \
    do {                                                                                                       \
        if (bf32)                                                                                              \
            SSYM_EXACT_CELLS2(D_, float);                                                                      \
        else                                                                                                   \
            SSYM_EXACT_CELLS2(D_, double);                                                                     \
    } while (0)
            switch (dimr) {
            case 12: SSYM_EXACT_CELLS(12); break;
            case 14: SSYM_EXACT_CELLS(14); break;
            case 16: SSYM_EXACT_CELLS(16); break;
            case 40: SSYM_EXACT_CELLS(40); break;
            default: SSYM_EXACT_CELLS(48); break;
            }
#undef SSYM_EXACT_CELLS
#undef SSYM_EXACT_CELLS2
            SSYM_HIP_CHECK(ctx, hipGetLastError());
            if (total <= cellsMax)          // (max_pairs bounds the list: nothing longer can turn up)
                return SSYM_OK;
            lowBound = cellsMax + 1;
        }
    }
    uint64_t regLo = 0;
    // sources of 65...512 frames: row chunks pipelined over the waves of a workgroup (dtw_exact_pipe_kernel)
    const uint32_t pipeW = (src.max_frames + 63) / 64;
    const int ldp = ldr;
    const size_t pipeLds = (size_t)pipeW * fbEven * sizeof(double) + (size_t)fbEven * ldp * (bf32 ? sizeof(float) : sizeof(double)) +
                           (pipeW + 1) * sizeof(int);
    // It pays while the list is short enough to be bound by the length of a pair's wavefront rather than by
    // issue (measured: 2x at 1024 pairs of 512 frames, 1.4x at 2048 of 256, even at 4096 of 128), and it loses
    // once the chip is full, because a waiting wave holds its slot -- and inside a band, where a chunk
    // needs two thirds of the chunk above before it can start.  The list length lives on the device, so
    // both kernels are launched and the length decides which of them works.
    static const bool pipeOff = ssym_knob("SSYM_EXACT_PIPE") && atoi(ssym_knob("SSYM_EXACT_PIPE")) == 0;
    const uint64_t pipeMax = (uint64_t)ctx->num_cus * 64 / std::max<uint32_t>(pipeW, 1);
    const bool pipeOk = !pipeOff && ctx->band < 0 && dimr && dimr <= 48 && pipeW >= 2 && pipeW <= 8 &&
                        pipeLds <= 150 * 1024 && regLds <= 64 * 1024;
    const unsigned *redoFlag = nullptr;
    if (pipeOk && lowBound <= pipeMax && !(pairs == nullptr && total > pipeMax)) {
        int32_t rcf = ensure(ctx, ctx->pipe_flag, 8 * sizeof(unsigned));
        if (rcf != SSYM_OK)
            return rcf;
        const int slot = ctx->pipe_slot++ & 7;
        unsigned *failCount = (unsigned *)ctx->pipe_flag.ptr + slot;
        rcf = zero_words(ctx, failCount, sizeof(unsigned));
        if (rcf != SSYM_OK)
            return rcf;
        ctx->pipe_mask |= 1u << slot;
        redoFlag = failCount;
        const char *fg = ssym_knob("SSYM_EXACT_PIPE_FORCE_GIVEUP");
        const int forceGiveUp = fg && atoi(fg) != 0 ? 1 : 0;
#define SSYM_EXACT_PIPE2(D_, T_)                                                                               \
    do {                                                                                                       \
        auto kern = dtw_exact_pipe_kernel<D_, T_>;                                                             \
        if (pipeLds > 64 * 1024)                                                                               \
            SSYM_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                    (int)pipeLds));                                            \
        kern<<<std::min<unsigned>(grid, (unsigned)pipeMax), 64 * pipeW, pipeLds, st>>>(                       \
            src.raw, src.off, tgt.raw, tgt.off, src.n, tgt.n, dim, ctx->band, ctx->squared, pairs, count_dev,  \
            max_pairs, fbEven, out, lowBound, pipeMax, failCount, forceGiveUp);                             \
    } while (0)
#define SSYM_EXACT_PIPE(D_)                                                                                    \
    do {                                                                                                       \
        if (bf32)                                                                                              \
            SSYM_EXACT_PIPE2(D_, float);                                                                       \
        else                                                                                                   \
            SSYM_EXACT_PIPE2(D_, double);                                                                      \
    } while (0)
        switch (dimr) {
        case 12: SSYM_EXACT_PIPE(12); break;
        case 14: SSYM_EXACT_PIPE(14); break;
        case 16: SSYM_EXACT_PIPE(16); break;
        case 40: SSYM_EXACT_PIPE(40); break;
        default: SSYM_EXACT_PIPE(48); break;
        }
#undef SSYM_EXACT_PIPE
#undef SSYM_EXACT_PIPE2
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        // longer lists: the one-wave-per-pair kernel below; a list the pipelined kernel certainly took: that kernel
        // only as the redo of a give-up (it leaves at once otherwise)
        regLo = (!pairs || max_pairs <= pipeMax) ? ~0ull : pipeMax + 1;
    } else {
        regLo = lowBound;
    }
    if (dimr && regLds <= (size_t)(dimr >= 64 ? 150 : 64) * 1024) {
#define SSYM_EXACT_REG(...)                                                                                    \
    do {                                                                                                       \
        if (bf32)                                                                                              \
            SSYM_EXACT_REG2(__VA_ARGS__, float);                                                               \
        else                                                                                                   \
            SSYM_EXACT_REG2(__VA_ARGS__, double);                                                              \
    } while (0)
#define SSYM_EXACT_REG2(...)                                                                                   \
    do {                                                                                                       \
        auto kern = dtw_exact_reg_kernel<__VA_ARGS__>;                                                         \
        if (regLds > 64 * 1024)                                                                                \
            SSYM_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                    (int)regLds));                                             \
        kern<<<grid, 64, regLds, st>>>(src.raw, src.off, tgt.raw, tgt.off, src.n, tgt.n, dim, ctx->band,      \
                                       ctx->squared, pairs, count_dev, max_pairs, fbEven, out, regLo, ~0ull,   \
                                       redoFlag);                                                              \
    } while (0)
        switch (dimr) {
        case 12: SSYM_EXACT_REG(12, 1); break;
        case 14: SSYM_EXACT_REG(14, 1); break;
        case 16: SSYM_EXACT_REG(16, 1); break;
        case 40: SSYM_EXACT_REG(40, 1); break;
        case 48: SSYM_EXACT_REG(48, 1); break;
        case 64: SSYM_EXACT_REG(64, 2); break;      // 49..64 values: the frame is fetched in two halves
        default: SSYM_EXACT_REG(96, 3); break;      // 65..96 values: three fetches of 32
        }
#undef SSYM_EXACT_REG
#undef SSYM_EXACT_REG2
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        return SSYM_OK;
    }
    const bool ldsFrames = boundBytes + frameBytes <= 64 * 1024;
    const size_t lds = boundBytes + (ldsFrames ? frameBytes : 0);
    if (ldsFrames)
        dtw_exact_kernel<true><<<grid, 64, lds, st>>>(src.raw, src.off, tgt.raw, tgt.off, src.n, tgt.n,
                                                     dim, ctx->band, ctx->squared, pairs, count_dev,
                                                     max_pairs, fbCap, out);
    else
        dtw_exact_kernel<false><<<grid, 64, lds, st>>>(src.raw, src.off, tgt.raw, tgt.off, src.n, tgt.n,
                                                      dim, ctx->band, ctx->squared, pairs, count_dev,
                                                      max_pairs, fbCap, out);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
