// chain.hip -- SoundSequence::from_distances (src/sound.rs:405-417) with the dictionary resident
// and NO host round trip between steps.
//
// The reference chains at_distance calls: the sound matched at step i is the query of step i+1.
// The first query is the caller's `start` sound; every later query is a dictionary entry, so
//   refcos: all later steps are row lookups in the dictionary's self-similarity matrix
//           S[s][t] = cosine_sim(sounds[s], sounds[t]) (computed once per dictionary by the same
//           kernel that serves match_queries, bit-identical, and symmetric bit for bit: products and
//           the norm product commute), followed by a one-workgroup first-minimum scan whose result
//           stays in device memory as the next step's row index;
//   dtw:    a step re-scores the N pairs (s, current) with the exact f64 kernel -- the pair list is
//           written on the device from the current index -- and runs the same scan.
// All steps are enqueued back to back on the context's stream; the host waits once at the end.
#include "ssym_internal.hpp"

#include <algorithm>
#include <cmath>

namespace ssym {

__global__ void chain_pairs_kernel(const uint32_t *__restrict__ cur, uint32_t n, uint2 *__restrict__ pairs,
                                   uint32_t *__restrict__ count)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n)
        pairs[s] = make_uint2(s, *cur);
    if (s == 0)
        *count = n;
}

// One workgroup: first minimum of |values[s] - distance| over s < n with the reference's fold
// (start (0, init), strict '<': lowest index wins ties, NaN never wins; src/sound.rs:361-367).
// values = base + (*rowSel) * rowStride when rowSel is given (row of the self-similarity matrix).
__global__ __launch_bounds__(1024) void chain_argmin_kernel(const double *__restrict__ base, size_t rowStride,
                                                            const uint32_t *rowSel, uint32_t n, double distance,
                                                            double init, int reportValue, uint32_t step,
                                                            uint32_t *cur, uint32_t *__restrict__ outIdx,
                                                            double *__restrict__ outCost)
{
    __shared__ double sKey[16];
    __shared__ uint32_t sIdx[16];
    const double *v = base + (rowSel ? (size_t)(*rowSel) * rowStride : 0);
    double bestKey = init;
    uint32_t bestIdx = 0xffffffffu;
    for (uint32_t s = threadIdx.x; s < n; s += blockDim.x) {
        const double key = fabs(v[s] - distance);
        if (key < bestKey) {        // ascending s within a thread: the first of equal keys stays
            bestKey = key;
            bestIdx = s;
        }
    }
    // lexicographic (key, index) minimum across the workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double k2 = __shfl_xor(bestKey, o);
        const uint32_t i2 = __shfl_xor(bestIdx, o);
        if (k2 < bestKey || (k2 == bestKey && i2 < bestIdx)) {
            bestKey = k2;
            bestIdx = i2;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sKey[wave] = bestKey;
        sIdx[wave] = bestIdx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (sKey[w] < bestKey || (sKey[w] == bestKey && sIdx[w] < bestIdx)) {
                bestKey = sKey[w];
                bestIdx = sIdx[w];
            }
        const bool found = bestIdx != 0xffffffffu;
        const uint32_t idx = found ? bestIdx : 0u;         // fold start: index 0
        *cur = idx;
        outIdx[step] = idx;
        if (outCost)
            outCost[step] = found ? (reportValue ? v[idx] : bestKey) : init;
    }
}

int32_t launch_chain_argmin(ssym_ctx *ctx, const double *base, size_t row_stride, const uint32_t *row_sel,
                            uint32_t n, double distance, double init, bool report_value, uint32_t step,
                            uint32_t *cur, uint32_t *out_idx, double *out_cost)
{
    chain_argmin_kernel<<<1, 1024, 0, ctx->stream>>>(base, row_stride, row_sel, n, distance, init,
                                                     report_value ? 1 : 0, step, cur, out_idx, out_cost);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

int32_t launch_chain_pairs(ssym_ctx *ctx, const uint32_t *cur, uint32_t n, uint2 *pairs, uint32_t *count)
{
    chain_pairs_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(cur, n, pairs, count);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

}  // namespace ssym
