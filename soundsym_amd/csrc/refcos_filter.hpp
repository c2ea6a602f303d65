// refcos_filter.hpp -- what the two filters of the refcos search share (refcos_mfma.hip: every dot on the f64 matrix
// pipe; refcos_q8.hip: every dot as exact integer products of 8-bit digits on the i8 matrix pipe): the entries of list 1,
// the per-segment values of an epilogue, and the key interval (derivation: top of refcos_mfma.hip).
#pragma once
#include <cstdint>

namespace ssym {

constexpr int rm_wait_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n), nothing else
constexpr unsigned long long kInfBitsU = 0x7ff0000000000000ull;

struct PairEntry {                  // list 1: a pair that may hold its target's first minimum
    uint32_t s, t;
    double key_lo;
};

struct PairEntryK {                 // list 1 of a top-k search: the interval's upper end travels too (the k-th smallest of a
    uint32_t s, t;                  // target's upper ends, over ALL its listed pairs, is the target's threshold)
    double key_lo, key_hi;
};

struct RowInfo {                    // per segment of the tile
    double sq;                      // >= sqrt(norm)
    double inv;                     // fl(1 / norm)
    double norm;
    double dist;                    // targets: the distance |sim - dist| is taken to
};

// [key_lo, key_hi] for the reference's key from a dot product `dotm` that is within
//   (3 L + 16) u 1.02 sqrt(na nb)  [cL * sasb]   +   extra / inv
// of the reference's own (extra: a filter's further error, already multiplied by inv = fl(ia ib); 0 for the f64 pipe).
__device__ __forceinline__ void refcos_key_interval(double dotm, double sasb, double inv, double nrm, double cL, double d,
                                                    double &klo, double &khi, double extra = 0.0)
{
    const double INF = __builtin_inf();
    const double u = 1.1102230246251565e-16;
    const double s = dotm * inv;
    const double z = fabs(s - d);
    const double R = 1.0001 * ((cL * sasb) * inv + extra) + 9.0 * u * (fabs(s) + fabs(d)) + 1e-290;
    klo = z > R ? (z - R) * (1.0 - 4.0 * u) : 0.0;
    khi = (z + R) * (1.0 + 4.0 * u);
    // something is not finite (or NaN), or the norms are so large or small that 1 / nrm or single products leave the
    // normal range (the relative bounds above need it): the pair stays in, bounds nothing
    if (!(khi < INF) || !(inv > 1e-280 && inv < 1e280)) {
        klo = 0.0;
        khi = INF;
    }
    if (nrm == 0.0 || nrm != nrm || d != d)        // the reference's key is NaN or +inf: never a winner
        klo = khi = INF;
}

}  // namespace ssym
