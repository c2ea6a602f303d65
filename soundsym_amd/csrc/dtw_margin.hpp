// dtw_margin.hpp -- the filter's error model as code (derivation: the comment at the top of select.hip),
// shared by the selection kernels and by prune.hip.
#pragma once
#include "ssym_internal.hpp"

namespace ssym {

struct MarginParams {
    double inv_scale2;  // 1 / s^2
    double in_round_a;  // relative rounding of the SOURCE frames the filter sees: 2^-22 (two f16 pieces) or 2^-11 (one)
    double in_round_b;  // ... of the TARGET frames (layout 3 of ssym_internal.hpp: source two pieces, target one)
    int squared;
    int lower_only;     // frames wider than the filter takes in: its cost bounds a pair's cost from BELOW only
    // element offsets of the per-slot record residuals behind the sets' max_sqnorm arrays (2 x n_pad, pack.hip): the
    // largest |frame - represented frame| of the slot, measured when the records were built (dtw_filter.hip)
    uint32_t src_resid_off, tgt_resid_off;
};

// worst error of one local cost of the pair (xmin = the pair's certificate, 0 = none).  ra, rb: how far the records'
// frames lie from the source's / target's own frames at most -- the MEASURED residual of the two segments (round 4; < 0:
// not known, the worst case in_round * |frame| of the layout stands in, as it did for every pair before).  The filter sees
// |a~ - b~|, and ||a~ - b~| - |a - b|| <= |a~ - a| + |b~ - b|.  For cepstral frames whose first two values -- kept in both
// pieces by layout 3 -- carry most of the norm, the measured residual is several times below 2^-11 |b|.
__host__ __device__ __forceinline__ double dtw_cell_error(const MarginParams &mp, double xmin, double na, double nb,
                                                          double ra = -1.0, double rb = -1.0)
{
    const double u = 5.9604644775390625e-8;   // 2^-24
    const double E = 256.0 * u * (na + nb) + 0.000244140625 * mp.inv_scale2;
    double cell;
    // operands rounded to their f16 piece(s) move a frame by <= in_round * |frame|, hence c by
    // <= in_round_a |a| + in_round_b |b| and c^2 by <= 2.05 rho (|a| + |b|)^2 <= 4.1 rho (|a|^2 + |b|^2), rho the larger
    if (mp.squared)
        cell = E + 4.1 * fmax(mp.in_round_a, mp.in_round_b) * (na + nb);
    else
        cell = (xmin > 6.0 * E ? E / (2.0 * sqrt(xmin - 2.0 * E)) : sqrt(E)) +
               1.001 * ((ra >= 0.0 ? fmin(ra, mp.in_round_a * sqrt(na)) : mp.in_round_a * sqrt(na)) +
                        (rb >= 0.0 ? fmin(rb, mp.in_round_b * sqrt(nb)) : mp.in_round_b * sqrt(nb)));
    // f16 pieces below 2^-14 are subnormal: their absolute rounding 2^-25 (scaled units) per value,
    // over at most 42 values of both frames
    cell += 9.5367431640625e-07 * sqrt(mp.inv_scale2);
    return cell;
}

__device__ __forceinline__ void dtw_key_interval(const MarginParams &mp, double cst, double xmin,
                                                 double na, double nb, int fa, int fb, double delta,
                                                 double &key_lo, double &key_hi, double ra = -1.0, double rb = -1.0)
{
    const double INF = __builtin_inf();
    if (!(cst < INF)) {          // unreachable / empty / NaN: never a candidate
        key_lo = INF;
        key_hi = INF;
        return;
    }
    const double u = 5.9604644775390625e-8;   // 2^-24
    const double cell = dtw_cell_error(mp, xmin, na, nb, ra, rb);
    const double L = (double)(fa + fb - 1);
    const double err = 1.02 * L * cell + (L + 6.0) * u * cst + 1e-300;
    const double lo = fmax(cst - err, 0.0), hi = mp.lower_only ? INF : cst + err;
    key_lo = fmax(fmax(lo - delta, delta - hi), 0.0);
    key_hi = fmax(fabs(lo - delta), fabs(hi - delta));
}

inline MarginParams margin_params(const ssym_ctx *ctx, const SegmentSet &src, const SegmentSet &tgt)
{
    MarginParams mp;
    mp.inv_scale2 = src.rec_scale > 0.0 ? 1.0 / (src.rec_scale * src.rec_scale) : 1.0;
    const int pieces = filter_pieces(filter_dim_used((int)src.dim));
    mp.in_round_a = pieces == 1 ? 4.8828125e-04 : 2.384185791015625e-07;
    mp.in_round_b = pieces == 2 ? 2.384185791015625e-07 : 4.8828125e-04;
    mp.squared = ctx->squared;
    mp.lower_only = filter_lower_bound_only(ctx, src, tgt) ? 1 : 0;
    mp.src_resid_off = 2u * src.n_pad;
    mp.tgt_resid_off = 2u * tgt.n_pad;
    return mp;
}

}  // namespace ssym
