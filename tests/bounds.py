"""The dtw filter's error model restated for the tests (csrc/dtw_filter.hip common_scale, csrc/dtw_margin.hpp)."""
import math
import os

import numpy as np


def common_scale(vmax, sqmax):
    """csrc/dtw_filter.hip common_scale: max |s v| < 64 and s^2 * max |frame|^2 * 1.01 < 65000."""
    if not vmax > 0:
        return 1.0
    e = 6 - math.frexp(vmax)[1]
    while sqmax * 2.0 ** (2 * e) * 1.01 >= 65000.0:
        e -= 1
    return 2.0 ** e


def input_rounding(dim_used):
    """csrc/dtw_margin.hpp margin_params: relative rounding of the (source, target) frames the filter sees.
    Up to 13 values the records hold the source in two f16 pieces and the target in one (layout 3 of
    csrc/ssym_internal.hpp; SSYM_FILTER_K48=1 keeps two on both sides); wider frames one piece on both sides."""
    if dim_used > 13:
        return 2.0 ** -11, 2.0 ** -11
    return 2.0 ** -22, (2.0 ** -22 if os.environ.get("SSYM_FILTER_K48") else 2.0 ** -11)


def worst_case_bound(src, tgt, dim_used, fa, fb):
    """|C~ - C| of the filter with the worst-case cell error (csrc/dtw_margin.hpp, xmin = 0)."""
    u = 2.0 ** -24
    in_a, in_b = input_rounding(dim_used)
    sq = lambda a: float((a.astype(np.float64)[..., :dim_used] ** 2).sum(-1).max())
    na, nb = sq(src), sq(tgt)
    full = lambda a: float((np.float32(1.000001) * (a.astype(np.float64) ** 2).sum(-1).astype(np.float32)).max())
    vmax = max(float(np.abs(src).max()), float(np.abs(tgt).max())) * 1.000001
    s = common_scale(vmax, max(full(src), full(tgt)))
    E = 256 * u * (na + nb) + 2.0 ** -12 / s ** 2
    cell = math.sqrt(E) + 1.001 * (in_a * math.sqrt(na) + in_b * math.sqrt(nb)) + 2.0 ** -20 / s
    return 1.02 * (fa + fb - 1) * cell, s


def record_residual(frames, s, dim_used, is_source):
    """|frame - the frame its filter record represents| per frame, unscaled (csrc/dtw_filter.hip
    build_filter_records_kernel): values are rounded to one f16 piece, or two where the layout keeps both -- up to 13
    values the source everywhere and the target in its first two values (layout 3; SSYM_FILTER_K48: both sides
    everywhere), wider frames one piece on both sides."""
    f16 = lambda x: np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)
    v = np.asarray(frames, dtype=np.float64)[..., :dim_used] * s
    h1 = f16(v)
    two = np.zeros(v.shape[-1], dtype=bool)
    if dim_used <= 13:
        if os.environ.get("SSYM_FILTER_K48") or is_source:
            two[:] = True
        two[:2] = True
    vh = h1 + np.where(two, f16(v - h1), 0.0)
    return np.sqrt(((v - vh) ** 2).sum(-1)) / s


def pair_bound_matrix(src, tgt, dim_used):
    """[n_src, n_tgt] |C~ - C| bounds of the filter as the selection's first stage prices a pair since round 4
    (csrc/dtw_margin.hpp, xmin = 0): the square-root term from the two SEGMENTS' largest squared frame norms and, for
    what the records round away, the segments' MEASURED residuals (never above the layout's worst case)."""
    u = 2.0 ** -24
    in_a, in_b = input_rounding(dim_used)
    sq = lambda a: float((np.asarray(a, dtype=np.float64)[..., :dim_used] ** 2).sum(-1).max()) * 1.000002 if np.asarray(a).size else 0.0
    full = lambda a: float((np.asarray(a, dtype=np.float64) ** 2).sum(-1).max()) * 1.000002 if np.asarray(a).size else 0.0
    vmax = max(max(float(np.abs(a).max()) for a in src if np.asarray(a).size),
               max(float(np.abs(a).max()) for a in tgt if np.asarray(a).size)) * 1.000001
    s = common_scale(vmax, max(max(full(a) for a in src), max(full(a) for a in tgt)))
    na = np.array([sq(a) for a in src])
    nb = np.array([sq(a) for a in tgt])
    ra = np.array([float(record_residual(a, s, dim_used, True).max()) * 1.000001 if np.asarray(a).size else 0.0 for a in src])
    rb = np.array([float(record_residual(a, s, dim_used, False).max()) * 1.000001 if np.asarray(a).size else 0.0 for a in tgt])
    fa = np.array([np.asarray(a).shape[0] for a in src])
    fb = np.array([np.asarray(a).shape[0] for a in tgt])
    E = 256 * u * (na[:, None] + nb[None, :]) + 2.0 ** -12 / s ** 2
    cell = np.sqrt(E) + 1.001 * (np.minimum(ra, in_a * np.sqrt(na))[:, None] + np.minimum(rb, in_b * np.sqrt(nb))[None, :]) + 2.0 ** -20 / s
    return 1.02 * (fa[:, None] + fb[None, :] - 1) * cell, ra, rb, na, nb


# ---- the refcos integer filter (csrc/refcos_q8.hip), restated --------------------------------------------------------
def q8_quantise(a):
    """A segment's values as 23-bit fixed point in three balanced base-256 digits (refcos_q8_records_kernel).
    Returns (q1, q2, q3 as int64 arrays, E, Eseg = sum|n| / 2 + 2^15 sum|q2|); None where the kernel says `outside`."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    if a.size > 32768 or not np.isfinite(a).all():
        return None
    amax = float(np.abs(a).max()) if a.size else 0.0
    if amax == 0.0:
        z = np.zeros(a.size, dtype=np.int64)
        return z, z, z, 0, 0.0
    if not (2.0 ** -120 <= amax <= 2.0 ** 120):
        return None
    e = math.frexp(amax)[1]
    E = 22 - e
    n = np.rint(np.ldexp(a, E)).astype(np.int64)
    q3 = ((n + 128) & 255) - 128
    n1 = (n - q3) >> 8
    q2 = ((n1 + 128) & 255) - 128
    q1 = (n1 - q2) >> 8
    assert (np.abs(q1) <= 65).all() and (q1 * 65536 + q2 * 256 + q3 == n).all()
    eseg = 0.5 * float(np.abs(n).sum()) + 32768.0 * float(np.abs(q2).sum())
    return q1, q2, q3, E, eseg


def q8_dot_and_extra(qa, qb, la, lb, ia, ib):
    """(dq, extra): the integer filter's dot of a pair and its error term in similarity units, as the kernel forms them."""
    q1, q2, q3, Ea, esa = qa
    r1, r2, r3, Eb, esb = qb
    L = min(la, lb)
    k0 = int((q1[:L] * r1[:L]).sum())
    k1 = int((q1[:L] * r2[:L]).sum() + (q2[:L] * r1[:L]).sum())
    k2 = int((q1[:L] * r3[:L]).sum() + (q2[:L] * r2[:L]).sum() + (q3[:L] * r1[:L]).sum())
    assert max(abs(k0), abs(k1), abs(k2)) < 2 ** 31
    gk = float(k0) * 2.0 ** 32 + float(k1) * 2.0 ** 24 + float(k2) * 2.0 ** 16
    sa, sb = 2.0 ** -Ea, 2.0 ** -Eb
    a2, b2 = sa * ia, sb * ib
    a1, b1 = esa * sa * ia, esb * sb * ib
    a3 = math.sqrt(16384.25 * la) * (1.0 + 2.0 ** -50) * sa * ia
    b3 = math.sqrt(16384.25 * lb) * (1.0 + 2.0 ** -50) * sb * ib
    return gk * (sa * sb), gk, a2 * b2, a1 * b2 + a2 * b1 + a3 * b3
