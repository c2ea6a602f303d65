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
