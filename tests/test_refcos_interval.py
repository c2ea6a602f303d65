"""The key interval of the refcos matrix-pipe filter (csrc/refcos_mfma.hip), restated on the CPU and held against
adversarial evaluation orders: whatever order (and fusing) the matrix pipe uses for the L-term dot, the reference's key
|fl(fl(dot_ref / nrm) - distance)| must lie inside [key_lo, key_hi] computed from the other order's dot.

Reference order: rulinalg's eight running sums, products and sums rounded separately (oracle/ssym_oracle.c,
src/sound.rs:31).  Other orders: a fused multiply-add chain (one rounding per step, emulated exactly with Fractions),
the same in reverse, four interleaved FMA chains (the MFMA's lane groups), pairwise summation of rounded products."""
from fractions import Fraction

import numpy as np
import pytest

U = 2.0 ** -53


def dot_reference(a, b):
    n = a.size
    q, r = divmod(n, 8)
    p = [np.float64(0.0)] * 8
    for k in range(q):
        for i in range(8):
            p[i] = p[i] + a[8 * k + i] * b[8 * k + i]          # product rounded, then the sum
    s = np.float64(0.0)
    s = s + (p[0] + p[4])
    s = s + (p[1] + p[5])
    s = s + (p[2] + p[6])
    s = s + (p[3] + p[7])
    for i in range(r):
        s = s + a[8 * q + i] * b[8 * q + i]
    return float(s)


def fma(x, y, z):
    return float(Fraction(x) * Fraction(y) + Fraction(z))          # one rounding (float(Fraction) rounds correctly)


def dot_fma_chain(a, b, order):
    s = 0.0
    for i in order:
        s = fma(float(a[i]), float(b[i]), s)
    return s


def dot_four_chains(a, b):
    acc = [0.0] * 4
    for i in range(a.size):
        acc[(i // 2) % 4] = fma(float(a[i]), float(b[i]), acc[(i // 2) % 4])
    return (acc[0] + acc[1]) + (acc[2] + acc[3])


def dot_pairwise(a, b):
    v = [float(x) for x in (a * b)]
    while len(v) > 1:
        v = [v[i] + v[i + 1] if i + 1 < len(v) else v[i] for i in range(0, len(v), 2)]
    return v[0] if v else 0.0


def key_interval(dotm, na, nb, length, d, extra=0.0):
    """refcos_key_interval (csrc/refcos_filter.hpp) + the per-segment values of pack.hip, same operations in the same
    order; `extra`: a filter's further error in similarity units (the integer filter's quantisation bound)."""
    sa = float(np.sqrt(np.float64(na))) * (1.0 + 4.5e-16)
    sb = float(np.sqrt(np.float64(nb))) * (1.0 + 4.5e-16)
    with np.errstate(all="ignore"):
        ia, ib = float(np.float64(1.0) / np.float64(na)), float(np.float64(1.0) / np.float64(nb))   # 1 / 0 = inf on the GPU
    nrm = na * nb
    cL = (3.0 * length + 16.0) * (U * 1.02)
    with np.errstate(all="ignore"):
        inv = float(np.float64(ia) * np.float64(ib))
        s = float(np.float64(dotm) * np.float64(inv))
        z = abs(float(np.float64(s) - np.float64(d)))
        R = float(np.float64(1.0001) * ((np.float64(cL) * (np.float64(sa) * np.float64(sb))) * np.float64(inv) + np.float64(extra))
                  + np.float64(9.0 * U) * (abs(np.float64(s)) + abs(np.float64(d))) + np.float64(1e-290))
    klo = (z - R) * (1.0 - 4.0 * U) if z > R else 0.0
    khi = (z + R) * (1.0 + 4.0 * U)
    if not (khi < np.inf) or not (1e-280 < inv < 1e280):
        klo, khi = 0.0, np.inf
    if nrm == 0.0 or nrm != nrm or d != d:
        klo = khi = np.inf
    return klo, khi


def key_interval_plain(dotm, na, nb, length, d):
    """The epilogue's short form for PLAIN segments (norms in [1e-139, 1e139], finite distance: refcos_epilogue): the
    same bound with the factors of a row or a column alone taken out of the pair, one fused multiply-add, max for the
    select -- same operations in the same order as the kernel."""
    f64 = np.float64
    sa = float(np.sqrt(f64(na))) * (1.0 + 4.5e-16)
    sb = float(np.sqrt(f64(nb))) * (1.0 + 4.5e-16)
    ia, ib = float(f64(1.0) / f64(na)), float(f64(1.0) / f64(nb))
    rq, cq = float(f64(sa) * f64(ia)), float(f64(sb) * f64(ib))
    cl = float(f64(1.0001) * ((f64(3.0) * f64(length) + f64(16.0)) * f64(U * 1.02)))       # min(cLa, cLb): equal lengths here
    sv = float(f64(dotm) * (f64(ia) * f64(ib)))
    z = abs(float(f64(sv) - f64(d)))
    R = float(f64(fma(9.0 * U, float(abs(f64(sv)) + abs(f64(d))), float(f64(cl) * (f64(rq) * f64(cq))))) + f64(1e-290))
    klo = max(float(f64(z - R) * f64(1.0 - 4.0 * U)), 0.0)
    khi = float(f64(z + R) * f64(1.0 + 4.0 * U))
    return klo, khi


def is_plain(na, nb, d):
    return 1e-139 <= na <= 1e139 and 1e-139 <= nb <= 1e139 and abs(d) <= 1e300


@pytest.mark.parametrize("length", [1, 7, 8, 9, 100, 1536])
@pytest.mark.parametrize("scale", [1e-120, 1e-3, 1.0, 3e4, 1e100])
def test_reference_key_lies_inside_the_interval_for_any_order(length, scale):
    rng = np.random.default_rng(length * 1000 + int(np.log10(scale)) + 500)
    for trial in range(4):
        a = rng.standard_normal(length) * scale
        b = rng.standard_normal(length) * scale
        if trial == 1:
            b = a * (1 + 1e-9 * rng.standard_normal(length))        # nearly parallel: sim at its largest
        if trial == 2:
            b = b - a * (a @ b) / (a @ a)                           # nearly orthogonal: heavy cancellation in the dot
        if trial == 3:
            a[::2] *= 1e6                                           # mixed magnitudes
        na = 0.0
        for x in a:                                                 # norm: sequential fold, src/sound.rs:35-38
            na = float(np.float64(x) * np.float64(x) + np.float64(na))
        nb = 0.0
        for x in b:
            nb = float(np.float64(x) * np.float64(x) + np.float64(nb))
        d_ref = dot_reference(a, b)
        nrm = na * nb
        for d in (1.0, 0.0, 0.37):
            with np.errstate(all="ignore"):
                k_ref = abs(float(np.float64(d_ref) / np.float64(nrm)) - d) if nrm != 0 else np.nan
            orders = [dot_fma_chain(a, b, range(length)), dot_fma_chain(a, b, range(length - 1, -1, -1)),
                      dot_four_chains(a, b), dot_pairwise(a, b)]
            for dotm in orders:
                klo, khi = key_interval(dotm, na, nb, length, d)
                if nrm == 0.0 or nrm != nrm:
                    assert klo == np.inf                            # the reference's key is NaN / inf: never a winner
                    continue
                assert klo <= k_ref <= khi, (length, scale, trial, d, dotm, d_ref, klo, k_ref, khi)
                if is_plain(na, nb, d):                             # the short form holds the same key, about as tightly
                    plo, phi = key_interval_plain(dotm, na, nb, length, d)
                    assert plo <= k_ref <= phi, (length, scale, trial, d, dotm, d_ref, plo, k_ref, phi)
                    assert abs(plo - klo) <= 1e-3 * (khi - klo) and abs(phi - khi) <= 1e-3 * (khi - klo)
                if scale == 1.0 and np.isfinite(khi):               # and it is an interval worth having
                    assert khi - klo <= 1e-9 * (abs(d_ref / nrm) + abs(d)) + 1e-11 * length / np.sqrt(nrm)


@pytest.mark.parametrize("length", [1, 7, 33, 100, 1536])
@pytest.mark.parametrize("scale", [1e-30, 1e-3, 1.0, 3e4, 1e25])
def test_reference_key_lies_inside_the_integer_filters_interval(length, scale):
    """csrc/refcos_q8.hip: 23-bit fixed point per segment, six exact integer digit products, the three cheapest left
    out -- the bound it adds to the interval (tests/bounds.py restates records and kernel) must hold the reference's
    key for parallel, orthogonal and wildly mixed segments, and be worth having on ordinary ones."""
    import bounds
    rng = np.random.default_rng(length * 77 + int(np.log10(scale)) + 900)
    for trial in range(5):
        la = length
        lb = length if trial != 4 else max(1, length // 2)          # unequal lengths: the common prefix (src/sound.rs:24-28)
        a = rng.standard_normal(la) * scale
        b = rng.standard_normal(lb) * scale
        if trial == 1:
            b = a[:lb] * (1 + 1e-9 * rng.standard_normal(lb))
        if trial == 2 and lb > 1:
            b = b - a[:lb] * (a[:lb] @ b) / (a[:lb] @ a[:lb])
        if trial == 3:
            a[::2] *= 1e6                                           # a crest factor of 1e6: half the digits carry nothing
        na = 0.0
        for x in a:
            na = float(np.float64(x) * np.float64(x) + np.float64(na))
        nb = 0.0
        for x in b:
            nb = float(np.float64(x) * np.float64(x) + np.float64(nb))
        L = min(la, lb)
        d_ref = dot_reference(a[:L], b[:L])
        nrm = na * nb
        qa, qb = bounds.q8_quantise(a), bounds.q8_quantise(b)
        assert qa is not None and qb is not None
        ia, ib = float(np.float64(1.0) / np.float64(na)), float(np.float64(1.0) / np.float64(nb))
        dq, gk, a2b2, extra = bounds.q8_dot_and_extra(qa, qb, la, lb, ia, ib)
        exact = sum(Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a[:L], b[:L]))
        # the bound on the dot itself, before anything of the key: |D - dq| <= extra / (ia ib)
        assert abs(exact - Fraction(dq)) <= Fraction(extra) / (Fraction(ia) * Fraction(ib)) * Fraction(1 + 1e-12)
        for d in (1.0, 0.0, 0.37):
            k_ref = abs(float(np.float64(d_ref) / np.float64(nrm)) - d)
            klo, khi = key_interval(dq, na, nb, L, d, extra)
            assert klo <= k_ref <= khi, (length, scale, trial, d, dq, d_ref, klo, k_ref, khi)
        if trial == 0 and length >= 100:                            # ~1e-5 of a similarity's scale on ordinary data
            assert extra * np.sqrt(nrm) <= 2e-5, extra * np.sqrt(nrm)


def test_column_constants_of_the_integer_filters_short_form_dominate_every_rows_own_bound():
    """The kernel's plain waves do not form a row's own half-width R but R_ub(z) = c9 (z + 2 |dist|) + C from per-COLUMN
    constants: the column's numbers against the largest a1..a4 among the wave's 64 rows, and |s| <= z + |dist|
    (csrc/refcos_q8.hip).  R_ub must be at least every row's R, the threshold it gives at least the smallest key_hi, and
    the key_lo it gives at most every row's own -- on 64 ragged rows against one column, same operations as the kernel."""
    import math
    import bounds
    rng = np.random.default_rng(4242)
    c9 = 9.0 * U * 1.0000001
    for trial in range(6):
        lens = rng.integers(8, 400, 64)
        rows = [rng.standard_normal(int(l)) * 10.0 ** rng.uniform(-2, 2) for l in lens]
        col = rng.standard_normal(int(rng.integers(8, 400))) * 10.0 ** rng.uniform(-2, 2)
        d = float(rng.uniform(-0.5, 1.5))

        def consts(v):
            q = bounds.q8_quantise(v)
            nv = 0.0
            for x in v:
                nv = float(np.float64(x) * np.float64(x) + np.float64(nv))
            inv = float(np.float64(1.0) / np.float64(nv))
            sq = float(np.sqrt(np.float64(nv))) * (1.0 + 4.5e-16)
            scl = 2.0 ** -q[3]
            cl = (3.0 * v.size + 16.0) * (U * 1.02)
            return dict(q=q, n=nv, inv=inv, a1=q[4] * scl * inv, a2=scl * inv,
                        a3=math.sqrt(16384.25 * v.size) * (1.0 + 2.0 ** -50) * scl * inv,
                        a4=math.sqrt(cl) * (1.0 + 2.0 ** -50) * (sq * inv), len=v.size)

        cr = [consts(v) for v in rows]
        cc = consts(col)
        m1, m2, m3, m4 = (max(c[k] for c in cr) for k in ("a1", "a2", "a3", "a4"))
        C = 1.0001 * (m4 * cc["a4"] + (m1 * cc["a2"] + (m2 * cc["a1"] + m3 * cc["a3"]))) + 1e-290
        K = (2.0 * c9 * abs(d) + C) * (1.0 + 4.0 * U)
        zs, his, los = [], [], []
        for v, c in zip(rows, cr):
            dq, gk, a2b2, extra = bounds.q8_dot_and_extra(c["q"], cc["q"], c["len"], cc["len"], c["inv"], cc["inv"])
            sv = gk * (c["a2"] * cc["a2"])
            z = abs(sv - d)
            R_row = 9.0 * U * (abs(sv) + abs(d)) + 1.0001 * (c["a4"] * cc["a4"] + extra) + 1e-290
            R_ub = c9 * (z + 2.0 * abs(d)) + C
            assert R_ub >= R_row, (trial, R_ub, R_row)
            zs.append(z)
            his.append((z + R_row) * (1.0 + 4.0 * U))
            los.append(max((z - R_row) * (1.0 - 4.0 * U), 0.0))
            assert max((z * (1.0 - c9) - K) * (1.0 - 4.0 * U), 0.0) <= los[-1] + 1e-300
        zmin = min(zs)
        assert (zmin + (c9 * (zmin + 2.0 * abs(d)) + C)) * (1.0 + 4.0 * U) >= min(his)
