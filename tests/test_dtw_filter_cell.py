"""The dtw filter's per-cell error model (csrc/select.hip, csrc/dtw_margin.hpp) held against a CPU emulation of the
records (csrc/dtw_filter.hip build_filter_records_kernel, all three layouts) and of the f32 accumulation of the matrix
pipe under adversarial orders and rounding modes: whatever order the K slots are summed in, and whether each partial sum
is rounded to nearest or truncated, the local cost the filter sees must lie within `dtw_cell_error` of the true one --
with the worst-case term sqrt(E) and with the certificate term E / (2 sqrt(m - 2E)).

This is the host-side half of tests/test_gpu_numerics.py (which asks the same of the hardware): it pins the BOUND, not
the kernel -- record slots, piece splits and the margin's constants are restated here from the sources named above."""
import numpy as np
import pytest

from bounds import common_scale

U = 2.0 ** -24


def f16(x):
    return np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)     # round to nearest even, like (_Float16)v


def build(frames, s, layout, is_source):
    """-> products' operand rows [n, K] (f64 holding f16 values) and the represented squared norms, scaled units."""
    n, d = frames.shape
    K = 32 if layout == 3 else 48
    out = np.zeros((n, K))
    v = frames * s
    h1 = f16(v)
    two = np.zeros(d, dtype=bool)
    if layout == 2:
        two[:] = True
    elif layout == 3:
        two[:] = True if is_source else False
        two[:2] = True
    h2 = np.where(two[None, :], f16(v - h1), 0.0)
    vh = h1 + h2
    nrm = np.zeros(n)
    for e in range(d):                       # e ascending, f64, like the kernel
        nrm = nrm + vh[:, e] * vh[:, e]
    m1, m2 = f16(-2.0 * h1), f16(-2.0 * h2)  # exact (power of two)
    for e in range(d):
        if layout == 2:
            out[:, 3 * e + 0] = m1[:, e] if is_source else h1[:, e]
            out[:, 3 * e + 1] = m1[:, e] if is_source else h2[:, e]
            out[:, 3 * e + 2] = m2[:, e] if is_source else h1[:, e]
        elif layout == 3:
            out[:, 2 * e + 0] = m1[:, e] if is_source else h1[:, e]
            out[:, 2 * e + 1] = m2[:, e] if is_source else h1[:, e]
            if e < 2:
                out[:, 26 + e] = m1[:, e] if is_source else h2[:, e]
        else:
            out[:, e] = m1[:, e] if is_source else h1[:, e]
    nbase = 28 if layout == 3 else (3 if layout == 2 else 1) * d
    npieces = 2 if layout == 3 else 3
    p = [f16(nrm)]
    p.append(f16(nrm - p[0]))
    p.append(f16(nrm - p[0] - p[1]))
    mine = nbase + (0 if is_source else npieces)
    other = nbase + (npieces if is_source else 0)
    for i in range(npieces):
        out[:, mine + i] = p[i]
        out[:, other + i] = 1.0
    return out, vh


def accumulate(prod, order, truncate):
    """f32 running sum of exact f32 products in `order`; every partial sum rounded to nearest or toward zero."""
    acc = np.zeros(prod.shape[0], dtype=np.float32)
    for k in order:
        exact = acc.astype(np.float64) + prod[:, k]                   # exact in f64 (two f32 values)
        r = exact.astype(np.float32)
        if truncate:
            over = np.abs(r.astype(np.float64)) > np.abs(exact)
            r = np.where(over, np.nextafter(r, np.float32(0.0)), r)
        acc = r.astype(np.float32)
    return acc.astype(np.float64)


def orders(K):
    asc = list(range(K))
    # plane by plane (three chained MFMAs of K = 16), inside a plane the two K halves interleaved
    inter = [16 * m + 8 * (i & 1) + (i >> 1) for m in range(K // 16) for i in range(16)]
    rng = np.random.default_rng(7)
    return [asc, asc[::-1], inter, list(rng.permutation(K))]


def cell_error(E, xmin, in_a, in_b, na, nb, s):
    cell = np.where(xmin > 6 * E, E / (2 * np.sqrt(np.maximum(xmin - 2 * E, 1e-300))), np.sqrt(E))
    return cell + 1.001 * (in_a * np.sqrt(na) + in_b * np.sqrt(nb)) + 2.0 ** -20 / s


CASES = [
    # (layout, dim, how the pair of frames is made)
    (3, 13, "gauss"), (3, 13, "close"), (3, 13, "c0"), (3, 13, "tiny2"), (3, 1, "gauss"), (3, 2, "close"), (3, 7, "c0"),
    (2, 13, "gauss"), (2, 13, "close"), (2, 13, "tiny2"),
    (1, 14, "gauss"), (1, 40, "gauss"), (1, 40, "close"), (1, 42, "c0"),
]


@pytest.mark.parametrize("layout,dim,kind", CASES)
def test_filter_cell_stays_inside_the_margin(layout, dim, kind):
    rng = np.random.default_rng(1000 * layout + 17 * dim + len(kind))
    n = 1500
    amp = 10.0 ** rng.uniform(-3, 3)
    a = rng.standard_normal((n, dim)) * amp
    if kind == "close":                       # near-copies: the cancellation regime the certificate is for
        b = a + rng.standard_normal((n, dim)) * amp * 10.0 ** rng.uniform(-7, -1, size=(n, 1))
    elif kind == "c0":                        # cepstral shape: the first value carries the norm
        a[:, 0] = amp * (20.0 + rng.standard_normal(n))
        b = rng.standard_normal((n, dim)) * amp
        b[:, 0] = amp * (20.0 + rng.standard_normal(n))
    elif kind == "tiny2":                     # values 2^8 and more below the largest: subnormal second pieces
        a = a * 2.0 ** -9
        a[0, 0] = amp * 3.0
        b = rng.standard_normal((n, dim)) * amp * 2.0 ** -9
    else:
        b = rng.standard_normal((n, dim)) * amp
    a = a.astype(np.float32).astype(np.float64)
    b = b.astype(np.float32).astype(np.float64)
    na_f = (a ** 2).sum(-1)
    nb_f = (b ** 2).sum(-1)
    vmax = max(np.abs(a).max(), np.abs(b).max()) * 1.000001
    s = common_scale(vmax, max(na_f.max(), nb_f.max()) * 1.000002)
    ra, ah = build(a, s, layout, True)
    rb, bh = build(b, s, layout, False)
    # what the records round away, measured (dtw_filter.hip stores the maximum per segment; here the frames' own)
    res_a = np.sqrt(((a * s - ah) ** 2).sum(-1)) / s * 1.0000002
    res_b = np.sqrt(((b * s - bh) ** 2).sum(-1)) / s * 1.0000002
    prod = ra * rb                                                    # f16 x f16: exact in f32, hence in f64
    assert np.array_equal(prod, prod.astype(np.float32).astype(np.float64))
    c_true = np.sqrt(((a - b) ** 2).sum(-1))
    # the margin's inputs: per-SEGMENT maxima in the library, here the frames' own norms, rounded up like the kernel's f32
    na, nb = na_f * 1.000002, nb_f * 1.000002
    E = 256 * U * (na + nb) + 2.0 ** -12 / s ** 2
    in_a = 2.0 ** -11 if layout == 1 else 2.0 ** -22
    in_b = 2.0 ** -22 if layout == 2 else 2.0 ** -11
    worst = 0.0
    for order in orders(prod.shape[1]):
        for truncate in (False, True):
            x = accumulate(prod, order, truncate) / s ** 2            # the accumulator, unscaled
            c_filt = np.sqrt(np.abs(x))
            slack = 2 * U * c_filt                                    # v_sqrt_f32 within 1 ulp: priced by the path term
            for xmin in (np.zeros_like(x), np.abs(x)):                # no certificate / the cell as its own certificate
                bound = cell_error(E, xmin, in_a, in_b, na, nb, s) + slack
                err = np.abs(c_filt - c_true)
                assert (err <= bound).all(), (layout, dim, kind, truncate, float((err / bound).max()))
                # round 4: the rounding term from the MEASURED residuals of the two frames (dtw_margin.hpp ra, rb)
                tight = cell_error(E, xmin, 0.0, 0.0, na, nb, s) + 1.001 * (np.minimum(res_a, in_a * np.sqrt(na)) +
                                                                             np.minimum(res_b, in_b * np.sqrt(nb))) + slack
                assert (tight <= bound * (1 + 1e-12)).all()
                assert (err <= tight).all(), (layout, dim, kind, truncate, float((err / tight).max()))
                worst = max(worst, float((err / tight).max()))
    assert worst > 1e-4          # the emulation is not vacuous: the error is a visible fraction of the bound somewhere


def test_layout3_represented_target_is_what_the_norm_describes():
    # the accumulator of layout 3 is |a~ - b~|^2 up to accumulation error: cross terms and norms describe ONE rounded frame
    rng = np.random.default_rng(5)
    a = rng.standard_normal((200, 13)).astype(np.float32).astype(np.float64)
    b = rng.standard_normal((200, 13)).astype(np.float32).astype(np.float64)
    s = common_scale(4.0, 60.0)
    ra, ah = build(a, s, 3, True)
    rb, bh = build(b, s, 3, False)
    exact = (ra * rb).sum(-1)                                         # the K = 32 dot in exact arithmetic
    rep = ((ah - bh) ** 2).sum(-1)                                    # |a~ - b~|^2, scaled
    # the only term left out is (a_e)2 (b_e)2 for e < 2, and the norms' third piece
    assert np.abs(exact - rep).max() <= 4 * 2.0 ** -22 * ((ah ** 2).sum(-1) + (bh ** 2).sum(-1)).max() + 2.0 ** -20
    assert np.abs(bh[:, 2:] - f16(b[:, 2:] * s)).max() == 0.0         # one piece from the third value on
