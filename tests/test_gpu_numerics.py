"""GPU numerics corner cases of the dtw filter's f16 records (csrc/dtw_filter.hip, csrc/select.hip).

* Flat spectra: frames of 17...42 values whose scaled squared norm would pass the f16 maximum with
  max |s v| in [32, 64) -- the norm pieces must stay finite (the common scale steps down), the pair
  must stay in the search and the filter matrix inside its bound.
* Subnormal second pieces: values 2^8 and more below the set's largest have a subnormal second f16
  piece; the error model prices it at 2^-25 (scaled units) per value, which holds only if the matrix
  pipe does not flush subnormal inputs.  Single-frame segments make the filter's output one local
  cost, so the per-cell claim of select.hip is checked directly.
"""
import math
import os

import numpy as np
import pytest

from soundsym_amd import Engine
from soundsym_amd.engine import pack_segments
from bounds import input_rounding, worst_case_bound as _worst_case_bound

pytestmark = pytest.mark.gpu
EXACT_RTOL = 1e-12


@pytest.mark.parametrize("dim", [18, 40, 42, 64])
@pytest.mark.parametrize("kind", ["constant", "uniform"])
def test_dtw_flat_spectrum_norm_pieces_stay_finite(oracle, dim, kind):
    rng = np.random.default_rng(0xF1A7 + dim)
    n, m, f = 24, 16, 20
    if kind == "constant":       # every value near the set's maximum: |frame|^2 = dim * max^2
        src = (1.9 + 0.01 * rng.standard_normal((n, f, dim))).astype(np.float32)   # scaled by 32: ~61
        src *= np.where(rng.random((n, f, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    else:
        src = rng.uniform(-1.0, 1.0, (n, f, dim)).astype(np.float32)
    pick = rng.permutation(n)[:m]
    tgt = (src[pick] + 0.01 * rng.standard_normal((m, f, dim))).astype(np.float32)
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(src.reshape(-1), so, dim), e.queries(tgt.reshape(-1), to, dim)
    idx, cost = e.match(d, q)
    assert e.timings()["used_filter"] == 1
    want_idx, want_cost, mat = oracle.dtw_match_all(src.reshape(-1).astype(np.float64), so,
                                                    tgt.reshape(-1).astype(np.float64), to, dim, want_matrix=True)
    assert np.array_equal(idx, want_idx) and np.array_equal(idx, pick)
    assert np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    filt = e.pair_matrix(d, q, exact=False)
    assert np.isfinite(filt).all(), "a norm piece overflowed f16: the pair dropped out of the filter"
    used = min(dim, 42)
    bound, s = _worst_case_bound(src, tgt, used, f, f)
    if dim <= 42:
        assert (np.abs(filt - mat) <= bound + 1e-5 * mat).all(), (np.abs(filt - mat).max(), bound)
    else:                         # the first 42 values only: a lower bound of the cost
        assert (filt <= mat + bound + 1e-5 * mat).all()
    if kind == "constant":
        assert s * float(np.abs(src).max()) < 32.0, "the scale should have stepped down for the norms"
    e.close()


def test_dtw_filter_subnormal_second_pieces_stay_inside_the_cell_bound():
    # single-frame segments: the filter's cost IS one local cost c~(a, b)
    dim = 13
    # sources: every value 2^-4 + ~2^-15, so H1 = 2^-4 and H2 ~ 2^-15 (subnormal f16); a few variations
    k = np.arange(64, dtype=np.float64)
    a = 2.0 ** -4 + (0.45 + 0.049 * ((k[:, None] * 7 + np.arange(dim)[None, :] * 3) % 10) / 10.0) * 2.0 ** -14
    a[1::2] *= -1.0
    # one extra source whose single large value fixes the common scale at 1 (max |v| = 40)
    big = np.zeros((1, dim))
    big[0, 0] = 40.0
    src = np.concatenate([a, big]).astype(np.float32)
    # targets: mid-sized values (|b| well below 29, where the norm term of E does not cover a flush)
    tv = np.array([0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 2.0 ** -4])
    tgt = np.repeat(tv[:, None], dim, axis=1).astype(np.float32)
    tgt[::2, ::2] *= -1.0
    so = np.arange(src.shape[0] + 1, dtype=np.uint64)
    to = np.arange(tgt.shape[0] + 1, dtype=np.uint64)
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(src.reshape(-1), so, dim), e.queries(tgt.reshape(-1), to, dim)
    filt = e.pair_matrix(d, q, exact=False)
    exact = e.pair_matrix(d, q, exact=True)
    e.close()
    a64, b64 = src.astype(np.float64), tgt.astype(np.float64)
    want = np.sqrt(((a64[:, None, :] - b64[None, :, :]) ** 2).sum(-1))
    assert np.allclose(exact, want, rtol=1e-14, atol=0)
    u = 2.0 ** -24
    na = (a64 ** 2).sum(-1)[:, None] * 1.000002      # the kernel's norms are f32, rounded up
    nb = (b64 ** 2).sum(-1)[None, :] * 1.000002
    E = 256 * u * (na + nb) + 2.0 ** -12             # common scale 1: max |v| = 40 in [32, 64)
    x = want ** 2
    certified = x > 8 * E                            # (x >= m - E with m > 6E certain)
    cell = np.where(certified, E / (2 * np.sqrt(np.maximum(x - 3 * E, 1e-300))), np.sqrt(E))
    # (the targets are f16 values: whatever the record layout keeps of a target, nothing is rounded away here)
    in_a, in_b = input_rounding(dim)[0], 0.0
    assert (tgt.astype(np.float16).astype(np.float32) == tgt).all()
    cell = cell + 1.001 * (in_a * np.sqrt(na) + in_b * np.sqrt(nb)) + 2.0 ** -20
    bound = 1.02 * cell + 7 * u * filt
    worst = np.abs(filt - want) / bound
    assert (worst <= 1.0).all(), ("filter cell outside the bound of select.hip -- subnormal f16 pieces flushed?",
                                  float(worst.max()), np.unravel_index(worst.argmax(), worst.shape))


def test_dtw_exact_pipelined_wave_giving_up_is_redone_not_dropped(oracle, monkeypatch):
    # dtw_exact_pipe_kernel bounds its spin; a wave that gives up must not turn its pair into "never a
    # candidate".  SSYM_EXACT_PIPE_FORCE_GIVEUP=1 makes wave 1 of the first pair give up at once: the
    # one-wave-per-pair kernel behind it scores the list again, results stay exact, exact_redone says so.
    from soundsym_amd import synth
    dim, n, m = 13, 6, 5
    st = synth.Stream(0x5EED6100)
    ls = 200 + st.integers(n, 200)
    lt = 100 + st.integers(m, 200)
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in ls]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in lt]
    tgt[0] = src[0][3:].copy()           # pair 0 of the all-pairs list holds target 0's neighbour
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim)
    assert want_idx[0] == 0
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost = e.match(d, q, force_exact=True)
    assert e.timings()["exact_redone"] == 0
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    monkeypatch.setenv("SSYM_EXACT_PIPE_FORCE_GIVEUP", "1")
    idx2, cost2 = e.match(d, q, force_exact=True)
    assert e.timings()["exact_redone"] == 1
    assert np.array_equal(idx2, idx) and np.array_equal(cost2, cost)
    # through the filter path as well (the candidates' re-scoring uses the same launcher)
    idx3, cost3 = e.match(d, q)
    assert e.timings()["used_filter"] == 1 and e.timings()["exact_redone"] >= 1
    assert np.array_equal(idx3, idx) and np.array_equal(cost3, cost)
    monkeypatch.delenv("SSYM_EXACT_PIPE_FORCE_GIVEUP")
    e.close()


# ---------------------------------------------------------------------------------------------
# refcos through the f64 matrix pipe (csrc/refcos_mfma.hip): filter + exact keys, results bit for bit
# ---------------------------------------------------------------------------------------------
def _refcos_sets(seed, n, m, fmin, fmax, dim, scale=0.05):
    from soundsym_amd import synth
    rs, rt = synth.make_ragged(n, m, fmin, fmax, dim, seed)
    src = [s.astype(np.float64) * scale for s in rs]
    tgt = [t.astype(np.float64) * scale for t in rt]
    return src, tgt


@pytest.mark.parametrize("n,m,fmin,fmax,dim", [(300, 260, 1, 40, 12), (520, 130, 20, 21, 13), (256, 256, 100, 128, 12)])
def test_refcos_mfma_search_is_bit_exact(oracle, n, m, fmin, fmax, dim):
    src, tgt = _refcos_sets(0x5EED7000 + n, n, m, fmin, fmax, dim)
    # planted near-duplicates, exact duplicates (ties: lowest index wins), an all-zero and an empty segment
    tgt[0] = src[7].copy()
    src[11] = src[7].copy()
    tgt[1] = src[3][: max(1, src[3].shape[0] // 2)].copy()
    src[5] = np.zeros_like(src[5])
    tgt[2] = np.zeros_like(tgt[2])
    src[9] = np.zeros((0, dim))
    tgt[3] = np.zeros((0, dim))
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, val = e.match(d, q)
    tm = e.timings()
    assert tm["used_filter"] == 1, "the search should have gone through the matrix pipe"
    assert tm["refcos_filter"] == 2, "finite values of ordinary size: the integer filter (csrc/refcos_q8.hip)"
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    assert tm["n_refined"] < 8 * m + 64, tm            # a handful of candidates per target, not the matrix
    # per-target distances (morph_to, src/sound.rs:440-446), NaN distance included
    dist = np.linspace(-0.5, 1.5, m)
    dist[4] = np.nan
    idx2, val2 = e.match(d, q, distance=dist)
    w_idx, w_val = oracle.refcos_match_all(sf, so, tf, to, dim, distance=dist)
    assert np.array_equal(idx2, w_idx) and np.array_equal(val2, w_val)
    # the filter's similarities against the bit-exact ones, inside the bound the kernel uses
    exact = e.pair_matrix(d, q)
    filt = nat_pair_matrix(e, d, q, 2)
    na = np.array([float((s.reshape(-1) ** 2).sum()) for s in src])
    nb = np.array([float((t.reshape(-1) ** 2).sum()) for t in tgt])
    L = np.minimum.outer(np.array([s.size for s in src]), np.array([t.size for t in tgt]))
    nrm = np.outer(na, nb)
    with np.errstate(divide="ignore", invalid="ignore"):
        bound = (3 * L + 16) * 2.0 ** -53 * 1.02 * np.sqrt(nrm) / nrm + 4 * 2.0 ** -53 * np.abs(exact)
    ok = nrm > 0
    assert np.array_equal(np.isnan(filt[~ok]), np.isnan(exact[~ok]))
    assert (np.abs(filt[ok] - exact[ok]) <= bound[ok]).all(), float((np.abs(filt[ok] - exact[ok]) / bound[ok]).max())
    # the integer filter's similarities (exact = 3) inside ITS bound, on a sample of pairs: records and error term
    # restated in tests/bounds.py; and they are the integer dots themselves -- the number the host forms from the digits
    import bounds
    qfilt = nat_pair_matrix(e, d, q, 3)
    rng = np.random.default_rng(n)
    picks = [(7, 0), (11, 0), (3, 1), (5, 4), (6, 2), (9, 5), (8, 3)] + [(int(rng.integers(n)), int(rng.integers(m))) for _ in range(120)]
    qs = {i: bounds.q8_quantise(src[i]) for i, _ in picks}
    qt = {j: bounds.q8_quantise(tgt[j]) for _, j in picks}
    for i, j in picks:
        if not nrm[i, j] > 0:
            assert np.isnan(qfilt[i, j]) == np.isnan(exact[i, j])
            continue
        with np.errstate(all="ignore"):
            ia, ib = float(np.float64(1.0) / np.float64(na[i])), float(np.float64(1.0) / np.float64(nb[j]))
        dq, gk, a2b2, extra = bounds.q8_dot_and_extra(qs[i], qt[j], src[i].size, tgt[j].size, ia, ib)
        # (to 12 digits: this test's norms are numpy sums, the library's the reference's sequential fold)
        assert abs(qfilt[i, j] - dq / nrm[i, j]) <= 1e-12 * abs(dq / nrm[i, j]), (i, j, qfilt[i, j], dq / nrm[i, j])
        assert abs(qfilt[i, j] - exact[i, j]) <= 1.0001 * extra + bound[i, j], (i, j, qfilt[i, j], exact[i, j], extra)
    e.close()


def test_refcos_mfma_segments_outside_the_plain_range_take_the_long_epilogue(oracle):
    """The epilogue has a short form for waves whose 64 + 64 segments are all plain (norm in [1e-139, 1e139], finite
    distance) and the full refcos_key_interval for the rest: tiny, huge, infinite and NaN segments are spread over the
    quadrants of the 128 x 128 tiles so that both forms run inside one workgroup, and every index and value must
    still be the oracle's bit for bit."""
    n = m = 384
    dim = 12
    src, tgt = _refcos_sets(0x5EED7A00, n, m, 90, 128, dim)
    src[70] = src[70] * 1e-80                     # norm ~1e-157: 1 / norm finite, products with huge partners fine
    src[140] = src[140] * 1e80                    # norm ~1e163
    src[141] = src[141] * 1e-160                  # norm underflows to 0 (or a subnormal)
    src[300][3, 2] = np.inf
    tgt[66] = tgt[66] * 1e-75
    tgt[130][0, 0] = np.nan
    tgt[200][5, 1] = -np.inf
    tgt[260] = tgt[260] * 1e78
    tgt[10] = src[20].copy()                      # (plain near-duplicates elsewhere)
    tgt[333] = src[350].copy()
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    with np.errstate(all="ignore"):
        want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    idx, val = e.match(d, q)
    assert e.timings()["used_filter"] == 1
    assert e.timings()["refcos_filter"] == 1, "values that are not finite: the f64 matrix pipe, not the integer filter"
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val, equal_nan=True)
    assert idx[10] == 20 and idx[333] == 350
    dist = np.linspace(-0.5, 1.5, m)
    dist[77] = np.nan
    dist[150] = np.inf
    dist[290] = 1e301
    with np.errstate(all="ignore"):
        w_idx, w_val = oracle.refcos_match_all(sf, so, tf, to, dim, distance=dist)
    idx2, val2 = e.match(d, q, distance=dist)
    assert np.array_equal(idx2, w_idx) and np.array_equal(val2, w_val, equal_nan=True)
    e.close()


def test_refcos_both_filters_give_the_same_search(oracle, monkeypatch):
    """SSYM_REFCOS_Q8=0 (read per call) sends the search through the f64 matrix pipe instead of the integer filter:
    the two candidate lists differ, indices and values may not -- here against each other and against the oracle, on
    near-ties closer than either filter resolves."""
    n, m, dim = 384, 256, 12
    src, tgt = _refcos_sets(0x5EED7B00, n, m, 60, 128, dim)
    for j in range(0, m, 8):                       # targets that are copies of a source; every third with a last-place change
        tgt[j] = src[(5 * j) % n].copy()
        if j % 3 == 0:
            tgt[j][0, 0] = np.nextafter(tgt[j][0, 0], np.inf)
    for i in range(0, n, 16):                      # duplicated sources: the lower index must win
        src[(i + 200) % n] = src[i].copy()
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    got = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("SSYM_REFCOS_Q8", knob)
        idx, val = e.match(d, q)
        tm = e.timings()
        assert tm["refcos_filter"] == (2 if knob == "1" else 1), tm
        assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
        got[knob] = tm["n_refined"]
    assert got["1"] >= got["0"] >= m               # the integer filter keeps a little more, both at least a winner per target
    assert got["1"] < 4 * m, got
    e.close()


def nat_pair_matrix(e, d, q, mode):
    import ctypes
    from soundsym_amd import _native as nat
    out = np.zeros((d.n, q.n), dtype=np.float64)
    nat.check(nat.lib().ssym_pair_matrix(e.ctx, d.ptr, q.ptr, mode, out.ctypes.data), e.ctx)
    return out


@pytest.mark.parametrize("k", [2, 5, 8, 17, 64])
def test_refcos_topk_through_the_matrix_pipe_is_bit_exact(oracle, k):
    # ssym_match_topk on the reference's metric: threshold = the k-th smallest distinct key_hi a wave's rows offer,
    # exact keys of what it cannot exclude, k rounds of the first-minimum fold -- rows equal to the oracle's bit for bit
    n, m, dim = 420, 200, 12
    src, tgt = _refcos_sets(0x5EED7100 + k, n, m, 2, 30, dim)
    tgt[0] = src[7].copy()
    src[11] = src[7].copy()
    src[300] = src[7].copy()                                        # a three-way tie: lowest index first at every rank
    src[5] = np.zeros_like(src[5])                                  # norm 0: never an entry
    src[9] = np.zeros((0, dim))
    tgt[3] = np.zeros((0, dim))                                     # an empty target: its row is all NO_MATCH
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    dist = np.linspace(-0.2, 1.4, m)
    for dd in (None, dist):
        idx, key = e.match_topk(d, q, k, dd)
        tm = e.timings()
        # (the waves' own k-th smallest bounds only decide what is listed; the candidates are selected by the k-th smallest
        #  bound over ALL of a target's listed pairs: csrc/refcos_mfma.hip, refcos_topk_*)
        want_filter = 1 if os.environ.get("SSYM_REFCOS_Q8") == "0" else 2     # (finite values: the integer filter, any k up to 64)
        assert tm["used_filter"] == 1 and tm["refcos_filter"] == want_filter, tm
        assert tm["n_refined"] <= (k + 6) * m, tm                            # k candidates per target and what ties with them
        if k <= 8:
            assert tm["n_refined"] < n * m // 2, tm                   # candidates, not the matrix
        want_idx, want_key = oracle.topk(oracle.refcos_matrix(sf, so, tf, to, dim), k, distance=dd)
        have = want_idx >= 0
        got_idx = idx.astype(np.int64)
        got_idx[idx == 0xFFFFFFFF] = -1
        assert np.array_equal(got_idx, want_idx)
        assert np.array_equal(key[have], want_key[have]) and np.isnan(key[~have]).all()
    one, val = e.match(d, q)
    idx, key = e.match_topk(d, q, k)
    won = idx[:, 0] != 0xFFFFFFFF
    assert np.array_equal(one[won], idx[won, 0]) and np.array_equal(val[won], key[won, 0])
    e.close()


def test_refcos_mfma_overflowing_list_falls_back_to_the_exact_kernel(oracle):
    # every source identical: every pair ties, list 1 cannot hold them -> the exact tile kernel takes the call
    dim, f, n, m = 12, 6, 2200, 600
    rng = np.random.default_rng(3)
    one = rng.standard_normal((f, dim)) * 0.1
    src = [one.copy() for _ in range(n)]
    tgt = [rng.standard_normal((f, dim)) * 0.1 for _ in range(m)]
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    idx, val = e.match(e.dictionary(sf, so, dim), e.queries(tf, to, dim))
    assert e.timings()["used_filter"] == 0 and e.timings()["refcos_filter"] == 0      # (both filters' lists overflowed)
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    e.close()


def test_refcos_integer_filter_at_its_longest_segments(oracle):
    """Segments of 32768 values -- the length up to which three digit products per value sum exactly in 32 bits, 1024
    chunks through the kernel's ring --, values at full amplitude with alternating signs (digits near their bounds):
    the integer filter takes the search and the result is the oracle's; one value more per segment and the sets go to
    the f64 filter."""
    n = m = 256
    dim, f = 128, 256
    rng = np.random.default_rng(77)
    def seg():
        x = rng.uniform(0.97, 1.0, (f, dim)) * rng.choice([-1.0, 1.0], (f, dim))
        return x * 3.7
    src = [seg() for _ in range(n)]
    tgt = [seg() for _ in range(m)]
    tgt[3] = src[200].copy()
    tgt[9] = -src[17]
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, val = e.match(d, q)
    assert e.timings()["refcos_filter"] == 2, e.timings()
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    assert idx[3] == 200
    d.close(); q.close()
    src[5] = np.concatenate([src[5], np.ones((1, dim))])            # 32896 values: beyond the records
    sf, so = pack_segments(src, dim)
    d = e.dictionary(sf, so, dim)
    q = e.queries(tf, to, dim)
    idx, val = e.match(d, q)
    assert e.timings()["refcos_filter"] == 1, e.timings()
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    e.close()


def test_refcos_integer_filter_overflowing_hands_the_search_to_the_f64_filter(oracle):
    """Sources that differ from one another by 1e-9 of their size: 23 bits of fixed point cannot tell them apart -- the
    integer filter lists every pair and its list overflows --, f64 can: the f64 filter takes the search (not the exact
    tile kernel on every pair), and the result is the oracle's."""
    dim, f, n, m = 12, 6, 2200, 600
    rng = np.random.default_rng(31)
    one = rng.standard_normal((f, dim)) * 0.1
    src = [one * (1.0 + 1e-9 * rng.standard_normal((f, dim))) for _ in range(n)]
    tgt = [rng.standard_normal((f, dim)) * 0.1 for _ in range(m)]
    tgt[5] = src[1234].copy()
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    e = Engine(metric="refcos", dtype="f64")
    idx, val = e.match(e.dictionary(sf, so, dim), e.queries(tf, to, dim))
    tm = e.timings()
    assert tm["used_filter"] == 1 and tm["refcos_filter"] == 1, tm
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    e.close()


def test_dtw_alternating_amplitudes_reuse_the_dictionary_records(oracle):
    # quiet and loud target batches in turn: the dictionary's f16 records are built for the loud scale once and
    # then kept (a smaller common scale is always admissible); results stay exact either way
    from soundsym_amd import synth
    g = synth.make_grid(96, 40, 24, 13, 0x5EED0B00)
    so = np.arange(97, dtype=np.uint64) * 24
    to = np.arange(41, dtype=np.uint64) * 24
    e = Engine(metric="dtw", dtype="f32")
    d = e.dictionary(g.sources.reshape(-1), so, 13)
    for rep in range(3):
        for amp in (1.0, 6.0):
            tgt = (g.targets * np.float32(amp)).astype(np.float32)
            q = e.queries(tgt.reshape(-1), to, 13)
            idx, cost = e.match(d, q)
            assert e.timings()["used_filter"] == 1
            want_idx, want_cost = oracle.dtw_match_all(g.sources.reshape(-1).astype(np.float64), so,
                                                       tgt.reshape(-1).astype(np.float64), to, 13)
            assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
            filt = e.pair_matrix(d, q, exact=False)
            _, _, mat = oracle.dtw_match_all(g.sources.reshape(-1).astype(np.float64), so,
                                             tgt.reshape(-1).astype(np.float64), to, 13, want_matrix=True)
            # the bound with the scale actually in use: at most 16 times smaller than this batch's ideal scale
            bound, s = _worst_case_bound(g.sources, tgt, 13, 24, 24)
            assert (np.abs(filt - mat) <= 16 * bound + 1e-5 * mat).all()
            q.close()
    e.close()


_PK_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from soundsym_amd import Engine
rng = np.random.default_rng(77)
dim = 13
def ragged(n, lo, hi, scale):
    lens = rng.integers(lo, hi + 1, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return (rng.standard_normal((int(off[-1]), dim)) * scale).astype(np.float32).reshape(-1), off
sf, so = ragged(40, 49, 200, 1.0)        # more than 48 frames: the 64-row passes, one to four of them
tf, to = ragged(70, 1, 150, 1.0)
e = Engine(metric="dtw", dtype="f32")
d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
idx, cost = e.match(d, q)
np.savez(sys.argv[2], filt=e.pair_matrix(d, q, exact=False), idx=idx, cost=cost)
"""


def test_dtw_filter_packed_add_experiment_is_bit_identical_to_the_product_kernel(tmp_path):
    # SSYM_FILTER_PK=1 (csrc/dtw_filter_pk_kernel.hpp, off in the product) performs the same IEEE operations per cell in
    # another order of cells: its whole filter matrix must equal the product kernel's bit for bit
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, env in (("plain", {"SSYM_FILTER_PK": "0"}), ("packed", {"SSYM_FILTER_PK": "1"})):
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _PK_SCRIPT, root, path], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = np.load(path)
    assert np.isfinite(out["plain"]["filt"]).any()
    assert np.array_equal(out["plain"]["filt"], out["packed"]["filt"], equal_nan=True)
    assert np.array_equal(out["plain"]["idx"], out["packed"]["idx"]) and np.array_equal(out["plain"]["cost"], out["packed"]["cost"])
