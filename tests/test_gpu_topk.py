"""ssym_match_topk against the oracle's sort-based restatement.

refcos: indices and keys bit-exact.  dtw: indices identical, costs at the exact kernel's tolerance,
through the MFMA filter (whose candidate rule must keep every member of the exact top k) and
through the all-pairs exact path.
"""
import numpy as np
import pytest

from soundsym_amd import Engine, SsymError, synth
from soundsym_amd._native import NO_MATCH
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu
EXACT_RTOL = 1e-12


@pytest.fixture(scope="module")
def refcos():
    e = Engine(metric="refcos", dtype="f64")
    yield e
    e.close()


@pytest.fixture(scope="module")
def dtw():
    e = Engine(metric="dtw", dtype="f32")
    yield e
    e.close()


def _check_rows(idx, cost, want_idx, want_val, exact):
    assert idx.shape == want_idx.shape
    got = np.where(idx == NO_MATCH, -1, idx.astype(np.int64))
    assert np.array_equal(got, want_idx)
    have = want_idx >= 0
    assert np.isnan(cost[~have]).all()
    if exact:
        assert np.array_equal(cost[have], want_val[have])
    else:
        assert np.allclose(cost[have], want_val[have], rtol=EXACT_RTOL, atol=0)


@pytest.mark.parametrize("k", [1, 3, 8])
def test_refcos_topk_bit_exact(refcos, oracle, k):
    rng = np.random.default_rng(0x70B0 + k)
    segs = [rng.normal(size=(int(rng.integers(1, 9)), 12)) for _ in range(150)]
    segs[40] = segs[7].copy()
    segs[90] = segs[7].copy()                                   # a three-way tie
    segs.append(np.full((3, 12), np.nan))
    tg = [rng.normal(size=(int(rng.integers(1, 9)), 12)) for _ in range(33)]
    tg[0] = segs[7].copy()
    sf, so = pack_segments(segs, 12)
    tf, to = pack_segments(tg, 12)
    dist = rng.uniform(0.0, 1.2, size=len(tg))
    for d in (None, dist):
        idx, key = refcos.match_topk(refcos.dictionary(sf, so, 12), refcos.queries(tf, to, 12), k, d)
        want_idx, want_key = oracle.topk(oracle.refcos_matrix(sf, so, tf, to, 12), k, distance=d)
        _check_rows(idx, key, want_idx, want_key, exact=True)
    # entry 0 is the plain match
    one, val = refcos.match(refcos.dictionary(sf, so, 12), refcos.queries(tf, to, 12), dist)
    assert np.array_equal(one, idx[:, 0]) and np.array_equal(val, key[:, 0])


def test_refcos_topk_fewer_candidates_than_k(refcos, oracle):
    # two sources, one of them NaN: rows hold one entry and NO_MATCH / NaN after it
    segs = [np.array([[0.5, 0.25]]), np.array([[np.nan, 1.0]])]
    sf, so = pack_segments(segs, 2)
    tf, to = pack_segments([np.array([[0.5, 0.5]]), np.array([[1.0, 0.0]])], 2)
    idx, key = refcos.match_topk(refcos.dictionary(sf, so, 2), refcos.queries(tf, to, 2), 4)
    want_idx, want_key = oracle.topk(oracle.refcos_matrix(sf, so, tf, to, 2), 4)
    _check_rows(idx, key, want_idx, want_key, exact=True)
    assert (idx[:, 1:] == NO_MATCH).all()


@pytest.mark.parametrize("k,planted", [(2, True), (5, True), (5, False), (16, False)])
def test_dtw_topk_through_the_filter(dtw, oracle, k, planted):
    if planted:
        g = synth.make_grid(96, 40, 24, 13, 0x5EED0700 + k)
        src, tgt = g.sources, g.targets
    else:
        src = synth.make_grid(96, 1, 24, 13, 0x5EED0710 + k).sources
        tgt = synth.make_grid(40, 1, 24, 13, 0x5EED0720 + k).sources
    src = src.copy()
    src[50] = src[3]                                            # exact duplicate: index order decides
    n, m, f = src.shape[0], tgt.shape[0], src.shape[1]
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    d, q = dtw.dictionary(src.reshape(-1), so, 13), dtw.queries(tgt.reshape(-1), to, 13)
    _, _, mat = oracle.dtw_match_all(src.reshape(-1).astype(np.float64), so, tgt.reshape(-1).astype(np.float64),
                                     to, 13, want_matrix=True)
    rng = np.random.default_rng(k)
    for dist in (None, rng.uniform(0.0, float(np.median(mat)), size=m)):
        idx, cost = dtw.match_topk(d, q, k, dist)
        assert dtw.timings()["used_filter"] == 1
        want_idx, _ = oracle.topk(mat, k, distance=dist, default_distance=0.0, fold_start=float("inf"))
        want_cost = np.where(want_idx >= 0, mat[np.maximum(want_idx, 0), np.arange(m)[:, None]], np.nan)
        _check_rows(idx, cost, want_idx, want_cost, exact=False)
        idx2, cost2 = dtw.match_topk(d, q, k, dist, force_exact=True)     # all-pairs exact path
        _check_rows(idx2, cost2, want_idx, want_cost, exact=False)
    one, c1 = dtw.match(d, q, dist)
    assert np.array_equal(one, idx[:, 0]) and np.array_equal(c1, cost[:, 0])


def test_dtw_topk_ragged_with_k_above_n(dtw, oracle):
    rng = np.random.default_rng(0x70B5)
    segs = [rng.normal(size=(int(rng.integers(1, 30)), 13)).astype(np.float32) for _ in range(6)]
    tg = [rng.normal(size=(int(rng.integers(1, 30)), 13)).astype(np.float32) for _ in range(9)]
    sf, so = pack_segments(segs, 13, np.float32)
    tf, to = pack_segments(tg, 13, np.float32)
    idx, cost = dtw.match_topk(dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13), 8)
    _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13, want_matrix=True)
    want_idx, _ = oracle.topk(mat, 8, default_distance=0.0, fold_start=float("inf"))
    want_cost = np.where(want_idx >= 0, mat[np.maximum(want_idx, 0), np.arange(9)[:, None]], np.nan)
    _check_rows(idx, cost, want_idx, want_cost, exact=False)
    assert (idx[:, 6:] == NO_MATCH).all()


def test_topk_rejects_bad_k(refcos):
    d = refcos.dictionary(np.ones(4), [0, 2], 2)
    q = refcos.queries(np.ones(4), [0, 2], 2)
    for k in (0, 65):
        with pytest.raises(SsymError):
            refcos.match_topk(d, q, k)


def test_dtw_topk_medium_unplanted(dtw, oracle):
    # no near-duplicates: ~10^2 pairs per target survive the worst-case stage, the certificates
    # shrink that to about k; every exact top-10 member must still be there
    n, m, f, k = 512, 128, 32, 10
    src = synth.make_grid(n, 1, f, 13, 0x5EED0730).sources
    tgt = synth.make_grid(m, 1, f, 13, 0x5EED0731).sources
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    idx, cost = dtw.match_topk(dtw.dictionary(src.reshape(-1), so, 13), dtw.queries(tgt.reshape(-1), to, 13), k)
    tm = dtw.timings()
    assert tm["used_filter"] == 1 and tm["n_refined"] < n * m // 4
    _, _, mat = oracle.dtw_match_all(src.reshape(-1).astype(np.float64), so, tgt.reshape(-1).astype(np.float64),
                                     to, 13, nthreads=oracle.max_threads(), want_matrix=True)
    want_idx, _ = oracle.topk(mat, k, default_distance=0.0, fold_start=float("inf"))
    want_cost = mat[want_idx, np.arange(m)[:, None]]
    _check_rows(idx, cost, want_idx, want_cost, exact=False)


@pytest.mark.parametrize("dim,f,k,band", [(50, 40, 3, -1), (64, 70, 5, -1), (90, 33, 2, -1), (64, 48, 4, 12)])
def test_dtw_topk_on_wide_frames_uses_the_lower_bound_cascade(oracle, dim, f, k, band):
    # frames wider than the filter takes in: the k pairs the filter likes best are scored exactly, the largest of
    # their costs bounds the k-th best from above, and every pair whose lower bound stays below it is re-scored
    g = synth.make_grid(96, 256 if band >= 0 else 80, f, dim, 0x5EED0700 + dim)
    src = [x for x in g.sources.astype(np.float64)]
    tgt = [x for x in g.targets.astype(np.float64)]
    src[17] = src[5].copy()                                    # ties inside the top k: index order decides
    tgt[3] = src[5].copy()
    tgt[4] = np.zeros((0, dim))
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost = e.match_topk(d, q, k)
    assert e.timings()["used_filter"] == 1
    _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band, want_matrix=True,
                                     nthreads=oracle.max_threads())
    want_idx, _ = oracle.topk(mat, k, default_distance=0.0, fold_start=float("inf"))
    want_val = np.where(want_idx >= 0, mat[np.maximum(want_idx, 0), np.arange(len(tgt))[:, None]], np.nan)
    _check_rows(idx, cost, want_idx, want_val, exact=False)
    # fewer finite pairs than k: the rest of the row says so
    few = e.dictionary(sf[: int(so[2]) * dim], so[:3], dim)
    i2, c2 = e.match_topk(few, q, k)
    assert (i2[0, 2:] == NO_MATCH).all() and np.isnan(c2[0, 2:]).all()
    e.close()
