"""Seeded random differential tests: small ragged problems of every shape class against the oracle.

Covers what the hand-written cases may miss: empty segments anywhere, 1-frame segments, lengths
straddling tile / pass / chunk boundaries (16, 48, 64, 128), dims 1..90 (both record layouts and the
lower-bound cascade for frames wider than 42 values), bands
from 0 to beyond the filter's reach, squared local cost, f32 and f64 inputs, per-target distances.
"""
import numpy as np
import pytest

from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu
# SSYM_FUZZ_CASES=N widens the seeded sweeps (default 120 dtw / 24 refcos cases keep the suite short)
import os
_N_DTW = int(os.environ.get("SSYM_FUZZ_CASES", "120"))
_N_REFCOS = max(24, _N_DTW // 5)


def _ragged(st, n, lo, hi, dim, scale):
    lens = lo + st.integers(n, hi - lo + 1)
    return [(st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) * scale) for f in lens]


@pytest.mark.parametrize("case", range(_N_DTW))
def test_dtw_random_shapes(oracle, case):
    st = synth.Stream(0x5EED1000 + case)
    dim = int([1, 2, 5, 12, 13, 14, 20, 40, 42, 43, 64, 90][st.integers(1, 12)[0]])
    hi = int([3, 17, 33, 50, 66, 130, 150][st.integers(1, 7)[0]])
    lo = int(st.integers(1, 2)[0])                      # 0 or 1: empty segments allowed
    n, m = int(2 + st.integers(1, 30)[0]), int(1 + st.integers(1, 40)[0])
    band = int([-1, -1, -1, 0, 1, 5, 20, 47, 60][st.integers(1, 9)[0]])
    squared = bool(st.integers(1, 4)[0] == 0)
    dtype = ["f32", "f64"][int(st.integers(1, 2)[0])]
    scale = float([1e-3, 1.0, 300.0][st.integers(1, 3)[0]])
    src = _ragged(st, n, lo, hi, dim, scale)
    tgt = _ragged(st, m, lo, hi, dim, scale)
    if m > 2 and n > 3 and src[3].shape[0] > 0:
        tgt[1] = src[3].copy()                           # an exact match
        tgt[2] = src[3][: max(1, src[3].shape[0] - 1)].copy()
    if n > 6:
        src[6] = src[2].copy()                           # duplicates: first index wins
    npdt = np.float32 if dtype == "f32" else np.float64
    sf, so = pack_segments(src, dim, npdt)
    tf, to = pack_segments(tgt, dim, npdt)
    use_dist = bool(st.integers(1, 3)[0] == 0)
    e = Engine(metric="dtw", dtype=dtype, band=band, squared=squared)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    want_idx, want_cost, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim,
                                                    band=band, squared=squared, want_matrix=True, nthreads=4)
    if use_dist:
        fin = np.where(np.isfinite(mat), mat, np.nan)
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                dist = np.nan_to_num(np.nanmedian(fin, axis=0), nan=1.0)
        key = np.abs(mat - dist[None, :])
        key = np.where(np.isnan(key), np.inf, key)
        want_idx = np.where(np.isfinite(key).any(axis=0), key.argmin(axis=0), 0)
        want_cost = np.where(np.isfinite(key).any(axis=0), mat[want_idx, np.arange(m)], np.inf)
        idx, cost = e.match(d, q, distance=dist)
    else:
        idx, cost = e.match(d, q)
    info = dict(case=case, dim=dim, hi=hi, n=n, m=m, band=band, squared=squared, dtype=dtype, scale=scale,
                dist=use_dist, filter=e.timings()["used_filter"])
    assert np.array_equal(idx, want_idx), info
    assert np.array_equal(np.isinf(cost), np.isinf(want_cost)), info
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=1e-12, atol=0), info
    e.close()


@pytest.mark.parametrize("case", range(_N_REFCOS))
def test_refcos_random_shapes(oracle, case):
    st = synth.Stream(0x5EED2000 + case)
    dim = int([1, 3, 12, 13, 40][st.integers(1, 5)[0]])
    hi = int([2, 9, 40, 100][st.integers(1, 4)[0]])
    n, m = int(1 + st.integers(1, 70)[0]), int(1 + st.integers(1, 70)[0])
    src = _ragged(st, n, 0, hi, dim, 0.05)
    tgt = _ragged(st, m, 0, hi, dim, 0.05)
    sf, so = pack_segments(src, dim)
    tf, to = pack_segments(tgt, dim)
    dist = None if case % 2 == 0 else np.abs(st.normal(m)) * 0.01
    e = Engine(metric="refcos", dtype="f64")
    idx, val = e.match(e.dictionary(sf, so, dim), e.queries(tf, to, dim), distance=dist)
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim, dist)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val), (case, dim, hi, n, m)
    e.close()


_N_MEDIUM = int(os.environ.get("SSYM_FUZZ_MEDIUM", "1"))


@pytest.mark.parametrize("seed", range(_N_MEDIUM))
@pytest.mark.parametrize("dim,band,use_dist", [(13, -1, False), (13, -1, True), (40, 32, False), (12, 8, False)])
def test_dtw_medium_ragged(oracle, dim, band, use_dist, seed):
    # enough segments for several workgroups, task ranges and XCD counters; lengths from 1 frame to
    # five row passes, targets beyond 128 frames (certificate column groups), slots reordered by length
    st = synth.Stream(0x5EED2000 + dim + band + 7919 * seed)
    n, m = 300, 96
    src = _ragged(st, n, 1, 260, dim, 1.0)
    tgt = _ragged(st, m, 1, 260, dim, 1.0)
    for t in range(0, m, 3):                                 # near-copies, in scattered positions
        s = (t * 7) % n
        tgt[t] = src[s] + 0.01 * st.normal(src[s].size).reshape(src[s].shape)
    src[211] = src[17].copy()                                # duplicate far apart in slot order
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                     want_matrix=True, nthreads=oracle.max_threads())
    dist = None
    if use_dist:
        dist = np.nan_to_num(np.nanmedian(np.where(np.isfinite(mat), mat, np.nan), axis=0), nan=1.0)
    key = np.abs(mat - (dist[None, :] if dist is not None else 0.0))
    key = np.where(np.isnan(key), np.inf, key)
    want_idx = np.where(np.isfinite(key).any(axis=0), key.argmin(axis=0), 0)
    want_cost = np.where(np.isfinite(key).any(axis=0), mat[want_idx, np.arange(m)], np.inf)
    idx, cost = e.match(d, q, distance=dist)
    assert e.timings()["used_filter"] == 1
    assert np.array_equal(idx, want_idx)
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=1e-12, atol=0) and np.isinf(cost[~fin]).all()
    # the filter matrix, back in the caller's order, stays inside the derived bound of the exact one
    fm = e.pair_matrix(d, q)
    ok = np.isfinite(mat)
    # (frames wider than 13 values are fed in ONE f16 piece: 2^-11 (|a| + |b|) absolute per cell)
    lens = np.diff(so).astype(np.float64)[:, None] + np.diff(to).astype(np.float64)[None, :]
    slack = 2e-3 * (1.0 + mat) + (0.0 if dim <= 13 else 2.0 ** -11 * 2 * 12.0 * lens)
    assert np.all(np.abs(fm[ok] - mat[ok]) <= slack[ok])
    # top-3 through the same path
    ti, tc = e.match_topk(d, q, 3, dist)
    o_idx, _ = oracle.topk(mat, 3, distance=dist, default_distance=0.0, fold_start=float("inf"))
    from soundsym_amd._native import NO_MATCH
    assert np.array_equal(np.where(ti == NO_MATCH, -1, ti.astype(np.int64)), o_idx)
    e.close()


@pytest.mark.parametrize("case", range(max(30, _N_DTW // 4)))
def test_topk_and_chain_random_shapes(oracle, case):
    from soundsym_amd._native import NO_MATCH
    st = synth.Stream(0x5EED3000 + case)
    metric = ["refcos", "dtw"][int(st.integers(1, 2)[0])]
    dim = int([2, 12, 13, 20][st.integers(1, 4)[0]])
    hi = int([4, 18, 40, 70][st.integers(1, 4)[0]])
    n, m = int(2 + st.integers(1, 40)[0]), int(1 + st.integers(1, 12)[0])
    k = int(1 + st.integers(1, 8)[0])
    src = _ragged(st, n, 1, hi, dim, 1.0)
    tgt = _ragged(st, m, 1, hi, dim, 1.0)
    if n > 5:
        src[5] = src[1].copy()
    tgt[0] = src[min(3, n - 1)].copy()
    sf, so = pack_segments(src, dim, np.float64)
    tf, to = pack_segments(tgt, dim, np.float64)
    e = Engine(metric=metric, dtype="f64")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    if metric == "refcos":
        mat = oracle.refcos_matrix(sf, so, tf, to, dim)
        dist = 0.2 + 1.2 * (st.integers(m, 1000) / 1000.0)
        want_idx, want_val = oracle.topk(mat, k, distance=dist)
    else:
        _, _, mat = oracle.dtw_match_all(sf, so, tf, to, dim, want_matrix=True)
        dist = np.nanmedian(mat, axis=0) * (st.integers(m, 1000) / 500.0)
        want_idx, _ = oracle.topk(mat, k, distance=dist, default_distance=0.0, fold_start=float("inf"))
        want_val = np.where(want_idx >= 0, mat[np.maximum(want_idx, 0), np.arange(m)[:, None]], np.nan)
    idx, val = e.match_topk(d, q, k, dist)
    assert np.array_equal(np.where(idx == NO_MATCH, -1, idx.astype(np.int64)), want_idx)
    have = want_idx >= 0
    if metric == "refcos":
        assert np.array_equal(val[have], want_val[have])
    else:
        assert np.allclose(val[have], want_val[have], rtol=1e-12, atol=0)
    # the greedy chain from target 0 with the same distances as step sizes
    steps = dist[: max(1, min(m, 6))]
    ci, cv = e.chain(d, tgt[0].reshape(-1), steps)
    wi, wv = oracle.chain(sf, so, dim, tgt[0].reshape(-1), steps, metric=metric)
    assert np.array_equal(ci, wi)
    assert np.array_equal(cv, wv) if metric == "refcos" else np.allclose(cv, wv, rtol=1e-12, atol=0)
    e.close()


_N_REFCOS_MFMA = max(16, _N_DTW // 6)


@pytest.mark.parametrize("case", range(_N_REFCOS_MFMA))
def test_refcos_matrix_pipe_random_shapes(oracle, case):
    """The refcos search through the matrix pipes (from 65 536 pairs up) -- the integer filter (csrc/refcos_q8.hip) where
    the values allow it, the f64 filter (csrc/refcos_mfma.hip) at amplitudes 1e-150 and 1e150, whose values the fixed-point
    records do not hold: random ragged sets, empty / all-zero segments, duplicates and prefixes, per-target distances,
    f32 inputs -- index and value bit for bit the oracle's."""
    st = synth.Stream(0x5EED3000 + case)
    dim = int([1, 5, 12, 13, 40][st.integers(1, 5)[0]])
    hi = int([3, 20, 60, 130][st.integers(1, 4)[0]])
    if dim == 40:
        hi = min(hi, 60)
    n = int(260 + st.integers(1, 400)[0])
    m = int((65536 + n - 1) // n + st.integers(1, 120)[0])
    scale = float([1e-150, 1e-3, 0.05, 1.0, 300.0, 1e150][st.integers(1, 6)[0]])
    dtype = "f64" if scale in (1e-150, 1e150) else ["f32", "f64"][int(st.integers(1, 2)[0])]
    src = _ragged(st, n, 0, hi, dim, scale)
    tgt = _ragged(st, m, 0, hi, dim, scale)
    for k in range(0, min(m, 40), 3):                    # planted copies, prefixes, scaled copies
        j = int(st.integers(1, n)[0])
        if src[j].shape[0]:
            tgt[k] = [src[j].copy(), src[j][: max(1, src[j].shape[0] // 2)].copy(), src[j] * 0.5][k % 3]
    src[n // 2] = src[1].copy()                          # duplicates: the lower index wins
    src[n // 3] = np.zeros_like(src[n // 3])             # norm 0: never a winner
    tgt[m // 2] = np.zeros_like(tgt[m // 2])
    npdt = np.float32 if dtype == "f32" else np.float64
    sf, so = pack_segments(src, dim, npdt)
    tf, to = pack_segments(tgt, dim, npdt)
    use_dist = bool(st.integers(1, 3)[0] == 0)
    dist = None
    if use_dist:
        dist = np.asarray(st.normal(m)) * 0.7 + 0.5
        dist[0] = 0.0
    e = Engine(metric="refcos", dtype=dtype)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, val = e.match(d, q, distance=dist)
    tm = e.timings()
    want_idx, want_val = oracle.refcos_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, distance=dist)
    info = dict(case=case, dim=dim, hi=hi, n=n, m=m, scale=scale, dtype=dtype, dist=use_dist, mfma=tm["used_filter"],
                filter=tm["refcos_filter"], refined=tm["n_refined"])
    assert tm["refcos_filter"] == (1 if scale in (1e-150, 1e150) else 2), info
    assert np.array_equal(idx, want_idx), info
    assert np.array_equal(val, want_val), info
    e.close()
