"""Early abandoning (SSYM_DTW_PRUNE, soundsym_amd/csrc/prune.hip + the PRUNE variant of the filter kernel):
the same indices and costs as the full search -- and as the oracle -- on planted grids (where it
abandons nearly everything), on data without any close pair (where it abandons little), on ragged
segments with empties, exact duplicates and multi-pass sources; ignored where it does not apply."""
import numpy as np
import pytest

from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu
import os
_N = int(os.environ.get("SSYM_FUZZ_CASES", "40"))      # SSYM_FUZZ_CASES=N widens the seeded sweeps


def _both(e, d, q, **kw):
    i0, c0 = e.match(d, q, **kw)
    t0 = e.timings()
    i1, c1 = e.match(d, q, prune=True, **kw)
    t1 = e.timings()
    assert np.array_equal(i0, i1) and np.array_equal(c0, c1)
    return i1, c1, t0, t1


@pytest.mark.parametrize("n,m,f,d", [(256, 192, 64, 13), (192, 96, 128, 13), (96, 64, 200, 13), (128, 64, 48, 20),
                                      (64, 40, 16, 5)])
def test_prune_planted_grid(oracle, n, m, f, d):
    g = synth.make_grid(n, m, f, d, 0x5EED0400 + f)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    e = Engine(metric="dtw", dtype="f32")
    dd, q = e.dictionary(sf, so, d), e.queries(tf, to, d)
    idx, cost, t0, t1 = _both(e, dd, q)
    assert t0["pruned"] == 0 and t1["pruned"] == 1 and t1["used_filter"] == 1
    assert np.array_equal(idx, g.planted)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, d,
                                               nthreads=oracle.max_threads())
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    # the planted pairs are far below everything else, so every task WITHOUT a planted pair is dropped
    # early (at these sizes a good share of the 64-pair tasks holds one and runs to its end)
    rows = 16 * (4 if f > 48 else (f + 15) // 16) * ((f + 63) // 64 if f > 48 else 1)
    full = ((n + 7) // 8 * 8) * ((m + 31) // 32 * 32) * rows * f
    assert 0 < t1["n_filter_cells"] < 0.9 * full
    e.close()


@pytest.mark.parametrize("case", range(_N))
def test_prune_random_shapes(oracle, case):
    st = synth.Stream(0x5EED4000 + case)
    dim = int([1, 2, 5, 12, 13, 14, 20, 40, 42, 43, 64, 90][st.integers(1, 12)[0]])
    hi = int([3, 17, 33, 50, 66, 130, 150][st.integers(1, 7)[0]])
    lo = int(st.integers(1, 2)[0])
    n, m = int(2 + st.integers(1, 60)[0]), int(1 + st.integers(1, 70)[0])
    squared = bool(st.integers(1, 4)[0] == 0)
    dtype = ["f32", "f64"][int(st.integers(1, 2)[0])]
    scale = float([1e-3, 1.0, 300.0][st.integers(1, 3)[0]])
    lens_s = lo + st.integers(n, hi - lo + 1)
    lens_t = lo + st.integers(m, hi - lo + 1)
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) * scale for f in lens_s]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) * scale for f in lens_t]
    for t in range(0, m, 2):                                   # near copies (tight thresholds) next to unrelated targets
        s = (5 * t + 1) % n
        if src[s].shape[0] > 0:
            tgt[t] = src[s] + 0.02 * scale * st.normal(src[s].size).reshape(src[s].shape)
    if n > 6:
        src[6] = src[2].copy()                                 # duplicates: the first index wins
    if m > 3 and n > 2:
        tgt[3] = src[2].copy()
    npdt = np.float32 if dtype == "f32" else np.float64
    sf, so = pack_segments(src, dim, npdt)
    tf, to = pack_segments(tgt, dim, npdt)
    e = Engine(metric="dtw", dtype=dtype, squared=squared)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost, _, t1 = _both(e, d, q)
    assert t1["pruned"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim,
                                               squared=squared, nthreads=4)
    info = dict(case=case, dim=dim, hi=hi, n=n, m=m, squared=squared, dtype=dtype, scale=scale)
    assert np.array_equal(idx, want_idx), info
    assert np.array_equal(np.isinf(cost), np.isinf(want_cost)), info
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=1e-12, atol=0), info
    e.close()


@pytest.mark.parametrize("n,m,f,d,band", [(192, 256, 96, 13, 8), (128, 256, 128, 40, 32), (96, 256, 64, 13, 20),
                                           (64, 256, 150, 20, 47)])
def test_prune_banded_planted_grid(oracle, n, m, f, d, band):
    g = synth.make_grid(n, m, f, d, 0x5EED0500 + band)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    e = Engine(metric="dtw", dtype="f32", band=band)
    dd, q = e.dictionary(sf, so, d), e.queries(tf, to, d)
    idx, cost, t0, t1 = _both(e, dd, q)
    assert t0["pruned"] == 0 and t1["pruned"] == 1 and t1["used_filter"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, d, band=band,
                                               nthreads=oracle.max_threads())
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    assert t1["n_filter_cells"] > 0
    e.close()


@pytest.mark.parametrize("case", range(max(24, _N // 2)))
def test_prune_banded_random_shapes(oracle, case):
    st = synth.Stream(0x5EED4200 + case)
    dim = int([2, 13, 14, 40][st.integers(1, 4)[0]])
    hi = int([17, 50, 66, 130][st.integers(1, 4)[0]])
    lo = int(st.integers(1, 2)[0])
    band = int([0, 1, 5, 8, 20, 32, 47][st.integers(1, 7)[0]])
    n, m = int(2 + st.integers(1, 40)[0]), int(1 + st.integers(1, 70)[0])
    squared = bool(st.integers(1, 4)[0] == 0)
    lens_s = lo + st.integers(n, hi - lo + 1)
    lens_t = lo + st.integers(m, hi - lo + 1)
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in lens_s]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in lens_t]
    for t in range(0, m, 2):
        s = (5 * t + 1) % n
        if src[s].shape[0] > 0:
            tgt[t] = src[s] + 0.02 * st.normal(src[s].size).reshape(src[s].shape)
    if n > 6:
        src[6] = src[2].copy()
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band, squared=squared)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost, _, t1 = _both(e, d, q)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                               squared=squared, nthreads=4)
    info = dict(case=case, dim=dim, hi=hi, n=n, m=m, band=band, squared=squared, filter=t1["used_filter"])
    assert t1["pruned"] == t1["used_filter"], info
    assert np.array_equal(idx, want_idx), info
    assert np.array_equal(np.isinf(cost), np.isinf(want_cost)), info
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=1e-12, atol=0), info
    e.close()


@pytest.mark.parametrize("seed", range(max(2, _N // 40)))
@pytest.mark.parametrize("dim,band,nt", [(13, -1, 0), (13, -1, 2), (20, -1, 4), (13, 24, 0)])
def test_prune_medium_ragged(oracle, dim, band, nt, seed, monkeypatch):
    # several workgroups and task ranges, lengths from 1 frame to five 64-row (ten 32-row) passes, sources
    # end-aligned at every offset inside a pass, near copies of every closeness next to unrelated targets:
    # passes stop at different columns per wave, tasks are dropped after different passes
    st = synth.Stream(0x5EED4300 + 31 * dim + band + 7919 * seed + nt)
    n, m = 300, 96
    hi = 260 if band < 0 else 120
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in 1 + st.integers(n, hi)]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in 1 + st.integers(m, hi)]
    for t in range(0, m, 3):
        s = (t * 7) % n
        eps = [0.0, 0.01, 0.3, 1.0][(t // 3) % 4]
        tgt[t] = src[s] + eps * st.normal(src[s].size).reshape(src[s].shape) * synth.sigma(dim)
    src[211] = src[17].copy()
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    if nt:
        monkeypatch.setenv("SSYM_PRUNE_NT", str(nt))       # pins the pass height (read at every launch)
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost, _, t1 = _both(e, d, q)
    idx2, cost2 = e.match(d, q, prune=True)                 # second pruned call: the pass height may have changed
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)
    assert t1["pruned"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                               nthreads=oracle.max_threads())
    assert np.array_equal(idx, want_idx)
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=1e-12, atol=0) and np.isinf(cost[~fin]).all()
    e.close()


@pytest.mark.parametrize("dim,f,band", [(50, 64, -1), (64, 100, -1), (90, 40, -1), (64, 96, 16)])
def test_prune_wide_frames(oracle, dim, f, band):
    # frames wider than the filter takes in: its cost is a lower bound, which is all abandoning needs
    g = synth.make_grid(160, 256 if band >= 0 else 96, f, dim, 0x5EED0600 + dim)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    e = Engine(metric="dtw", dtype="f32", band=band)
    dd, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost, t0, t1 = _both(e, dd, q)
    assert t1["pruned"] == t1["used_filter"] == 1
    assert np.array_equal(idx, g.planted)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                               nthreads=oracle.max_threads())
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    e.close()


def test_prune_without_close_pairs_and_index_base(oracle):
    st = synth.Stream(0x5EED4100)
    dim, n, m = 13, 200, 70
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in 30 + st.integers(n, 100)]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in 30 + st.integers(m, 100)]
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost, _, t1 = _both(e, d, q, index_base=1000)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, nthreads=4)
    assert np.array_equal(idx, want_idx + 1000) and np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    assert t1["pruned"] == 1
    e.close()


def test_prune_is_ignored_where_it_does_not_apply(oracle):
    g = synth.make_grid(64, 32, 40, 13, 0x5EED0444)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    # per-target distances: the search is not a plain minimum
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    dist = np.full(32, 50.0)
    i0, c0 = e.match(d, q, distance=dist)
    i1, c1 = e.match(d, q, distance=dist, prune=True)
    assert e.timings()["pruned"] == 0 and np.array_equal(i0, i1) and np.array_equal(c0, c1)
    e.close()
    # top-k
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    e.match_topk(d, q, 3)
    assert e.timings()["pruned"] == 0
    e.close()
    # refcos: the flag means nothing
    r = Engine(metric="refcos", dtype="f64")
    sf64, tf64 = sf.astype(np.float64) * 0.02, tf.astype(np.float64) * 0.02
    d, q = r.dictionary(sf64, so, 13), r.queries(tf64, to, 13)
    i0, v0 = r.match(d, q)
    i1, v1 = r.match(d, q, prune=True)
    assert np.array_equal(i0, i1) and np.array_equal(v0, v1)
    r.close()


def test_prune_after_append_uses_the_new_sources(oracle):
    g = synth.make_grid(48, 24, 32, 13, 0x5EED0455)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    e = Engine(metric="dtw", dtype="f32")
    d = e.dictionary(sf[: 24 * 32 * 13], so[:25], 13)
    q = e.queries(tf, to, 13)
    e.match(d, q, prune=True)                                   # centroids of the first 24 sources are cached now
    e.dictionary_append(d, sf[24 * 32 * 13:], so[24:] - so[24])
    idx, cost, _, _ = _both(e, d, q)
    assert np.array_equal(idx, g.planted)
    e.close()


def test_prune_as_the_context_default_reaches_the_entry_points_without_flags(oracle):
    g = synth.make_grid(128, 96, 48, 13, 0x5EED0466)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    plain = Engine(metric="dtw", dtype="f32")
    want_idx, want_cost = plain.match_batch(plain.dictionary(sf, so, 13), tf, to)
    assert plain.timings()["pruned"] == 0
    plain.close()
    e = Engine(metric="dtw", dtype="f32", prune=True)             # ssym_config.dtw_prune
    d = e.dictionary(sf, so, 13)
    idx, cost = e.match_batch(d, tf, to)                          # ssym_match_batch has no flags argument
    assert e.timings()["pruned"] == 1
    assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost)
    few_idx, few_cost = e.match_batch(d, tf[: 5 * 48 * 13], to[:6])   # a handful of targets: not worth the extra launches
    assert e.timings()["pruned"] == 0
    assert np.array_equal(few_idx, want_idx[:5]) and np.array_equal(few_cost, want_cost[:5])
    e.close()
