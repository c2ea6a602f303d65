"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars: refcos -- similarities, winning values and indices BIT-EXACT (f64, same operation order);
dtw -- indices identical, costs within 1e-5 relative (north-star tolerance; the exact kernel is
expected to be bit-exact and is checked at 1e-12), the f32 MFMA filter within its derived bound.
"""
import json
import math
import os
from fractions import Fraction

import numpy as np
import pytest

from soundsym_amd import Engine, EmptyDictionaryError, SsymError, synth
from soundsym_amd.engine import pack_segments
from bounds import input_rounding, pair_bound_matrix

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DTW_RTOL = 1e-5      # BASELINE.json north_star: "DTW costs within 1e-5 relative f32"
EXACT_RTOL = 1e-12   # exact f64 kernel vs f64 oracle (same operation order)


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


# ---------------------------------------------------------------------------------------------
# refcos
# ---------------------------------------------------------------------------------------------
def test_refcos_kats(refcos):
    kat = json.load(open(os.path.join(GOLD, "refcos_kat.json")))
    for case in kat["cosine_sim"]:
        d = refcos.dictionary(np.array(case["me"], dtype=np.float64), [0, len(case["me"])], 1)
        q = refcos.queries(np.array(case["you"], dtype=np.float64), [0, len(case["you"])], 1)
        sim = refcos.pair_matrix(d, q)[0, 0]
        assert sim == float(Fraction(case["num"], case["den"])), case
    for case in kat["at_distance"]:
        dim = 1 if len(case["dict"][0]) == 1 else 2
        flat, off = pack_segments([np.array(s, dtype=np.float64) for s in case["dict"]], dim)
        d = refcos.dictionary(flat, off, dim)
        you = np.array(case["you"], dtype=np.float64)
        idx, val = refcos.match_one(d, you, case["distance"])
        assert idx == case["idx"], case
        if "val" in case:
            assert val == case["val"], case


def test_refcos_golden_ragged_bit_exact(refcos, oracle):
    g = np.load(os.path.join(GOLD, "refcos_ragged.npz"))
    d = refcos.dictionary(g["src"], g["src_off"], 12)
    q = refcos.queries(g["tgt"], g["tgt_off"], 12)
    sims = refcos.pair_matrix(d, q)
    assert np.array_equal(sims, g["sims"])                 # bit for bit, NaN-free fixture
    idx, val = refcos.match(d, q)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(val, g["val"])
    idx_d, val_d = refcos.match(d, q, distance=g["dist"])  # morph_to: per-target distance
    assert np.array_equal(idx_d, g["idx_d"]) and np.array_equal(val_d, g["val_d"])
    # ssym_match_batch and ssym_match_one are the same path
    idx_b, val_b = refcos.match_batch(d, g["tgt"], g["tgt_off"])
    assert np.array_equal(idx_b, g["idx"]) and np.array_equal(val_b, g["val"])
    t0 = g["tgt"][int(g["tgt_off"][5]) * 12:int(g["tgt_off"][6]) * 12]
    assert refcos.match_one(d, t0, 1.0) == (int(g["idx"][5]), float(g["val"][5]))


@pytest.mark.parametrize("n,m,fmin,fmax,dim", [(70, 45, 1, 30, 12), (33, 65, 5, 9, 13), (8, 8, 40, 41, 40)])
def test_refcos_random_ragged_vs_oracle(refcos, oracle, n, m, fmin, fmax, dim):
    src, tgt = synth.make_ragged(n, m, fmin, fmax, dim, 0x5EED0200 + n)
    sf, so = pack_segments([s.astype(np.float64) * 0.03 for s in src], dim)
    tf, to = pack_segments([t.astype(np.float64) * 0.03 for t in tgt], dim)
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, dim)
    d, q = refcos.dictionary(sf, so, dim), refcos.queries(tf, to, dim)
    idx, val = refcos.match(d, q)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    sims = refcos.pair_matrix(d, q)
    for s in (0, n // 2, n - 1):
        for t in (0, m - 1):
            assert sims[s, t] == oracle.cosine_sim(src[s].astype(np.float64) * 0.03,
                                                   tgt[t].astype(np.float64) * 0.03)


def test_refcos_f32_inputs_are_widened_exactly(oracle):
    e = Engine(metric="refcos", dtype="f32")
    src, tgt = synth.make_ragged(20, 10, 2, 12, 12, 0x5EED0210)
    sf, so = pack_segments(src, 12, np.float32)
    tf, to = pack_segments(tgt, 12, np.float32)
    idx, val = e.match(e.dictionary(sf, so, 12), e.queries(tf, to, 12))
    want_idx, want_val = oracle.refcos_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 12)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    e.close()


def test_refcos_edge_cases(refcos, oracle):
    # empty dictionary: the reference panics (src/sound.rs:369), the ABI reports it
    d0 = refcos.dictionary(np.zeros(0), [0], 12)
    q = refcos.queries(np.ones(12), [0, 1], 12)
    with pytest.raises(EmptyDictionaryError):
        refcos.match(d0, q)
    # zero-length target and zero-length dictionary entries: NaN never wins -> (0, 2.0)
    flat, off = pack_segments([np.ones(12), np.zeros(0), np.full(24, 2.0)], 12)
    d = refcos.dictionary(flat, off, 12)
    tflat, toff = pack_segments([np.zeros(0), np.ones(12)], 12)
    idx, val = refcos.match(d, refcos.queries(tflat, toff, 12))
    want_idx, want_val = oracle.refcos_match_all(flat, off, tflat, toff, 12)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    assert idx[0] == 0 and val[0] == 2.0
    # no targets: nothing to do
    idx, val = refcos.match(d, refcos.queries(np.zeros(0), [0], 12))
    assert idx.size == 0
    # append == add_segments: indices continue
    refcos.dictionary_append(d, np.full(12, 0.5), [0, 1])
    assert d.n == 4
    idx, _ = refcos.match(d, refcos.queries(np.full(12, 0.5), [0, 1], 12))
    f2, o2 = pack_segments([np.ones(12), np.zeros(0), np.full(24, 2.0), np.full(12, 0.5)], 12)
    assert idx[0] == oracle.refcos_match_all(f2, o2, np.full(12, 0.5), np.array([0, 1], dtype=np.uint64), 12)[0][0]
    with pytest.raises(SsymError):
        refcos.match(d, refcos.queries(np.ones(13), [0, 1], 13))   # dim mismatch


# ---------------------------------------------------------------------------------------------
# dtw
# ---------------------------------------------------------------------------------------------
def _filter_bound(src, tgt, fa, fb, dim=13):
    """|C~ - C| bound of the f16-split MFMA filter as derived in soundsym_amd/csrc/select.hip
    (worst case over pairs: no use of the per-pair smallest-cell certificate)."""
    u = 2.0 ** -24
    in_a, in_b = input_rounding(min(dim, 42))
    na = max(float((s.astype(np.float64) ** 2).sum(-1).max()) for s in src if s.size)
    nb = max(float((t.astype(np.float64) ** 2).sum(-1).max()) for t in tgt if t.size)
    vmax = max(max(float(np.abs(s).max()) for s in src if s.size), max(float(np.abs(t).max()) for t in tgt if t.size))
    scale = 2.0 ** (6 - math.frexp(vmax)[1]) if vmax > 0 else 1.0
    E = 256 * u * (na + nb) + 2.0 ** -12 / scale ** 2
    cell = math.sqrt(E) + 1.001 * (in_a * math.sqrt(na) + in_b * math.sqrt(nb))
    return (fa + fb - 1) * cell


def test_dtw_kats(dtw):
    kat = json.load(open(os.path.join(GOLD, "refcos_kat.json")))
    for case in kat["dtw"]:
        a, b = np.array(case["a"], dtype=np.float32), np.array(case["b"], dtype=np.float32)
        e = Engine(metric="dtw", dtype="f32", band=case["band"], squared=bool(case.get("squared", False)))
        d = e.dictionary(a.reshape(-1), [0, a.shape[0]], a.shape[1])
        q = e.queries(b.reshape(-1), [0, b.shape[0]], a.shape[1])
        idx, cost = e.match(d, q)
        want = float("inf") if case["cost"] == "inf" else case["cost"]
        assert idx[0] == 0 and cost[0] == want, case
        e.close()


def test_dtw_golden_grid(dtw, oracle):
    g = np.load(os.path.join(GOLD, "dtw_grid_32x32x16x13.npz"))
    n, f, dim = g["sources"].shape
    so = np.arange(n + 1, dtype=np.uint64) * f
    d = dtw.dictionary(g["sources"].reshape(-1), so, dim)
    q = dtw.queries(g["targets"].reshape(-1), so, dim)
    idx, cost = dtw.match(d, q)
    assert dtw.timings()["used_filter"] == 1
    assert np.array_equal(idx, g["idx"]) and np.array_equal(idx, g["planted"])
    assert np.allclose(cost, g["cost"], rtol=DTW_RTOL, atol=0)
    assert np.allclose(cost, g["cost"], rtol=EXACT_RTOL, atol=0)     # refined in f64
    # the f16-split MFMA filter's whole cost matrix against the oracle's, within the derived bound
    filt = dtw.pair_matrix(d, q, exact=False)
    bound = _filter_bound(list(g["sources"]), list(g["targets"]), f, f)
    err = np.abs(filt - g["matrix"])
    assert (err <= bound + 1e-5 * g["matrix"]).all(), (err.max(), bound)
    # ... and within the bound the selection's first stage now gives every PAIR: its two segments' norms and the
    # measured residual of their records (what the target's one f16 piece really rounds away, not 2^-11 |b|)
    pb, ra, rb, _, nb = pair_bound_matrix(list(g["sources"]), list(g["targets"]), dim)
    assert (err <= pb + 1e-5 * g["matrix"]).all(), float((err / (pb + 1e-5 * g["matrix"])).max())
    assert (pb <= 1.03 * bound + 1e-12).all()                       # never looser than the set-wide worst case
    if not os.environ.get("SSYM_FILTER_K48"):
        assert np.median(rb / (2.0 ** -11 * np.sqrt(nb))) < 0.8     # and it does tighten: cepstral decay, two values kept whole
    # non-planted pairs are far from cancellation: there the filter is f32-accurate already, up to what the record
    # layout rounds away of the TARGET frames (one f16 piece: 2^-11 per value, far from adding up along a path)
    off_diag = g["matrix"] > 4 * g["cost"].max()
    assert (err[off_diag] <= (2e-5 + 0.25 * input_rounding(dim)[1]) * g["matrix"][off_diag]).all()
    # exact kernel on every pair == oracle matrix
    exact = dtw.pair_matrix(d, q, exact=True)
    assert np.allclose(exact, g["matrix"], rtol=EXACT_RTOL, atol=0)
    # filter bypassed: same answers
    idx2, cost2 = dtw.match(d, q, force_exact=True)
    assert np.array_equal(idx2, idx) and np.array_equal(cost2, cost)


def test_dtw_golden_band_and_squared(oracle):
    g = np.load(os.path.join(GOLD, "dtw_grid_32x32x16x13.npz"))
    n, f, dim = g["sources"].shape
    so = np.arange(n + 1, dtype=np.uint64) * f
    eb = Engine(metric="dtw", dtype="f32", band=3)
    idx, cost = eb.match(eb.dictionary(g["sources"].reshape(-1), so, dim),
                         eb.queries(g["targets"].reshape(-1), so, dim))
    assert np.array_equal(idx, g["idx_band3"]) and np.allclose(cost, g["cost_band3"], rtol=EXACT_RTOL)
    eb.close()
    es = Engine(metric="dtw", dtype="f32", squared=True)
    idx, cost = es.match(es.dictionary(g["sources"].reshape(-1), so, dim),
                         es.queries(g["targets"].reshape(-1), so, dim))
    assert es.timings()["used_filter"] == 1
    assert np.array_equal(idx, g["idx_sq"]) and np.allclose(cost, g["cost_sq"], rtol=EXACT_RTOL)
    es.close()


def test_dtw_golden_ragged(dtw):
    g = np.load(os.path.join(GOLD, "dtw_ragged.npz"))
    d = dtw.dictionary(g["src"], g["src_off"], 13)
    q = dtw.queries(g["tgt"], g["tgt_off"], 13)
    idx, cost = dtw.match(d, q)
    assert dtw.timings()["used_filter"] == 1
    assert np.array_equal(idx, g["idx"]) and np.allclose(cost, g["cost"], rtol=EXACT_RTOL, atol=0)
    filt = dtw.pair_matrix(d, q, exact=False)
    fa = int(np.diff(g["src_off"]).max())
    fb = int(np.diff(g["tgt_off"]).max())
    src = [g["src"][int(a) * 13:int(b) * 13].reshape(-1, 13) for a, b in zip(g["src_off"][:-1], g["src_off"][1:])]
    tgt = [g["tgt"][int(a) * 13:int(b) * 13].reshape(-1, 13) for a, b in zip(g["tgt_off"][:-1], g["tgt_off"][1:])]
    bound = _filter_bound(src, tgt, fa, fb)
    assert (np.abs(filt - g["matrix"]) <= bound + 1e-5 * g["matrix"]).all()
    pb = pair_bound_matrix(src, tgt, 13)[0]                          # ragged lengths: every pair under ITS bound
    fin = np.isfinite(g["matrix"])
    assert (np.abs(filt - g["matrix"])[fin] <= (pb + 1e-5 * g["matrix"])[fin]).all()


@pytest.mark.parametrize("n,m,f,dim", [(40, 70, 64, 13), (17, 33, 100, 12), (9, 5, 128, 13), (64, 32, 7, 5)])
def test_dtw_grid_vs_oracle(dtw, oracle, n, m, f, dim):
    g = synth.make_grid(n, m, f, dim, 0x5EED0300 + f)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, nthreads=8)
    idx, cost = dtw.match(dtw.dictionary(sf, so, dim), dtw.queries(tf, to, dim))
    assert dtw.timings()["used_filter"] == 1
    assert np.array_equal(idx, want_idx)
    assert np.allclose(cost, want_cost, rtol=DTW_RTOL, atol=0)
    assert np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    if m <= n:
        assert np.array_equal(idx, g.planted)


def test_dtw_unplanted_targets_keep_the_candidate_set_small(dtw, oracle):
    # targets that have NO near-identical source (what a rank sees when a target's neighbour lives in
    # another shard): all costs are of similar size, and only the per-pair error bound -- built on
    # the smallest cell the filter saw in each pair -- keeps the exact re-scoring to a few pairs
    src = synth.make_grid(192, 1, 64, 13, 0x5EED0350).sources
    tgt = synth.make_grid(96, 1, 64, 13, 0x5EED0351).sources
    so = np.arange(src.shape[0] + 1, dtype=np.uint64) * 64
    to = np.arange(tgt.shape[0] + 1, dtype=np.uint64) * 64
    sf, tf = src.reshape(-1), tgt.reshape(-1)
    idx, cost = dtw.match(dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13))
    tm = dtw.timings()
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13, nthreads=8)
    assert np.array_equal(idx, want_idx)
    assert np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    assert tm["used_filter"] == 1 and tm["n_refined"] <= 3 * tgt.shape[0], tm


def test_dtw_ties_take_the_lowest_index(dtw, oracle):
    g = synth.make_grid(24, 8, 20, 13, 0x5EED0310)
    g.sources[19] = g.sources[3]                   # exact duplicates among the sources
    g.sources[11] = g.sources[3]
    g.targets[0] = g.sources[3]
    g.planted[0] = 3
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    idx, cost = dtw.match(dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13))
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and idx[0] == 3 and cost[0] == 0.0
    assert np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)


def test_dtw_long_segments_take_several_row_block_passes(oracle):
    # 150 and 300 source frames: 3 and 5 passes of 64 rows with the boundary row handed over
    e = Engine(metric="dtw", dtype="f32")
    for n, m, f in ((6, 5, 150), (4, 3, 300)):
        g = synth.make_grid(n, m, f, 13, 0x5EED0320 + f)
        sf, so = g.flat("sources")
        tf, to = g.flat("targets")
        d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
        idx, cost = e.match(d, q)
        assert e.timings()["used_filter"] == 1
        want_idx, want_cost, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13,
                                                        want_matrix=True)
        assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
        filt = e.pair_matrix(d, q, exact=False)
        bound = _filter_bound(list(g.sources), list(g.targets), f, f)
        assert (np.abs(filt - mat) <= bound + 1e-5 * mat).all()
    e.close()


@pytest.mark.parametrize("dim,f", [(40, 40), (20, 70), (42, 17)])
def test_dtw_wide_frames_use_single_piece_records(oracle, dim, f):
    # 14..42 values per frame: one f16 piece per value in the MFMA records, exact refine as always
    e = Engine(metric="dtw", dtype="f32")
    g = synth.make_grid(24, 16, f, dim, 0x5EED0370 + dim)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost = e.match(d, q)
    tm = e.timings()
    assert tm["used_filter"] == 1
    want_idx, want_cost, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim,
                                                    want_matrix=True)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    assert np.array_equal(idx, g.planted)
    filt = e.pair_matrix(d, q, exact=False)
    bound = _filter_bound(list(g.sources), list(g.targets), f, f, dim)
    assert (np.abs(filt - mat) <= bound + 1e-5 * mat).all(), (np.abs(filt - mat).max(), bound)
    assert tm["n_refined"] <= 4 * 16, tm
    e.close()


@pytest.mark.parametrize("n,m,f,dim,band", [(24, 40, 30, 13, 3), (16, 33, 64, 13, 8), (10, 12, 100, 40, 32),
                                              (12, 9, 48, 12, 0), (8, 8, 70, 20, 40), (9, 11, 90, 13, 28),
                                              (12, 10, 256, 40, 32)])      # the last: configs[4]'s segment shape
def test_dtw_banded_filter_vs_oracle(oracle, n, m, f, dim, band):
    # Sakoe-Chiba band on the MFMA path: diagonal-coordinate kernel, source pair in LDS
    e = Engine(metric="dtw", dtype="f32", band=band)
    g = synth.make_grid(n, m, f, dim, 0x5EED0380 + band)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost = e.match(d, q)
    assert e.timings()["used_filter"] == 1
    want_idx, want_cost, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim,
                                                    band=band, want_matrix=True, nthreads=8)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    filt = e.pair_matrix(d, q, exact=False)
    bound = _filter_bound(list(g.sources), list(g.targets), f, f, dim)
    finite = np.isfinite(mat)
    assert np.array_equal(np.isfinite(filt), finite)
    assert (np.abs(filt[finite] - mat[finite]) <= bound + 1e-5 * mat[finite]).all()
    e.close()


def test_dtw_banded_ragged_lengths(oracle):
    # unequal lengths: the end cell (fa-1, fb-1) is outside the band when |fa - fb| > r -> +inf
    src, tgt = synth.make_ragged(40, 70, 1, 45, 13, 0x5EED0390)
    tgt[5] = src[11].copy()
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    for band in (2, 10):
        e = Engine(metric="dtw", dtype="f32", band=band)
        idx, cost = e.match(e.dictionary(sf, so, 13), e.queries(tf, to, 13))
        assert e.timings()["used_filter"] == 1
        want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13,
                                                   band=band, nthreads=8)
        assert np.array_equal(idx, want_idx)
        assert np.array_equal(np.isinf(cost), np.isinf(want_cost))
        fin = np.isfinite(want_cost)
        assert np.allclose(cost[fin], want_cost[fin], rtol=EXACT_RTOL, atol=0)
        e.close()


def test_dtw_very_wide_bands_use_the_exact_kernel_and_wide_frames_the_bound(oracle):
    # more than 42 values per frame: the filter bounds the cost from below (its first 42 values); a band
    # wider than 6 tiles of diagonals is outside the BANDED kernel: the unbanded filter bounds the banded cost from
    # below, the banded exact kernel scores the survivors; forced exact: every pair, including the exact kernel's
    # own 64-row chunking (150 frames = 3 chunks)
    g = synth.make_grid(5, 4, 20, 50, 0x5EED0323)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    ew = Engine(metric="dtw", dtype="f32")
    idx, cost = ew.match(ew.dictionary(sf, so, 50), ew.queries(tf, to, 50))
    assert ew.timings()["used_filter"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 50)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    ew.close()
    g = synth.make_grid(6, 5, 120, 13, 0x5EED0321)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    eb = Engine(metric="dtw", dtype="f32", band=60)
    idx, cost = eb.match(eb.dictionary(sf, so, 13), eb.queries(tf, to, 13))
    assert eb.timings()["used_filter"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13, band=60)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    eb.close()
    e = Engine(metric="dtw", dtype="f32")
    g = synth.make_grid(5, 4, 150, 13, 0x5EED0322)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    idx, cost = e.match(e.dictionary(sf, so, 13), e.queries(tf, to, 13), force_exact=True)
    assert e.timings()["used_filter"] == 0
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    e.close()


def test_dtw_f64_inputs_and_distance(oracle):
    e = Engine(metric="dtw", dtype="f64")
    src, tgt = synth.make_ragged(30, 14, 4, 33, 13, 0x5EED0330)
    rng = np.random.default_rng(3)
    src = [s.astype(np.float64) + 1e-9 * rng.normal(size=s.shape) for s in src]   # not f32-representable
    tgt = [t.astype(np.float64) for t in tgt]
    sf, so = pack_segments(src, 13)
    tf, to = pack_segments(tgt, 13)
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    idx, cost = e.match(d, q)
    assert e.timings()["used_filter"] == 1
    want_idx, want_cost, mat = oracle.dtw_match_all(sf, so, tf, to, 13, want_matrix=True)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    # per-target distance (morph_to semantics on the dtw metric): argmin |cost - distance|
    dist = np.median(mat, axis=0)
    idx_d, cost_d = e.match(d, q, distance=dist)
    want = np.abs(mat - dist[None, :]).argmin(axis=0)
    assert np.array_equal(idx_d, want)
    assert np.allclose(cost_d, mat[want, np.arange(mat.shape[1])], rtol=EXACT_RTOL, atol=0)
    e.close()


def test_dtw_edge_cases(dtw, oracle):
    d0 = dtw.dictionary(np.zeros(0, dtype=np.float32), [0], 13)
    q = dtw.queries(np.ones(13, dtype=np.float32), [0, 1], 13)
    with pytest.raises(EmptyDictionaryError):
        dtw.match(d0, q)
    # empty segments on either side cost +inf; all-inf keeps the fold start (0, +inf)
    flat, off = pack_segments([np.zeros(0), np.ones(26)], 13, np.float32)
    d = dtw.dictionary(flat, off, 13)
    tflat, toff = pack_segments([np.zeros(0), np.ones(13), np.zeros(39)], 13, np.float32)
    idx, cost = dtw.match(d, dtw.queries(tflat, toff, 13))
    want_idx, want_cost = oracle.dtw_match_all(flat.astype(np.float64), off, tflat.astype(np.float64), toff, 13)
    assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost)
    assert idx[0] == 0 and cost[0] == float("inf") and idx[1] == 1
    # single frames, single pair
    idx, cost = dtw.match_batch(d, np.full(13, 3.0, dtype=np.float32), [0, 1])
    assert idx[0] == 1 and abs(cost[0] - 2 * math.sqrt(13 * 4.0)) < 1e-12
    # index_base shifts the returned indices (source shard of a multi-GPU run)
    idx_b, _ = dtw.match(d, dtw.queries(tflat, toff, 13), index_base=1000)
    assert np.array_equal(idx_b, want_idx + 1000)


def test_dtw_degenerate_values(oracle):
    e = Engine(metric="dtw", dtype="f32")
    # all zeros: every cost is 0, the first minimum is index 0
    z = np.zeros((5, 9, 13), dtype=np.float32)
    so = np.arange(6, dtype=np.uint64) * 9
    idx, cost = e.match(e.dictionary(z.reshape(-1), so, 13), e.queries(z[:3].reshape(-1), so[:4], 13))
    assert idx.tolist() == [0, 0, 0] and cost.tolist() == [0.0, 0.0, 0.0]
    # wide dynamic range: large values next to tiny ones (common scale, f16 pieces, subnormal tails)
    g = synth.make_grid(40, 24, 30, 13, 0x5EED0360)
    src = g.sources.copy()
    tgt = g.targets.copy()
    src[::3] *= 4096.0
    tgt[::2] *= 1e-3
    sf, tf = src.reshape(-1), tgt.reshape(-1)
    so = np.arange(41, dtype=np.uint64) * 30
    to = np.arange(25, dtype=np.uint64) * 30
    idx, cost = e.match(e.dictionary(sf, so, 13), e.queries(tf, to, 13))
    assert e.timings()["used_filter"] == 1
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    # non-finite features: the filter steps aside, the exact kernel keeps IEEE semantics
    bad = g.sources.copy()
    bad[3, 5, 2] = np.nan
    bad[7, 0, 0] = np.inf
    idx, cost = e.match(e.dictionary(bad.reshape(-1), so, 13), e.queries(g.targets.reshape(-1), to, 13))
    assert e.timings()["used_filter"] == 0
    want_idx, want_cost = oracle.dtw_match_all(bad.reshape(-1).astype(np.float64), so,
                                               g.targets.reshape(-1).astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost)
    e.close()


def test_dtw_dictionary_append_and_device_create(oracle):
    import torch
    e = Engine(metric="dtw", dtype="f32")
    g = synth.make_grid(30, 10, 20, 13, 0x5EED0361)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d = e.dictionary(sf[:10 * 20 * 13], so[:11], 13)
    e.dictionary_append(d, sf[10 * 20 * 13:], so[10:] - so[10])       # add_segments: indices continue
    assert d.n == 30
    q = e.queries(torch.from_numpy(tf).cuda(), to, 13)                  # device-resident targets
    idx, cost = e.match(d, q)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    e.close()


def test_dtw_candidate_overflow_is_redone_with_the_reported_size(dtw, oracle):
    # every source identical: every pair is an exact tie, so list 1 wants all 600 x 200 pairs, more
    # than its first capacity (max(256 per target, 65536)); stage 1 reports the size, the later stages
    # skip, and the selection is redone with that room; the first index wins every tie
    n, m, f = 600, 200, 8
    one = synth.make_grid(1, 1, f, 13, 0x5EED0395).sources[0]
    src = np.repeat(one[None], n, axis=0)
    tgt = synth.make_grid(m, 1, f, 13, 0x5EED0396).sources
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    idx, cost = dtw.match(dtw.dictionary(src.reshape(-1), so, 13), dtw.queries(tgt.reshape(-1), to, 13))
    tm = dtw.timings()
    assert tm["n_refined"] == n * m and (idx == 0).all()
    want_idx, want_cost = oracle.dtw_match_all(src.reshape(-1).astype(np.float64), so,
                                               tgt.reshape(-1).astype(np.float64), to, 13)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    # the same through top-k: every source ties, so each row is the first k indices in order
    d, q = dtw.dictionary(src.reshape(-1), so, 13), dtw.queries(tgt.reshape(-1), to, 13)
    ti, tc = dtw.match_topk(d, q, 5)
    assert dtw.timings()["n_refined"] == n * m
    assert np.array_equal(ti, np.tile(np.arange(5, dtype=np.uint32), (m, 1)))
    assert np.allclose(tc, want_cost[:, None], rtol=EXACT_RTOL, atol=0)


def test_merge_shards_kernel(dtw):
    import torch
    from soundsym_amd import sharding
    rng = np.random.default_rng(11)
    g, m = 8, 1000
    costs = rng.integers(0, 5, size=(g, m)).astype(np.float64)       # many exact ties
    costs[rng.random((g, m)) < 0.05] = np.inf
    idx = (np.arange(g)[:, None] * 100000 + rng.integers(0, 100000, size=(g, m))).astype(np.int32)
    oi, oc = sharding.merge_shards(dtw, torch.from_numpy(costs).cuda(), torch.from_numpy(idx).cuda())
    order = np.lexsort((idx, costs), axis=0)[0]
    assert np.array_equal(oi.cpu().numpy(), idx[order, np.arange(m)])
    assert np.array_equal(oc.cpu().numpy(), costs[order, np.arange(m)])


def test_device_resident_inputs_and_outputs(dtw, oracle):
    import torch
    g = synth.make_grid(48, 40, 32, 13, 0x5EED0340)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d = dtw.dictionary(torch.from_numpy(sf).cuda(), so, 13)
    q = dtw.queries(torch.from_numpy(tf).cuda(), to, 13)
    out_idx = torch.empty(40, dtype=torch.int32, device="cuda")
    out_cost = torch.empty(40, dtype=torch.float64, device="cuda")
    dtw.match(d, q, out_idx=out_idx, out_cost=out_cost)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13)
    assert np.array_equal(out_idx.cpu().numpy(), want_idx)
    assert np.allclose(out_cost.cpu().numpy(), want_cost, rtol=EXACT_RTOL, atol=0)


def test_rows_f_golden_fixture(refcos):
    # committed vectors for top-k, the MFCC front-end and the greedy chain (tests/golden/rows_f.npz)
    from soundsym_amd._native import NO_MATCH
    g = np.load(os.path.join(GOLD, "refcos_ragged.npz"))
    f = np.load(os.path.join(GOLD, "rows_f.npz"))
    d = refcos.dictionary(g["src"], g["src_off"], 12)
    q = refcos.queries(g["tgt"], g["tgt_off"], 12)
    idx, key = refcos.match_topk(d, q, 4, g["dist"])
    assert np.array_equal(np.where(idx == NO_MATCH, -1, idx.astype(np.int64)), f["top_idx"])
    assert np.array_equal(key, f["top_key"], equal_nan=True)
    m = refcos.mfcc(f["wave"], 44100.0)
    assert np.all(np.abs(m - f["mfcc"]) <= 1e-12 * (1 + np.abs(f["mfcc"])))
    ci, cv = refcos.chain(d, f["chain_start"], f["chain_dist"])
    assert np.array_equal(ci, f["chain_idx"]) and np.array_equal(cv, f["chain_val"])


@pytest.mark.parametrize("dim,band", [(45, -1), (48, 6), (64, -1), (100, 3), (64, 40)])
def test_dtw_wide_frames_use_the_filter_as_a_lower_bound(oracle, dim, band):
    # frames wider than the filter's 42 values: the filter scores the first 42 (a lower bound of the
    # cost), one exact evaluation per target gives the upper bound, the survivors are re-scored
    rng = np.random.default_rng(dim)
    sig = synth.sigma(dim)
    src = [(rng.normal(size=(int(rng.integers(1, 90)), dim)) * sig).astype(np.float32) for _ in range(61)]
    tgt = [(rng.normal(size=(int(rng.integers(1, 90)), dim)) * sig).astype(np.float32) for _ in range(19)]
    tgt[4] = src[8].copy()
    src[30] = src[8].copy()                                  # duplicate: the lower index wins
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    idx, cost = e.match(d, q)
    tm = e.timings()
    want_idx, want_cost, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim,
                                                    band=band, want_matrix=True)
    assert np.array_equal(idx, want_idx)
    fin = np.isfinite(want_cost)
    assert np.allclose(cost[fin], want_cost[fin], rtol=EXACT_RTOL, atol=0) and np.isinf(cost[~fin]).all()
    assert tm["used_filter"] == 1                            # (a band beyond the banded kernel: the unbanded filter as a bound)
    if band <= 47:
        assert tm["n_refined"] < len(src) * len(tgt) // 2
        # the filter matrix never exceeds the exact cost by more than its own error
        fm = e.pair_matrix(d, q)
        ok = np.isfinite(mat)
        lens = np.diff(so).astype(np.float64)[:, None] + np.diff(to).astype(np.float64)[None, :]
        assert np.all(fm[ok] <= mat[ok] + 2e-3 * (1.0 + mat[ok]) + 2.0 ** -11 * 2 * 12.0 * lens[ok])
    # per-target distances on wide frames: the same cascade with key intervals that are open above (the filter's cost
    # bounds a pair from below only); distances in the middle of the costs, near zero and beyond everything
    med = np.nan_to_num(np.nanmedian(np.where(np.isfinite(mat), mat, np.nan), axis=0), nan=1.0)
    for dist in (med, 0.05 * med, 3.0 * med + 1.0):
        i2, c2 = e.match(d, q, distance=dist)
        assert e.timings()["used_filter"] == 1
        key = np.where(np.isfinite(mat), np.abs(mat - dist[None, :]), np.inf)
        want = np.where(np.isfinite(key).any(axis=0), key.argmin(axis=0), 0)
        assert np.array_equal(i2, want)
        have = np.isfinite(key).any(axis=0)
        assert np.allclose(c2[have], mat[want, np.arange(len(tgt))][have], rtol=EXACT_RTOL, atol=0)
    # ... and with top-k
    ti, tc = e.match_topk(d, q, 3, med)
    o_idx, _ = oracle.topk(mat, 3, distance=med, default_distance=0.0, fold_start=float("inf"))
    from soundsym_amd._native import NO_MATCH
    assert np.array_equal(np.where(ti == NO_MATCH, -1, ti.astype(np.int64)), o_idx)
    e.close()


def test_dtw_very_wide_frames_generic_exact_kernel(oracle):
    # wider than 48 values with per-target distances: the generic exact kernel on every pair
    rng = np.random.default_rng(100)
    src = [rng.normal(size=(int(rng.integers(1, 40)), 100)).astype(np.float32) for _ in range(15)]
    tgt = [rng.normal(size=(int(rng.integers(1, 40)), 100)).astype(np.float32) for _ in range(6)]
    sf, so = pack_segments(src, 100, np.float32)
    tf, to = pack_segments(tgt, 100, np.float32)
    e = Engine(metric="dtw", dtype="f32")
    idx, cost = e.match(e.dictionary(sf, so, 100), e.queries(tf, to, 100), force_exact=True)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 100)
    assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=EXACT_RTOL, atol=0)
    e.close()


@pytest.mark.parametrize("dim,dtype,lo,hi", [(13, "f32", 300, 512), (13, "f64", 65, 200), (40, "f32", 129, 400),
                                             (12, "f64", 449, 512), (48, "f32", 70, 130)])
def test_dtw_exact_chunks_pipelined_over_waves(oracle, dim, dtype, lo, hi):
    # sources of 65...512 frames and a short list: dtw_exact_pipe_kernel (one wave per 64-row chunk, bottom rows
    # handed over through LDS as they are produced).  Every pair against the oracle, exact kernel on all of them.
    st = synth.Stream(0x5EED6000 + dim + hi)
    n, m = 9, 7
    ls = lo + st.integers(n, hi - lo + 1)
    lt = 1 + st.integers(m, hi)
    ls[0], ls[1] = hi, lo                                    # the longest (all waves busy) and the shortest
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in ls]
    tgt = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in lt]
    tgt[2] = src[3][5:].copy()
    npdt = np.float32 if dtype == "f32" else np.float64
    sf, so = pack_segments(src, dim, npdt)
    tf, to = pack_segments(tgt, dim, npdt)
    e = Engine(metric="dtw", dtype=dtype)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    _, _, want = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, want_matrix=True,
                                      nthreads=oracle.max_threads())
    got = e.pair_matrix(d, q, exact=True)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    idx, cost = e.match(d, q, force_exact=True)
    assert np.array_equal(idx, want.argmin(axis=0)) and np.allclose(cost, want.min(axis=0), rtol=1e-12, atol=0)
    e.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_refcos_match_one_single_launch_equals_the_batched_path(oracle, dtype):
    # ssym_match_one with the refcos metric is one kernel (refcos_match_one_kernel); it must return bit for bit what the
    # batched path and the oracle return: ragged entries, empty entries, an empty query, lengths around the blocks of
    # eight, distances on both sides of the similarities, a query longer than every entry
    st = synth.Stream(0x5EED6100)
    dim, n = 12, 333
    lens = st.integers(n, 41)
    lens[[5, 77]] = 0
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * 0.05 for f in lens]
    npdt = np.float64 if dtype == "f64" else np.float32
    sf, so = pack_segments(src, dim, npdt)
    e = Engine(metric="refcos", dtype=dtype)
    d = e.dictionary(sf, so, dim)
    queries = [st.normal(int(f) * dim).reshape(int(f), dim) * 0.05 for f in [0, 1, 2, 7, 8, 9, 16, 23, 40, 41, 60, 300]]
    queries.append(src[9].copy())
    queries.append(np.full((3, dim), np.nan))
    for qi, qf in enumerate(queries):
        qv = np.ascontiguousarray(qf, dtype=npdt).reshape(-1)
        for dist in (1.0, 0.0, 0.37, 1.9, -3.0):
            idx, val = e.match_one(d, qv, dist)
            assert e.timings()["main_launches"] == 1
            tf, to = pack_segments([qf], dim, npdt)
            bi, bv = e.match(d, e.queries(tf, to, dim), distance=np.array([dist]))
            wi, wv = oracle.refcos_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, np.array([dist]))
            assert idx == int(bi[0]) == int(wi[0]), (qi, dist)
            assert np.array_equal(np.array([val]), bv) and np.array_equal(bv, wv), (qi, dist, val, bv, wv)
    # the same kernel takes a small batch (ssym_match_batch with up to 64 short queries), with and without distances
    tf, to = pack_segments(queries, dim, npdt)
    for dist in (None, 0.2 + 0.1 * np.arange(len(queries))):
        bi, bv = e.match_batch(d, tf, to, dist)
        assert e.timings()["main_launches"] == 1
        ri, rv = e.match(d, e.queries(tf, to, dim), distance=dist)            # resident queries: the tiled path
        wi, wv = oracle.refcos_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, dist)
        assert np.array_equal(bi, ri) and np.array_equal(bi, wi)
        assert np.array_equal(bv, rv) and np.array_equal(bv, wv)
    # offsets that do not start at 0 (frame_offsets index into the buffer they come with)
    full_i, full_v = e.match_batch(d, tf, to, None)
    bi2, bv2 = e.match_batch(d, tf, to[3:], None)
    assert np.array_equal(bi2, full_i[3:]) and np.array_equal(bv2, full_v[3:])
    e.close()


@pytest.mark.parametrize("dtype,dim,band,squared", [("f64", 12, -1, False), ("f32", 13, -1, False), ("f64", 16, 5, False),
                                                    ("f32", 5, -1, True), ("f64", 13, 0, True)])
def test_dtw_few_queries_single_launch_equals_the_batched_path(oracle, dtype, dim, band, squared):
    # ssym_match_one / ssym_match_batch with up to 4 short dtw queries against short entries is one kernel
    # (dtw_match_few_kernel): same indices and costs as the resident-queries path and as the oracle
    st = synth.Stream(0x5EED6200 + dim + band)
    n = 257
    lens = st.integers(n, 65)
    lens[[3, 100]] = 0
    lens[7] = 64
    src = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in lens]
    src[20] = src[11].copy()                                   # a tie: the lower index wins
    npdt = np.float64 if dtype == "f64" else np.float32
    sf, so = pack_segments(src, dim, npdt)
    e = Engine(metric="dtw", dtype=dtype, band=band, squared=squared)
    d = e.dictionary(sf, so, dim)
    queries = [st.normal(int(f) * dim).reshape(int(f), dim) * synth.sigma(dim) for f in [0, 1, 9, 33, 64, 100, 128]]
    queries.append(src[11].astype(npdt).astype(np.float64))
    queries.append(src[7][:40].astype(npdt).astype(np.float64) + 0.01)
    for qi, qf in enumerate(queries):
        qv = np.ascontiguousarray(qf, dtype=npdt).reshape(-1)
        for dist in (0.0, 25.0):
            idx, val = e.match_one(d, qv, dist)
            assert e.timings()["main_launches"] == 1, qi
            tf, to = pack_segments([qf], dim, npdt)
            bi, bv = e.match(d, e.queries(tf, to, dim), distance=np.array([dist]), force_exact=True)
            assert idx == int(bi[0]), (qi, dist, idx, bi)
            assert np.array_equal(np.array([val]), bv), (qi, dist, val, bv)
            _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                             squared=squared, want_matrix=True)
            key = np.abs(mat[:, 0] - dist)
            key = np.where(np.isnan(key), np.inf, key)
            if np.isfinite(key).any():
                assert idx == int(key.argmin()) and np.isclose(val, mat[idx, 0], rtol=1e-12, atol=0)
            else:
                assert idx == 0 and np.isinf(val)
    # four queries at once, per-query distances
    four = queries[2:6]
    tf, to = pack_segments(four, dim, npdt)
    dist = np.array([0.0, 10.0, 0.0, 40.0])
    bi, bv = e.match_batch(d, tf, to, dist)
    assert e.timings()["main_launches"] == 1
    ri, rv = e.match(d, e.queries(tf, to, dim), distance=dist, force_exact=True)
    assert np.array_equal(bi, ri) and np.array_equal(bv, rv)
    e.close()


@pytest.mark.parametrize("band,squared,dim,with_dist,k", [(-1, False, 13, False, 1), (-1, False, 13, True, 1),
                                                          (-1, True, 13, False, 1), (8, False, 13, True, 1),
                                                          (-1, False, 50, False, 1), (-1, False, 13, True, 4),
                                                          (-1, False, 20, False, 3)])
def test_selection_pretests_keep_the_lists(oracle, band, squared, dim, with_dist, k):
    # dtw_colmin_kernel / dtw_mark_kernel form a pair's key interval only when a cheap necessary condition on its filter
    # cost holds (select.hip): bounds, candidate lists, indices and costs are those of the run that forms every interval
    # (SSYM_SELECT_PRETEST=0), and the oracle's -- unrelated and planted ragged segments, per-target distances on both
    # sides of the costs (the interval test then has a lower AND an upper cut), bands, squared costs, frames wider than the
    # filter takes in (its cost is a lower bound only: no lower cut), top-k rounds
    import os
    src, tgt = synth.make_ragged(300, 200, 3, 70, dim, 0x5EED0C00 + dim + (band & 0xff) + 7 * k)
    for t in range(0, 200, 3):
        tgt[t] = (src[(11 * t) % 300] * np.float32(1.0 + 0.002 * (t % 5))).astype(np.float32)
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band, squared=squared)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    dist = None
    if with_dist:
        base = e.match(d, q)[1]
        fin = np.where(np.isfinite(base), base, 0.0)
        dist = fin * np.tile([0.0, 0.5, 1.0, 1.5, 3.0], 40) + np.tile([0.0, 0.1, 0.0, 7.0, 0.0], 40)

    def run():
        if k > 1:
            r = e.match_topk(d, q, k, distance=dist)
        else:
            r = e.match(d, q, distance=dist)
        return r[0].copy(), r[1].copy(), int(e.timings()["n_refined"])

    on = run()
    os.environ["SSYM_SELECT_PRETEST"] = "0"
    try:
        off = run()
    finally:
        del os.environ["SSYM_SELECT_PRETEST"]
    assert np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1], equal_nan=True) and on[2] == off[2]
    if k == 1:
        _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, dim, band=band,
                                         squared=squared, nthreads=8, want_matrix=True)
        key = np.where(np.isfinite(mat), np.abs(mat - (dist[None, :] if dist is not None else 0.0)), np.inf)
        have = np.isfinite(key).any(axis=0)
        want = np.where(have, key.argmin(axis=0), 0)
        assert np.array_equal(on[0], want)
        assert np.allclose(on[1][have], mat[want, np.arange(len(tgt))][have], rtol=EXACT_RTOL, atol=0)
    e.close()


def test_multi_pair_tasks_small_and_forced(oracle):
    # sources of at most 16 frames run three pairs per wave only when such tasks fill the chip (dtw_filter.hip); a small
    # search takes one pair per wave.  SSYM_SP_MULTIPAIR=2 forces the multi-pair kernel onto the small search: the filter's
    # matrix is the same bits either way -- ragged lengths 0...16, pair counts that do not divide by three, an empty source
    import os
    st = synth.Stream(0x5EED0C77)
    sig = synth.sigma(13)
    for n, m in ((37, 70), (128, 33), (5, 5)):
        ls = st.integers(n, 17)
        lt = 1 + st.integers(m, 60)
        src = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in ls]
        tgt = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in lt]
        sf, so = pack_segments(src, 13, np.float32)
        tf, to = pack_segments(tgt, 13, np.float32)
        e = Engine(metric="dtw", dtype="f32")
        d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
        plain = e.pair_matrix(d, q, exact=False)
        idx, cost = e.match(d, q)
        os.environ["SSYM_SP_MULTIPAIR"] = "2"
        try:
            forced = e.pair_matrix(d, q, exact=False)
            idx2, cost2 = e.match(d, q)
        finally:
            del os.environ["SSYM_SP_MULTIPAIR"]
        assert np.array_equal(plain, forced)
        assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2, equal_nan=True)
        want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tf.astype(np.float64), to, 13, nthreads=8)
        assert np.array_equal(idx, want_idx)
        fin = np.isfinite(want_cost)
        assert np.array_equal(np.isfinite(cost), fin) and np.allclose(cost[fin], want_cost[fin], rtol=EXACT_RTOL, atol=0)
        e.close()
