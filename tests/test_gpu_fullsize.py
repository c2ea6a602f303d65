"""BASELINE.json sizes: the oracle cannot finish these, so size-independent properties are used.

  * planted neighbours: expected indices are known from the generator alone;
  * sampled pairs / targets re-scored by the oracle;
  * idempotence (same call twice -> identical bits: any difference would be a race);
  * permutation equivariance of the dictionary order.
"""
import numpy as np
import pytest

from soundsym_amd import Engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dtw():
    e = Engine(metric="dtw", dtype="f32")
    yield e
    e.close()


def test_config2_1024x1024x64x13(dtw, oracle):
    g = synth.make_grid(1024, 1024, 64, 13, 0x5EED0002)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d, q = dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13)
    idx, cost = dtw.match(d, q)
    assert np.array_equal(idx, g.planted)
    idx2, cost2 = dtw.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)      # run-to-run identical
    # a sample of targets checked end to end against the oracle (all 1024 sources each)
    pick = np.arange(0, 1024, 128)
    tsel, tosel = synth.Grid(g.sources, g.targets[pick], g.planted[pick], 64, 13).flat("targets")
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, tsel.astype(np.float64), tosel, 13, nthreads=16)
    assert np.array_equal(idx[pick], want_idx)
    assert np.allclose(cost[pick], want_cost, rtol=1e-5, atol=0)
    assert np.allclose(cost[pick], want_cost, rtol=1e-12, atol=0)


def test_config3_4096x4096x128x13(dtw, oracle):
    g = synth.make_grid(4096, 4096, 128, 13, 0x5EED0003)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d, q = dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13)
    idx, cost = dtw.match(d, q)
    tm = dtw.timings()
    assert tm["used_filter"] == 1 and tm["n_pairs"] == 4096 * 4096
    assert np.array_equal(idx, g.planted)
    # winners' costs against the oracle on the planted pairs (cheap: one pair per sampled target)
    for t in range(0, 4096, 256):
        want = oracle.dtw(g.sources[g.planted[t]].astype(np.float64), g.targets[t].astype(np.float64), 13)
        assert abs(cost[t] - want) <= 1e-12 * want
    # permuting the dictionary permutes the answers
    perm = synth.Stream(99).permutation(4096)
    d2 = dtw.dictionary(np.ascontiguousarray(g.sources[perm]).reshape(-1), so, 13)
    idx_p, cost_p = dtw.match(d2, q)
    assert np.array_equal(perm[idx_p], idx) and np.array_equal(cost_p, cost)


def test_refcos_4096x4096x128x12(oracle):
    e = Engine(metric="refcos", dtype="f64")
    g = synth.make_grid(4096, 4096, 128, 12, 0x5EED0013)
    sf, so = g.flat("sources", np.float64)
    tf, to = g.flat("targets", np.float64)
    sf *= 0.02
    tf *= 0.02
    d, q = e.dictionary(sf, so, 12), e.queries(tf, to, 12)
    idx, val = e.match(d, q)
    pick = np.arange(0, 4096, 512)
    tsel = np.concatenate([tf[int(to[t]) * 12:int(to[t + 1]) * 12] for t in pick])
    tosel = np.arange(pick.size + 1, dtype=np.uint64) * 128
    want_idx, want_val = oracle.refcos_match_all(sf, so, tsel, tosel, 12)
    assert np.array_equal(idx[pick], want_idx) and np.array_equal(val[pick], want_val)
    idx2, val2 = e.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(val, val2)
    e.close()


def test_config5_shape_banded_512x512x256x40():
    # BASELINE configs[4] shape (256 frames x 40 dims, Sakoe-Chiba r = 32) at 512 x 512 segments:
    # planted neighbours recovered, run-to-run identical
    e = Engine(metric="dtw", dtype="f32", band=32)
    g = synth.make_grid(512, 512, 256, 40, 0x5EED0005)
    sf, so = g.flat("sources")
    tf, to = g.flat("targets")
    d, q = e.dictionary(sf, so, 40), e.queries(tf, to, 40)
    idx, cost = e.match(d, q)
    tm = e.timings()
    assert tm["used_filter"] == 1
    assert np.array_equal(idx, g.planted)
    idx2, cost2 = e.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)
    e.close()


def test_config4_16384x4096_source_sharded_on_one_gpu(dtw):
    """BASELINE configs[3]: 16384 x 4096 segments, 128 frames x 13 dims, source axis split in 8
    shards.  The 8 ranks' work is run one after the other on this GPU (same calls a rank makes:
    match with index_base, then the gathered [8, M] candidates through the HIP merge kernel);
    the planted neighbours are known from the generator alone."""
    import torch
    from soundsym_amd import sharding
    g = synth.make_grid(16384, 4096, 128, 13, 0x5EED0004)
    tf, to = g.flat("targets")
    q = dtw.queries(torch.from_numpy(tf).cuda(), to, 13)
    costs = torch.empty((8, 4096), dtype=torch.float64, device="cuda")
    idxs = torch.empty((8, 4096), dtype=torch.int32, device="cuda")
    refined = 0
    for rank in range(8):
        lo, hi = sharding.shard_range(16384, 8, rank)
        sf = np.ascontiguousarray(g.sources[lo:hi]).reshape(-1)
        so = np.arange(hi - lo + 1, dtype=np.uint64) * 128
        d = dtw.dictionary(torch.from_numpy(sf).cuda(), so, 13)
        dtw.match(d, q, index_base=lo, out_idx=idxs[rank], out_cost=costs[rank])
        refined += dtw.timings()["n_refined"]
        d.close()
    out_idx, out_cost = sharding.merge_shards(dtw, costs, idxs)
    assert np.array_equal(out_idx.cpu().numpy().astype(np.int64), g.planted)
    # a rank that does not hold a target's neighbour must not re-score many pairs
    assert refined <= 8 * 4096 * 3, refined
    c = out_cost.cpu().numpy()
    assert np.isfinite(c).all() and (c > 0).all()
