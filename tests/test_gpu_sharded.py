"""ssym_match_begin / ssym_match_finish: the bound exchange of a source-sharded dtw match.

Two shards are played by two contexts in one process; the all-reduce(MIN) of the ranks is an
elementwise minimum here (the collective itself is covered by the gloo test in test_host.py and by
bench.py's multi-rank path).  Results must equal the unsharded match, and the shard that does not
hold a target's neighbour must re-score (almost) nothing for it.
"""
import numpy as np
import pytest

from soundsym_amd import Engine, SsymError, sharding, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _sets(e, g, lo, hi, f, dim):
    so = np.arange(hi - lo + 1, dtype=np.uint64) * f
    to = np.arange(g.targets.shape[0] + 1, dtype=np.uint64) * f
    return (e.dictionary(np.ascontiguousarray(g.sources[lo:hi]).reshape(-1), so, dim),
            e.queries(g.targets.reshape(-1), to, dim))


def test_begin_finish_without_exchange_equals_match():
    g = synth.make_grid(256, 96, 32, 13, 0x5EED0900)
    e = Engine(metric="dtw", dtype="f32")
    d, q = _sets(e, g, 0, 256, 32, 13)
    m = 96
    want_idx, want_cost = e.match(d, q, index_base=7)
    bounds = torch.empty(m, dtype=torch.float64, device="cuda")
    oi = torch.empty(m, dtype=torch.int32, device="cuda")
    oc = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_begin(d, q, bounds, index_base=7)
    assert bool((bounds >= 0).all()) and bool(torch.isfinite(bounds).all())
    e.match_finish(bounds, oi, oc)
    assert np.array_equal(oi.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(oc.cpu().numpy(), want_cost)
    with pytest.raises(SsymError):                      # finish needs a begin
        e.match_finish(bounds, oi, oc)
    e.close()


def test_two_shards_agree_on_bounds_and_merge_to_the_unsharded_answer(oracle):
    n, m, f, dim = 512, 128, 32, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0901)
    g.sources[300] = g.sources[40]                      # duplicate across the shard boundary
    g.targets[5] = g.sources[40]
    whole = Engine(metric="dtw", dtype="f32")
    d, q = _sets(whole, g, 0, n, f, dim)
    want_idx, want_cost = whole.match(d, q)
    whole.close()

    shards, refined = [], []
    for r in range(2):
        lo, hi = sharding.shard_range(n, 2, r)
        e = Engine(metric="dtw", dtype="f32")
        dd, qq = _sets(e, g, lo, hi, f, dim)
        b = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_begin(dd, qq, b, index_base=lo)
        shards.append((e, dd, qq, b, lo))
    agreed = torch.minimum(shards[0][3], shards[1][3])   # what all_reduce(MIN) leaves on every rank
    costs, idxs = [], []
    for e, dd, qq, b, lo in shards:
        b.copy_(agreed)
        oi = torch.empty(m, dtype=torch.int32, device="cuda")
        oc = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_finish(b, oi, oc)
        refined.append(e.timings()["n_refined"])
        costs.append(oc)
        idxs.append(oi)
    e0 = shards[0][0]
    out_idx, out_cost = sharding.merge_shards(e0, torch.stack(costs), torch.stack(idxs))
    assert np.array_equal(out_idx.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(out_cost.cpu().numpy(), want_cost)
    assert int(out_idx[5]) == 40                         # the lower of the two duplicates
    # every target's neighbour lives in exactly one shard: together the shards re-score about one
    # pair per target, not ~10^2 per target on the shard without the neighbour
    assert sum(refined) <= 2 * m
    # a shard with no candidate for a target reports the fold start there
    far = costs[1].cpu().numpy()[np.asarray(g.planted) < 256]
    assert np.isinf(far).all()
    for e, *_ in shards:
        e.close()


def test_begin_finish_falls_back_where_the_filter_does_not_apply():
    rng = np.random.default_rng(3)
    e = Engine(metric="refcos", dtype="f64")
    src = rng.normal(size=(20, 6, 12))
    tgt = rng.normal(size=(9, 6, 12))
    d = e.dictionary(src.reshape(-1), np.arange(21, dtype=np.uint64) * 6, 12)
    q = e.queries(tgt.reshape(-1), np.arange(10, dtype=np.uint64) * 6, 12)
    want_idx, want_val = e.match(d, q)
    b = torch.empty(9, dtype=torch.float64, device="cuda")
    oi = torch.empty(9, dtype=torch.int32, device="cuda")
    oc = torch.empty(9, dtype=torch.float64, device="cuda")
    e.match_begin(d, q, b)
    assert bool(torch.isinf(b).all())
    e.match_finish(b, oi, oc)
    assert np.array_equal(oi.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(oc.cpu().numpy(), want_val)
    e.close()


def test_two_shards_with_per_target_distances():
    # morph_to's per-target distances in a sharded run: the bound exchange and the merge both work on
    # |cost - distance|
    n, m, f, dim = 256, 64, 24, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0902)
    whole = Engine(metric="dtw", dtype="f32")
    d, q = _sets(whole, g, 0, n, f, dim)
    base_idx, base_cost = whole.match(d, q)
    rng = np.random.default_rng(9)
    dist = rng.uniform(0.5, 2.0, size=m) * np.median(whole.pair_matrix(d, q), axis=0)
    want_idx, want_cost = whole.match(d, q, distance=dist)
    assert not np.array_equal(want_idx, base_idx)         # the distances do change the answer
    whole.close()
    costs, idxs, engines = [], [], []
    bounds = []
    for r in range(2):
        lo, hi = sharding.shard_range(n, 2, r)
        e = Engine(metric="dtw", dtype="f32")
        dd, qq = _sets(e, g, lo, hi, f, dim)
        b = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_begin(dd, qq, b, distance=dist, index_base=lo)
        engines.append((e, dd, qq))
        bounds.append(b)
    agreed = torch.minimum(bounds[0], bounds[1])
    for (e, dd, qq), b in zip(engines, bounds):
        b.copy_(agreed)
        oi = torch.empty(m, dtype=torch.int32, device="cuda")
        oc = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_finish(b, oi, oc)
        costs.append(oc)
        idxs.append(oi)
    out_idx, out_cost = sharding.merge_shards(engines[0][0], torch.stack(costs), torch.stack(idxs), dist)
    assert np.array_equal(out_idx.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(out_cost.cpu().numpy(), want_cost)
    for e, *_ in engines:
        e.close()


@pytest.mark.parametrize("band", [-1, 8])
def test_two_shards_pruned_exchange_candidates_first(oracle, band):
    # early abandoning across shards: candidates' costs reduced with MIN before the filters run, so the shard
    # WITHOUT a target's neighbour abandons against the neighbour's cost too; results stay those of one
    # unsharded, unpruned match
    n, m, f, dim = 512, 128, 64, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0911)
    g.sources[300] = g.sources[40]
    g.targets[5] = g.sources[40]
    whole = Engine(metric="dtw", dtype="f32", band=band)
    d, q = _sets(whole, g, 0, n, f, dim)
    want_idx, want_cost = whole.match(d, q)
    whole.close()

    shards = []
    for r in range(2):
        lo, hi = sharding.shard_range(n, 2, r)
        e = Engine(metric="dtw", dtype="f32", band=band)
        dd, qq = _sets(e, g, lo, hi, f, dim)
        c = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_candidates(dd, qq, c)
        shards.append([e, dd, qq, c, lo])
    cand = torch.minimum(shards[0][3], shards[1][3])
    assert bool(torch.isfinite(cand).all())
    cells = []
    for s in shards:
        e, dd, qq, _, lo = s
        b = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_begin_pruned(dd, qq, b, cand, index_base=lo)
        tm = e.timings()
        assert tm["pruned"] == 1
        cells.append(tm["n_filter_cells"])
        s.append(b)
    agreed = torch.minimum(shards[0][5], shards[1][5])
    costs, idxs = [], []
    for e, dd, qq, _, lo, b in shards:
        b.copy_(agreed)
        oi = torch.empty(m, dtype=torch.int32, device="cuda")
        oc = torch.empty(m, dtype=torch.float64, device="cuda")
        e.match_finish(b, oi, oc)
        costs.append(oc)
        idxs.append(oi)
    out_idx, out_cost = sharding.merge_shards(shards[0][0], torch.stack(costs), torch.stack(idxs))
    assert np.array_equal(out_idx.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(out_cost.cpu().numpy(), want_cost)
    assert int(out_idx[5]) == 40
    # with only its OWN candidates a shard could not drop the targets whose neighbour lives elsewhere
    e, dd, qq, _, lo, _ = shards[1]
    oi = torch.empty(m, dtype=torch.int32, device="cuda")
    oc = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match(dd, qq, index_base=lo, out_idx=oi, out_cost=oc, prune=True)
    assert e.timings()["n_filter_cells"] > cells[1]
    # begin_pruned without candidates for these sets is a plain begin
    b = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_begin_pruned(dd, qq, b, cand, index_base=lo)
    assert e.timings()["pruned"] == 0
    e.match_finish(b, oi, oc)
    for s in shards:
        s[0].close()


def test_candidates_are_dropped_when_another_match_comes_in_between(oracle):
    # ssym_match_candidates leaves its pairs in the context; a different pruned match overwrites them, so the
    # begin that follows must not trust them any more (it runs as a plain begin -- results stay right)
    g = synth.make_grid(256, 96, 48, 13, 0x5EED0922)
    e = Engine(metric="dtw", dtype="f32")
    d, q = _sets(e, g, 0, 256, 48, 13)
    want_idx, want_cost = e.match(d, q)
    other = synth.make_grid(128, 80, 48, 13, 0x5EED0923)
    d2, q2 = _sets(e, other, 0, 128, 48, 13)
    m = 96
    c = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_candidates(d, q, c)
    e.match(d2, q2, prune=True)                                  # reuses the candidate buffers
    b = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_begin_pruned(d, q, b, c)
    assert e.timings()["pruned"] == 0
    oi = torch.empty(m, dtype=torch.int32, device="cuda")
    oc = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_finish(b, oi, oc)
    assert np.array_equal(oi.cpu().numpy().astype(np.int64), want_idx.astype(np.int64))
    assert np.array_equal(oc.cpu().numpy(), want_cost)
    # and a match between begin and finish ends the pair: finish says so instead of folding foreign scratch
    e.match_begin(d, q, b)
    e.match(d2, q2)
    with pytest.raises(SsymError):
        e.match_finish(b, oi, oc)
    e.close()
