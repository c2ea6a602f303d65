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


def test_config5_full_size_4096x4096x256x40_band32(oracle):
    """BASELINE configs[4] at FULL size on one GPU: 4096 x 4096 segments, 256 frames x 40 dims, Sakoe-Chiba
    r = 32.  Planted indices, run-to-run identical bits, 16 sampled targets' winners against the oracle over
    all 4096 sources, and 1024 sampled pairs of the banded filter's matrix against ssym_oracle_dtw(band = 32)
    under the filter's derived bound."""
    from bounds import worst_case_bound
    n = m = 4096
    f, dim, band = 256, 40, 32
    # the bench's workload itself: same generator, same seed (0x5EED0000 + config number; `bench.py
    # --workload c5` draws exactly this grid)
    g = synth.make_grid(n, m, f, dim, 0x5EED0005)
    src, tgt, perm = g.sources, g.targets, g.planted
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(src.reshape(-1), so, dim), e.queries(tgt.reshape(-1), to, dim)
    idx, cost = e.match(d, q)
    tm = e.timings()
    assert tm["used_filter"] == 1 and tm["n_pairs"] == n * m
    assert np.array_equal(idx, perm)
    idx2, cost2 = e.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)              # run-to-run identical
    # 16 sampled targets end to end against the oracle (all 4096 sources each)
    pick = np.arange(100, m, 256)
    assert pick.size == 16
    tsel = np.ascontiguousarray(tgt[pick]).reshape(-1).astype(np.float64)
    tosel = np.arange(pick.size + 1, dtype=np.uint64) * f
    want_idx, want_cost = oracle.dtw_match_all(src.reshape(-1).astype(np.float64), so, tsel, tosel, dim, band=band,
                                               nthreads=oracle.max_threads())
    assert np.array_equal(idx[pick], want_idx)
    assert np.allclose(cost[pick], want_cost, rtol=1e-5, atol=0)                  # the north star's tolerance
    assert np.allclose(cost[pick], want_cost, rtol=1e-12, atol=0)                 # what the exact kernel delivers
    # 1024 sampled pairs of the filter's matrix (winners, near and far pairs) under its bound
    filt = e.pair_matrix(d, q, exact=False)
    e.close()
    rng = np.random.default_rng(5)
    ss = np.concatenate([rng.integers(0, n, 960), perm[:64]])
    tt = np.concatenate([rng.integers(0, m, 960), np.arange(64)])
    bound, _ = worst_case_bound(src, tgt, dim, f, f)
    worst = 0.0
    for s_, t_ in zip(ss, tt):
        want = oracle.dtw(src[s_].astype(np.float64), tgt[t_].astype(np.float64), dim, band=band)
        assert np.isfinite(filt[s_, t_]) and abs(filt[s_, t_] - want) <= bound + 1e-5 * want, (s_, t_, filt[s_, t_], want, bound)
        worst = max(worst, abs(filt[s_, t_] - want) / want)
    assert worst < 1e-2, worst        # (in practice the f16 filter is within ~1e-3 relative at this shape)


def test_config4_16384x4096_source_sharded_on_one_gpu(dtw, oracle):
    """BASELINE configs[3]: 16384 x 4096 segments, 128 frames x 13 dims, source axis split in 8
    shards.  The 8 ranks' work is run one after the other on this GPU (same calls a rank makes:
    match with index_base, then the gathered [8, M] candidates through the HIP merge kernel);
    the planted neighbours are known from the generator alone, and 16 sampled targets' winners and
    costs are checked against the oracle over ALL 16 384 sources (`bench.py --workload c4` draws this grid)."""
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
    # 16 sampled targets end to end against the oracle (all 16 384 sources each: 262 144 pairs)
    pick = np.arange(77, 4096, 256)
    assert pick.size == 16
    sf, so = g.flat("sources", np.float64)
    tsel = np.ascontiguousarray(g.targets[pick], dtype=np.float64).reshape(-1)
    tosel = np.arange(pick.size + 1, dtype=np.uint64) * 128
    want_idx, want_cost = oracle.dtw_match_all(sf, so, tsel, tosel, 13, nthreads=oracle.max_threads())
    assert np.array_equal(out_idx.cpu().numpy().view(np.uint32)[pick].astype(np.int64), want_idx)
    assert np.allclose(c[pick], want_cost, rtol=1e-5, atol=0)                     # the north star's tolerance
    assert np.allclose(c[pick], want_cost, rtol=1e-12, atol=0)                    # what the exact kernel delivers


def test_config4_eight_ranks_through_ssym_match_sharded(dtw):
    """BASELINE configs[3] as it is stated -- 16384 x 4096 segments, source-sharded across 8 ranks -- through the call the
    bench makes per rank, ssym_match_sharded: eight ranks as eight threads of this process on the one GPU (the in-process
    transport in RCCL's place: same block layout, same bound exchange, same merge over 8 shards).  Every rank must return
    the whole answer: the planted neighbours, identical costs on all ranks, equal to what the unsharded search of the
    same 16384 sources returns, and a shard that does not hold a target's neighbour re-scores next to nothing."""
    from test_gpu_comm import _run_ranks
    from soundsym_amd import sharding
    n, m, f, dim, world = 16384, 4096, 128, 13, 8
    g = synth.make_grid(n, m, f, dim, 0x5EED0004)
    to = np.arange(m + 1, dtype=np.uint64) * f
    shards, bases = [], []
    for r in range(world):
        lo, hi = sharding.shard_range(n, world, r)
        shards.append((np.ascontiguousarray(g.sources[lo:hi]).reshape(-1), np.arange(hi - lo + 1, dtype=np.uint64) * f))
        bases.append(lo)
    res = _run_ranks(world, "dtw", "f32", shards, dim, g.targets.reshape(-1), to, bases)
    so = np.arange(n + 1, dtype=np.uint64) * f
    want_idx, want_cost = dtw.match(dtw.dictionary(g.sources.reshape(-1), so, dim), dtw.queries(g.targets.reshape(-1), to, dim))
    assert np.array_equal(want_idx, g.planted)
    refined = 0
    for r in range(world):
        idx, cost, tm = res[r]
        assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost), r
        assert tm["attempts"] == 1 and tm["used_filter"] == 1
        refined += tm["n_refined"]
    assert refined <= 3 * m, refined


# --- the reference's real segment shape: 4096 x 4096 ragged segments of 5...40 frames (bench.py secondary.ragged) -----

def _ragged_pick(src, tgt, pick, dim, dtype):
    from soundsym_amd.engine import pack_segments
    sf, so = pack_segments(src, dim, dtype)
    tf, to = pack_segments(tgt, dim, dtype)
    ts, tos = pack_segments([tgt[i] for i in pick], dim, dtype)
    return (sf, so), (tf, to), (ts, tos)


@pytest.mark.parametrize("planted", [False, True])
def test_ragged_4096x4096_dtw(dtw, oracle, planted):
    """SoundDictionary::add_segments emits segments of seg / HOP frames (src/sound.rs:330-343, src/lib.rs:137): short
    and ragged.  The full-size ragged grid of bench.py's secondary.ragged through the filter path (one launch per
    class of source lengths on dtw_filter_sp_kernel), 16 sampled targets against the oracle over all 4096 sources,
    run-to-run identical; the same search with SSYM_FILTER_SP=0 semantics is covered by test_gpu_numerics."""
    import bench
    if planted:
        src, tgt, pi = synth.make_ragged(4096, 4096, 5, 40, 13, bench.RAGGED_SEED + 1, planted=True)
    else:
        src, tgt = synth.make_ragged(4096, 4096, 5, 40, 13, bench.RAGGED_SEED)
    pick = np.arange(7, 4096, 256)
    (sf, so), (tf, to), (ts, tos) = _ragged_pick(src, tgt, pick, 13, np.float32)
    d, q = dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13)
    idx, cost = dtw.match(d, q)
    tm = dtw.timings()
    assert tm["used_filter"] == 1 and tm["n_pairs"] == 4096 * 4096 and tm["main_launches"] == 3
    true_cells = float(np.diff(so).astype(np.float64).sum()) * float(np.diff(to).astype(np.float64).sum())
    assert true_cells <= tm["n_filter_cells"] <= 1.35 * true_cells      # padding: row blocks of 4, a group's longest target
    idx2, cost2 = dtw.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)    # run-to-run identical
    if planted:
        assert np.array_equal(idx.astype(np.int64), pi)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, ts.astype(np.float64), tos, 13, nthreads=16)
    assert np.array_equal(idx[pick], want_idx)
    assert np.allclose(cost[pick], want_cost, rtol=1e-5, atol=0)        # the north star's tolerance
    assert np.allclose(cost[pick], want_cost, rtol=1e-12, atol=0)       # what the exact re-scoring delivers


def test_ragged_4096x4096_refcos(oracle):
    """The reference's own metric on the same shape (12 values per frame, f64, the dot over the common prefix,
    src/sound.rs:24-31): indices and values bit for bit against the oracle for 16 sampled targets, run-to-run identical."""
    import bench
    src, tgt = synth.make_ragged(4096, 4096, 5, 40, 12, bench.RAGGED_SEED + 2)
    src = [a.astype(np.float64) * 0.05 for a in src]
    tgt = [a.astype(np.float64) * 0.05 for a in tgt]
    pick = np.arange(3, 4096, 256)
    (sf, so), (tf, to), (ts, tos) = _ragged_pick(src, tgt, pick, 12, np.float64)
    r = Engine(metric="refcos", dtype="f64")
    d, q = r.dictionary(sf, so, 12), r.queries(tf, to, 12)
    idx, val = r.match(d, q)
    assert r.timings()["used_filter"] == 1
    idx2, val2 = r.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(val, val2)
    want_idx, want_val = oracle.refcos_match_all(sf, so, ts, tos, 12)
    assert np.array_equal(idx[pick], want_idx) and np.array_equal(val[pick], want_val)
    r.close()


def test_ragged_long_sources_multi_pass_classes(dtw, oracle):
    """Sources beyond 48 frames of ragged length run in classes by the row passes that pad them least (passes of 48 or 64
    rows, dtw_filter.hip): 1024 sources of 49...200 frames against 256 targets of 20...150 frames -- several multi-pass
    launches -- with 16 sampled targets held against the oracle over all sources, and the same search with every long pair
    in the set's own shape (SSYM_FILTER_LONG_CLASSES semantics are a measurement knob; here the answers must simply be the
    oracle's)."""
    from soundsym_amd.engine import pack_segments
    st = synth.Stream(0x5EED0A90)
    sig = synth.sigma(13)
    ls = 49 + st.integers(1024, 152)
    lt = 20 + st.integers(256, 131)
    src = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in ls]
    tgt = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in lt]
    for t in range(0, 256, 5):                                   # plant neighbours of another length for a fifth of the targets
        a = src[(7 * t) % 1024]
        rows = np.rint(np.linspace(0.0, a.shape[0] - 1.0, max(20, a.shape[0] - 3))).astype(np.int64)
        tgt[t] = (a[rows] + 0.05 * sig * st.normal(rows.size * 13).reshape(rows.size, 13)).astype(np.float32)
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    d, q = dtw.dictionary(sf, so, 13), dtw.queries(tf, to, 13)
    idx, cost = dtw.match(d, q)
    tm = dtw.timings()
    assert tm["used_filter"] == 1 and tm["main_launches"] >= 3, tm
    true_cells = float(np.diff(so).astype(np.float64).sum()) * float(np.diff(to).astype(np.float64).sum())
    assert true_cells <= tm["n_filter_cells"] <= 1.45 * true_cells, tm["n_filter_cells"] / true_cells
    idx2, cost2 = dtw.match(d, q)
    assert np.array_equal(idx, idx2) and np.array_equal(cost, cost2)
    pick = np.arange(0, 256, 16)
    ts, tos = pack_segments([tgt[i] for i in pick], 13, np.float32)
    want_idx, want_cost = oracle.dtw_match_all(sf.astype(np.float64), so, ts.astype(np.float64), tos, 13, nthreads=16)
    assert np.array_equal(idx[pick], want_idx)
    assert np.allclose(cost[pick], want_cost, rtol=1e-12, atol=0)
    # a task's first pass skips the cells of a first tile that holds only padding (dtw_filter_kernel SKIP0): the filter's
    # values are the same bits with the skip compiled out of the launch, and fewer cells are evaluated with it
    import os
    filt = dtw.pair_matrix(d, q, exact=False)
    os.environ["SSYM_FILTER_SKIP0"] = "0"
    try:
        filt0 = dtw.pair_matrix(d, q, exact=False)
        idx0, cost0 = dtw.match(d, q)
        cells0 = dtw.timings()["n_filter_cells"]
    finally:
        del os.environ["SSYM_FILTER_SKIP0"]
    assert np.array_equal(filt, filt0) and np.isfinite(filt).all()
    assert np.array_equal(idx, idx0) and np.array_equal(cost, cost0)
    assert tm["n_filter_cells"] < 0.97 * cells0, (tm["n_filter_cells"], cells0)
