"""ssym_comm_* / ssym_match_sharded: the source-sharded match with its RCCL collectives inside the library.

One GPU is all a test box has, and RCCL refuses two ranks on one device, so what runs here is a
WORLD-1 communicator: the all-reduce and the all-gather are real RCCL calls on the context's stream,
the merge sees one shard, and the result must be bit for bit ssym_match_queries' -- for every path
the step has (filter, per-target distances, early abandoning, bands, shapes outside the filter,
refcos, an empty shard, an overflowing candidate list that makes every rank repeat the tail).
Range / gather / merge logic with two ranks: tests/test_host.py (gloo), tests/test_gpu_sharded.py.
MORE than one rank through ssym_match_sharded itself: the second half of this file -- a thread per rank on the one
GPU with the in-process transport (ssym_comm_create_local) in RCCL's place.
"""
import numpy as np
import pytest

from soundsym_amd import Engine, sharding, synth
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _grid_sets(e, g, f, dim):
    so = np.arange(g.sources.shape[0] + 1, dtype=np.uint64) * f
    to = np.arange(g.targets.shape[0] + 1, dtype=np.uint64) * f
    return e.dictionary(g.sources.reshape(-1), so, dim), e.queries(g.targets.reshape(-1), to, dim)


@pytest.mark.parametrize("band,prune,with_dist", [(-1, False, False), (-1, True, False), (-1, False, True),
                                                   (6, False, False), (6, True, False)])
def test_world1_sharded_equals_match_queries(band, prune, with_dist):
    n, m, f, dim = 384, 160, 40, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0A00 + (band & 0xff))
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = _grid_sets(e, g, f, dim)
    dist = np.linspace(0.0, 30.0, m) if with_dist else None
    want_idx, want_cost = e.match(d, q, distance=dist, index_base=1000)
    comm = sharding.init_comm(e, 0, 1)
    idx, cost = sharding.match_sharded(e, comm, d, q, 1000, distance=dist, prune=prune)
    tm = e.timings()
    assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost)
    assert tm["used_filter"] == 1 and tm["attempts"] == 1 and tm["collective_ms"] > 0
    assert tm["pruned"] == (1 if prune else 0)
    if not with_dist:
        assert np.array_equal(idx.astype(np.int64) - 1000, g.planted)
    # device outputs, twice (buffers and communicator are reused)
    oi = torch.empty(m, dtype=torch.int32, device="cuda")
    oc = torch.empty(m, dtype=torch.float64, device="cuda")
    for _ in range(2):
        sharding.match_sharded(e, comm, d, q, 1000, out_idx=oi, out_cost=oc, distance=dist, prune=prune)
        e.synchronize()
        assert np.array_equal(oi.cpu().numpy().view(np.uint32), want_idx) and np.array_equal(oc.cpu().numpy(), want_cost)
    comm.close()
    e.close()


def test_world1_sharded_outside_the_filter_and_refcos(oracle):
    # frames of 100 values with per-target distances: the exact kernel on every pair (no filter, no bounds)
    rsrc, rtgt = synth.make_ragged(40, 24, 3, 20, 100, 0x5EED0A10)
    sf, so = pack_segments(rsrc, 100, np.float32)
    tf, to = pack_segments(rtgt, 100, np.float32)
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(sf, so, 100), e.queries(tf, to, 100)
    dist = np.linspace(1.0, 50.0, 24)
    want = e.match(d, q, distance=dist, index_base=5)
    comm = sharding.init_comm(e, 0, 1)
    got = sharding.match_sharded(e, comm, d, q, 5, distance=dist)
    assert e.timings()["used_filter"] == 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    comm.close()
    e.close()
    # refcos: the keys travel, the merge compares them as they are
    r = Engine(metric="refcos", dtype="f64")
    rs, rt = synth.make_ragged(90, 33, 2, 25, 12, 0x5EED0A11)
    sf, so = pack_segments([s.astype(np.float64) * 0.05 for s in rs], 12)
    tf, to = pack_segments([t.astype(np.float64) * 0.05 for t in rt], 12)
    d, q = r.dictionary(sf, so, 12), r.queries(tf, to, 12)
    want = r.match(d, q)
    comm = sharding.init_comm(r, 0, 1)
    got = sharding.match_sharded(r, comm, d, q, 0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, 12)
    assert np.array_equal(got[0], want_idx) and np.array_equal(got[1], want_val)
    comm.close()
    r.close()


def test_world1_empty_shard_reports_the_fold_start():
    e = Engine(metric="dtw", dtype="f32")
    g = synth.make_grid(8, 5, 10, 13, 0x5EED0A20)
    d = e.dictionary(np.zeros(0, dtype=np.float32), np.zeros(1, dtype=np.uint64), 13)
    q = e.queries(g.targets.reshape(-1), np.arange(6, dtype=np.uint64) * 10, 13)
    comm = sharding.init_comm(e, 0, 1)
    idx, cost = sharding.match_sharded(e, comm, d, q, 77)
    assert (idx == 77).all() and np.isinf(cost).all()
    idx, cost = sharding.match_sharded(e, comm, d, q, 77, prune=True)
    assert (idx == 77).all() and np.isinf(cost).all()
    comm.close()
    e.close()


def test_world1_overflowing_candidate_list_repeats_the_tail(oracle):
    # every source identical: list 1 wants all 600 x 200 pairs; the gathered status tells every rank,
    # and the tail (selection ... merge) is repeated once with the room asked for
    n, m, f = 600, 200, 8
    one = synth.make_grid(1, 1, f, 13, 0x5EED0395).sources[0]
    src = np.repeat(one[None], n, axis=0)
    tgt = synth.make_grid(m, 1, f, 13, 0x5EED0396).sources
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    e = Engine(metric="dtw", dtype="f32")
    d, q = e.dictionary(src.reshape(-1), so, 13), e.queries(tgt.reshape(-1), to, 13)
    want = e.match(d, q)
    comm = sharding.init_comm(e, 0, 1)
    got = sharding.match_sharded(e, comm, d, q, 0)
    tm = e.timings()
    assert tm["attempts"] == 2 and tm["n_refined"] == n * m
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (got[0] == 0).all()
    comm.close()
    e.close()


# ---------------------------------------------------------------------------------------------------------
# More than one rank on the one GPU: a thread per rank, every rank its own context, the in-process transport
# (ssym_comm_create_local) instead of RCCL.  Same ssym_match_sharded, G > 1: gather layout and strides, merge over
# G shards, bounds exchange, the agreed repeat after ONE rank's overflow, an empty shard among full ones.
# ---------------------------------------------------------------------------------------------------------
def _run_ranks(world, metric, dtype, shards, dim, tflat, toff, bases, band=-1, distance=None, prune=False):
    """shards: per rank (flat features, frame offsets).  Returns per rank (idx, cost, timings)."""
    import threading
    from soundsym_amd.engine import LocalGroup
    group = LocalGroup(world)
    out, err = [None] * world, [None] * world
    ready = threading.Barrier(world)

    def rank_main(r):
        try:
            e = Engine(metric=metric, dtype=dtype, band=band)
            d = e.dictionary(shards[r][0], shards[r][1], dim)
            q = e.queries(tflat, toff, dim)
            comm = e.comm_create_local(group, r)
            ready.wait(timeout=120)
            for _ in range(2):                                   # twice: buffers and group are reused
                idx, cost = e.match_sharded(comm, d, q, distance=distance, index_base=bases[r], prune=prune)
            out[r] = (idx.copy(), cost.copy(), e.timings())
            comm.close()
            e.close()
        except Exception as ex:                                  # noqa: BLE001
            err[r] = ex
            try:
                ready.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    group.close()
    for r in range(world):
        assert err[r] is None, (r, err[r])
        assert out[r] is not None, f"rank {r} did not finish"
    return out


def _split(segs, world, dim, dtype):
    parts, bases = [], []
    for r in range(world):
        lo, hi = sharding.shard_range(len(segs), world, r)
        parts.append(pack_segments(segs[lo:hi], dim, dtype))
        bases.append(lo)
    return parts, bases


@pytest.mark.parametrize("world,band,prune,with_dist", [(2, -1, False, False), (3, -1, True, False), (4, -1, False, True),
                                                         (2, 6, False, False), (3, 6, True, False)])
def test_multi_rank_in_process_equals_the_unsharded_match(world, band, prune, with_dist):
    n, m, f, dim = 300, 140, 36, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0A40 + world)
    g.sources[200] = g.sources[17]                       # a duplicate across shard boundaries: the lowest index wins
    g.targets[3] = g.sources[17]
    dist = np.linspace(0.0, 25.0, m) if with_dist else None
    e = Engine(metric="dtw", dtype="f32", band=band)
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    want_idx, want_cost = e.match(e.dictionary(g.sources.reshape(-1), so, dim), e.queries(g.targets.reshape(-1), to, dim),
                                  distance=dist)
    e.close()
    shards, bases = _split(list(g.sources), world, dim, np.float32)
    res = _run_ranks(world, "dtw", "f32", shards, dim, g.targets.reshape(-1), to, bases, band=band, distance=dist, prune=prune)
    refined = 0
    for r in range(world):
        idx, cost, tm = res[r]
        assert np.array_equal(idx, want_idx) and np.array_equal(cost, want_cost), r       # every rank: the whole answer
        assert tm["attempts"] == 1 and tm["used_filter"] == 1
        refined += tm["n_refined"]
    assert want_idx[3] == 17
    if not with_dist and not prune:
        assert refined <= 3 * m, refined                 # the bound exchange: a shard without the neighbour re-scores (almost) nothing


def test_multi_rank_refcos_with_distances_and_an_empty_shard(oracle):
    # refcos shards report the key |sim - distance| itself: the merge compares keys as they are; rank 2 holds nothing
    rs, rt = synth.make_ragged(90, 40, 1, 25, 12, 0x5EED0A50)
    src = [s.astype(np.float64) * 0.05 for s in rs]
    tgt = [t.astype(np.float64) * 0.05 for t in rt]
    src[70] = src[5].copy()
    tgt[0] = src[5].copy()
    tf, to = pack_segments(tgt, 12)
    sf, so = pack_segments(src, 12)
    dist = np.linspace(0.2, 1.4, 40)
    want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, 12, distance=dist)
    cuts = [(0, 40), (40, 90), (90, 90)]
    shards = [pack_segments(src[a:b], 12) for a, b in cuts]
    res = _run_ranks(3, "refcos", "f64", shards, 12, tf, to, [a for a, _ in cuts], distance=dist)
    for r in range(3):
        assert np.array_equal(res[r][0], want_idx) and np.array_equal(res[r][1], want_val), r


def test_multi_rank_one_rank_overflows_and_every_rank_repeats(oracle):
    # rank 0's sources are all identical (its candidate list wants every pair), rank 1's are ordinary: the gathered
    # status makes BOTH ranks repeat the tail; the identical sources tie, the lowest global index wins
    m, f, dim = 200, 8, 13
    one = synth.make_grid(1, 1, f, dim, 0x5EED0395).sources[0]
    same = np.repeat(one[None], 600, axis=0)
    other = synth.make_grid(64, 1, f, dim, 0x5EED0A60).sources
    rng = np.random.default_rng(4)
    tgt = (one[None] + 0.01 * rng.standard_normal((m, f, dim))).astype(np.float32)    # every target's best: the 600 ties
    allsrc = np.concatenate([same, other])
    so = np.arange(allsrc.shape[0] + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    want_idx, want_cost = oracle.dtw_match_all(allsrc.reshape(-1).astype(np.float64), so, tgt.reshape(-1).astype(np.float64), to, dim)
    shards = [(same.reshape(-1), np.arange(601, dtype=np.uint64) * f), (other.reshape(-1), np.arange(65, dtype=np.uint64) * f)]
    res = _run_ranks(2, "dtw", "f32", shards, dim, tgt.reshape(-1), to, [0, 600])
    for r in range(2):
        idx, cost, tm = res[r]
        assert tm["attempts"] == 2, (r, tm)
        assert np.array_equal(idx, want_idx) and np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    assert (want_idx == 0).all()


def test_sharded_refcos_goes_through_the_matrix_pipe(oracle):
    """A refcos shard of 65 536 pairs and more runs its search on the f64 matrix pipe inside ssym_match_sharded too
    (filter + exact keys of the candidates; the list's header travels in the gathered status): world-1 RCCL and two
    in-process ranks, plain and with per-target distances, bit for bit the oracle's answer; a shard whose list
    overflows (every source tied with every other) makes every rank repeat the tail on the exact tile kernel."""
    rs, rt = synth.make_ragged(640, 260, 2, 24, 12, 0x5EED0A80)
    src = [s_.astype(np.float64) * 0.05 for s_ in rs]
    tgt = [t_.astype(np.float64) * 0.05 for t_ in rt]
    src[400] = src[9].copy()
    tgt[0] = src[9].copy()                                    # two equal keys across the shard boundary (refcos: not necessarily the winners)
    sf, so = pack_segments(src, 12)
    tf, to = pack_segments(tgt, 12)
    dist = np.linspace(0.1, 1.3, 260)
    r = Engine(metric="refcos", dtype="f64")
    d, q = r.dictionary(sf, so, 12), r.queries(tf, to, 12)
    comm = sharding.init_comm(r, 0, 1)
    for dd in (None, dist):
        want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, 12, distance=dd)
        got = sharding.match_sharded(r, comm, d, q, 0, distance=dd)
        tm = r.timings()
        assert tm["used_filter"] == 1 and tm["attempts"] == 1 and tm["n_refined"] < 8 * 260, tm
        assert tm["refcos_filter"] == 2, tm           # (the integer filter: its records are built when the step starts)
        assert np.array_equal(got[0], want_idx) and np.array_equal(got[1], want_val)
    comm.close()
    r.close()
    cuts = [(0, 320), (320, 640)]
    shards = [pack_segments(src[a:b], 12) for a, b in cuts]
    for dd in (None, dist):
        want_idx, want_val = oracle.refcos_match_all(sf, so, tf, to, 12, distance=dd)
        res = _run_ranks(2, "refcos", "f64", shards, 12, tf, to, [a for a, _ in cuts], distance=dd)
        for rk in range(2):
            assert res[rk][2]["used_filter"] == 1 and res[rk][2]["attempts"] == 1 and res[rk][2]["refcos_filter"] == 2
            assert np.array_equal(res[rk][0], want_idx) and np.array_equal(res[rk][1], want_val), rk
    # rank 0's sources all identical: its list 1 overflows, both ranks repeat the tail, rank 0 on the exact tile kernel
    rng = np.random.default_rng(8)
    one = rng.standard_normal((6, 12)) * 0.1
    same = [one.copy() for _ in range(2200)]
    other = [rng.standard_normal((6, 12)) * 0.1 for _ in range(300)]
    tg2 = [rng.standard_normal((6, 12)) * 0.1 for _ in range(600)]
    tf2, to2 = pack_segments(tg2, 12)
    sfa, soa = pack_segments(same + other, 12)
    want_idx, want_val = oracle.refcos_match_all(sfa, soa, tf2, to2, 12)
    res = _run_ranks(2, "refcos", "f64", [pack_segments(same, 12), pack_segments(other, 12)], 12, tf2, to2, [0, 2200])
    for rk in range(2):
        assert res[rk][2]["attempts"] == 2, (rk, res[rk][2])
        assert np.array_equal(res[rk][0], want_idx) and np.array_equal(res[rk][1], want_val), rk


# ---------------------------------------------------------------------------------------------------------
# Failure containment (include/soundsym_amd.h, "FAILURE on one rank is part of the protocol"): no rank may be left
# waiting, and a failure that can still be reported is reported by EVERY rank with the same status.
# ---------------------------------------------------------------------------------------------------------
def _run_ranks_with_fault(world, fault_rank, phase, kind, timeout_ms=20000, absent_rank=None, steps_after=1, prune=False,
                          distance=None):
    """A thread per rank on the one GPU (in-process transport).  Returns per rank a dict: the status code the faulty
    step raised (None = no error), the seconds it took, whether the communicator is dead, and what a further step on
    the same communicator did."""
    import threading
    import time
    from soundsym_amd import _native as nat
    from soundsym_amd.engine import LocalGroup
    n, m, f, dim = 240, 96, 24, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0A70)
    to = np.arange(m + 1, dtype=np.uint64) * f
    e0 = Engine(metric="dtw", dtype="f32")
    so = np.arange(n + 1, dtype=np.uint64) * f
    want = e0.match(e0.dictionary(g.sources.reshape(-1), so, dim), e0.queries(g.targets.reshape(-1), to, dim), distance=distance)
    e0.close()
    shards, bases = _split(list(g.sources), world, dim, np.float32)
    group = LocalGroup(world)
    res = [dict() for _ in range(world)]
    ready = threading.Barrier(world)

    def rank_main(r):
        try:
            e = Engine(metric="dtw", dtype="f32")
            d = e.dictionary(shards[r][0], shards[r][1], dim)
            q = e.queries(g.targets.reshape(-1), to, dim)
            comm = e.comm_create_local(group, r)
            comm.set_timeout(timeout_ms)
            idx, cost = e.match_sharded(comm, d, q, index_base=bases[r], prune=prune, distance=distance)         # a healthy step first
            res[r]["healthy"] = bool(np.array_equal(idx, want[0]) and np.array_equal(cost, want[1]))
            if r == fault_rank:
                comm.inject_fault(phase, kind)
            ready.wait(timeout=120)
            t0 = time.perf_counter()
            code = None
            if r != absent_rank:
                try:
                    e.match_sharded(comm, d, q, index_base=bases[r], prune=prune, distance=distance)
                except nat.SsymError as ex:
                    code, res[r]["msg"] = ex.code, str(ex)
            res[r]["code"], res[r]["seconds"] = code, time.perf_counter() - t0
            res[r]["dead"] = comm.dead
            after = []
            for _ in range(steps_after):
                try:
                    idx, cost = e.match_sharded(comm, d, q, index_base=bases[r], prune=prune, distance=distance)
                    after.append(bool(np.array_equal(idx, want[0]) and np.array_equal(cost, want[1])))
                except nat.SsymError as ex:
                    after.append(ex.code)
            res[r]["after"] = after
            comm.close()
            e.close()
        except Exception as ex:                                  # noqa: BLE001
            res[r]["crash"] = repr(ex)
            try:
                ready.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "a rank is still waiting: the containment failed"
    group.close()
    for r in range(world):
        assert "crash" not in res[r], (r, res[r])
        assert res[r]["healthy"], r
    return res


@pytest.mark.parametrize("phase", [1, 2])
def test_a_failing_rank_takes_part_and_every_rank_returns_its_status(phase):
    from soundsym_amd import _native as nat
    res = _run_ranks_with_fault(3, fault_rank=1, phase=phase, kind=0)
    for r in range(3):
        assert res[r]["code"] == nat.SSYM_E_NOMEM, (r, res[r])             # the same status on all three
        assert res[r]["seconds"] < 5.0, (r, res[r])                        # nobody waited for a deadline
        assert f"rank 1 of 3, phase {phase}" in res[r]["msg"], res[r]["msg"]
        assert not res[r]["dead"]
        assert res[r]["after"] == [True], (r, res[r])                      # the communicator is still in step
    assert "injected failure" in res[1]["msg"] and "injected failure" not in res[0]["msg"]


def test_a_failing_rank_in_a_pruned_step_and_in_a_step_with_distances():
    # the same rule through the step's other shapes: early abandoning (one more all-reduce, of the candidates' costs, in
    # front of the filter) and per-target distances (uploaded by the phase that failed: the merge still needs them)
    from soundsym_amd import _native as nat
    for kw in (dict(prune=True), dict(distance=np.linspace(0.0, 20.0, 96))):
        res = _run_ranks_with_fault(3, fault_rank=0, phase=1, kind=0, **kw)
        for r in range(3):
            assert res[r]["code"] == nat.SSYM_E_NOMEM and not res[r]["dead"], (kw, r, res[r])
            assert res[r]["after"] == [True], (kw, r, res[r])


@pytest.mark.parametrize("phase", [1, 2])
def test_a_rank_that_leaves_the_step_does_not_hang_its_peers(phase):
    from soundsym_amd import _native as nat
    res = _run_ranks_with_fault(3, fault_rank=2, phase=phase, kind=1)
    assert res[2]["code"] == nat.SSYM_E_HIP and res[2]["dead"]
    for r in (0, 1):
        assert res[r]["code"] == nat.SSYM_E_TIMEOUT, (r, res[r])
        assert res[r]["seconds"] < 20.0 and res[r]["dead"], (r, res[r])
    for r in range(3):
        assert res[r]["after"] == [nat.SSYM_E_COMM], (r, res[r])          # an aborted communicator answers SSYM_E_COMM


def test_a_rank_that_never_calls_is_met_by_the_deadline():
    from soundsym_amd import _native as nat
    res = _run_ranks_with_fault(3, fault_rank=None, phase=0, kind=0, timeout_ms=1500, absent_rank=0, steps_after=0)
    for r in (1, 2):
        assert res[r]["code"] == nat.SSYM_E_TIMEOUT and res[r]["dead"], (r, res[r])
        assert 1.0 < res[r]["seconds"] < 10.0, (r, res[r])


def test_world1_rccl_fault_injection_and_abort():
    """The same rules on a real RCCL communicator (world 1: all one GPU allows): a reported failure leaves the
    communicator usable, a rank that leaves the step aborts it (ncclCommAbort) and later calls answer SSYM_E_COMM."""
    from soundsym_amd import _native as nat
    n, m, f, dim = 200, 64, 20, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0A71)
    e = Engine(metric="dtw", dtype="f32")
    d, q = _grid_sets(e, g, f, dim)
    want = e.match(d, q)
    comm = sharding.init_comm(e, 0, 1)
    comm.set_timeout(5000)
    for phase in (1, 2):
        comm.inject_fault(phase, 0)
        with pytest.raises(nat.SsymError) as ei:
            sharding.match_sharded(e, comm, d, q, 0)
        assert ei.value.code == nat.SSYM_E_NOMEM and not comm.dead
        got = sharding.match_sharded(e, comm, d, q, 0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    comm.inject_fault(2, 1)
    with pytest.raises(nat.SsymError) as ei:
        sharding.match_sharded(e, comm, d, q, 0)
    assert ei.value.code == nat.SSYM_E_HIP and comm.dead
    with pytest.raises(nat.SsymError) as ei:
        sharding.match_sharded(e, comm, d, q, 0)
    assert ei.value.code == nat.SSYM_E_COMM
    comm.close()
    # the context survives its communicator
    got = e.match(d, q)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    e.close()


def test_test_hooks_need_the_environment_and_replayed_bounds_cut_the_rescoring(monkeypatch):
    """ssym_comm_inject_fault / ssym_comm_replay_bounds answer only to a process with SSYM_TEST_HOOKS=1; with the FULL
    dictionary's bounds replayed, a one-rank step on a shard that lacks most targets' neighbours re-scores (almost)
    nothing for them and still returns the shard's own answer (what bench.py --replay-world measures)."""
    torch = pytest.importorskip("torch")
    from soundsym_amd import _native as nat
    n, m, f, dim = 512, 256, 24, 13
    g = synth.make_grid(n, m, f, dim, 0x5EED0A73)
    e = Engine(metric="dtw", dtype="f32")
    d, q = _grid_sets(e, g, f, dim)
    comm = sharding.init_comm(e, 0, 1)
    monkeypatch.delenv("SSYM_TEST_HOOKS", raising=False)
    with pytest.raises(nat.SsymError) as ei:
        comm.inject_fault(1, 0)
    assert ei.value.code == nat.SSYM_E_UNSUPPORTED
    with pytest.raises(nat.SsymError) as ei:
        comm.replay_bounds(torch.zeros(m, dtype=torch.float64, device="cuda"))
    assert ei.value.code == nat.SSYM_E_UNSUPPORTED
    monkeypatch.setenv("SSYM_TEST_HOOKS", "1")
    # the bounds of the full dictionary = what the ranks of a run over its shards would all-reduce
    full_bounds = torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_begin(d, q, full_bounds)
    fi, fc = torch.empty(m, dtype=torch.int32, device="cuda"), torch.empty(m, dtype=torch.float64, device="cuda")
    e.match_finish(full_bounds.clone(), fi, fc)
    # a shard: the first quarter of the sources
    sf, so = g.flat("sources")
    k = n // 4
    ds = e.dictionary(sf[:k * f * dim], so[:k + 1], dim)
    own_idx, own_cost = sharding.match_sharded(e, comm, ds, q, 0)
    plain = e.timings()["n_refined"]
    comm.replay_bounds(full_bounds)
    r_idx, r_cost = sharding.match_sharded(e, comm, ds, q, 0)
    replayed = e.timings()["n_refined"]
    assert replayed < plain and replayed <= plain // 2, (plain, replayed)
    # targets whose neighbour lives in the shard keep their answer; the others report the fold start (+inf: cannot win)
    mine = g.planted < k
    assert np.array_equal(r_idx[mine], own_idx[mine]) and np.array_equal(r_cost[mine], own_cost[mine])
    assert np.all(np.isinf(r_cost[~mine]) | (r_idx[~mine] == own_idx[~mine]))
    with pytest.raises(nat.SsymError):
        comm.replay_bounds(torch.zeros(m + 1, dtype=torch.float64, device="cuda"))
        sharding.match_sharded(e, comm, ds, q, 0)
    comm.replay_bounds(None)
    comm.close()
    e.close()


def test_library_communicator_beside_a_torch_nccl_process_group():
    """The process shape of a bench.py rank at N > 1: torch.distributed's "nccl" group initialised and used first,
    then the library's communicator, then both in turn (tools/rccl_beside_torch.py, its own process because a
    process group is process-wide state).  The library must bind the RCCL that torch already loaded -- one copy of
    RCCL in the process, not two -- and its sharded step must equal the unsharded match."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_beside_torch.py")], capture_output=True,
                         text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["comm_available"] and rec["sharded_equals_match"] and rec["planted"] and rec["torch_allreduce"] == 1.0
    libs = [p for p in rec["rccl_mapped_after"] if "rccl" in p]
    assert len(libs) == 1 and libs == [p for p in rec["rccl_mapped_before"] if "rccl" in p], rec
