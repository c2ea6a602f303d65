"""Host-side logic: segment packing, the synthetic generator, the API containers, sharding."""
import os
import socket

import numpy as np
import pytest

from soundsym_amd import api, sharding, synth
from soundsym_amd.engine import pack_segments


def test_pack_segments_offsets_are_in_frames():
    flat, off = pack_segments([np.ones((3, 12)), np.zeros((0, 12)), np.full((2, 12), 2.0)], 12)
    assert off.tolist() == [0, 3, 3, 5] and flat.size == 60 and flat[36] == 2.0
    with pytest.raises(ValueError):
        pack_segments([np.ones(13)], 12)


def test_splitmix64_known_answers():
    # splitmix64 reference outputs for seed 0 (Vigna's published test values)
    got = synth.splitmix64(0, 3)
    assert [int(v) for v in got] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_grid_is_deterministic_and_planted():
    a = synth.make_grid(16, 8, 12, 13, 0x5EED0002)
    b = synth.make_grid(16, 8, 12, 13, 0x5EED0002)
    assert np.array_equal(a.sources, b.sources) and np.array_equal(a.targets, b.targets)
    assert a.sources.dtype == np.float32 and a.targets.shape == (8, 12, 13)
    assert len(set(a.planted.tolist())) == 8          # injection when n_tgt <= n_src
    c = synth.make_grid(16, 8, 12, 13, 0x5EED0003)
    assert not np.array_equal(a.sources, c.sources)


def test_grid_shard_rows_equal_the_full_grid():
    # what bench.py builds per rank at N > 1: only its source rows, same targets / planted indices
    for (n, m, f, d, seed) in [(40, 16, 20, 13, 0x5EED0001), (7, 9, 5, 3, 11), (33, 33, 3, 2, 5)]:
        full = synth.make_grid(n, m, f, d, seed)
        for lo, hi in [(0, n), (n // 3, 2 * n // 3), (n - 1, n)]:
            part = synth.make_grid(n, m, f, d, seed, src_range=(lo, hi))
            assert np.array_equal(part.sources, full.sources[lo:hi])
            assert np.array_equal(part.targets, full.targets)
            assert np.array_equal(part.planted, full.planted)


def test_add_segments_slices_like_the_reference():
    # src/sound.rs:330-343: seg samples and seg / HOP * NCOEFFS feature values per segment
    samples = np.arange(256 * 5, dtype=np.float64)
    mfccs = np.arange(5 * 12, dtype=np.float64)
    parent = api.Sound(samples, 44100.0, mfccs)
    d = api.SoundDictionary.new()
    d.add_segments(parent, [512, 256, 300])
    assert [s.samples().size for s in d.sounds] == [512, 256, 300]
    assert [s.num_frames() for s in d.sounds] == [2, 1, 1]      # 300 // 256 = 1 frame
    assert d.sounds[1].mfccs()[0] == 24.0 and d.sounds[2].samples()[0] == 768.0


def test_length_fit_matches_oracle(oracle):
    m = np.arange(1, 8, dtype=np.float64)
    for n in (0, 3, 7, 12):
        assert np.array_equal(api.length_fit(m, n), oracle.length_fit(m, n))


def test_sound_without_features_raises():
    s = api.Sound(np.zeros(10), 44100.0, None)
    assert not s.has_mfccs()
    with pytest.raises(ValueError):
        s.mfccs()


def test_from_samples_none_needs_the_gpu_and_says_so():
    # Sound::from_samples(.., None, ..) analyses (src/sound.rs:92-107): here on the GPU only --
    # without a device the call fails loudly, there is no CPU fallback
    from soundsym_amd import SsymError
    with pytest.raises(SsymError) as ei:
        api.Sound.from_samples(np.zeros(4096), 44100.0, None)
    assert "no CPU path" in str(ei.value)


def test_empty_dictionary_errors_like_the_reference_panics():
    d = api.SoundDictionary.new()
    with pytest.raises(api.EmptyDictionaryError):
        d.match_sound(api.Sound(np.zeros(4), 44100.0, np.ones(12)))


def test_shard_range_covers_everything_in_order():
    for n, g in ((16384, 8), (10, 3), (5, 8), (0, 2)):
        spans = [sharding.shard_range(n, g, r) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, n_src, q):
    import torch
    import torch.distributed as dist
    import oracle as oracle_pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle_pkg.load()
        g = synth.make_grid(n_src, 12, 8, 13, 0x5EED0004)
        g.sources[5] = g.sources[2]      # duplicate ACROSS the shard boundary: lowest index must win
        g.targets[0] = g.sources[2]
        lo, hi = sharding.shard_range(n_src, world, rank)
        sf, so = pack_segments(list(g.sources[lo:hi]), 13)
        tf, to = pack_segments(list(g.targets), 13)
        # per-shard results come from the oracle here: this test rehearses the sharding and
        # exchange logic on CPU, the GPU tests cover the kernels
        idx, cost = o.dtw_match_all(sf, so, tf, to, 13)
        costs, idxs = sharding.gather_candidates(torch.from_numpy(cost),
                                                 torch.from_numpy((idx + lo).astype(np.int32)))
        q.put((rank, costs.numpy().copy(), idxs.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_source_sharding_world2_gloo(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, n_src = _free_port(), 10
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, n_src, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the same G x M candidates
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    costs, idxs = got[0][1], got[0][2]
    # reference merge rule: smallest cost, lowest global index on equal cost
    order = np.lexsort((idxs, costs), axis=0)[0]
    merged = idxs[order, np.arange(costs.shape[1])]
    g = synth.make_grid(n_src, 12, 8, 13, 0x5EED0004)
    g.sources[5] = g.sources[2]
    g.targets[0] = g.sources[2]
    sf, so = pack_segments(list(g.sources), 13)
    tf, to = pack_segments(list(g.targets), 13)
    want, _ = oracle.dtw_match_all(sf, so, tf, to, 13)
    assert np.array_equal(merged, want)
    assert merged[0] == 2     # the duplicate at index 5 (other shard) must not win


def test_from_timestamps_rounds_half_away_from_zero_and_checks_the_range():
    # f64::round (src/sound.rs:422-423) is half away from zero; the slice [a, b] inclusive panics out of range (:424)
    from soundsym_amd.api import _round_half_away, Sound, SoundSequence
    assert [_round_half_away(x) for x in (0.5, 1.5, 2.5, 2.4999, -0.5, -1.5)] == [1, 2, 3, 2, -1, -2]
    assert [round(x) for x in (0.5, 1.5, 2.5)] == [0, 2, 2]            # what Python's round would have given
    s = Sound(np.arange(100, dtype=np.float64), 10.0, np.zeros(12))
    import pytest
    with pytest.raises(IndexError):
        SoundSequence.from_timestamps(s, [(0.0, 10.0, "past the end")])   # samples [0, 100] of 100
    # `as usize` saturates: a negative or NaN time is sample 0 and the call proceeds (src/sound.rs:422-424)
    from soundsym_amd.api import _round_as_usize
    assert [_round_as_usize(x) for x in (-3.7, float("nan"), 0.4, 2.5, float("inf"))] == [0, 0, 0, 3, 2 ** 64 - 1]
    s = Sound(np.arange(100, dtype=np.float64), 10.0, np.zeros(12))

    class _NoEngine:
        def mfcc(self, samples, rate, ncoeffs):
            return np.zeros((0, ncoeffs))
    seq = SoundSequence.from_timestamps(s, [(-1.0, 0.3, "negative start"), (float("nan"), 0.0, "nan start")], engine=_NoEngine())
    assert [x.samples().tolist() for x in seq.sounds()] == [[0.0, 1.0, 2.0, 3.0], [0.0]]
    with pytest.raises(IndexError):
        SoundSequence.from_timestamps(s, [(0.5, 0.1, "start beyond end + 1")], engine=_NoEngine())


def test_dictionary_residency_follows_the_content_not_the_length():
    # `sounds` is public and mutable: swapping an entry in place keeps the length but must invalidate the GPU copy
    from soundsym_amd.api import Sound, SoundDictionary
    d = SoundDictionary(engine=object())
    a, b, c = (Sound(np.zeros(4), 1.0, np.full(12, v)) for v in (1.0, 2.0, 3.0))
    d.sounds += [a, b]
    key = d._content_key()
    assert d._same(key, d._content_key())
    d.sounds[1] = c
    assert not d._same(key, d._content_key())
    d.sounds[1] = b
    d.sounds.reverse()
    assert not d._same(key, d._content_key())
    # every way of changing the list moves the key; looking at it does not; assigning a new list does
    for change in (lambda: d.sounds.append(c), lambda: d.sounds.pop(), lambda: d.sounds.insert(0, c), lambda: d.sounds.remove(c),
                   lambda: d.sounds.extend([c]), lambda: d.sounds.sort(key=id), lambda: d.sounds.__delitem__(0),
                   lambda: setattr(d, "sounds", [a, b]), lambda: d.sounds.clear()):
        key = d._content_key()
        _ = d.sounds[0] if d.sounds else None, len(d.sounds), list(d.sounds)
        assert d._same(key, d._content_key())
        change()
        assert not d._same(key, d._content_key())


def _failing_step_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = 6

        class StubEngine:                      # the C-ABI calls of one rank; rank 1's filter phase fails
            def match_begin(self, d, q_, bounds, distance=None, index_base=0):
                if rank == 1:
                    raise RuntimeError("injected: out of memory in the filter phase")
                bounds.fill_(3.0)

            def match_finish(self, bounds, out_idx, out_cost):
                out_idx.fill_(rank)
                out_cost.fill_(float(rank + 1))

            def merge_shards(self, costs, idx, out_idx, out_cost, distance=None):
                raise AssertionError("no merge after a failed step")

        bounds = torch.zeros(m, dtype=torch.float64)
        out_idx = torch.zeros(m, dtype=torch.int32)
        out_cost = torch.zeros(m, dtype=torch.float64)
        try:
            sharding.match_sharded_torch(StubEngine(), None, None, 0, out_idx, out_cost, bounds)
            q.put((rank, "no error"))
        except sharding.ShardedStepError as ex:
            q.put((rank, str(ex)))
    finally:
        dist.destroy_process_group()


def test_torch_collective_step_fails_on_every_rank_together():
    # the rehearsal path (torch.distributed collectives around the two-phase calls) follows ssym_match_sharded's rule: a
    # rank whose local phase raises still takes part in every collective, and ALL ranks raise afterwards
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "this rank's local phase failed" in got[1] and "injected" in got[1]
    assert "another rank's local phase failed" in got[0]
