"""ssym_chain = SoundSequence::from_distances (src/sound.rs:405-417) on the device, against the
oracle's loop of at_distance calls.  refcos: indices and values bit-exact (this also pins the claim
that the self-similarity matrix is symmetric bit for bit); dtw: indices identical, exact costs."""
import numpy as np
import pytest

from soundsym_amd import Engine, EmptyDictionaryError
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu


def _ragged(rng, n, dim, lo=1, hi=9, dtype=np.float64):
    return [rng.normal(size=(int(rng.integers(lo, hi)), dim)).astype(dtype) for _ in range(n)]


def test_refcos_chain_bit_exact(oracle):
    rng = np.random.default_rng(0xC4A1)
    segs = _ragged(rng, 300, 12)
    segs[17] = segs[5].copy()                                    # ties along the way
    segs.append(np.zeros((2, 12)))                               # zero norm: NaN similarity, never wins
    sf, so = pack_segments(segs, 12)
    start = rng.normal(size=(5, 12))
    dist = np.concatenate([[1.0], rng.uniform(0.0, 1.6, size=63)])
    e = Engine(metric="refcos", dtype="f64")
    d = e.dictionary(sf, so, 12)
    idx, val = e.chain(d, start, dist)
    want_idx, want_val = oracle.chain(sf, so, 12, start, dist)
    assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    # same chain, one host call per step (the pre-existing path): identical
    cur, loop = start.reshape(-1), []
    for dd in dist[:8]:
        i, _ = e.match_one(d, cur, float(dd))
        loop.append(i)
        cur = sf[int(so[i]) * 12:int(so[i + 1]) * 12]
    assert loop == list(idx[:8])
    # the cached self-similarity matrix follows the dictionary
    more = _ragged(rng, 20, 12)
    mf, mo = pack_segments(more, 12)
    e.dictionary_append(d, mf, mo)
    sf2, so2 = pack_segments(segs + more, 12)
    idx2, val2 = e.chain(d, start, dist)
    want2, wval2 = oracle.chain(sf2, so2, 12, start, dist)
    assert np.array_equal(idx2, want2) and np.array_equal(val2, wval2)
    e.close()


def test_dtw_chain(oracle):
    rng = np.random.default_rng(0xC4A2)
    segs = _ragged(rng, 70, 13, 2, 20, np.float32)
    sf, so = pack_segments(segs, 13, np.float32)
    start = rng.normal(size=(7, 13)).astype(np.float32)
    e = Engine(metric="dtw", dtype="f32")
    d = e.dictionary(sf, so, 13)
    _, _, mat = oracle.dtw_match_all(sf.astype(np.float64), so, sf.astype(np.float64), so, 13, want_matrix=True)
    dist = rng.uniform(0.0, float(np.median(mat)), size=10)
    dist[0] = 0.0
    idx, cost = e.chain(d, start, dist)
    want_idx, want_cost = oracle.chain(sf.astype(np.float64), so, 13, start.astype(np.float64), dist, metric="dtw")
    assert np.array_equal(idx, want_idx)
    assert np.allclose(cost, want_cost, rtol=1e-12, atol=0)
    e.close()


def test_chain_edge_cases():
    e = Engine(metric="refcos", dtype="f64")
    empty = e.dictionary(np.zeros(0), [0], 12)
    with pytest.raises(EmptyDictionaryError):                    # the reference panics, src/sound.rs:369
        e.chain(empty, np.ones(12), [1.0])
    d = e.dictionary(np.ones(24), [0, 1, 2], 12)
    idx, val = e.chain(d, np.ones(12), [])                       # no steps: nothing to do
    assert idx.size == 0 and val.size == 0
    idx, val = e.chain(d, np.ones(12), [1.0])                    # identical entries: the first wins
    assert list(idx) == [0]
    e.close()


def test_from_distances_uses_the_chain(oracle):
    from soundsym_amd import api
    rng = np.random.default_rng(0xC4A3)
    eng = Engine(metric="refcos", dtype="f64")
    sd = api.SoundDictionary(eng)
    parent = api.Sound(rng.normal(size=40 * 256), 44100.0, rng.normal(size=(40, 12)).reshape(-1))
    sd.add_segments(parent, [256 * k for k in (3, 5, 2, 7, 4, 6, 1, 8, 4)])
    start = api.Sound(rng.normal(size=1024), 44100.0, rng.normal(size=(4, 12)).reshape(-1))
    dist = rng.uniform(0.2, 1.2, size=12)
    seq = api.SoundSequence.from_distances(dist, start, sd)
    flat, off = pack_segments([s.mfccs() for s in sd.sounds], 12)
    want, _ = oracle.chain(flat, off, 12, start.mfccs(), dist)
    assert seq.sounds()[0] is start
    assert [sd.sounds.index(s) for s in seq.sounds()[1:]] == list(want)
    eng.close()
