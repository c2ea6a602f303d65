"""The oracle against hand-derived known answers and against its own second restatement.

The reference's tests hold no golden vector with numbers for this path:
these KATs are derived by hand from the reference source text and are what pins the oracle.
"""
import json
import math
import os
from fractions import Fraction

import numpy as np
import pytest

import oracle as oracle_pkg
from oracle.oracle import pack_segments

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLD, "refcos_kat.json")) as f:
        return json.load(f)


def test_cosine_sim_kats(oracle, kat):
    for case in kat["cosine_sim"]:
        want = float(Fraction(case["num"], case["den"]))
        assert oracle.cosine_sim(case["me"], case["you"]) == want, case
        assert oracle_pkg.np_cosine_sim(case["me"], case["you"]) == want, case


def test_reference_test_angular_distance_arithmetic(oracle, kat):
    # src/sound.rs:611-615: cosine_sim(m, m) must exceed 1 (squared norms) for the clamp to give 0.0
    a = kat["angular"]
    got = oracle.cosine_sim(a["m"], a["m"])
    assert got > 1.0
    assert abs(got - a["approx"]) <= a["rtol"] * a["approx"]
    clamped = 1.0 if (got > 1.0 or got < -1.0) else got      # src/sound.rs:63-67
    assert math.acos(clamped) / math.pi == 0.0


def test_at_distance_kats(oracle, kat):
    for case in kat["at_distance"]:
        dim = 1 if len(case["dict"][0]) == 1 else 2
        flat, off = pack_segments([np.array(s, dtype=np.float64) for s in case["dict"]], dim)
        idx, val = oracle.at_distance(flat, off, dim, case["distance"], np.array(case["you"], dtype=np.float64))
        assert idx == case["idx"], case
        if "val" in case:
            assert val == case["val"], case
        idx2, val2 = oracle_pkg.np_at_distance([np.array(s, dtype=np.float64) for s in case["dict"]],
                                               case["distance"], np.array(case["you"], dtype=np.float64))
        assert (idx2, val2) == (idx, val)


def test_empty_dictionary_is_an_error(oracle):
    # the reference panics at src/sound.rs:369
    idx, _ = oracle.at_distance(np.zeros(0), np.zeros(1, dtype=np.uint64), 12, 1.0, np.ones(12))
    assert idx == -1
    with pytest.raises(IndexError):
        oracle_pkg.np_at_distance([], 1.0, np.ones(12))


def test_dtw_kats(oracle, kat):
    for case in kat["dtw"]:
        a, b = np.array(case["a"], dtype=np.float64), np.array(case["b"], dtype=np.float64)
        want = float("inf") if case["cost"] == "inf" else case["cost"]
        sq = bool(case.get("squared", False))
        assert oracle.dtw(a, b, a.shape[1], case["band"], sq) == want, case
        assert oracle_pkg.np_dtw(a, b, case["band"], sq) == want, case


def test_c_oracle_equals_python_restatement_refcos(oracle):
    rng = np.random.default_rng(7)
    for _ in range(50):
        n1, n2 = rng.integers(0, 60, size=2)
        x, y = rng.normal(size=n1), rng.normal(size=n2)
        a, b = oracle.cosine_sim(x, y), oracle_pkg.np_cosine_sim(x, y)
        assert (a == b) or (math.isnan(a) and math.isnan(b))


def test_c_oracle_equals_python_restatement_dtw(oracle):
    rng = np.random.default_rng(8)
    for band in (-1, 0, 2, 5):
        for _ in range(6):
            fa, fb = rng.integers(1, 12, size=2)
            a, b = rng.normal(size=(fa, 5)), rng.normal(size=(fb, 5))
            assert oracle.dtw(a, b, 5, band) == oracle_pkg.np_dtw(a, b, band)
    assert oracle.dtw(np.zeros((0, 3)), np.ones((2, 3)), 3) == float("inf")


def test_dot_order_is_the_eight_accumulator_order(oracle):
    # a vector where the summation order changes the f64 result: the oracle must follow
    # rulinalg's order, not a left-to-right sum
    x = np.array([1e16, 1.0, -1e16, 1.0] * 5, dtype=np.float64)
    y = np.ones_like(x)
    from oracle.oracle import _np_dot
    assert oracle.dot(x, y) == _np_dot(x, y, x.size)
    assert oracle.dot(x, y) != float(np.sum(x))  # naive order differs here


def test_golden_fixtures_match_oracle(oracle):
    g = np.load(os.path.join(GOLD, "refcos_ragged.npz"))
    idx, val = oracle.refcos_match_all(g["src"], g["src_off"], g["tgt"], g["tgt_off"], 12)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(val, g["val"])
    assert g["idx"][3] == 7 or g["val"][3] <= abs(oracle.cosine_sim(
        g["src"][int(g["src_off"][7]) * 12:int(g["src_off"][8]) * 12],
        g["tgt"][int(g["tgt_off"][3]) * 12:int(g["tgt_off"][4]) * 12]) - 1.0)
    d = np.load(os.path.join(GOLD, "dtw_grid_32x32x16x13.npz"))
    # planted neighbours are recovered: expected indices known independently of any DTW code
    assert np.array_equal(d["idx"], d["planted"])
    assert np.array_equal(d["idx_sq"], d["planted"])


def test_length_fit(oracle):
    m = np.arange(1, 6, dtype=np.float64)
    assert np.array_equal(oracle.length_fit(m, 8), [1, 2, 3, 4, 5, 0, 0, 0])   # src/sound.rs:457-459
    assert np.array_equal(oracle.length_fit(m, 3), [1, 2, 3])                  # src/sound.rs:460-462
    assert np.array_equal(oracle.length_fit(m, 5), m)                          # src/sound.rs:463-464


def test_reconstruct_and_pcm32(oracle):
    # src/sound.rs:456-465 + :475-480 on three targets; :139 conversion incl. saturation and NaN
    src = np.arange(1, 11, dtype=np.float64)             # sounds: [1,2,3], [4..10]
    src_off = np.array([0, 3, 10], dtype=np.uint64)
    out = oracle.reconstruct(src, src_off, [1, 0, 0], np.array([0, 2, 7, 10], dtype=np.uint64))
    assert out.tolist() == [4, 5, 1, 2, 3, 0, 0, 1, 2, 3]
    assert oracle.pcm32([0.0, 0.5, -0.5, 1.0, 2.0, -2.0, float("nan"), 1e-10]).tolist() == [
        0, 1073741823, -1073741823, 2147483647, 2147483647, -2147483648, 0, 0]


def test_topk_first_entry_is_at_distance_and_rows_are_sorted(oracle):
    # row F1: entry 0 of the top-k is the reference's at_distance whenever something beats the
    # fold start; rows are ordered by (key, index); NaN keys and keys >= 2.0 never enter
    rng = np.random.default_rng(0x70B)
    segs = [rng.normal(size=(int(rng.integers(1, 6)), 3)) for _ in range(23)]
    segs[5] = segs[2].copy()                                     # an exact tie
    segs.append(np.full((2, 3), np.nan))                         # NaN similarity to everything
    tg = [rng.normal(size=(int(rng.integers(1, 6)), 3)) for _ in range(7)]
    from oracle.oracle import pack_segments
    sf, so = pack_segments(segs, 3)
    tf, to = pack_segments(tg, 3)
    sims = oracle.refcos_matrix(sf, so, tf, to, 3)
    dist = rng.uniform(0.0, 1.5, size=len(tg))
    idx, key = oracle.topk(sims, 6, distance=dist)
    first, val = oracle.refcos_match_all(sf, so, tf, to, 3, distance=dist)
    for t in range(len(tg)):
        keys_t = np.abs(sims[:, t] - dist[t])
        ok = np.where(keys_t < 2.0)[0]                           # NaN < 2.0 is False
        want = sorted(ok, key=lambda s: (keys_t[s], s))[:6]
        got = [int(i) for i in idx[t] if i >= 0]
        assert got == [int(w) for w in want]
        if got:
            assert got[0] == first[t] and key[t, 0] == val[t]
        assert len(segs) - 1 not in got                          # the NaN segment
    # fewer candidates than k: the tail is (-1, NaN)
    idx2, key2 = oracle.topk(sims[:2], 4)
    assert (idx2[:, 2:] == -1).all() and np.isnan(key2[:, 2:]).all()


def test_mfcc_oracle_against_a_numpy_fft_restatement(oracle):
    # row F3 (parity unpinned against the reference): the C restatement (own radix-2 FFT, explicit
    # filter loops) against the same definition written with numpy's FFT and matrix products
    rng = np.random.default_rng(0xF3)
    rate, nc = 44100.0, 12
    x = rng.normal(size=6000) * 0.1 + np.sin(2 * np.pi * 440 * np.arange(6000) / rate)
    got = oracle.mfcc(x, rate, nc)
    frames = (x.size - 1024) // 256 + 1
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)
    fr = np.stack([x[t * 256:t * 256 + 1024] * win for t in range(frames)])
    power = np.abs(np.fft.rfft(fr, axis=1)) ** 2
    nf = 2 * nc + 2
    mel = lambda f: 1127 * np.log(1 + f / 700)          # noqa: E731
    hz = lambda m: 700 * (np.exp(m / 1127) - 1)         # noqa: E731
    pts = hz(mel(100.0) + (mel(8000.0) - mel(100.0)) * np.arange(nf + 2) / (nf + 1))
    f = np.arange(513) * rate / 1024
    w = np.zeros((nf, 513))
    for m in range(nf):
        h0, h1, h2 = pts[m], pts[m + 1], pts[m + 2]
        up, dn = (f > h0) & (f <= h1), (f > h1) & (f < h2)
        w[m, up] = (f[up] - h0) / (h1 - h0)
        w[m, dn] = (h2 - f[dn]) / (h2 - h1)
    loge = np.log(np.maximum(power @ w.T, 1e-30))
    dct = np.cos(np.pi * np.arange(1, nc + 1)[:, None] * (np.arange(nf) + 0.5)[None, :] / nf)
    assert got.shape == (frames, nc)
    assert np.abs(got - loge @ dct.T).max() < 1e-9
    assert oracle.mfcc(x[:1023], rate).shape == (0, 12)
    assert oracle.mfcc(x[:1023], rate, pad_tail=True).shape == (3, 12)


def test_rows_f_golden_fixture_matches_oracle(oracle):
    # top-k / MFCC / chain fixtures of tests/golden/rows_f.npz (make_golden.py): the oracle today
    # still produces what was committed; MFCC values may move by libm ulps between machines
    g = np.load(os.path.join(GOLD, "refcos_ragged.npz"))
    f = np.load(os.path.join(GOLD, "rows_f.npz"))
    sims = oracle.refcos_matrix(g["src"], g["src_off"], g["tgt"], g["tgt_off"], 12)
    assert np.array_equal(sims, g["sims"])
    idx, key = oracle.topk(sims, 4, distance=g["dist"])
    assert np.array_equal(idx, f["top_idx"]) and np.array_equal(key, f["top_key"], equal_nan=True)
    assert np.array_equal(idx[:, 0], g["idx_d"]) and np.array_equal(key[:, 0], g["val_d"])
    m = oracle.mfcc(f["wave"], 44100.0)
    assert m.shape == f["mfcc"].shape and np.all(np.abs(m - f["mfcc"]) <= 1e-12 * (1 + np.abs(f["mfcc"])))
    ci, cv = oracle.chain(g["src"], g["src_off"], 12, f["chain_start"], f["chain_dist"])
    assert np.array_equal(ci, f["chain_idx"]) and np.array_equal(cv, f["chain_val"])


def test_rulinalg_combine_is_one_shared_constant(oracle, tmp_path):
    """include/ssym_rulinalg.h: the association of rulinalg's combine step (s + (p0 + p4) against (s + p0) + p4; which
    of the two rulinalg 0.4.2 uses could not be checked in this image).  The C oracle, the Python restatement and
    the product read the same constant; the two choices really differ (so the constant is not vacuous), and the C
    oracle built with the other value equals the Python restatement of that value."""
    import ctypes
    import subprocess
    import oracle.oracle as om
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ssym_rulinalg.h")).read()
    assert "#define SSYM_RULINALG_COMBINE 0" in hdr or "#define SSYM_RULINALG_COMBINE 1" in hdr
    assert oracle.lib.ssym_oracle_rulinalg_combine() == om.RULINALG_COMBINE
    for f in ("soundsym_amd/csrc/refcos.hip", "soundsym_amd/csrc/refcos_mfma.hip", "oracle/ssym_oracle.c"):
        text = open(os.path.join(root, f)).read()
        assert "ssym_rulinalg.h" in text and "SSYM_RULINALG_STEP" in text, f
    other = 1 - om.RULINALG_COMBINE
    out = str(tmp_path / "libssym_oracle_other.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "-B", f"EXTRA=-DSSYM_RULINALG_COMBINE={other}",
                           f"OUT={out}"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(out)
    lib.ssym_oracle_dot.restype = ctypes.c_double
    lib.ssym_oracle_dot.argtypes = [ctypes.POINTER(ctypes.c_double)] * 2 + [ctypes.c_size_t]
    assert lib.ssym_oracle_rulinalg_combine() == other

    def py_dot(x, y, combine):
        p = [0.0] * 8
        n8 = x.size // 8 * 8
        for i in range(0, n8, 8):
            for k in range(8):
                p[k] = p[k] + float(x[i + k]) * float(y[i + k])
        s = 0.0
        for a, b in ((0, 4), (1, 5), (2, 6), (3, 7)):
            s = s + (p[a] + p[b]) if combine == 0 else (s + p[a]) + p[b]
        for i in range(n8, x.size):
            s = s + float(x[i]) * float(y[i])
        return s

    rng = np.random.default_rng(7)
    differ = 0
    for _ in range(200):
        n = int(rng.integers(8, 200))
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        px, py = x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), y.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        d_this, d_other = oracle.dot(x, y), lib.ssym_oracle_dot(px, py, n)
        assert d_this == py_dot(x, y, om.RULINALG_COMBINE) and d_other == py_dot(x, y, other)
        differ += d_this != d_other
    assert differ > 20, differ          # the two associations part in the last place often enough to matter
