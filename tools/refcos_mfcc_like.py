#!/usr/bin/env python3
"""tools/refcos_mfcc_like.py -- the refcos search on features that look like the reference's own (Sound::mfccs(): 12 MFCCs per
frame of real audio, strongly correlated across frames and across segments), not on white noise: a long synthetic
recording -- a harmonic voice with a gliding pitch, vibrato, moving formant-like amplitude envelope and noise -- goes
through the library's MFCC front-end (ssym_mfcc), is cut into 4096 dictionary segments of 40...128 frames and 4096
targets (other stretches of the same recording), and is searched through the integer filter and through the f64 filter:
pairs keyed exactly per target, times, and that both return the same indices and values."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine
from soundsym_amd.engine import pack_segments

rate = 44100.0
rng = np.random.default_rng(0x5EED0F0)
secs = 700
n = int(rate * secs)
t = np.arange(n) / rate
f0 = 110.0 * 2.0 ** (np.cumsum(rng.standard_normal(n)) * 2e-4 % 1.5) * (1.0 + 0.01 * np.sin(2 * np.pi * 5.5 * t))
phase = 2 * np.pi * np.cumsum(f0) / rate
x = np.zeros(n)
for h in range(1, 14):
    env = 0.5 + 0.5 * np.sin(2 * np.pi * (0.13 * h + 0.07) * t + h)
    x += env / h * np.sin(h * phase)
x += 0.05 * rng.standard_normal(n)
e = Engine(metric="refcos", dtype="f64")
feats = e.mfcc(x, rate)
print("mfcc frames:", feats.shape, flush=True)
dim = feats.shape[1]
N = M = 4096
lens = rng.integers(40, 129, N + M)
starts = rng.integers(0, feats.shape[0] - 130, N + M)
segs = [feats[s:s + l].copy() for s, l in zip(starts, lens)]
src, tgt = segs[:N], segs[N:]
sf, so = pack_segments(src, dim)
tf, to = pack_segments(tgt, dim)
d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
res = {}
for knob, name in (("1", "integer filter"), ("0", "f64 filter")):
    os.environ["SSYM_REFCOS_Q8"] = knob
    e.match(d, q)
    t0 = time.perf_counter()
    for _ in range(5):
        idx, val = e.match(d, q)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    tm = e.timings()
    res[knob] = (idx, val)
    print("%-15s filter %d: main %.3f ms, tail %.3f ms, %.3f ms per call; %d pairs keyed exactly (%.2f per target)" % (
        name, tm["refcos_filter"], tm["main_ms"], tm["reduce_ms"], wall, tm["n_refined"], tm["n_refined"] / M), flush=True)
print("same indices and values:", bool(np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])))
sims = np.sort(val)
print("winning keys |sim - 1|: min %.3g, median %.3g, max %.3g" % (sims[0], sims[len(sims) // 2], sims[-1]))
