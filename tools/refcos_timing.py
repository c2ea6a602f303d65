#!/usr/bin/env python3
"""tools/refcos_timing.py -- the reference's own metric (refcos_sims_kernel + fold) on the benchmark's shape and on ragged segments."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = m = 4096
f, d = 128, 12
g = synth.make_grid(n, m, f, d, 0x5EED0103)
e = Engine(metric="refcos", dtype="f64")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.astype(np.float64).reshape(-1) * 0.02, off, d)
q = e.queries(g.targets.astype(np.float64).reshape(-1) * 0.02, off, d)
for _ in range(3):
    e.match(dd, q)
tm = e.timings()
print("4096x4096x128f x12d refcos: sims %.3f ms, total %.3f ms (%.3g pairs/s); list 1 / list 2: %d / %d pairs" % (
    tm["main_ms"], tm["total_ms"], n * m / tm["total_ms"] * 1e3, tm.get("n_candidates", -1), tm.get("n_refined", -1)), flush=True)
src, tgt = synth.make_ragged(2048, 2048, 4, 160, d, 0x5EED0A77)
sf2, so2 = pack_segments([x * 0.02 for x in src], d, np.float64)
tf2, to2 = pack_segments([x * 0.02 for x in tgt], d, np.float64)
d2, q2 = e.dictionary(sf2, so2, d), e.queries(tf2, to2, d)
for _ in range(3):
    e.match(d2, q2)
print("2048x2048 ragged 4..160 frames: sims %.3f ms" % e.timings()["main_ms"])
