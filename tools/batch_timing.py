#!/usr/bin/env python3
"""tools/batch_timing.py -- ssym_match_batch (host targets: upload, pack, records, match, release -- what a drop-in caller
of clone_from_dictionary pays per batch) against ssym_match_queries on resident targets, configs[2]'s shape."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 4096
f, d = 128, 13
g = synth.make_grid(n, m, f, d, 0x5EED0003)
off = np.arange(n + 1, dtype=np.uint64) * f
e = Engine(metric="dtw", dtype="f32")
dd = e.dictionary(g.sources.reshape(-1), off, d)
tflat = np.ascontiguousarray(g.targets.reshape(-1))
q = e.queries(tflat, off, d)
for _ in range(3):
    e.match(dd, q)
t0 = time.perf_counter()
for _ in range(5):
    e.match(dd, q)
res = (time.perf_counter() - t0) / 5 * 1e3
for _ in range(2):
    idx, _ = e.match_batch(dd, tflat, off)
t0 = time.perf_counter()
for _ in range(5):
    idx, _ = e.match_batch(dd, tflat, off)
bat = (time.perf_counter() - t0) / 5 * 1e3
tm = e.timings()
print("resident targets %.2f ms per call; host targets (ssym_match_batch) %.2f ms per call: pack %.2f ms, filter %.2f ms; planted ok %s" % (
    res, bat, tm["pack_ms"], tm["main_ms"], bool(np.array_equal(idx, g.planted))))
