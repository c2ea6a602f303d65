#!/usr/bin/env python3
"""tools/dtw_topk_timing.py -- ssym_match_topk with the dtw metric on the headline's shape, wall clock per call and phases."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
n = m = 4096
f, d = 128, 13
g = synth.make_grid(n, m, f, d, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.reshape(-1), off, d)
q = e.queries(g.targets.reshape(-1), off, d)
for k in (1, 2, 8, 16):
    for _ in range(2):
        (e.match_topk(dd, q, k) if k > 1 else e.match(dd, q))
    t0 = time.perf_counter()
    for _ in range(3):
        out = e.match_topk(dd, q, k) if k > 1 else e.match(dd, q)
    dt = (time.perf_counter() - t0) / 3 * 1e3
    tm = e.timings()
    print(f"k = {k:2d}: {dt:7.3f} ms per call; main {tm['main_ms']:.2f} select {tm['select_ms']:.2f} refine {tm['refine_ms']:.2f} "
          f"reduce {tm['reduce_ms']:.2f} ms; pairs re-scored exactly: {tm['n_refined']}", flush=True)
