#!/usr/bin/env python3
"""tools/refcos_topk_timing.py -- ssym_match_topk with the reference's metric on the benchmark's shape (4096 x 4096
segments of 128 frames x 12 values), wall clock per call: through the f64 matrix pipe (default) or, with
SSYM_REFCOS_MFMA=0 (read once per process), on the exact tile kernel over every pair."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
n = m = 4096
f, d = 128, 12
g = synth.make_grid(n, m, f, d, 0x5EED0103)
e = Engine(metric="refcos", dtype="f64")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.astype(np.float64).reshape(-1) * 0.02, off, d)
q = e.queries(g.targets.astype(np.float64).reshape(-1) * 0.02, off, d)
for k in (1, 2, 4, 8, 16, 64):
    for _ in range(2):
        (e.match_topk(dd, q, k) if k > 1 else e.match(dd, q))
    t0 = time.perf_counter()
    for _ in range(5):
        out = e.match_topk(dd, q, k) if k > 1 else e.match(dd, q)
    dt = (time.perf_counter() - t0) / 5 * 1e3
    tm = e.timings()
    print(f"k = {k:2d}: {dt:7.3f} ms per call, main kernel {tm['main_ms']:.3f} ms, through the matrix pipe: {bool(tm['used_filter'])}, "
          f"pairs keyed exactly: {tm['n_refined']}", flush=True)
