#!/usr/bin/env python3
"""tools/refcos_profile_cmd.py -- the refcos search on the benchmark's shape, a few steps: the command the refcos profile
in profiles/ is taken over (SSYM_PROFILE_PY=tools/refcos_profile_cmd.py tools/profile_bench.sh <tag> 5)."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = m = 4096
f, d = 128, 12
g = synth.make_grid(n, m, f, d, 0x5EED0103)
e = Engine(metric="refcos", dtype="f64")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.astype(np.float64).reshape(-1) * 0.02, off, d)
q = e.queries(g.targets.astype(np.float64).reshape(-1) * 0.02, off, d)
for _ in range(steps):
    idx, val = e.match(dd, q)
tm = e.timings()
print("refcos 4096x4096x128f x12d: main %.3f ms, tail %.3f ms, candidates %d" % (
    tm["main_ms"], tm["reduce_ms"], tm["n_refined"]), flush=True)
