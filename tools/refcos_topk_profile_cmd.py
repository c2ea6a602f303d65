#!/usr/bin/env python3
"""tools/refcos_topk_profile_cmd.py <k> -- three ssym_match_topk calls (refcos, the benchmark's shape) for a kernel trace."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = m = 4096
f, d = 128, 12
g = synth.make_grid(n, m, f, d, 0x5EED0103)
e = Engine(metric="refcos", dtype="f64")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.astype(np.float64).reshape(-1) * 0.02, off, d)
q = e.queries(g.targets.astype(np.float64).reshape(-1) * 0.02, off, d)
for _ in range(3):
    e.match_topk(dd, q, k)
print(e.timings())
