#!/usr/bin/env python3
"""tools/c2_phases.py -- phase times of BASELINE configs[1] (1024 x 1024 segments, 64 frames x 13 dims): a call that short
is a chain of ~20 launches, and the gaps between them show in total_ms."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 1024
f, d = 64, 13
g = synth.make_grid(n, m, f, d, 0x5EED0002)
e = Engine(metric="dtw", dtype="f32")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(torch.from_numpy(g.sources.reshape(-1)).cuda(), off, d)
q = e.queries(torch.from_numpy(g.targets.reshape(-1)).cuda(), off, d)
oi = torch.empty(m, dtype=torch.int32, device="cuda"); oc = torch.empty(m, dtype=torch.float64, device="cuda")
for _ in range(5):
    e.match(dd, q, out_idx=oi, out_cost=oc)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    e.match(dd, q, out_idx=oi, out_cost=oc)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e3
tm = e.timings()
print({k: round(float(v), 3) for k, v in tm.items() if k.endswith("_ms")}, "wall %.3f ms per call -> %.3g pairs/s" % (wall, n * m / wall * 1e3))
assert np.array_equal(oi.cpu().numpy(), g.planted)
