#!/usr/bin/env python3
"""tools/filter_scaling.py -- the filter's time against the number of sources (4096 targets, 128 frames x 13 dims): the
fixed cost of a launch (ramp-up, the tail of a persistent grid whose tasks are 64 pairs each) is what bounds the
strong-scaling efficiency of the source-sharded step, where every rank runs N / G sources."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

m, f, d = 4096, 128, 13
fs = int(sys.argv[1]) if len(sys.argv) > 1 else f      # source frames (64: single-pass tasks of half the length)
g = synth.make_grid(4096, m, f, d, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32")
to = np.arange(m + 1, dtype=np.uint64) * f
q = e.queries(torch.from_numpy(g.targets.reshape(-1)).cuda(), to, d)
oi = torch.empty(m, dtype=torch.int32, device="cuda"); oc = torch.empty(m, dtype=torch.float64, device="cuda")
rows = []
for n in (128, 256, 512, 1024, 2048, 4096):
    so = np.arange(n + 1, dtype=np.uint64) * fs
    dd = e.dictionary(torch.from_numpy(np.ascontiguousarray(g.sources[:n, :fs]).reshape(-1)).cuda(), so, d)
    ms = []
    for it in range(8):
        e.match(dd, q, out_idx=oi, out_cost=oc)
        if it >= 2:
            ms.append(e.timings()["main_ms"])
    rows.append((n, float(np.mean(ms)), float(np.min(ms))))
    dd.close()
(n1, t1, _), (n2, t2, _) = rows[2], rows[-1]
b = (t2 - t1) / (n2 - n1)
for n, t, tmin in rows:
    print("%5d sources: filter %.3f ms (min %.3f)  = %.3f ms fixed + %.4f ms per source; per-pair rate %.3g pairs/s" % (
        n, t, tmin, t - b * n, b, n * m / t * 1e3))
