#!/usr/bin/env python3
"""tools/refcos_call_overhead.py -- wall time of one refcos search call against its device time (ssym_get_timings), with
the results coming back to host arrays and with the results left on the device: what the host side of a call costs."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 4096
f, d = 128, 12
g = synth.make_grid(n, m, f, d, 0x5EED0103)
e = Engine(metric="refcos", dtype="f64")
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(g.sources.astype(np.float64).reshape(-1) * 0.02, off, d)
q = e.queries(g.targets.astype(np.float64).reshape(-1) * 0.02, off, d)
oi = torch.empty(m, dtype=torch.int32, device="cuda"); oc = torch.empty(m, dtype=torch.float64, device="cuda")
for name, kw in (("host outputs", {}), ("device outputs", {"out_idx": oi, "out_cost": oc})):
    for _ in range(5):
        e.match(dd, q, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev = []
    for _ in range(50):
        e.match(dd, q, **kw)
        dev.append(e.timings()["total_ms"])
    wall = (time.perf_counter() - t0) / 50 * 1e3
    print("%-15s wall %.3f ms per call, device %.3f ms, host side %.3f ms" % (name, wall, float(np.mean(dev)), wall - float(np.mean(dev))))
