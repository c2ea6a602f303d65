#!/usr/bin/env python3
"""One-off timing of the refcos path (the reference's own metric) at 4096x4096x128fx12d, f64."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
import oracle

g = synth.make_grid(4096, 4096, 128, 12, 0x5EED0013)
sf, so = g.flat("sources", np.float64)
tf, to = g.flat("targets", np.float64)
sf *= 0.02
tf *= 0.02
e = Engine(metric="refcos", dtype="f64")
d, q = e.dictionary(sf, so, 12), e.queries(tf, to, 12)
for _ in range(3):
    t0 = time.perf_counter()
    idx, val = e.match(d, q)
    dt = time.perf_counter() - t0
    print("gpu refcos: wall %.2f ms, timings %s" % (dt * 1e3, {k: round(v, 3) if isinstance(v, float) else v for k, v in e.timings().items()}))
o = oracle.load()
n = 64
t0 = time.perf_counter()
o.refcos_match_all(sf, so, tf[: n * 128 * 12], to[: n + 1], 12)
dt = time.perf_counter() - t0
print("cpu oracle refcos (1 thread, as the reference executes it): %.3e pairs/s (%d targets x 4096 sources in %.2f s)" % (n * 4096 / dt, n, dt))
print("gpu refcos: %.3e pairs/s (kernel+fold)" % (4096 * 4096 / (e.timings()["total_ms"] * 1e-3)))
