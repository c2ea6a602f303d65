#!/usr/bin/env python3
"""refcos_q8_kernel: persistent workgroups with the next tile's chunks requested before the epilogue (default) against one
workgroup per tile (SSYM_REFCOS_Q8_PERSIST=0 is read once per process: run this script twice)."""
import os, sys
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
ms = []
for _ in range(40):
    e.match(dd, q)
    ms.append(e.timings()["main_ms"])
tm = e.timings()
print("persist=%s: main kernel min %.4f median %.4f ms, filter %d, refined %d" % (
    os.environ.get("SSYM_REFCOS_Q8_PERSIST", "1"), min(ms), float(np.median(ms)), tm["refcos_filter"], tm["n_refined"]))
