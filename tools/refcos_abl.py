#!/usr/bin/env python3
"""tools/refcos_abl.py -- main_ms of the refcos search on the benchmark's shape under whatever library SSYM_LIB names
(ablation builds: -DSSYM_RM_NOEPI, -DSSYM_RM_NOFETCH: wrong values, valid timing -- results are not checked here)."""
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
for _ in range(8):
    try:
        e.match(dd, q)
    except Exception as ex:
        pass
    ms.append(e.timings()["main_ms"])
print(os.environ.get("SSYM_LIB", "product").split("/")[-2:], "main_ms", [round(x, 3) for x in ms[3:]])
