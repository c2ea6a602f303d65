#!/usr/bin/env python3
"""tools/mfma_drop_timing.py -- the filter's time on configs[2] under whatever library SSYM_LIB names: run it with the
product library and with one built with -DSSYM_ABL_DROP_MFMA=1 (two of the three MFMAs of a tile: wrong values, valid
timing) to see what a K = 32 record layout could save at most."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 4096
f, d = 128, 13
g = synth.make_grid(n, m, f, d, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32")
off = np.arange(n + 1, dtype=np.uint64) * f
dd, q = e.dictionary(g.sources.reshape(-1), off, d), e.queries(g.targets.reshape(-1), off, d)
ms = []
for it in range(8):
    try:
        e.match(dd, q)
    except Exception as ex:            # (wrong filter values may overflow the candidate list: the timing is still valid)
        print("match raised:", type(ex).__name__)
    ms.append(e.timings()["main_ms"])
print(os.environ.get("SSYM_LIB", "product library"), "filter main_ms:", [round(x, 2) for x in ms[2:]])
