#!/usr/bin/env python3
"""tools/band_abl_timing.py -- the banded filter's time on configs[4] (256 frames x 40 values, r = 32) under whatever
library SSYM_LIB names: the product library, or one built with -DSSYM_BAND_ABL=1|2|3 (no MFMAs / no LDS operand reads /
no target loads: wrong values, valid timing) to see what each stream of the kernel costs beside the recurrence."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 4096
f, d, r = 256, 40, 32
if len(sys.argv) > 1:
    f, d, r = (int(x) for x in sys.argv[1:4])
if len(sys.argv) > 5:
    n, m = int(sys.argv[4]), int(sys.argv[5])
g = synth.make_grid(n, m, f, d, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32", band=r)
dd = e.dictionary(g.sources.reshape(-1), np.arange(n + 1, dtype=np.uint64) * f, d)
q = e.queries(g.targets.reshape(-1), np.arange(m + 1, dtype=np.uint64) * f, d)
ms = []
for it in range(6):
    try:
        e.match(dd, q)
    except Exception as ex:            # (wrong filter values may overflow the candidate list: the timing is still valid)
        print("match raised:", type(ex).__name__)
    ms.append(e.timings()["main_ms"])
print(os.environ.get("SSYM_LIB", "product library"), (n, m, f, d, r), "filter main_ms:", [round(x, 2) for x in ms[2:]])
