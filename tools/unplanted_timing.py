#!/usr/bin/env python3
"""Phase timings when no target has a near-duplicate source (the per-rank view in a sharded run)."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
src = synth.make_grid(n, 1, 128, 13, 0x5EED0801).sources
tgt = synth.make_grid(n, 1, 128, 13, 0x5EED0802).sources
off = np.arange(n + 1, dtype=np.uint64) * 128
e = Engine(metric="dtw", dtype="f32")
d, q = e.dictionary(src.reshape(-1), off, 13), e.queries(tgt.reshape(-1), off, 13)
for _ in range(3):
    idx, cost = e.match(d, q)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in e.timings().items()})
