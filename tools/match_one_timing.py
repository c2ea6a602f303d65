#!/usr/bin/env python3
"""tools/match_one_timing.py -- ssym_match_one (the reference's own call pattern: one query at a time) per call, refcos and dtw."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments
dd = 12
rsrc, rtgt = synth.make_ragged(1024, 32, 5, 40, dd, 0x5EED0700)
for metric, dtype in (("refcos", "f64"), ("dtw", "f32")):
    e = Engine(metric=metric, dtype=dtype)
    npdt = np.float64 if dtype == "f64" else np.float32
    sf1, so1 = pack_segments([s_.astype(np.float64) * 0.05 for s_ in rsrc], dd, npdt)
    d1 = e.dictionary(sf1, so1, dd)
    qs = [t_.astype(npdt).reshape(-1) * npdt(0.05) for t_ in rtgt]
    e.match_one(d1, qs[0], 1.0 if metric == "refcos" else 0.0)
    t0 = time.perf_counter()
    for _ in range(16):
        for q1 in qs:
            e.match_one(d1, q1, 1.0 if metric == "refcos" else 0.0)
    print("%s: %.1f us per ssym_match_one call" % (metric, (time.perf_counter() - t0) / (16 * len(qs)) * 1e6))
    e.close()
