#!/usr/bin/env python3
"""Selection time (bounds, lists, certificates: ssym_timings.select_ms) on the ragged grid at two sizes and on the headline
grid; SSYM_SELECT_PRETEST=0 forms every pair's key interval as before round 4 (same lists)."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

tag = os.environ.get("SSYM_SELECT_PRETEST", "1")
for n in (4096, 16384):
    e = Engine(metric="dtw", dtype="f32")
    src, tgt = synth.make_ragged(n, n, 5, 40, 13, 0x5EED0A28)
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    best = None
    for _ in range(10):
        e.match(d, q)
        tm = e.timings()
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = dict(tm)
    print("pretest", tag, "ragged 5...40", n, {k: round(float(v), 3) for k, v in best.items() if k.endswith("_ms") and v},
          int(best["n_refined"]), flush=True)
    e.close()
g = synth.make_grid(4096, 4096, 128, 13, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32")
sf, so = g.flat("sources")
tf, to = g.flat("targets")
d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
for _ in range(4):
    e.match(d, q)
    tm = e.timings()
print("pretest", tag, "headline", {k: round(float(v), 3) for k, v in tm.items() if k.endswith("_ms") and v}, int(tm["n_refined"]))
