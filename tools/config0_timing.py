#!/usr/bin/env python3
"""BASELINE configs[0] (284 x 55 segments of the reference's recordings): where a ssym_match_batch call's time goes."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from soundsym_amd import Engine

r = Engine(metric="refcos", dtype="f64")
(sflat, soff), (tflat, toff) = bench.config0_features(r)
for metric, eng in (("refcos", r), ("dtw", Engine(metric="dtw", dtype="f64"))):
    d0 = eng.dictionary(sflat, soff, 12)
    for _ in range(5):
        eng.match_batch(d0, tflat, toff)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.match_batch(d0, tflat, toff)
    dt = (time.perf_counter() - t0) / 50
    print(metric, "%.3f ms per call" % (dt * 1e3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in eng.timings().items()})
    q = eng.queries(tflat, toff, 12)
    for _ in range(5):
        eng.match(d0, q)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.match(d0, q)
    print(metric, "resident queries: %.3f ms per call" % ((time.perf_counter() - t0) / 50 * 1e3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in eng.timings().items() if k.endswith("_ms")})
    if metric == "dtw":
        for _ in range(5):
            eng.match(d0, q, force_exact=True)
        t0 = time.perf_counter()
        for _ in range(50):
            eng.match(d0, q, force_exact=True)
        print(metric, "resident queries, exact kernel on every pair: %.3f ms per call" % ((time.perf_counter() - t0) / 50 * 1e3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in eng.timings().items() if k.endswith("_ms")})
