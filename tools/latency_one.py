#!/usr/bin/env python3
"""Latency of ssym_match_one (the reference's one-query-at-a-time call pattern)."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

src, tgt = synth.make_ragged(1024, 64, 5, 40, 12, 0x5EED0700)
for metric, dtype in (("refcos", "f64"), ("dtw", "f64")):
    e = Engine(metric=metric, dtype=dtype)
    sf, so = pack_segments([s.astype(np.float64) * 0.05 for s in src], 12)
    d = e.dictionary(sf, so, 12)
    qs = [t.astype(np.float64).reshape(-1) * 0.05 for t in tgt]
    e.match_one(d, qs[0], 1.0)
    t0 = time.perf_counter()
    for q in qs:
        e.match_one(d, q, 1.0 if metric == "refcos" else 0.0)
    dt = (time.perf_counter() - t0) / len(qs)
    print(f"{metric}: match_one on a 1024-entry dictionary: {dt * 1e6:.0f} us per call; timings {e.timings()['total_ms']:.3f} ms device")
    e.close()

# from_distances: the chain on the device (ssym_chain) against one host call per step
rng = np.random.default_rng(7)
for metric, dtype, steps in (("refcos", "f64", 512), ("dtw", "f64", 64)):
    e = Engine(metric=metric, dtype=dtype)
    sf, so = pack_segments([s.astype(np.float64) * 0.05 for s in src], 12)
    d = e.dictionary(sf, so, 12)
    dist = rng.uniform(0.2, 1.2, size=steps) if metric == "refcos" else rng.uniform(0.0, 5.0, size=steps)
    start = tgt[0].astype(np.float64).reshape(-1) * 0.05
    e.chain(d, start, dist[:4])                       # builds the self-similarity matrix (refcos)
    t0 = time.perf_counter()
    idx, _ = e.chain(d, start, dist)
    t_chain = time.perf_counter() - t0
    t0 = time.perf_counter()
    cur, loop = start, []
    for dd in dist:
        i, _ = e.match_one(d, cur, float(dd))
        loop.append(i)
        cur = sf[int(so[i]) * 12:int(so[i + 1]) * 12]
    t_loop = time.perf_counter() - t0
    assert loop == list(idx)
    print(f"{metric}: from_distances, {steps} steps on a 1024-entry dictionary: ssym_chain {t_chain / steps * 1e6:.1f} us/step, "
          f"match_one loop {t_loop / steps * 1e6:.1f} us/step")
    e.close()
