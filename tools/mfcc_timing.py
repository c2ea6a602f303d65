#!/usr/bin/env python3
"""Throughput of ssym_mfcc (row F3) on a long signal, with the single-threaded CPU restatement beside it."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine
import oracle

rate = 44100.0
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(rate * secs)
rng = np.random.default_rng(3)
x = 0.3 * np.sin(2 * np.pi * 440 * np.arange(n) / rate) + 0.05 * rng.normal(size=n)
e = Engine(metric="refcos", dtype="f64")
e.mfcc(x[:44100], rate)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    got = e.mfcc(x, rate)
    best = min(best, time.perf_counter() - t0)
frames = got.shape[0]
print(f"gpu: {frames} frames ({secs:.0f} s of audio) in {best * 1e3:.1f} ms incl. H2D/D2H -> {frames / best:.3e} frames/s, "
      f"{secs / best:.0f}x real time")
o = oracle.load()
m = int(rate * min(secs, 20.0))
t0 = time.perf_counter()
want = o.mfcc(x[:m], rate)
dt = time.perf_counter() - t0
print(f"cpu oracle (1 thread): {want.shape[0]} frames in {dt * 1e3:.1f} ms -> {want.shape[0] / dt:.3e} frames/s")
print("max |gpu - cpu| on the common prefix:", float(np.abs(got[:want.shape[0]] - want).max()))
