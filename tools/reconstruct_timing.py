#!/usr/bin/env python3
"""Throughput of the reconstruction gather (row F2): kernel time from HIP events against HBM bytes."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine

n, seg = 4096, 128 * 256                      # 4096 sounds of 128 frames x 256 samples
rng = np.random.default_rng(2)
smp = rng.uniform(-1, 1, size=n * seg)
off = np.arange(n + 1, dtype=np.uint64) * seg
e = Engine(metric="refcos", dtype="f64")
h = e.samples(smp, off)
idx = rng.integers(0, n, size=n).astype(np.uint32)
lens = rng.integers(seg // 2, seg * 3 // 2, size=n)          # some truncated, some zero-padded
ooff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
for want_pcm in (False, True):
    e.reconstruct(h, idx, ooff, want_pcm32=want_pcm)
    e.reconstruct(h, idx, ooff, want_pcm32=want_pcm)
    ms = e.timings()["main_ms"]
    total = int(ooff[-1])
    rd = int(np.minimum(lens, seg).sum()) * 8
    wr = total * (8 + (4 if want_pcm else 0))
    print(f"reconstruct {total} samples (pcm32 {want_pcm}): kernel {ms:.3f} ms, "
          f"{(rd + wr) / ms / 1e6:.0f} GB/s of HBM traffic ({rd / 1e6:.0f} MB read + {wr / 1e6:.0f} MB written)")
