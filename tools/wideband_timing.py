#!/usr/bin/env python3
"""tools/wideband_timing.py -- a Sakoe-Chiba band beyond the banded kernel (r = 64): the unbanded filter as a lower bound
against the exact kernel on every pair."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n, f, dim, band = 2048, 128, 13, 64
for planted in (True, False):
    if planted:
        g = synth.make_grid(n, n, f, dim, 0x5EED0C00)
        src, tgt = g.sources, g.targets
    else:
        src = synth.make_grid(n, 1, f, dim, 0x5EED0C10).sources
        tgt = synth.make_grid(n, 1, f, dim, 0x5EED0C20).sources
    off = np.arange(n + 1, dtype=np.uint64) * f
    e = Engine(metric="dtw", dtype="f32", band=band)
    d, q = e.dictionary(src.reshape(-1), off, dim), e.queries(tgt.reshape(-1), off, dim)
    e.match(d, q)
    idx, cost = e.match(d, q)
    tm = e.timings()
    print(f"r = {band}, {'planted' if planted else 'unplanted'}: total {tm['total_ms']:.1f} ms (filter {tm['main_ms']:.1f}, select {tm['select_ms']:.1f}, "
          f"refine {tm['refine_ms']:.1f}), refined {tm['n_refined']} of {n * n}", flush=True)
    m = 128
    qs = e.queries(tgt[:m].reshape(-1), off[:m + 1], dim)
    e.match(d, qs, force_exact=True)
    i2, c2 = e.match(d, qs, force_exact=True)
    t2 = e.timings()
    print(f"   exact kernel on every pair ({n}x{m}): {t2['total_ms']:.1f} ms -> x{n // m} = {t2['total_ms'] * n / m:.0f} ms for the whole set; "
          f"same answers: {bool(np.array_equal(i2, idx[:m]) and np.array_equal(c2, cost[:m]))}")
    e.close()
