#!/usr/bin/env python3
"""Frames wider than the filter's 42 values: the lower-bound cascade against the exact kernel on every pair."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n, f = 2048, 128
for dim in (64, 90):
    for planted in (True, False):
        if planted:
            g = synth.make_grid(n, n, f, dim, 0x5EED0B00 + dim)
            src, tgt = g.sources, g.targets
        else:
            src = synth.make_grid(n, 1, f, dim, 0x5EED0B10 + dim).sources
            tgt = synth.make_grid(n, 1, f, dim, 0x5EED0B20 + dim).sources
        off = np.arange(n + 1, dtype=np.uint64) * f
        e = Engine(metric="dtw", dtype="f32")
        d, q = e.dictionary(src.reshape(-1), off, dim), e.queries(tgt.reshape(-1), off, dim)
        e.match(d, q)
        idx, cost = e.match(d, q)
        tm = e.timings()
        line = (f"dim {dim} {'planted' if planted else 'unplanted'}: total {tm['total_ms']:.1f} ms (filter {tm['main_ms']:.1f}, "
                f"select {tm['select_ms']:.1f}, refine {tm['refine_ms']:.1f}), refined {tm['n_refined']} of {n * n}, "
                f"{n * n / tm['total_ms'] * 1e3:.3g} pairs/s")
        if planted:
            line += f", planted recovered: {bool(np.array_equal(idx, g.planted))}"
        print(line)
        e.match(d, q, prune=True)
        pi, pc = e.match(d, q, prune=True)
        tp = e.timings()
        print(f"   with early abandoning: total {tp['total_ms']:.1f} ms (thresholds {tp['prune_ms']:.1f}, filter {tp['main_ms']:.1f}, "
              f"select {tp['select_ms']:.1f}, refine {tp['refine_ms']:.1f}), same answers: "
              f"{bool(np.array_equal(pi, idx) and np.array_equal(pc, cost))}")
        e.match_topk(d, q, 3)
        ti, tc = e.match_topk(d, q, 3)
        tk = e.timings()
        print(f"   top-3: total {tk['total_ms']:.1f} ms (filter {tk['main_ms']:.1f}, select {tk['select_ms']:.1f}, refine {tk['refine_ms']:.1f}), "
              f"refined {tk['n_refined']}, best of three equals the match: {bool(np.array_equal(ti[:, 0], idx))}")
        if dim == 64 and planted:
            m = 256
            qs = e.queries(tgt[:m].reshape(-1), off[:m + 1], dim)
            e.match(d, qs, force_exact=True)
            i2, c2 = e.match(d, qs, force_exact=True)
            t2 = e.timings()
            print(f"   exact kernel on every pair ({n}x{m}): {t2['total_ms']:.1f} ms -> {n * m / t2['total_ms'] * 1e3:.3g} pairs/s; "
                  f"same answers: {bool(np.array_equal(i2, idx[:m]) and np.array_equal(c2, cost[:m]))}")
        e.close()
