#!/usr/bin/env python3
"""tools/exact_timing.py -- the exact f64 re-scoring step (ssym_timings.refine_ms) of the BASELINE shapes, for A/B runs of
the exact kernels (SSYM_EXACT_CELLS=0: the one-wave-per-pair / pipelined kernels of rounds 1-2; default: local costs
first, recurrence afterwards).  Shapes: the 8-GPU share of configs[2] (512 x 4096 x 128f x 13d, 4096 pairs re-scored
at world 1), configs[2] itself, configs[4]'s share (512 x 4096 x 256f x 40d, r = 32), short lists (early abandoning's
candidates: M pairs), and the exact kernel on every pair of a small grid."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth


def run(name, n, m, f, d, band, seed, exact_all=False, reps=5):
    g = synth.make_grid(n, m, f, d, seed)
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    e = Engine(metric="dtw", dtype="f32", band=band)
    dd, q = e.dictionary(g.sources.reshape(-1), so, d), e.queries(g.targets.reshape(-1), to, d)
    kw = dict(force_exact=True) if exact_all else {}
    e.match(dd, q, **kw)
    ref, tot, nref = [], [], 0
    for _ in range(reps):
        t0 = time.perf_counter()
        idx, cost = e.match(dd, q, **kw)
        tot.append((time.perf_counter() - t0) * 1e3)
        tm = e.timings()
        ref.append(tm["refine_ms"])
        nref = tm["n_refined"]
    ok = bool(np.array_equal(idx, g.planted))
    print(f"{name:34s} refine {np.mean(ref):8.3f} ms  ({nref} pairs)  step {np.mean(tot):8.3f} ms  planted {ok}  "
          f"checksum {float(np.sum(cost)):.12e}", flush=True)
    e.close()


which = sys.argv[1:] or ["c3share", "c3", "c5share", "short", "all"]
if "c3share" in which:
    run("configs[2] share 512x4096x128x13", 512, 4096, 128, 13, -1, 0x5EED0003)
if "c3" in which:
    run("configs[2] 4096x4096x128x13", 4096, 4096, 128, 13, -1, 0x5EED0003)
if "c5share" in which:
    run("configs[4] share 512x4096x256x40 r32", 512, 4096, 256, 40, 32, 0x5EED0005)
if "short" in which:
    run("512x512x128x13 (512 pairs)", 512, 512, 128, 13, -1, 0x5EED0031)
    run("512x512x256x40 r32 (512 pairs)", 512, 512, 256, 40, 32, 0x5EED0032)
    run("1024x1024x512x13 (1024 pairs)", 1024, 1024, 512, 13, -1, 0x5EED0033, reps=3)
if "all" in which:
    run("exact on all 256x256x128x13", 256, 256, 128, 13, -1, 0x5EED0034, exact_all=True, reps=3)
    run("exact on all 128x128x256x40 r32", 128, 128, 256, 40, 32, 0x5EED0035, exact_all=True, reps=3)
