#!/usr/bin/env python3
"""tools/refcos_matrix_timing.py -- the bit-exact similarity matrix (ssym_pair_matrix, refcos) on the benchmark's shape and
on ragged segments: the exact tile kernel of csrc/refcos.hip (refcos_sims8_kernel, eight lanes per pair).  Device time
is the match's main_ms with the matrix pipe switched off, which runs the same kernel on every pair."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
os.environ["SSYM_REFCOS_MFMA"] = "0"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments
e = Engine(metric="refcos", dtype="f64")
for name, n, m, fmin, fmax, d in (("4096 x 4096 x 128f x 12d", 4096, 4096, 128, 128, 12), ("2048 x 2048 ragged 4..160f x 12d", 2048, 2048, 4, 160, 12),
                                 ("1024 x 1024 ragged 1..40f x 13d", 1024, 1024, 1, 40, 13)):
    src, tgt = synth.make_ragged(n, m, fmin, fmax, d, 0x5EED0A90)
    sf, so = pack_segments([x * 0.02 for x in src], d, np.float64)
    tf, to = pack_segments([x * 0.02 for x in tgt], d, np.float64)
    dd, q = e.dictionary(sf, so, d), e.queries(tf, to, d)
    ms = []
    for _ in range(5):
        idx, val = e.match(dd, q)
        ms.append(e.timings()["main_ms"])
    print(f"{name}: tile kernel {np.mean(ms[2:]):.3f} ms, checksum {float(np.nansum(val)):.15e} {int(idx.sum())}", flush=True)
