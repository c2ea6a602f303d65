#!/usr/bin/env python3
"""Filter efficiency on ragged segment lengths (real audio segmentations are ragged)."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e = Engine(metric="dtw", dtype="f32")
for lo, hi in ((128, 129), (16, 129), (16, 257), (4, 65), (8, 41), (4, 25)):
    src, tgt = synth.make_ragged(n, n, lo, hi, 13, 0x5EED0A00 + hi)
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    e.match(d, q)
    e.match(d, q)
    tm = e.timings()
    ls = np.diff(so).astype(np.float64)
    lt = np.diff(to).astype(np.float64)
    cells = ls.sum() * lt.sum()
    print(f"frames {lo}..{hi - 1}: filter {tm['main_ms']:.2f} ms, total {tm['total_ms']:.2f} ms, refined {tm['n_refined']}, "
          f"{cells / tm['main_ms'] / 1e9:.2f} T true cells/s (uniform 128-frame reference: ~6.4)")
