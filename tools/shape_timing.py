#!/usr/bin/env python3
"""Filter time over source / target length ranges: where short segments lose against long ones.
usage: shape_timing.py n "slo-shi:tlo-thi" ...   (inclusive frame ranges)"""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1])
e = Engine(metric="dtw", dtype="f32")
for spec in sys.argv[2:]:
    s, t = spec.split(":")
    slo, shi = map(int, s.split("-"))
    tlo, thi = map(int, t.split("-"))
    st = synth.Stream(0x5EED0B00 + shi * 1000 + thi)
    sig = synth.sigma(13)
    ls = slo + st.integers(n, shi - slo + 1)
    lt = tlo + st.integers(n, thi - tlo + 1)
    src = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in ls]
    tgt = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in lt]
    sf, so = pack_segments(src, 13, np.float32)
    tf, to = pack_segments(tgt, 13, np.float32)
    d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
    best = 1e9
    for _ in range(12):
        e.match(d, q)
        best = min(best, e.timings()["main_ms"])
    cells = float(ls.sum()) * float(lt.sum())
    print(f"src {slo}..{shi} x tgt {tlo}..{thi}: filter {best:.3f} ms, {cells / best / 1e9:.2f} T true cells/s "
          f"({cells / best / 1e9 / 9.83:.3f} of the 16-cycle cell model at 2.4 GHz)", flush=True)
