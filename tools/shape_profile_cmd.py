#!/usr/bin/env python3
"""One dtw search on uniform or ragged lengths, repeated: a command for rocprofv3 (tools/kernel_pmc.sh, tools/profile_cmd.sh).
usage: shape_profile_cmd.py n "slo-shi:tlo-thi" [reps]"""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1])
s, t = sys.argv[2].split(":")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
slo, shi = map(int, s.split("-"))
tlo, thi = map(int, t.split("-"))
st = synth.Stream(0x5EED0B00 + shi * 1000 + thi)
sig = synth.sigma(13)
ls = slo + st.integers(n, shi - slo + 1)
lt = tlo + st.integers(n, thi - tlo + 1)
src = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in ls]
tgt = [(st.normal(int(f) * 13).reshape(int(f), 13) * sig).astype(np.float32) for f in lt]
e = Engine(metric="dtw", dtype="f32")
sf, so = pack_segments(src, 13, np.float32)
tf, to = pack_segments(tgt, 13, np.float32)
d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
for _ in range(reps):
    e.match(d, q)
print(sys.argv[2], "filter %.3f ms" % e.timings()["main_ms"])
