#!/usr/bin/env python3
"""One ragged dtw search, repeated: the command rocprofv3 profiles for the reference's real segment shape
(tools/profile_bench.sh with SSYM_PROFILE_PY=tools/ragged_profile_cmd.py).
usage: ragged_profile_cmd.py [n] [lo] [hi] [reps] [planted 0|1]"""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth
from soundsym_amd.engine import pack_segments

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 40
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
e = Engine(metric="dtw", dtype="f32")
src, tgt = synth.make_ragged(n, n, lo, hi, 13, 0x5EED0A28)      # bench.py RAGGED_SEED: secondary.ragged.dtw
sf, so = pack_segments(src, 13, np.float32)
tf, to = pack_segments(tgt, 13, np.float32)
d, q = e.dictionary(sf, so, 13), e.queries(tf, to, 13)
for _ in range(reps):
    e.match(d, q)
tm = e.timings()
cells = np.diff(so).astype(np.float64).sum() * np.diff(to).astype(np.float64).sum()
print(f"frames {lo}..{hi}: filter {tm['main_ms']:.3f} ms, total {tm['total_ms']:.3f} ms, refined {tm['n_refined']}, "
      f"{cells / tm['main_ms'] / 1e9:.2f} T true cells/s")
