import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from soundsym_amd import Engine, synth
e = Engine(metric="dtw", dtype="f32")
for n in (64, 128, 256, 512):
    g = synth.make_grid(n, n, 128, 13, 5)
    off = np.arange(n + 1, dtype=np.uint64) * 128
    d, q = e.dictionary(g.sources.reshape(-1), off, 13), e.queries(g.targets.reshape(-1), off, 13)
    e.match(d, q, force_exact=True)
    e.match(d, q, force_exact=True)
    tm = e.timings()
    print(n * n, "pairs: refine_ms", round(tm["refine_ms"], 3), "ns/pair", round(tm["refine_ms"] * 1e6 / (n * n), 1))
