#!/usr/bin/env python3
"""tools/big_shapes.py -- more than 2^32 pairs in one call (70 000 x 70 000 short segments): the index arithmetic of
every kernel on the path at sizes the 288 GB of an MI355X invite (cost matrix 19.6 GB).  Planted neighbours must come
back, dtw and refcos; not part of the default suite (it needs ~25 GB of device memory and a minute)."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine

n = m = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
f, d = 4, 13
rng = np.random.default_rng(7)
src = rng.standard_normal((n, f, d)).astype(np.float32)
perm = rng.permutation(n)[:m]
tgt = (src[perm] + 0.01 * rng.standard_normal((m, f, d))).astype(np.float32)
so = np.arange(n + 1, dtype=np.uint64) * f
to = np.arange(m + 1, dtype=np.uint64) * f
e = Engine(metric="dtw", dtype="f32")
dd, q = e.dictionary(src.reshape(-1), so, d), e.queries(tgt.reshape(-1), to, d)
t0 = time.perf_counter()
idx, cost = e.match(dd, q)
dt = time.perf_counter() - t0
tm = e.timings()
print("dtw %d x %d (%.3g pairs): %.3f s, filter %.1f ms, selection %.1f ms, re-scoring %.1f ms, planted ok %s, refined %d" % (
    n, m, float(n) * m, dt, tm["main_ms"], tm["select_ms"], tm["refine_ms"], bool(np.array_equal(idx, perm)), tm["n_refined"]),
    flush=True)
assert np.array_equal(idx, perm)
e.close()
r = Engine(metric="refcos", dtype="f64")
s64, t64 = src.astype(np.float64) * 0.3, tgt.astype(np.float64) * 0.3
dd, q = r.dictionary(s64.reshape(-1), so, d), r.queries(t64.reshape(-1), to, d)
t0 = time.perf_counter()
idx, val = r.match(dd, q)
dt = time.perf_counter() - t0
tm = r.timings()
# the planted source has the largest similarity unless another entry's |sim - 1| is smaller: check the key directly
a, b = s64.reshape(n, -1), t64.reshape(m, -1)
probe = np.arange(0, m, max(1, m // 64))
sims = (a[idx[probe]] * b[probe]).sum(1) / ((a[idx[probe]] ** 2).sum(1) * (b[probe] ** 2).sum(1))
simp = (a[perm[probe]] * b[probe]).sum(1) / ((a[perm[probe]] ** 2).sum(1) * (b[probe] ** 2).sum(1))
print("refcos %d x %d: %.2f s, main %.1f ms, through the matrix pipe %d, candidates %d, winners at least as close to 1 as the planted: %s" % (
    n, m, dt, tm["main_ms"], tm["used_filter"], tm["n_refined"], bool((np.abs(sims - 1) <= np.abs(simp - 1) + 1e-12).all())), flush=True)
assert (np.abs(sims - 1) <= np.abs(simp - 1) + 1e-12).all()
r.close()
