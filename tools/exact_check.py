#!/usr/bin/env python3
"""tools/exact_check.py -- the exact kernels on every pair against the oracle, shapes around the panel / chunk edges."""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from soundsym_amd import Engine
from soundsym_amd.engine import pack_segments
o = oracle.load()
rng = np.random.default_rng(3)
bad = 0
for (fa_lo, fa_hi, fb_lo, fb_hi, dim, band) in [(1, 70, 1, 70, 13, -1), (100, 200, 250, 300, 13, -1), (60, 130, 120, 135, 12, -1),
                                                 (500, 512, 500, 512, 13, -1), (1, 300, 1, 300, 40, 32), (200, 256, 200, 256, 40, 32),
                                                 (50, 140, 50, 140, 16, 5), (64, 64, 128, 128, 14, -1), (65, 65, 129, 129, 48, 63)]:
    src = [rng.standard_normal((int(rng.integers(fa_lo, fa_hi + 1)), dim)).astype(np.float32) for _ in range(12)]
    tgt = [rng.standard_normal((int(rng.integers(fb_lo, fb_hi + 1)), dim)).astype(np.float32) for _ in range(10)]
    sf, so = pack_segments(src, dim, np.float32)
    tf, to = pack_segments(tgt, dim, np.float32)
    for dtype in ("f32", "f64"):
        e = Engine(metric="dtw", dtype=dtype, band=band)
        npd = np.float32 if dtype == "f32" else np.float64
        d, q = e.dictionary(sf.astype(npd), so, dim), e.queries(tf.astype(npd), to, dim)
        got = e.pair_matrix(d, q, exact=True)
        e.close()
        want = np.array([[o.dtw(s.astype(np.float64), t.astype(np.float64), dim, band=band) for t in tgt] for s in src])
        same = np.array_equal(got, want)
        close = np.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)
        print((fa_lo, fa_hi, fb_lo, fb_hi, dim, band), dtype, "bit-equal", same, "1e-12", close, flush=True)
        if not close:
            bad += 1
            w = np.argwhere(~np.isclose(got, want, rtol=1e-12, atol=0))[:5]
            for a, b in w:
                print("   pair", a, b, "Fa", src[a].shape[0], "Fb", tgt[b].shape[0], got[a, b], want[a, b])
sys.exit(1 if bad else 0)
