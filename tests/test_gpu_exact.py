"""The exact f64 kernels on every pair against the oracle, at the shapes where their tiling has edges: the 64-row chunks
and 128-column panels of dtw_exact_cells_kernel (local costs first, recurrence afterwards: all banded shapes, and
unbanded lists of up to 4 pairs per CU), the wave-pipelined and one-wave-per-pair kernels behind it (longer lists,
SSYM_EXACT_CELLS=0), pairs whose last cell lies outside the band (+inf), frames padded to the kernels' register widths.
Costs must equal the oracle's to 1e-12 relative (they are bit-equal in practice: same operations, same order)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from soundsym_amd import Engine
from soundsym_amd.engine import pack_segments

pytestmark = pytest.mark.gpu

SHAPES = [  # (source frames lo..hi, target frames lo..hi, dim, band)
    (1, 70, 1, 70, 13, -1),            # one chunk, one panel, ragged
    (100, 200, 250, 300, 13, -1),      # 2-4 chunks x 2-3 panels (a chunk's boundary row is the next chunk's input)
    (60, 130, 120, 135, 12, -1),       # around the 64-row and 128-column edges
    (250, 256, 380, 390, 13, -1),      # four chunks, four panels
    (1, 300, 1, 300, 40, 32),          # configs[4]'s band, ragged: most pairs end outside the band
    (200, 256, 200, 256, 40, 32),
    (50, 140, 50, 140, 16, 5),         # a narrow band over chunk boundaries
    (64, 64, 128, 128, 14, -1),        # exactly one chunk, one panel
    (65, 65, 129, 129, 48, 63),        # the widest frames and the widest band the cells kernel takes
]


def _case(shape, seed):
    fa_lo, fa_hi, fb_lo, fb_hi, dim, band = shape
    rng = np.random.default_rng(seed)
    src = [rng.standard_normal((int(rng.integers(fa_lo, fa_hi + 1)), dim)).astype(np.float32) for _ in range(12)]
    tgt = [rng.standard_normal((int(rng.integers(fb_lo, fb_hi + 1)), dim)).astype(np.float32) for _ in range(10)]
    return src, tgt, dim, band


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_exact_kernels_every_pair_against_the_oracle(shape, dtype, oracle):
    src, tgt, dim, band = _case(shape, 3)
    npd = np.float32 if dtype == "f32" else np.float64
    sf, so = pack_segments(src, dim, npd)
    tf, to = pack_segments(tgt, dim, npd)
    e = Engine(metric="dtw", dtype=dtype, band=band)
    d, q = e.dictionary(sf, so, dim), e.queries(tf, to, dim)
    got = e.pair_matrix(d, q, exact=True)
    idx, cost = e.match(d, q, force_exact=True)
    e.close()
    want = np.array([[oracle.dtw(s.astype(np.float64), t.astype(np.float64), dim, band=band) for t in tgt] for s in src])
    assert np.array_equal(np.isinf(got), np.isinf(want))                     # pairs that end outside the band
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    fin = np.isfinite(want).any(axis=0)
    assert np.array_equal(idx[fin], np.argmin(want, axis=0)[fin]) and np.allclose(cost[fin], want.min(axis=0)[fin], rtol=1e-12, atol=0)


def test_both_families_of_exact_kernels_give_the_same_bits():
    # SSYM_EXACT_CELLS=0 (read once per process) keeps the one-wave-per-pair / pipelined kernels on every list: the two
    # families must agree bit for bit on the same problems
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_exact import SHAPES, _case
from soundsym_amd import Engine
from soundsym_amd.engine import pack_segments
out = []
for shape in SHAPES:
    src, tgt, dim, band = _case(shape, 5)
    sf, so = pack_segments(src, dim, np.float32); tf, to = pack_segments(tgt, dim, np.float32)
    e = Engine(metric="dtw", dtype="f32", band=band)
    out.append(e.pair_matrix(e.dictionary(sf, so, dim), e.queries(tf, to, dim), exact=True)); e.close()
np.save(sys.argv[1], np.concatenate([o.reshape(-1) for o in out]))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = []
        for knob in ("1", "0"):
            path = os.path.join(td, f"m{knob}.npy")
            env = dict(os.environ, SSYM_EXACT_CELLS=knob)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res.append(np.load(path))
    assert np.array_equal(res[0], res[1])
