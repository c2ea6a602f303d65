"""ctypes loader for ``libssym_oracle.so`` plus an independent pure-Python/numpy restatement.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The numpy functions re-derive the same
reference lines (src/sound.rs:22-38, 351-370) a second time, with Python-level loops, so that the
C oracle itself can be cross-checked on small cases without trusting either implementation.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSYM_ORACLE_LIB: a differently built oracle (tools/rulinalg_variants.sh: the other association of rulinalg's dot)
_LIB = os.environ.get("SSYM_ORACLE_LIB") or os.path.join(_HERE, "libssym_oracle.so")


def _rulinalg_combine() -> int:
    """SSYM_RULINALG_COMBINE: the association of rulinalg's combine step, from include/ssym_rulinalg.h (the constant
    the C oracle and the product are compiled with); the environment variable of the same name selects the other
    one for the variant run, in step with -DSSYM_RULINALG_COMBINE on the two native builds."""
    env = os.environ.get("SSYM_RULINALG_COMBINE")
    if env is not None:
        return int(env)
    import re
    text = open(os.path.join(os.path.dirname(_HERE), "include", "ssym_rulinalg.h")).read()
    return int(re.search(r"#ifndef SSYM_RULINALG_COMBINE\s*#define SSYM_RULINALG_COMBINE (\d)", text).group(1))


RULINALG_COMBINE = _rulinalg_combine()

_f64p = ctypes.POINTER(ctypes.c_double)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the library path."""
    src = os.path.join(_HERE, "ssym_oracle.c")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "ssym_rulinalg.h")
    if os.environ.get("SSYM_ORACLE_LIB"):
        return _LIB                                   # a variant built by its own recipe
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libssym_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def pack_segments(segments: Sequence[np.ndarray], dim: int) -> Tuple[np.ndarray, np.ndarray]:
    """List of [frames_i, dim] arrays -> (flat f64 values, frame offsets u64[n+1])."""
    off = np.zeros(len(segments) + 1, dtype=np.uint64)
    for i, s in enumerate(segments):
        s = np.asarray(s)
        assert s.size == 0 or s.reshape(-1, dim).shape[1] == dim
        off[i + 1] = off[i] + np.uint64(s.size // dim)
    flat = np.zeros(int(off[-1]) * dim, dtype=np.float64)
    for i, s in enumerate(segments):
        flat[int(off[i]) * dim:int(off[i + 1]) * dim] = np.asarray(s, dtype=np.float64).reshape(-1)
    return flat, off


class Oracle:
    """Thin wrapper over the C oracle.  All arrays are float64 / uint64, C-contiguous."""

    def __init__(self, path: Optional[str] = None):
        self.lib = ctypes.CDLL(path or build())
        L = self.lib
        L.ssym_oracle_norm.restype = ctypes.c_double
        L.ssym_oracle_norm.argtypes = [_f64p, ctypes.c_size_t]
        L.ssym_oracle_rulinalg_combine.restype = ctypes.c_int
        if L.ssym_oracle_rulinalg_combine() != RULINALG_COMBINE:
            raise RuntimeError("libssym_oracle.so was built with SSYM_RULINALG_COMBINE=%d, the Python restatement uses %d"
                               % (L.ssym_oracle_rulinalg_combine(), RULINALG_COMBINE))
        L.ssym_oracle_dot.restype = ctypes.c_double
        L.ssym_oracle_dot.argtypes = [_f64p, _f64p, ctypes.c_size_t]
        L.ssym_oracle_cosine_sim.restype = ctypes.c_double
        L.ssym_oracle_cosine_sim.argtypes = [_f64p, ctypes.c_size_t, _f64p, ctypes.c_size_t]
        L.ssym_oracle_at_distance.restype = ctypes.c_int64
        L.ssym_oracle_at_distance.argtypes = [_f64p, _u64p, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_double, _f64p, ctypes.c_uint64, _f64p]
        L.ssym_oracle_refcos_match_all.restype = ctypes.c_int
        L.ssym_oracle_refcos_match_all.argtypes = [_f64p, _u64p, ctypes.c_uint32, _f64p, _u64p,
                                                   ctypes.c_uint32, ctypes.c_uint32, _f64p,
                                                   _i64p, _f64p]
        L.ssym_oracle_length_fit.restype = None
        L.ssym_oracle_refcos_matrix.argtypes = [_f64p, _u64p, ctypes.c_uint32, _f64p, _u64p, ctypes.c_uint32,
                                                ctypes.c_uint32, _f64p]
        L.ssym_oracle_refcos_matrix.restype = None
        L.ssym_oracle_topk.argtypes = [_f64p, ctypes.c_uint32, ctypes.c_uint32, _f64p, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_uint32, _i64p, _f64p]
        L.ssym_oracle_topk.restype = ctypes.c_int
        L.ssym_oracle_mfcc_num_frames.argtypes = [ctypes.c_uint64, ctypes.c_int]
        L.ssym_oracle_mfcc_num_frames.restype = ctypes.c_uint64
        L.ssym_oracle_mfcc.argtypes = [_f64p, ctypes.c_uint64, ctypes.c_double, ctypes.c_uint32, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_int, _f64p]
        L.ssym_oracle_mfcc.restype = ctypes.c_int
        L.ssym_oracle_length_fit.argtypes = [_f64p, ctypes.c_uint64, ctypes.c_uint64, _f64p]
        L.ssym_oracle_reconstruct.restype = None
        L.ssym_oracle_reconstruct.argtypes = [_f64p, _u64p, _i64p, _u64p, ctypes.c_uint32, _f64p]
        L.ssym_oracle_pcm32.restype = ctypes.c_int32
        L.ssym_oracle_pcm32.argtypes = [ctypes.c_double]
        L.ssym_oracle_dtw.restype = ctypes.c_double
        L.ssym_oracle_dtw.argtypes = [_f64p, ctypes.c_uint64, _f64p, ctypes.c_uint64,
                                      ctypes.c_uint32, ctypes.c_int64, ctypes.c_int]
        L.ssym_oracle_dtw_match_all.restype = ctypes.c_int
        L.ssym_oracle_dtw_match_all.argtypes = [_f64p, _u64p, ctypes.c_uint32, _f64p, _u64p,
                                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_int, _i64p, _f64p, _f64p]
        L.ssym_oracle_max_threads.restype = ctypes.c_int

    # -- refcos ---------------------------------------------------------------------------
    def norm(self, x) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        return self.lib.ssym_oracle_norm(_ptr(x, _f64p), x.size)

    def dot(self, x, y) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        n = min(x.size, y.size)
        return self.lib.ssym_oracle_dot(_ptr(x, _f64p), _ptr(y, _f64p), n)

    def cosine_sim(self, me, you) -> float:
        me = np.ascontiguousarray(me, dtype=np.float64).reshape(-1)
        you = np.ascontiguousarray(you, dtype=np.float64).reshape(-1)
        return self.lib.ssym_oracle_cosine_sim(_ptr(me, _f64p), me.size, _ptr(you, _f64p), you.size)

    def at_distance(self, src_flat, src_off, dim, distance, you) -> Tuple[int, float]:
        src_flat = np.ascontiguousarray(src_flat, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        you = np.ascontiguousarray(you, dtype=np.float64).reshape(-1)
        v = ctypes.c_double(0.0)
        idx = self.lib.ssym_oracle_at_distance(_ptr(src_flat, _f64p), _ptr(src_off, _u64p),
                                               src_off.size - 1, dim, float(distance),
                                               _ptr(you, _f64p), you.size // dim, ctypes.byref(v))
        return int(idx), v.value

    def refcos_match_all(self, src_flat, src_off, tgt_flat, tgt_off, dim, distance=None):
        src_flat = np.ascontiguousarray(src_flat, dtype=np.float64)
        tgt_flat = np.ascontiguousarray(tgt_flat, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        tgt_off = np.ascontiguousarray(tgt_off, dtype=np.uint64)
        m = tgt_off.size - 1
        idx = np.zeros(m, dtype=np.int64)
        val = np.zeros(m, dtype=np.float64)
        dist = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            assert dist.size == m
        rc = self.lib.ssym_oracle_refcos_match_all(
            _ptr(src_flat, _f64p), _ptr(src_off, _u64p), src_off.size - 1,
            _ptr(tgt_flat, _f64p), _ptr(tgt_off, _u64p), m, dim,
            _ptr(dist, _f64p) if dist is not None else None, _ptr(idx, _i64p), _ptr(val, _f64p))
        if rc != 0:
            raise ValueError("empty dictionary (the reference panics here, src/sound.rs:369)")
        return idx, val

    def refcos_matrix(self, src_flat, src_off, tgt_flat, tgt_off, dim) -> np.ndarray:
        """cosine_sim of every (source, target) pair, [n_src][n_tgt]."""
        src_flat = np.ascontiguousarray(src_flat, dtype=np.float64)
        tgt_flat = np.ascontiguousarray(tgt_flat, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        tgt_off = np.ascontiguousarray(tgt_off, dtype=np.uint64)
        n, m = src_off.size - 1, tgt_off.size - 1
        out = np.zeros((n, m), dtype=np.float64)
        self.lib.ssym_oracle_refcos_matrix(_ptr(src_flat, _f64p), _ptr(src_off, _u64p), n, _ptr(tgt_flat, _f64p),
                                           _ptr(tgt_off, _u64p), m, dim, _ptr(out, _f64p))
        return out

    def topk(self, values, k: int, distance=None, default_distance: float = 1.0, fold_start: float = 2.0):
        """k best sources per target from a [n_src][n_tgt] value matrix: (idx [m][k] (-1 = none),
        key [m][k] (NaN = none)), ordered by (|value - distance|, index); keys must be < fold_start."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        n, m = values.shape
        idx = np.zeros((m, k), dtype=np.int64)
        key = np.zeros((m, k), dtype=np.float64)
        dist = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            assert dist.size == m
        rc = self.lib.ssym_oracle_topk(_ptr(values, _f64p), n, m, _ptr(dist, _f64p) if dist is not None else None,
                                       float(default_distance), float(fold_start), k, _ptr(idx, _i64p),
                                       _ptr(key, _f64p))
        if rc != 0:
            raise MemoryError
        return idx, key

    def length_fit(self, matched, n_target: int) -> np.ndarray:
        matched = np.ascontiguousarray(matched, dtype=np.float64)
        out = np.empty(n_target, dtype=np.float64)
        self.lib.ssym_oracle_length_fit(_ptr(matched, _f64p), matched.size, n_target,
                                        _ptr(out, _f64p))
        return out

    def reconstruct(self, src_samples, src_off, idx, out_off) -> np.ndarray:
        src_samples = np.ascontiguousarray(src_samples, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
        out = np.empty(int(out_off[-1]), dtype=np.float64)
        self.lib.ssym_oracle_reconstruct(_ptr(src_samples, _f64p), _ptr(src_off, _u64p), _ptr(idx, _i64p),
                                         _ptr(out_off, _u64p), idx.size, _ptr(out, _f64p))
        return out

    def pcm32(self, samples) -> np.ndarray:
        return np.array([self.lib.ssym_oracle_pcm32(float(v)) for v in np.asarray(samples).reshape(-1)],
                        dtype=np.int32)

    # -- dtw ------------------------------------------------------------------------------
    def dtw(self, a, b, dim, band: int = -1, squared: bool = False) -> float:
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        return self.lib.ssym_oracle_dtw(_ptr(a, _f64p), a.size // dim, _ptr(b, _f64p),
                                        b.size // dim, dim, band, int(squared))

    def dtw_match_all(self, src_flat, src_off, tgt_flat, tgt_off, dim, band: int = -1,
                      squared: bool = False, nthreads: int = 1, want_matrix: bool = False):
        src_flat = np.ascontiguousarray(src_flat, dtype=np.float64)
        tgt_flat = np.ascontiguousarray(tgt_flat, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        tgt_off = np.ascontiguousarray(tgt_off, dtype=np.uint64)
        n, m = src_off.size - 1, tgt_off.size - 1
        idx = np.zeros(m, dtype=np.int64)
        cost = np.zeros(m, dtype=np.float64)
        mat = np.zeros((n, m), dtype=np.float64) if want_matrix else None
        rc = self.lib.ssym_oracle_dtw_match_all(
            _ptr(src_flat, _f64p), _ptr(src_off, _u64p), n, _ptr(tgt_flat, _f64p),
            _ptr(tgt_off, _u64p), m, dim, band, int(squared), nthreads, _ptr(idx, _i64p),
            _ptr(cost, _f64p), _ptr(mat, _f64p) if mat is not None else None)
        if rc != 0:
            raise ValueError("empty dictionary")
        return (idx, cost, mat) if want_matrix else (idx, cost)

    def chain(self, src_flat, src_off, dim, start, distances, metric: str = "refcos"):
        """SoundSequence::from_distances (src/sound.rs:405-417): a loop of at_distance calls, the
        match of one step being the query of the next.  Returns (idx, val) per step."""
        src_flat = np.ascontiguousarray(src_flat, dtype=np.float64)
        src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
        cur = np.ascontiguousarray(start, dtype=np.float64).reshape(-1)
        idx, val = [], []
        for d in distances:
            if metric == "refcos":
                i, v = self.at_distance(src_flat, src_off, dim, float(d), cur)
            else:
                off1 = np.array([0, cur.size // dim], dtype=np.uint64)
                _, _, mat = self.dtw_match_all(src_flat, src_off, cur, off1, dim, want_matrix=True)
                keys = np.abs(mat[:, 0] - float(d))
                i, best = 0, float("inf")
                for s in range(keys.size):                       # src/sound.rs:361-367
                    if keys[s] < best:
                        i, best = s, float(keys[s])
                v = float(mat[i, 0]) if best < float("inf") else float("inf")
            idx.append(int(i))
            val.append(v)
            cur = src_flat[int(src_off[i]) * dim:int(src_off[i + 1]) * dim]
        return np.array(idx, dtype=np.int64), np.array(val, dtype=np.float64)

    def mfcc(self, samples, rate: float, ncoeffs: int = 12, f_lo: float = 100.0, f_hi: float = 8000.0,
             pad_tail: bool = False) -> np.ndarray:
        """[frames][ncoeffs] MFCCs by the definition in ssym_oracle.c (parity unpinned, row F3)."""
        x = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1)
        t = int(self.lib.ssym_oracle_mfcc_num_frames(x.size, int(pad_tail)))
        out = np.zeros((t, ncoeffs), dtype=np.float64)
        if t:
            rc = self.lib.ssym_oracle_mfcc(_ptr(x, _f64p), x.size, float(rate), ncoeffs, float(f_lo), float(f_hi),
                                           int(pad_tail), _ptr(out, _f64p))
            if rc != 0:
                raise MemoryError
        return out

    def max_threads(self) -> int:
        return self.lib.ssym_oracle_max_threads()


_cached: Optional[Oracle] = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        _cached = Oracle()
    return _cached


# -----------------------------------------------------------------------------------------------
# Independent pure-Python restatement (small cases only).  Python floats are IEEE f64 and every
# operation below is a single rounded operation, so these follow the reference's operation order.
# -----------------------------------------------------------------------------------------------
def _np_norm(me) -> float:                      # src/sound.rs:35-38
    memo = 0.0
    for item in me:
        memo = float(item) * float(item) + memo
    return memo


def _np_dot(xs, ys, n) -> float:                # rulinalg 0.4.2 utils::dot (see ssym_oracle.c)
    p = [0.0] * 8
    i = 0
    while i + 8 <= n:
        for k in range(8):
            p[k] = p[k] + float(xs[i + k]) * float(ys[i + k])
        i += 8
    s = 0.0
    for a, b in ((0, 4), (1, 5), (2, 6), (3, 7)):
        s = s + (p[a] + p[b]) if RULINALG_COMBINE == 0 else (s + p[a]) + p[b]      # include/ssym_rulinalg.h
    while i < n:
        s = s + float(xs[i]) * float(ys[i])
        i += 1
    return s


def np_cosine_sim(me, you) -> float:            # src/sound.rs:22-33
    me = np.asarray(me, dtype=np.float64).reshape(-1)
    you = np.asarray(you, dtype=np.float64).reshape(-1)
    n = min(me.size, you.size)
    nrm = _np_norm(me) * _np_norm(you)
    dot = _np_dot(me, you, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(dot) / np.float64(nrm))


def np_at_distance(segments, distance, you) -> Tuple[int, float]:   # src/sound.rs:351-370
    if len(segments) == 0:
        raise IndexError("empty dictionary: the reference panics (src/sound.rs:369)")
    min_idx, min_distance = 0, 2.0
    for idx, s in enumerate(segments):
        v = abs(np_cosine_sim(s, you) - distance)
        if v < min_distance:
            min_idx, min_distance = idx, v
    return min_idx, min_distance


def np_dtw(a, b, band: int = -1, squared: bool = False) -> float:
    """Full-matrix restatement of the DTW definition in ssym_oracle.c (a: [Fa,d], b: [Fb,d])."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fa, fb = a.shape[0], b.shape[0]
    if fa == 0 or fb == 0:
        return float("inf")
    D = np.full((fa, fb), np.inf)
    for i in range(fa):
        for j in range(fb):
            if band >= 0 and abs(i - j) > band:
                continue
            acc = 0.0
            for k in range(a.shape[1]):
                df = float(a[i, k]) - float(b[j, k])
                acc = acc + df * df
            c = acc if squared else float(np.sqrt(np.float64(acc)))
            if i == 0 and j == 0:
                best = 0.0
            else:
                best = min(D[i - 1, j] if i > 0 else np.inf,
                           D[i, j - 1] if j > 0 else np.inf,
                           D[i - 1, j - 1] if (i > 0 and j > 0) else np.inf)
            D[i, j] = c + best
    return float(D[fa - 1, fb - 1])
