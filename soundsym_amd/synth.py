"""Seeded synthetic segment grids for the BASELINE.json configurations.

splitmix64 counter stream -> 24-bit uniforms -> Box-Muller normals, all in numpy integer / f64
arithmetic, so the same seed gives the same grid on every box with this image.

Sources: x[s,f,k] = sigma_k * n + slowly varying per-segment offset, sigma_k = 4/(1+k) (MFCC-like
decay).  Targets: a seeded injection pi picks a source per target; the target is that source
time-warped (<= 10 % of frames repeated / dropped, length restored) plus noise 0.05 * sigma_k, so
every target has a planted nearest neighbour with a wide margin and the expected index pi(t) is
known independently of any implementation.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int, start: int = 0) -> np.ndarray:
    """n outputs of the splitmix64 stream with the given seed, from counter `start`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Stream:
    def __init__(self, seed: int):
        self.seed, self.pos = int(seed), 0

    def u64(self, n: int) -> np.ndarray:
        out = splitmix64(self.seed, n, self.pos)
        self.pos += n
        return out

    def uniform24(self, n: int) -> np.ndarray:
        """(k + 0.5) / 2^24 with k the top 24 bits: in (0, 1), never 0."""
        return ((self.u64(n) >> np.uint64(40)).astype(np.float64) + 0.5) / 16777216.0

    def normal(self, n: int) -> np.ndarray:
        m = (n + 1) // 2
        u = self.uniform24(2 * m)
        r = np.sqrt(-2.0 * np.log(u[:m]))
        th = 2.0 * np.pi * u[m:]
        return np.concatenate([r * np.cos(th), r * np.sin(th)])[:n]

    def integers(self, n: int, hi: int) -> np.ndarray:
        return (self.u64(n) % np.uint64(hi)).astype(np.int64)

    def permutation(self, n: int) -> np.ndarray:
        return np.argsort(self.u64(n), kind="stable")


@dataclass
class Grid:
    sources: np.ndarray     # [N, F, d] float32
    targets: np.ndarray     # [M, F, d] float32
    planted: np.ndarray     # [M] expected nearest source per target
    frames: int
    dim: int

    def flat(self, which: str, dtype=np.float32):
        a = self.sources if which == "sources" else self.targets
        off = (np.arange(a.shape[0] + 1, dtype=np.uint64) * np.uint64(a.shape[1]))
        return np.ascontiguousarray(a, dtype=dtype).reshape(-1), off


def sigma(dim: int) -> np.ndarray:
    return 4.0 / (1.0 + np.arange(dim, dtype=np.float64))


def _normal_at(seed: int, pos0: int, n_total: int, idx: np.ndarray) -> np.ndarray:
    """Elements `idx` of the array Stream(seed).normal(n_total) would return when drawn at stream
    position pos0, without drawing the rest (splitmix64 is counter-based)."""
    m = (n_total + 1) // 2
    k = idx % m
    with np.errstate(over="ignore"):
        def u_at(p):
            z = np.uint64(seed) + (p.astype(np.uint64) + np.uint64(pos0 + 1)) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            return ((z >> np.uint64(40)).astype(np.float64) + 0.5) / 16777216.0
        r = np.sqrt(-2.0 * np.log(u_at(k)))
        th = 2.0 * np.pi * u_at(k + m)
    return np.where(idx < m, r * np.cos(th), r * np.sin(th))


def make_grid(n_src: int, n_tgt: int, frames: int, dim: int, seed: int,
              noise: float = 0.05, warp: float = 0.10, src_range=None) -> Grid:
    """Seed convention: 0x5EED0000 + config number.

    src_range=(lo, hi): materialise only sources [lo, hi) (what one rank of a source-sharded run
    holds; `sources` then has hi - lo rows) -- targets and planted indices are those of the full grid,
    value for value, because the generator is counter-based."""
    st = Stream(seed)
    sig = sigma(dim)
    ramp = np.linspace(-1.0, 1.0, frames).reshape(1, frames, 1)
    fd = frames * dim
    n_x = n_src * fd
    pos_lvl = 2 * ((n_x + 1) // 2)                       # stream position after the sources' normals
    n_l = n_src * dim
    pos_slope = pos_lvl + 2 * ((n_l + 1) // 2)

    def source_rows(rows: np.ndarray) -> np.ndarray:
        rows = np.asarray(rows, dtype=np.int64)
        base = _normal_at(seed, 0, n_x, (rows[:, None] * fd + np.arange(fd)[None, :]).reshape(-1))
        lvl_ = _normal_at(seed, pos_lvl, n_l, (rows[:, None] * dim + np.arange(dim)[None, :]).reshape(-1))
        slp_ = _normal_at(seed, pos_slope, n_l, (rows[:, None] * dim + np.arange(dim)[None, :]).reshape(-1))
        out = base.reshape(rows.size, frames, dim) * sig
        return out + lvl_.reshape(rows.size, 1, dim) * (0.5 * sig) + slp_.reshape(rows.size, 1, dim) * (0.25 * sig) * ramp

    if src_range is None:
        x = st.normal(n_x).reshape(n_src, frames, dim) * sig
        # slowly varying per-segment offset: a random level plus a random slope across the segment
        lvl = st.normal(n_l).reshape(n_src, 1, dim) * (0.5 * sig)
        slope = st.normal(n_l).reshape(n_src, 1, dim) * (0.25 * sig)
        x = x + lvl + slope * ramp
    else:
        st.pos = pos_slope + 2 * ((n_l + 1) // 2)        # skip what the full grid would have drawn
        x = None

    # seeded injection target -> source (a permutation prefix when n_tgt <= n_src)
    if n_tgt <= n_src:
        planted = st.permutation(n_src)[:n_tgt]
    else:
        planted = st.integers(n_tgt, n_src)

    # time warp: start from the identity frame map, repeat / drop up to `warp` of the frames,
    # keep the length at `frames`
    idx = np.tile(np.arange(frames, dtype=np.int64), (n_tgt, 1))
    n_ops = int(frames * warp / 2)
    if n_ops > 0:
        rep = st.integers(n_tgt * n_ops, frames).reshape(n_tgt, n_ops)
        drp = st.integers(n_tgt * n_ops, frames).reshape(n_tgt, n_ops)
        for t in range(n_tgt):
            keep = np.ones(frames, dtype=bool)
            keep[drp[t]] = False
            m = np.concatenate([np.arange(frames)[keep], rep[t]])
            m.sort()
            if m.size < frames:
                m = np.concatenate([m, np.full(frames - m.size, frames - 1)])
            idx[t] = m[:frames]
    if x is None:
        px = source_rows(planted)                         # the planted sources only
        y = px[np.arange(n_tgt)[:, None], idx, :]
        lo, hi = src_range
        x = source_rows(np.arange(lo, hi))
    else:
        y = x[planted[:, None], idx, :]
    y = y + st.normal(n_tgt * frames * dim).reshape(n_tgt, frames, dim) * (noise * sig)
    return Grid(x.astype(np.float32), y.astype(np.float32), planted.astype(np.int64), frames, dim)


def make_ragged(n_src: int, n_tgt: int, min_frames: int, max_frames: int, dim: int, seed: int, planted: bool = False):
    """Variable-length segments (the reference's real shape, SURVEY.md D4: add_segments gives a segment seg / HOP frames,
    src/sound.rs:330-343): lists of [f_i, d] arrays, lengths uniform in [min_frames, max_frames].

    planted=False: sources and targets are unrelated (no target has a close source: the selection's worst case).
    planted=True: target t is source pi(t) resampled to a length within two frames of the source's (nearest frame along
    a straight warping line) plus noise 0.05 * sigma_k -- a planted neighbour of a different length; returns
    (src, tgt, pi).  (Frames here are independent draws, so a resampling to a very different length would leave whole
    frames unmatched and a short unrelated source could cost less: DTW costs grow with the path.)"""
    st = Stream(seed)
    sig = sigma(dim)
    ls = min_frames + st.integers(n_src, max_frames - min_frames + 1)
    lt = min_frames + st.integers(n_tgt, max_frames - min_frames + 1)
    src = [(st.normal(int(f) * dim).reshape(int(f), dim) * sig).astype(np.float32) for f in ls]
    if not planted:
        tgt = [(st.normal(int(f) * dim).reshape(int(f), dim) * sig).astype(np.float32) for f in lt]
        return src, tgt
    pi = st.permutation(n_src)[:n_tgt] if n_tgt <= n_src else st.integers(n_tgt, n_src)
    lt = np.clip(ls[pi] + (lt % 5) - 2, min_frames, max_frames)
    tgt = []
    for t in range(n_tgt):
        a, f = src[int(pi[t])], int(lt[t])
        rows = np.rint(np.linspace(0.0, a.shape[0] - 1.0, f)).astype(np.int64)
        tgt.append((a[rows].astype(np.float64) + st.normal(f * dim).reshape(f, dim) * (0.05 * sig)).astype(np.float32))
    return src, tgt, pi.astype(np.int64)
