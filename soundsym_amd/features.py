"""A stand-in frame feature extractor for the plumbing example -- NOT the reference's MFCC.

The reference's features come from `vox_box`'s MFCC (src/sound.rs:215-242), an un-vendored git
dependency whose arithmetic cannot be pinned here.  The matching path
only needs *some* deterministic `frames x 12` feature stream with the reference's framing:
1024-sample Hanning windows hopped by 256 (src/lib.rs:24-25, src/sound.rs:228-229).  This module
provides log band energies over 12 bands between 100 Hz and 8 kHz.  Host side, numpy only.
"""
from __future__ import annotations

import numpy as np

from .api import BIN, HOP, NCOEFFS


def frame_features(samples: np.ndarray, sample_rate: float, ncoeffs: int = NCOEFFS) -> np.ndarray:
    """[n_frames * ncoeffs] f64, frame-major, n_frames = len(samples) // HOP (zero-padded tail)."""
    x = np.asarray(samples, dtype=np.float64).reshape(-1)
    n_frames = x.size // HOP
    if n_frames == 0:
        return np.zeros(0)
    pad = np.concatenate([x, np.zeros(BIN)])
    idx = np.arange(n_frames)[:, None] * HOP + np.arange(BIN)[None, :]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(BIN) / BIN)
    spec = np.abs(np.fft.rfft(pad[idx] * win, axis=1)) ** 2
    freqs = np.fft.rfftfreq(BIN, 1.0 / sample_rate)
    edges = np.geomspace(100.0, min(8000.0, sample_rate / 2 - 1), ncoeffs + 1)
    feats = np.empty((n_frames, ncoeffs))
    for k in range(ncoeffs):
        band = (freqs >= edges[k]) & (freqs < edges[k + 1])
        feats[:, k] = np.log10(spec[:, band].sum(axis=1) + 1e-12)
    return feats.reshape(-1)
