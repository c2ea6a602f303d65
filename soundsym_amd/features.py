"""Frame features for whole sounds: the MFCC front-end of SURVEY.md section 8 row F3.

`analyze_mfccs` (src/sound.rs:215-242) in the reference: 1024-sample Hanning windows hopped by 256
(src/lib.rs:24-25), 12 MFCCs between 100 Hz and 8 kHz per window (src/sound.rs:218).  The
arithmetic runs on the GPU (`ssym_mfcc`, csrc/mfcc.hip); its definition is this package's own --
the reference's lives in the un-vendored `vox_box` / `sample` crates, so PARITY IS UNPINNED.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .api import NCOEFFS, default_engine
from .engine import Engine


def frame_features(samples: np.ndarray, sample_rate: float, ncoeffs: int = NCOEFFS,
                   engine: Optional[Engine] = None, pad_tail: bool = True) -> np.ndarray:
    """[n_frames * ncoeffs] f64, frame-major.  pad_tail=True gives len(samples) // 256 frames (the
    tail windows read zeros past the end), which is what the segment arithmetic of
    SoundDictionary::add_segments (`seg / HOP * NCOEFFS` values per segment, src/sound.rs:335)
    expects of a parent sound; pad_tail=False keeps full windows only."""
    e = engine or default_engine()
    return e.mfcc(samples, sample_rate, ncoeffs, pad_tail=pad_tail).reshape(-1)
