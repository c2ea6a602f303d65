"""soundsym_amd -- MI355X-native segment-distance matching (the soundsym matcher hot path).

The compute lives in ``libsoundsym_amd.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/soundsym_amd.h``).  This package is the host side: the ctypes binding, an array-level
engine, and a mirror of the reference's Sound / SoundDictionary / SoundSequence interface.
"""
from ._native import (  # noqa: F401
    ABI_SYMBOLS,
    EmptyDictionaryError,
    LIB_PATH,
    SsymError,
    build,
)
from .engine import Engine, pack_segments  # noqa: F401
from .api import (  # noqa: F401
    BIN,
    HOP,
    NCOEFFS,
    Sound,
    SoundDictionary,
    SoundSequence,
    length_fit,
)

__all__ = [
    "ABI_SYMBOLS", "BIN", "EmptyDictionaryError", "Engine", "HOP", "LIB_PATH", "NCOEFFS", "Sound",
    "SoundDictionary", "SoundSequence", "SsymError", "build", "length_fit", "pack_segments",
]
