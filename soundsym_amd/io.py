"""WAV and label I/O with the reference's conventions (host side only).

  read_wav        Sound::from_path        src/sound.rs:116-126  (integer PCM, mono, sample / (i32::MAX >> (32 - bits)))
  write_wav32     Sound::write_file       src/sound.rs:129-143  (32-bit integer PCM, mono)
  audacity_labels_to_timestamps           src/sound.rs:510-532  (start \\t end \\t label; bad numbers -> 0.0)
"""
from __future__ import annotations

import struct
from typing import List, Optional, Tuple

import numpy as np

I32_MAX = 2147483647


def read_wav(path: str) -> Tuple[np.ndarray, float]:
    """Integer-PCM WAV -> (f64 samples, sample rate).  Multi-channel files are read interleaved as
    hound's `samples::<i32>()` does (the reference only ever uses mono files)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("missing fmt or data chunk")
    tag, _channels, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE):
        raise ValueError("only integer PCM is supported (like the reference's use of hound)")
    if bits == 8:
        ints = np.frombuffer(pcm, dtype=np.uint8).astype(np.int64) - 128
    elif bits == 16:
        ints = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2").astype(np.int64)
    elif bits == 24:
        b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        ints = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        ints = np.where(ints >= 1 << 23, ints - (1 << 24), ints)
    elif bits == 32:
        ints = np.frombuffer(pcm[:len(pcm) // 4 * 4], dtype="<i4").astype(np.int64)
    else:
        raise ValueError(f"unsupported bit depth {bits}")
    div = float(I32_MAX >> (32 - bits))          # i32::max_value().wrapping_shr(32 - bits), :118-120
    return ints.astype(np.float64) / div, float(rate)


def pcm32(samples: np.ndarray) -> np.ndarray:
    """`(i32::max_value() as f64 * sample) as i32` (src/sound.rs:139): truncate, saturate, NaN -> 0."""
    v = np.asarray(samples, dtype=np.float64) * float(I32_MAX)
    v = np.where(np.isnan(v), 0.0, v)
    return np.trunc(np.clip(v, -2147483648.0, 2147483647.0)).astype(np.int64).clip(-2147483648, I32_MAX).astype("<i4")


def write_wav32(path: str, samples=None, sample_rate: float = 44100.0, pcm: Optional[np.ndarray] = None) -> None:
    """Mono 32-bit integer PCM (hound WavSpec of src/sound.rs:130-135).  `pcm` may carry samples
    already converted on the GPU (ssym_reconstruct's out_pcm32)."""
    q = np.asarray(pcm, dtype="<i4") if pcm is not None else pcm32(samples)
    rate = int(sample_rate)
    body = q.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, rate, rate * 4, 4, 32) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def write_wav16(path: str, pcm16, sample_rate: float = 44100.0) -> None:
    """Mono 16-bit integer PCM, the spec examples/matcher.rs writes (:22-27)."""
    q = np.asarray(pcm16, dtype="<i2")
    rate = int(sample_rate)
    body = q.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def max_power(samples) -> float:
    """analyze_max_power (src/sound.rs:244-256): the largest RMS over 128-sample rectangular windows
    hopped by 64 (full windows only, like the Windower; 0.0 when the sound is shorter than one)."""
    x = np.asarray(samples, dtype=np.float64).reshape(-1)
    if x.size < 128:
        return 0.0
    n = (x.size - 128) // 64 + 1
    idx = np.arange(n)[:, None] * 64 + np.arange(128)[None, :]
    return float(np.sqrt((x[idx] ** 2).sum(axis=1) / 128.0).max())


def audacity_labels_to_timestamps(path: str) -> List[Tuple[float, float, Optional[str]]]:
    """src/sound.rs:510-532: one Timestamp(start, end, label) per line of a tab-separated file;
    a missing or unparsable number becomes 0.0, a missing label None."""
    out = []
    with open(path, "r") as f:
        for line in f:
            if line == "":
                continue
            parts = line.strip().split("\t")

            def num(i):
                try:
                    return float(parts[i])
                except (IndexError, ValueError):
                    return 0.0
            out.append((num(0), num(1), parts[2] if len(parts) > 2 else None))
    return out
