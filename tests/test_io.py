"""WAV / label I/O conventions of the reference (host side), and the label fixture its tests hold."""
import os
import struct

import numpy as np

from soundsym_amd import io as sio

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _write_int_wav(path, ints, bits, rate=44100):
    if bits == 16:
        body = np.asarray(ints, dtype="<i2").tobytes()
    else:  # 24
        v = np.asarray(ints, dtype=np.int64) & 0xFFFFFF
        body = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], axis=1).astype(np.uint8).tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, rate, rate * bits // 8, bits // 8, bits) + b"data" + struct.pack("<I", len(body))
    open(path, "wb").write(hdr + body)


def test_audacity_labels_to_timestamps_reference_fixture():
    # the reference's own test (src/sound.rs:546-554) on its own data file
    ts = sio.audacity_labels_to_timestamps(os.path.join(GOLD, "vowel.txt"))
    assert abs(ts[0][0] - 0.7065779155923718) < 1e-10
    assert abs(ts[26][1] - 5.59353222977394) < 1e-10
    assert ts[44][2] == "ning"
    assert len(ts) == 55


def test_read_wav_divisor_matches_hound_convention(tmp_path):
    # src/sound.rs:118-120: sample / (i32::MAX >> (32 - bits))
    p16, p24 = str(tmp_path / "a16.wav"), str(tmp_path / "a24.wav")
    _write_int_wav(p16, [0, 32767, -32768, 12345], 16)
    _write_int_wav(p24, [0, 8388607, -8388608, -5], 24)
    s16, r = sio.read_wav(p16)
    assert r == 44100.0 and np.array_equal(s16, np.array([0, 32767, -32768, 12345]) / 32767.0)
    s24, _ = sio.read_wav(p24)
    assert np.array_equal(s24, np.array([0, 8388607, -8388608, -5]) / 8388607.0)


def test_write_wav32_roundtrip_and_conversion(tmp_path, oracle):
    x = np.array([0.0, 0.5, -0.5, 1.0, 1.5, -1.5, 1e-10, -0.123456789])
    assert np.array_equal(sio.pcm32(x), oracle.pcm32(x))           # src/sound.rs:139
    p = str(tmp_path / "o.wav")
    sio.write_wav32(p, x, 22050.0)
    back, rate = sio.read_wav(p)
    assert rate == 22050.0
    assert np.array_equal(back, sio.pcm32(x).astype(np.float64) / 2147483647.0)


def test_max_power_and_wav16(tmp_path):
    # analyze_max_power (src/sound.rs:244-256): largest RMS over 128-sample windows hopped by 64
    from soundsym_amd import io as sio
    x = np.zeros(1000)
    x[300:428] = 0.5                       # exactly one window (start 320 is not a hop multiple: 256 and 320 straddle it)
    rms = [np.sqrt(np.mean(x[s:s + 128] ** 2)) for s in range(0, 1000 - 128 + 1, 64)]
    assert sio.max_power(x) == max(rms) and 0.4 < sio.max_power(x) <= 0.5
    assert sio.max_power(np.ones(100)) == 0.0          # shorter than one window: no frame at all
    q = np.array([0, 1, -1, 32767, -32768], dtype=np.int16)
    sio.write_wav16(str(tmp_path / "a.wav"), q, 8000)
    back, rate = sio.read_wav(str(tmp_path / "a.wav"))
    assert rate == 8000 and np.array_equal(back, q.astype(np.float64) / 32767.0)


def test_reference_sample_wav_known_answers():
    # the reference's own known-answer test on its own data file (test_sound_from_samples, src/lib.rs:246-260):
    #   max |sample| of tests/sample.wav = 0.6503654301602161 (1e-9)  -- pins the WAV convention of
    #   src/sound.rs:116-126 (24-bit PCM divided by i32::MAX >> 8);
    # its second constant (max_power = 0.25781895526454907) is stale upstream: analyze_max_power as written
    # today (128-sample windows hopped by 64, src/sound.rs:244-256) gives 0.3263162680772736, and the
    # constant is reproduced by 2048-sample windows only -- both stated here.
    x, rate = sio.read_wav(os.path.join(GOLD, "audio", "sample.wav"))
    assert rate == 44100.0 and x.dtype == np.float64
    assert abs(np.abs(x).max() - 0.6503654301602161) < 1e-9
    assert abs(sio.max_power(x) - 0.3263162680772736) < 1e-12
    n = (x.size - 2048) // 1024 + 1
    idx = np.arange(n)[:, None] * 1024 + np.arange(2048)[None, :]
    assert abs(float(np.sqrt((x[idx] ** 2).sum(axis=1) / 2048.0).max()) - 0.25781895526454907) < 1e-12
    y, r2 = sio.read_wav(os.path.join(GOLD, "audio", "Section_7_1.wav"))
    assert y.size > 0 and r2 > 0
    # every label of the reference's vowel.txt lies inside its recording
    ts = sio.audacity_labels_to_timestamps(os.path.join(GOLD, "vowel.txt"))
    assert 0 <= ts[0][0] and ts[-1][1] * r2 <= y.size
