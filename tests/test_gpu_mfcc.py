"""ssym_mfcc against the oracle's restatement of the same definition.

Parity with the REFERENCE is unpinned (its MFCC lives in un-vendored crates); what is checked is
GPU == CPU oracle.  FFT, spectrum and filter sums share operation order and tables, only ln() is
each side's libm, so the tolerance is a few ulps of the log energies carried through the DCT:
|gpu - oracle| <= 1e-12 * (1 + |oracle|).
"""
import numpy as np
import pytest

from soundsym_amd import Engine

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    e = Engine(metric="refcos", dtype="f64")
    yield e
    e.close()


def _signal(n, rate, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    return 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3000.0 * t + 1.0) + 0.05 * rng.normal(size=n)


@pytest.mark.parametrize("n,rate,nc,pad", [(1024, 44100.0, 12, False), (1023, 44100.0, 12, False),
                                           (5000, 44100.0, 12, False), (5000, 44100.0, 12, True),
                                           (44100, 44100.0, 12, False), (30000, 16000.0, 13, True),
                                           (9000, 8000.0, 20, False), (255, 44100.0, 12, True)])
def test_mfcc_matches_oracle(eng, oracle, n, rate, nc, pad):
    x = _signal(n, rate, n)
    got, mean = eng.mfcc(x, rate, nc, pad_tail=pad, want_mean=True)
    want = oracle.mfcc(x, rate, nc, pad_tail=pad)
    assert got.shape == want.shape
    frames = n // 256 if pad else (0 if n < 1024 else (n - 1024) // 256 + 1)
    assert got.shape == (frames, nc)
    if frames:
        assert np.all(np.abs(got - want) <= TOL * (1.0 + np.abs(want)))
        acc = np.zeros(nc)
        for row in got:
            acc = acc + row
        assert np.array_equal(mean, acc / frames)          # analyze_mean_mfccs' fold, src/sound.rs:271-286


def test_mfcc_silence_and_band_limits(eng, oracle):
    # all-zero input: every filter sits on the 1e-30 floor, and a constant log-energy vector has a
    # vanishing DCT beyond c0, which is not kept
    z = eng.mfcc(np.zeros(4096), 44100.0)
    assert np.all(np.abs(z) < 1e-9)
    # rate below 2 * f_hi: the filterbank stops at rate / 2
    x = _signal(6000, 8000.0, 5)
    got, want = eng.mfcc(x, 8000.0), oracle.mfcc(x, 8000.0)
    assert np.all(np.abs(got - want) <= TOL * (1.0 + np.abs(want)))


def test_from_samples_none_analyses_like_the_reference(eng, oracle):
    from soundsym_amd import api
    x = _signal(8192, 44100.0, 9)
    s = api.Sound.from_samples(x, 44100.0, None, engine=eng)          # src/sound.rs:92-107 with None
    want = oracle.mfcc(x, 44100.0)
    assert s.num_frames() == want.shape[0] == (8192 - 1024) // 256 + 1
    assert np.all(np.abs(s.mfccs().reshape(want.shape) - want) <= TOL * (1.0 + np.abs(want)))
    assert np.allclose(s.mean_mfccs(), want.mean(axis=0), rtol=1e-10)


def test_mfcc_rejects_bad_arguments(eng):
    from soundsym_amd import SsymError
    with pytest.raises(SsymError):
        eng.mfcc(np.zeros(2048), 44100.0, 0)
    with pytest.raises(SsymError):
        eng.mfcc(np.zeros(2048), 44100.0, 12, f_lo=500.0, f_hi=100.0)


def test_mfcc_device_output_feeds_a_dictionary(eng, oracle):
    # SSYM_OUT_DEVICE: features stay on the GPU and go straight into ssym_dict_create_device
    import ctypes
    torch = pytest.importorskip("torch")
    from soundsym_amd import _native as nat
    x = _signal(256 * 40 + 768, 44100.0, 77)
    want = oracle.mfcc(x, 44100.0)
    frames = want.shape[0]
    out = torch.empty(frames * 12, dtype=torch.float64, device="cuda")
    mean = np.zeros(12)
    xs = np.ascontiguousarray(x)
    rc = nat.lib().ssym_mfcc(eng.ctx, xs.ctypes.data, xs.size, 44100.0, 12, 100.0, 8000.0, nat.OUT_DEVICE,
                             out.data_ptr(), mean.ctypes.data)
    nat.check(rc, eng.ctx)
    got = out.cpu().numpy().reshape(frames, 12)
    assert np.all(np.abs(got - want) <= TOL * (1.0 + np.abs(want)))
    assert np.allclose(mean, want.mean(axis=0), rtol=1e-10)
    off = np.arange(0, frames + 1, 8, dtype=np.uint64)          # 8-frame segments of the stream
    d = eng.dictionary(out, off, 12)
    q = eng.queries(got[:8 * (off.size - 1)].reshape(-1), off, 12)
    idx, _ = eng.match(d, q)
    want_idx, _ = oracle.refcos_match_all(got.reshape(-1), off, got.reshape(-1), off, 12)
    assert np.array_equal(idx, want_idx)
