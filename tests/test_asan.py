"""Host-side sanitizer run: tests/asan/asan_host.c -- the CPU oracle's edge cases and an
instrumented caller of the C ABI's no-device / bad-argument / struct-size paths -- under AddressSanitizer and
UndefinedBehaviorSanitizer.  CPU box only: GPU sanitizers are not available on this pool and nothing here needs a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_and_abi_no_device_paths_under_asan_ubsan():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("sanitizer runs are for the CPU box")
    except ImportError:
        pass
    d = os.path.join(ROOT, "tests", "asan")
    r = subprocess.run(["make", "-C", d, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "asan_host ok (oracle edge cases)" in r.stdout
    if os.path.exists(os.path.join(ROOT, "soundsym_amd", "libsoundsym_amd.so")):
        assert "asan_host ok (oracle edge cases + C ABI without a device)" in r.stdout
