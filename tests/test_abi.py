"""The C-ABI library loads on a CPU-only box and exports everything include/soundsym_amd.h declares."""
import ctypes
import os
import re

import soundsym_amd._native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "soundsym_amd.h")).read()
    return sorted(set(re.findall(r"SSYM_API\s+[\w\s\*]+?\b(ssym_\w+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(nat.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(native_lib):
    for name in _declared():
        assert hasattr(native_lib, name), name
    assert native_lib.ssym_abi_version() == 3


def test_struct_layouts_match_header():
    # ssym_config: u32 + 5 x i32 + pointer + 2 x i32; ssym_timings: 6 floats, 2 x u64, 2 x i32, float, i32, u64, float, 3 x i32
    assert ctypes.sizeof(nat.Config) == 40
    assert ctypes.sizeof(nat.Timings) == 80


def test_no_cpu_fallback_without_a_device(native_lib):
    import torch
    if torch.cuda.device_count() > 0:
        return  # a GPU box: covered by the gpu tests
    cfg = nat.Config(ctypes.sizeof(nat.Config), 0, nat.METRIC_DTW, nat.DTYPE_F32, -1, 0, None)
    out = ctypes.c_void_p()
    rc = native_lib.ssym_ctx_create(ctypes.byref(cfg), ctypes.byref(out))
    assert rc == nat.SSYM_E_NO_DEVICE and not out.value
    assert b"no CPU path" in native_lib.ssym_last_error(None)


def test_bad_config_is_rejected(native_lib):
    cfg = nat.Config(4, 0, nat.METRIC_DTW, nat.DTYPE_F32, -1, 0, None)   # wrong struct_size
    out = ctypes.c_void_p()
    assert native_lib.ssym_ctx_create(ctypes.byref(cfg), ctypes.byref(out)) == nat.SSYM_E_INVALID


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "soundsym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "ssym_oracle" not in text, f
