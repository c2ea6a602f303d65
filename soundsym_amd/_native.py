"""ctypes binding of ``libsoundsym_amd.so`` (the C ABI of ``include/soundsym_amd.h``).

This is plumbing: argument marshalling and error translation only.  There is no fallback of any
kind -- if the shared library is missing or no gfx950 device is usable, calls raise.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSYM_LIB: tuning experiments load a differently built library (tools/); the product path is the in-tree one
LIB_PATH = os.environ.get("SSYM_LIB") or os.path.join(_HERE, "libsoundsym_amd.so")
CSRC = os.path.join(_HERE, "csrc")

SSYM_OK = 0
SSYM_E_INVALID = -1
SSYM_E_EMPTY_DICT = -2
SSYM_E_NO_DEVICE = -3
SSYM_E_HIP = -4
SSYM_E_NOMEM = -5
SSYM_E_UNSUPPORTED = -6
SSYM_E_TIMEOUT = -7
SSYM_E_COMM = -8

METRIC_REFCOS = 0
METRIC_DTW = 1
DTYPE_F64 = 0
DTYPE_F32 = 1

OUT_DEVICE = 1
DTW_FORCE_EXACT = 2
DTW_PRUNE = 4

# every symbol include/soundsym_amd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ssym_abi_version", "ssym_ctx_create", "ssym_ctx_destroy", "ssym_last_error",
    "ssym_ctx_synchronize", "ssym_get_timings", "ssym_dict_create", "ssym_dict_create_device",
    "ssym_dict_append", "ssym_dict_size", "ssym_dict_destroy", "ssym_queries_create",
    "ssym_queries_create_device", "ssym_queries_destroy", "ssym_match_queries", "ssym_match_begin",
    "ssym_match_candidates", "ssym_match_begin_pruned",
    "ssym_match_finish", "ssym_match_topk",
    "ssym_match_batch",
    "ssym_match_one", "ssym_chain", "ssym_pair_matrix", "ssym_merge_shards", "ssym_merge_shards_at", "ssym_samples_create",
    "ssym_samples_destroy", "ssym_reconstruct", "ssym_mfcc_num_frames", "ssym_mfcc",
    "ssym_comm_unique_id", "ssym_comm_create", "ssym_comm_destroy", "ssym_match_sharded",
    "ssym_local_group_create", "ssym_local_group_destroy", "ssym_comm_create_local",
    "ssym_comm_available", "ssym_comm_set_timeout", "ssym_comm_is_dead", "ssym_comm_inject_fault", "ssym_comm_replay_bounds",
]
COMM_ID_BYTES = 128        # SSYM_COMM_ID_BYTES


NO_MATCH = 0xFFFFFFFF      # SSYM_NO_MATCH
MFCC_PAD_TAIL = 4           # SSYM_MFCC_PAD_TAIL
TOPK_MAX = 64              # SSYM_TOPK_MAX


class SsymError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"soundsym_amd error {code}: {msg}")
        self.code = code


class EmptyDictionaryError(SsymError):
    """The reference panics here (src/sound.rs:369); the C ABI returns SSYM_E_EMPTY_DICT."""


class Config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("device", ctypes.c_int32),
        ("metric", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("band", ctypes.c_int32),
        ("dtw_squared", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("dtw_prune", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class Timings(ctypes.Structure):
    _fields_ = [
        ("pack_ms", ctypes.c_float),
        ("main_ms", ctypes.c_float),
        ("select_ms", ctypes.c_float),
        ("refine_ms", ctypes.c_float),
        ("reduce_ms", ctypes.c_float),
        ("total_ms", ctypes.c_float),
        ("n_pairs", ctypes.c_uint64),
        ("n_refined", ctypes.c_uint64),
        ("main_launches", ctypes.c_int32),
        ("used_filter", ctypes.c_int32),
        ("prune_ms", ctypes.c_float),
        ("pruned", ctypes.c_int32),
        ("n_filter_cells", ctypes.c_uint64),
        ("collective_ms", ctypes.c_float),
        ("attempts", ctypes.c_int32),
        ("exact_redone", ctypes.c_int32),
        ("refcos_filter", ctypes.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib: Optional[ctypes.CDLL] = None
_hip: Optional[ctypes.CDLL] = None


def hip_runtime_path() -> str:
    """The ONE HIP runtime this process will use.  libsoundsym_amd.so carries no DT_NEEDED on
    libamdhip64 (see csrc/Makefile): a PyTorch wheel bundles its own runtime, and two runtimes in
    one process cannot both open the GPU.  When torch is installed its copy is used whether or
    not torch has been imported yet, so that import order never matters; otherwise ROCm's."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    for cand in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"),
                 "/opt/rocm/lib/libamdhip64.so"):
        if os.path.exists(cand):
            return cand
    raise ImportError("no libamdhip64.so found (torch/lib or $ROCM_PATH/lib)")


_rccl: Optional[ctypes.CDLL] = None


def load_rccl() -> None:
    """Make ONE RCCL visible to the library (it binds the symbols at run time, csrc/comm.hip).  Same rule as for
    the HIP runtime: when torch is installed its bundled librccl.so is the one, imported yet or not."""
    global _rccl
    if _rccl is not None:
        return
    lib()
    import importlib.util
    cands = []
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so"))
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librccl.so.1"), "librccl.so.1"]
    for cand in cands:
        if os.path.isabs(cand) and not os.path.exists(cand):
            continue
        try:
            _rccl = ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            return
        except OSError:
            continue
    raise ImportError("no librccl.so found (torch/lib or $ROCM_PATH/lib): the sharded match needs RCCL")


def lib() -> ctypes.CDLL:
    """Load the native library.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C soundsym_amd/csrc` (there is no CPU fallback)")
    global _hip
    _hip = ctypes.CDLL(hip_runtime_path(), mode=ctypes.RTLD_GLOBAL)
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, i32, u64, f64 = (ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64,
                              ctypes.c_double)
    pvp = ctypes.POINTER(ctypes.c_void_p)
    L.ssym_abi_version.restype = i32
    L.ssym_abi_version.argtypes = []
    L.ssym_ctx_create.restype = i32
    L.ssym_ctx_create.argtypes = [ctypes.POINTER(Config), pvp]
    L.ssym_ctx_destroy.restype = i32
    L.ssym_ctx_destroy.argtypes = [vp]
    L.ssym_last_error.restype = ctypes.c_char_p
    L.ssym_last_error.argtypes = [vp]
    L.ssym_ctx_synchronize.restype = i32
    L.ssym_ctx_synchronize.argtypes = [vp]
    L.ssym_get_timings.restype = i32
    L.ssym_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    for name in ("ssym_dict_create", "ssym_dict_create_device", "ssym_queries_create",
                 "ssym_queries_create_device"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, vp, vp, u32, u32, pvp]
    L.ssym_dict_append.restype = i32
    L.ssym_dict_append.argtypes = [vp, vp, vp, vp, u32]
    L.ssym_dict_size.restype = i32
    L.ssym_dict_size.argtypes = [vp, ctypes.POINTER(u32)]
    L.ssym_dict_destroy.restype = i32
    L.ssym_dict_destroy.argtypes = [vp, vp]
    L.ssym_queries_destroy.restype = i32
    L.ssym_queries_destroy.argtypes = [vp, vp]
    L.ssym_match_queries.restype = i32
    L.ssym_match_queries.argtypes = [vp, vp, vp, vp, u32, vp, vp, u32]
    L.ssym_match_begin.restype = i32
    L.ssym_match_begin.argtypes = [vp, vp, vp, vp, u32, vp]
    L.ssym_match_candidates.restype = i32
    L.ssym_match_candidates.argtypes = [vp, vp, vp, vp]
    L.ssym_match_begin_pruned.restype = i32
    L.ssym_match_begin_pruned.argtypes = [vp, vp, vp, u32, vp, vp]
    L.ssym_match_finish.restype = i32
    L.ssym_match_finish.argtypes = [vp, vp, vp, vp, u32]
    L.ssym_match_topk.restype = i32
    L.ssym_match_topk.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, u32]
    L.ssym_match_batch.restype = i32
    L.ssym_match_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    L.ssym_match_one.restype = i32
    L.ssym_match_one.argtypes = [vp, vp, vp, u64, f64, vp, vp]
    L.ssym_chain.restype = i32
    L.ssym_chain.argtypes = [vp, vp, vp, u64, vp, u32, vp, vp]
    L.ssym_merge_shards_at.restype = i32
    L.ssym_merge_shards_at.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp]
    L.ssym_mfcc_num_frames.restype = i32
    L.ssym_mfcc_num_frames.argtypes = [u64, u32, vp]
    L.ssym_mfcc.restype = i32
    L.ssym_mfcc.argtypes = [vp, vp, u64, f64, u32, f64, f64, u32, vp, vp]
    L.ssym_pair_matrix.restype = i32
    L.ssym_pair_matrix.argtypes = [vp, vp, vp, i32, vp]
    L.ssym_merge_shards.restype = i32
    L.ssym_merge_shards.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.ssym_samples_create.restype = i32
    L.ssym_samples_create.argtypes = [vp, vp, vp, u32, pvp]
    L.ssym_samples_destroy.restype = i32
    L.ssym_samples_destroy.argtypes = [vp, vp]
    L.ssym_reconstruct.restype = i32
    L.ssym_reconstruct.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    L.ssym_comm_unique_id.restype = i32
    L.ssym_comm_unique_id.argtypes = [vp]
    L.ssym_comm_create.restype = i32
    L.ssym_comm_create.argtypes = [vp, vp, i32, i32, pvp]
    L.ssym_comm_destroy.restype = i32
    L.ssym_comm_destroy.argtypes = [vp, vp]
    L.ssym_match_sharded.restype = i32
    L.ssym_match_sharded.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, u32]
    L.ssym_local_group_create.restype = i32
    L.ssym_local_group_create.argtypes = [i32, pvp]
    L.ssym_local_group_destroy.restype = i32
    L.ssym_local_group_destroy.argtypes = [vp]
    L.ssym_comm_create_local.restype = i32
    L.ssym_comm_create_local.argtypes = [vp, vp, i32, pvp]
    L.ssym_comm_available.restype = i32
    L.ssym_comm_available.argtypes = []
    L.ssym_comm_set_timeout.restype = i32
    L.ssym_comm_set_timeout.argtypes = [vp, ctypes.c_int64]
    L.ssym_comm_is_dead.restype = i32
    L.ssym_comm_is_dead.argtypes = [vp]
    L.ssym_comm_inject_fault.restype = i32
    L.ssym_comm_inject_fault.argtypes = [vp, i32, i32]
    L.ssym_comm_replay_bounds.restype = i32
    L.ssym_comm_replay_bounds.argtypes = [vp, vp, ctypes.c_uint32]
    _lib = L
    return L


def check(rc: int, ctx: Optional[int] = None) -> None:
    if rc == SSYM_OK:
        return
    msg = lib().ssym_last_error(ctx)
    text = msg.decode("utf-8", "replace") if msg else ""
    if rc == SSYM_E_EMPTY_DICT:
        raise EmptyDictionaryError(rc, text or "empty dictionary")
    raise SsymError(rc, text)
