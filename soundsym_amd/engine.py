"""Array-level host interface over the C ABI: contexts, resident dictionaries / query sets, match.

Everything here forwards to ``libsoundsym_amd.so``; no arithmetic of the matching path happens in
Python.  Feature buffers may be numpy arrays (host) or torch CUDA tensors (device, handed over as
raw pointers -- torch is only the owner of the memory).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat


def pack_segments(segments: Sequence[np.ndarray], dim: int, dtype=np.float64):
    """List of per-segment arrays (frames_i x dim, or flat) -> (flat values, frame offsets)."""
    off = np.zeros(len(segments) + 1, dtype=np.uint64)
    for i, s in enumerate(segments):
        n = int(np.asarray(s).size)
        if n % dim:
            raise ValueError(f"segment {i}: {n} values is not a whole number of {dim}-dim frames")
        off[i + 1] = off[i] + np.uint64(n // dim)
    flat = np.zeros(int(off[-1]) * dim, dtype=dtype)
    for i, s in enumerate(segments):
        flat[int(off[i]) * dim:int(off[i + 1]) * dim] = np.asarray(s, dtype=dtype).reshape(-1)
    return flat, off


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


class _Handle:
    def __init__(self, engine: "Engine", ptr: int, n: int, dim: int, kind: str):
        self.engine, self.ptr, self.n, self.dim, self.kind = engine, ptr, n, dim, kind

    def close(self):
        if self.ptr and self.engine.ctx:
            L = nat.lib()
            (L.ssym_dict_destroy if self.kind == "dict" else L.ssym_queries_destroy)(
                self.engine.ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _SamplesHandle:
    def __init__(self, engine, ptr, n):
        self.engine, self.ptr, self.n = engine, ptr, n

    def close(self):
        if self.ptr and self.engine.ctx:
            nat.lib().ssym_samples_destroy(self.engine.ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """One rank of a source-sharded run: an RCCL communicator bound to an Engine (ssym_comm)."""

    def __init__(self, engine: "Engine", ptr: int, rank: int, world: int):
        self.engine, self.ptr, self.rank, self.world = engine, ptr, rank, world

    def set_timeout(self, milliseconds: int) -> None:
        """ssym_comm_set_timeout: the deadline of one match_sharded step (a rank that never arrives)."""
        nat.check(nat.lib().ssym_comm_set_timeout(self.ptr, int(milliseconds)), self.engine.ctx)

    @property
    def dead(self) -> bool:
        """ssym_comm_is_dead: aborted by a failure; every further step raises SSYM_E_COMM."""
        return bool(self.ptr) and nat.lib().ssym_comm_is_dead(self.ptr) == 1

    def inject_fault(self, phase: int, kind: int) -> None:
        """ssym_comm_inject_fault (containment tests): the next step fails in `phase`; kind 0 = the local work
        reports an error and the rank takes part, kind 1 = the rank leaves the step without its collectives."""
        nat.check(nat.lib().ssym_comm_inject_fault(self.ptr, phase, kind), self.engine.ctx)

    def replay_bounds(self, bounds) -> None:
        """ssym_comm_replay_bounds (measurement hook, needs SSYM_TEST_HOOKS=1): a device tensor of per-target bounds
        (f64, one per target) that joins every following step's bound exchange; None clears.  The tensor is kept alive."""
        self._replay = bounds
        if bounds is None:
            nat.check(nat.lib().ssym_comm_replay_bounds(self.ptr, None, 0), self.engine.ctx)
        else:
            nat.check(nat.lib().ssym_comm_replay_bounds(self.ptr, bounds.data_ptr(), int(bounds.numel())), self.engine.ctx)

    def close(self):
        if self.ptr and self.engine.ctx:
            nat.lib().ssym_comm_destroy(self.engine.ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_available() -> bool:
    """ssym_comm_available: the library itself could bind every RCCL symbol it calls (what the ranks of a job agree
    on before any of them enters comm_create)."""
    try:
        nat.load_rccl()
    except ImportError:
        pass                  # (the library still looks for librccl.so.1 / $SSYM_RCCL_LIB by itself)
    return nat.lib().ssym_comm_available() == 1


class LocalGroup:
    """The ranks of one process without RCCL (ssym_local_group): a thread per rank, every rank its own Engine.
    RCCL refuses two ranks on one device; this is how the multi-rank logic of ssym_match_sharded runs on a one-GPU
    box.  Every rank's thread must be inside match_sharded at the same time (ctypes releases the GIL)."""

    def __init__(self, world: int):
        out = ctypes.c_void_p()
        nat.check(nat.lib().ssym_local_group_create(world, ctypes.byref(out)), None)
        self.ptr, self.world = out.value, world

    def close(self):
        if self.ptr:
            nat.lib().ssym_local_group_destroy(self.ptr)
        self.ptr = None


def comm_unique_id() -> bytes:
    """ssym_comm_unique_id: the 128-byte RCCL id rank 0 hands to the other ranks."""
    nat.load_rccl()
    buf = ctypes.create_string_buffer(nat.COMM_ID_BYTES)
    nat.check(nat.lib().ssym_comm_unique_id(buf), None)
    return buf.raw


class Engine:
    """One ssym_ctx: one GPU, one stream, one metric / dtype configuration."""

    def __init__(self, metric: str = "dtw", dtype: str = "f32", device: int = 0, band: int = -1,
                 squared: bool = False, stream: Optional[int] = None, prune: bool = False):
        self.metric, self.dtype = metric, dtype
        self.np_dtype = {"f64": np.float64, "f32": np.float32}[dtype]
        self.ctx = None
        cfg = nat.Config(ctypes.sizeof(nat.Config), device,
                         {"refcos": nat.METRIC_REFCOS, "dtw": nat.METRIC_DTW}[metric],
                         {"f64": nat.DTYPE_F64, "f32": nat.DTYPE_F32}[dtype], band,
                         1 if squared else 0, stream, 1 if prune else 0, 0)
        out = ctypes.c_void_p()
        nat.check(nat.lib().ssym_ctx_create(ctypes.byref(cfg), ctypes.byref(out)), None)
        self.ctx = out.value
        self.device = device

    def close(self):
        if self.ctx:
            nat.lib().ssym_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- resident segment sets ---------------------------------------------------------------
    def _make(self, kind: str, feats, offsets, dim: int) -> _Handle:
        L = nat.lib()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = off.size - 1
        if n < 0:
            raise ValueError("offsets must hold n+1 entries")
        out = ctypes.c_void_p()
        if _is_device_tensor(feats):
            want = {"f64": "torch.float64", "f32": "torch.float32"}[self.dtype]
            if str(feats.dtype) != want or not feats.is_contiguous():
                raise ValueError(f"device features must be contiguous {want}")
            fn = L.ssym_dict_create_device if kind == "dict" else L.ssym_queries_create_device
            fptr = feats.data_ptr()
            keep = feats
        else:
            keep = np.ascontiguousarray(feats, dtype=self.np_dtype).reshape(-1)
            fn = L.ssym_dict_create if kind == "dict" else L.ssym_queries_create
            fptr = keep.ctypes.data
        rc = fn(self.ctx, fptr, off.ctypes.data, n, dim, ctypes.byref(out))
        del keep
        nat.check(rc, self.ctx)
        return _Handle(self, out.value, n, dim, kind)

    def dictionary(self, feats, offsets, dim: int) -> _Handle:
        return self._make("dict", feats, offsets, dim)

    def queries(self, feats, offsets, dim: int) -> _Handle:
        return self._make("queries", feats, offsets, dim)

    def dictionary_append(self, d: _Handle, feats, offsets) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        f = np.ascontiguousarray(feats, dtype=self.np_dtype).reshape(-1)
        nat.check(nat.lib().ssym_dict_append(self.ctx, d.ptr, f.ctypes.data, off.ctypes.data,
                                             off.size - 1), self.ctx)
        d.n += off.size - 1

    # -- the hot path ------------------------------------------------------------------------
    def match(self, d: _Handle, q: _Handle, distance=None, index_base: int = 0,
              force_exact: bool = False, out_idx=None, out_cost=None, prune: bool = False
              ) -> Tuple[np.ndarray, np.ndarray]:
        """argmin per target.  With torch CUDA tensors in out_idx (int32/uint32 storage, n
        entries) and out_cost (float64) the results stay on the device.  prune=True asks for
        early abandoning (SSYM_DTW_PRUNE): same results, data-dependent time."""
        L = nat.lib()
        m = q.n
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            if dist.size != m:
                raise ValueError("distance must have one entry per target")
            dist_p = dist.ctypes.data
        flags = (nat.DTW_FORCE_EXACT if force_exact else 0) | (nat.DTW_PRUNE if prune else 0)
        if out_idx is not None and _is_device_tensor(out_idx):
            flags |= nat.OUT_DEVICE
            rc = L.ssym_match_queries(self.ctx, d.ptr, q.ptr, dist_p, index_base, out_idx.data_ptr(),
                                      out_cost.data_ptr() if out_cost is not None else None, flags)
            nat.check(rc, self.ctx)
            return out_idx, out_cost
        idx = np.zeros(m, dtype=np.uint32)
        cost = np.zeros(m, dtype=np.float64)
        rc = L.ssym_match_queries(self.ctx, d.ptr, q.ptr, dist_p, index_base, idx.ctypes.data,
                                  cost.ctypes.data, flags)
        nat.check(rc, self.ctx)
        return idx, cost

    def match_begin(self, d: _Handle, q: _Handle, bounds, distance=None, index_base: int = 0) -> None:
        """ssym_match_begin: filter + per-target bound into `bounds` (torch CUDA float64 [m])."""
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            if dist.size != q.n:
                raise ValueError("distance must have one entry per target")
            dist_p = dist.ctypes.data
        nat.check(nat.lib().ssym_match_begin(self.ctx, d.ptr, q.ptr, dist_p, index_base, bounds.data_ptr()),
                  self.ctx)

    def match_candidates(self, d: _Handle, q: _Handle, costs) -> None:
        """ssym_match_candidates: exact cost of this shard's candidate pair per target into `costs`
        (torch CUDA float64 [m]); all-reduce(MIN) it, then match_begin(..., candidate_costs=costs)."""
        nat.check(nat.lib().ssym_match_candidates(self.ctx, d.ptr, q.ptr, costs.data_ptr()), self.ctx)

    def match_begin_pruned(self, d: _Handle, q: _Handle, bounds, candidate_costs, index_base: int = 0) -> None:
        """ssym_match_begin_pruned: match_begin whose filter abandons pairs above the reduced costs."""
        nat.check(nat.lib().ssym_match_begin_pruned(self.ctx, d.ptr, q.ptr, index_base, candidate_costs.data_ptr(),
                                                    bounds.data_ptr()), self.ctx)

    def match_finish(self, bounds, out_idx, out_cost):
        """ssym_match_finish with device outputs (torch CUDA tensors)."""
        nat.check(nat.lib().ssym_match_finish(self.ctx, bounds.data_ptr(), out_idx.data_ptr(),
                                              out_cost.data_ptr() if out_cost is not None else None,
                                              nat.OUT_DEVICE), self.ctx)
        return out_idx, out_cost

    # -- source-sharded runs: the collectives inside the library (RCCL on the context's stream) ----------
    def comm_create(self, unique_id: bytes, rank: int, world: int) -> Comm:
        """ssym_comm_create: collective over all `world` ranks (ncclCommInitRank on this engine's GPU)."""
        nat.load_rccl()
        if len(unique_id) != nat.COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        out = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), nat.COMM_ID_BYTES)
        nat.check(nat.lib().ssym_comm_create(self.ctx, buf, rank, world, ctypes.byref(out)), self.ctx)
        return Comm(self, out.value, rank, world)

    def comm_create_local(self, group: LocalGroup, rank: int) -> Comm:
        """ssym_comm_create_local: this engine as rank `rank` of an in-process group (no RCCL)."""
        out = ctypes.c_void_p()
        nat.check(nat.lib().ssym_comm_create_local(self.ctx, group.ptr, rank, ctypes.byref(out)), self.ctx)
        return Comm(self, out.value, rank, group.world)

    def match_sharded(self, comm: Comm, d: _Handle, q: _Handle, distance=None, index_base: int = 0,
                      out_idx=None, out_cost=None, prune: bool = False, force_exact: bool = False):
        """ssym_match_sharded: this rank's shard `d` (global indices from index_base) against all targets `q`;
        every rank returns the merged answer.  Outputs as for match()."""
        L = nat.lib()
        m = q.n
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            if dist.size != m:
                raise ValueError("distance must have one entry per target")
            dist_p = dist.ctypes.data
        flags = (nat.DTW_FORCE_EXACT if force_exact else 0) | (nat.DTW_PRUNE if prune else 0)
        if out_idx is not None and _is_device_tensor(out_idx):
            rc = L.ssym_match_sharded(self.ctx, comm.ptr, d.ptr, q.ptr, dist_p, index_base, out_idx.data_ptr(),
                                      out_cost.data_ptr() if out_cost is not None else None, flags | nat.OUT_DEVICE)
            nat.check(rc, self.ctx)
            return out_idx, out_cost
        idx = np.zeros(m, dtype=np.uint32)
        cost = np.zeros(m, dtype=np.float64)
        rc = L.ssym_match_sharded(self.ctx, comm.ptr, d.ptr, q.ptr, dist_p, index_base, idx.ctypes.data,
                                  cost.ctypes.data, flags)
        nat.check(rc, self.ctx)
        return idx, cost

    def match_topk(self, d: _Handle, q: _Handle, k: int, distance=None, index_base: int = 0,
                   force_exact: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """ssym_match_topk: (idx [m][k] uint32, cost [m][k] f64); rows are ordered by
        (|value - distance|, index); missing entries are nat.NO_MATCH / NaN."""
        m = q.n
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            if dist.size != m:
                raise ValueError("distance must have one entry per target")
            dist_p = dist.ctypes.data
        idx = np.zeros((m, k), dtype=np.uint32)
        cost = np.zeros((m, k), dtype=np.float64)
        rc = nat.lib().ssym_match_topk(self.ctx, d.ptr, q.ptr, dist_p, k, index_base, idx.ctypes.data,
                                       cost.ctypes.data, nat.DTW_FORCE_EXACT if force_exact else 0)
        nat.check(rc, self.ctx)
        return idx, cost

    def match_batch(self, d: _Handle, feats, offsets, distance=None):
        """ssym_match_batch: pack host targets, match, release."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        f = np.ascontiguousarray(feats, dtype=self.np_dtype).reshape(-1)
        m = off.size - 1
        idx = np.zeros(m, dtype=np.uint32)
        cost = np.zeros(m, dtype=np.float64)
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            dist_p = dist.ctypes.data
        rc = nat.lib().ssym_match_batch(self.ctx, d.ptr, f.ctypes.data, off.ctypes.data, m, dist_p,
                                        idx.ctypes.data, cost.ctypes.data)
        nat.check(rc, self.ctx)
        return idx, cost

    def match_one(self, d: _Handle, feats, distance: float):
        f = np.ascontiguousarray(feats, dtype=self.np_dtype).reshape(-1)
        if f.size % d.dim:
            raise ValueError("query is not a whole number of frames")
        idx = ctypes.c_uint32(0)
        cost = ctypes.c_double(0.0)
        rc = nat.lib().ssym_match_one(self.ctx, d.ptr, f.ctypes.data, f.size // d.dim,
                                      float(distance), ctypes.byref(idx), ctypes.byref(cost))
        nat.check(rc, self.ctx)
        return int(idx.value), float(cost.value)

    def chain(self, d: _Handle, start_feats, distances) -> Tuple[np.ndarray, np.ndarray]:
        """ssym_chain: from_distances (src/sound.rs:405-417) without a host round trip per step.
        Returns (idx [n_steps], cost [n_steps]); idx[i] is the dictionary entry matched at step i."""
        f = np.ascontiguousarray(start_feats, dtype=self.np_dtype).reshape(-1)
        if f.size % d.dim:
            raise ValueError("start is not a whole number of frames")
        dist = np.ascontiguousarray(distances, dtype=np.float64).reshape(-1)
        idx = np.zeros(dist.size, dtype=np.uint32)
        cost = np.zeros(dist.size, dtype=np.float64)
        rc = nat.lib().ssym_chain(self.ctx, d.ptr, f.ctypes.data, f.size // d.dim, dist.ctypes.data, dist.size,
                                  idx.ctypes.data, cost.ctypes.data)
        nat.check(rc, self.ctx)
        return idx, cost

    def pair_matrix(self, d: _Handle, q: _Handle, exact: bool = False) -> np.ndarray:
        out = np.zeros((d.n, q.n), dtype=np.float64)
        nat.check(nat.lib().ssym_pair_matrix(self.ctx, d.ptr, q.ptr, 1 if exact else 0,
                                             out.ctypes.data), self.ctx)
        return out

    # -- feature front-end (F3) --------------------------------------------------------------
    def mfcc(self, samples, sample_rate: float, ncoeffs: int = 12, f_lo: float = 100.0, f_hi: float = 8000.0,
             pad_tail: bool = False, want_mean: bool = False):
        """ssym_mfcc: [frames][ncoeffs] f64 (analyze_mfccs, src/sound.rs:215-242; parity unpinned),
        optionally with the per-coefficient mean (analyze_mean_mfccs, :271-286)."""
        x = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1)
        flags = nat.MFCC_PAD_TAIL if pad_tail else 0
        t = ctypes.c_uint64(0)
        nat.check(nat.lib().ssym_mfcc_num_frames(x.size, flags, ctypes.byref(t)), self.ctx)
        out = np.zeros((int(t.value), ncoeffs), dtype=np.float64)
        mean = np.zeros(ncoeffs, dtype=np.float64)
        rc = nat.lib().ssym_mfcc(self.ctx, x.ctypes.data if x.size else None, x.size, float(sample_rate), ncoeffs,
                                 float(f_lo), float(f_hi), flags, out.ctypes.data if out.size else None,
                                 mean.ctypes.data if want_mean else None)
        nat.check(rc, self.ctx)
        return (out, mean) if want_mean else out

    # -- reconstruction tail (F2) ------------------------------------------------------------
    def samples(self, samples, sample_offsets):
        """Make the dictionary sounds' samples resident (ssym_samples_create)."""
        smp = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1)
        off = np.ascontiguousarray(sample_offsets, dtype=np.uint64)
        out = ctypes.c_void_p()
        nat.check(nat.lib().ssym_samples_create(self.ctx, smp.ctypes.data, off.ctypes.data, off.size - 1,
                                                ctypes.byref(out)), self.ctx)
        return _SamplesHandle(self, out.value, off.size - 1)

    def reconstruct(self, smp, idx, out_offsets, want_pcm32: bool = False):
        """Length-fitted, concatenated samples of the matched sounds (ssym_reconstruct)."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        off = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        total = int(off[-1])
        out = np.zeros(total, dtype=np.float64)
        pcm = np.zeros(total, dtype=np.int32) if want_pcm32 else None
        nat.check(nat.lib().ssym_reconstruct(self.ctx, smp.ptr, idx.ctypes.data, off.ctypes.data, idx.size,
                                             out.ctypes.data, pcm.ctypes.data if pcm is not None else None),
                  self.ctx)
        return (out, pcm) if want_pcm32 else out

    def merge_shards(self, costs, idx, out_idx, out_cost, distance=None) -> None:
        """costs [G, M] f64, idx [G, M] 32-bit, outputs [M]: torch CUDA tensors on this GPU; distance:
        the per-target distances the shards matched with (host array), or None."""
        g, m = costs.shape
        dist_p = None
        if distance is not None:
            dist = np.ascontiguousarray(distance, dtype=np.float64)
            dist_p = dist.ctypes.data
        nat.check(nat.lib().ssym_merge_shards_at(self.ctx, g, m, costs.data_ptr(), idx.data_ptr(), dist_p,
                                                 out_idx.data_ptr(), out_cost.data_ptr()), self.ctx)

    def timings(self) -> dict:
        t = nat.Timings()
        nat.check(nat.lib().ssym_get_timings(self.ctx, ctypes.byref(t)), self.ctx)
        return t.as_dict()

    def synchronize(self) -> None:
        nat.check(nat.lib().ssym_ctx_synchronize(self.ctx), self.ctx)
