"""Source-axis sharding across the GPUs of one node.

Rank g of G holds dictionary segments [lo, hi) = shard_range(N, G, g) and ALL targets.  The step itself
-- filter, all-reduce(MIN) of the per-target bounds, selection and exact re-scoring, all-gather of the
per-target (cost, global index), merge by the reference's first-minimum rule (src/sound.rs:361-367) --
is ONE C-ABI call, ssym_match_sharded: the collectives are RCCL calls enqueued by the library on its own
stream (csrc/comm.hip), the host synchronises once.  This module is its caller: `init_comm` distributes
the RCCL unique id over whatever process group the host already has and creates the communicator,
`match_sharded` forwards.

`gather_candidates` / `reduce_bounds` / `match_sharded_torch` are the same exchange spelled with
torch.distributed collectives around ssym_match_begin / _finish / ssym_merge_shards; they remain for
rehearsals over gloo (CPU tests here, more ranks than GPUs on a one-GPU box), where RCCL cannot run.

This module only moves bytes and computes ranges; it contains no arithmetic of the path and no
CPU substitute for the merge kernel.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, ordered, balanced split of [0, n): the first n % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return lo, hi


def gather_candidates(cost, idx, group=None):
    """all_gather every rank's per-target (cost f64 [M], global index int32-storage [M]).

    Returns (costs [G, M], idx [G, M]) on the same device as the inputs.  With world size 1 (or
    no process group) this is a reshape, no collective.  Costs and indices travel in ONE collective:
    the indices ride along as float64 (every 32-bit integer is exact in a double), which halves the
    number of latency-bound RCCL calls per step."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return cost.reshape(1, -1), idx.reshape(1, -1)
    world = dist.get_world_size(group)
    m = cost.numel()
    dev = cost.device
    # gloo (CPU rehearsal of the exchange, also with GPU-resident results) moves bytes through host
    # memory; nccl (= RCCL over xGMI) gathers device to device
    via_host = dist.get_backend(group) == "gloo" and cost.is_cuda
    packed = torch.empty(2 * m, dtype=torch.float64, device=dev)
    packed[:m] = cost.reshape(-1)
    # indices are u32 in the ABI; torch stores them as int32 bits, so widen through int64 and mask
    packed[m:] = (idx.reshape(-1).view(torch.int32).to(torch.int64) & 0xFFFFFFFF).to(torch.float64)
    if via_host:
        packed = packed.cpu()
    allp = torch.empty(world * 2 * m, dtype=torch.float64, device=packed.device)
    dist.all_gather_into_tensor(allp, packed, group=group)
    if via_host:
        allp = allp.to(dev)
    allp = allp.view(world, 2, m)
    gidx = allp[:, 1, :].to(torch.int64)
    gidx = torch.where(gidx >= 2 ** 31, gidx - 2 ** 32, gidx).to(torch.int32)      # back to the u32 bit pattern
    return allp[:, 0, :].contiguous(), gidx.view(idx.dtype).contiguous() if idx.dtype != torch.int32 else gidx.contiguous()


def reduce_bounds(bounds, group=None):
    """all_reduce(MIN) of the per-target bounds of ssym_match_begin, in place (M x 8 bytes; RCCL
    over xGMI on GPUs).  No process group: nothing to do."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return bounds
    if dist.get_backend(group) == "gloo" and bounds.is_cuda:
        host = bounds.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
        bounds.copy_(host)
    else:
        dist.all_reduce(bounds, op=dist.ReduceOp.MIN, group=group)
    return bounds


def init_comm(engine, rank: int, world: int, group=None):
    """Create this rank's RCCL communicator behind the C ABI (ssym_comm_create).  Rank 0 draws the unique
    id (ssym_comm_unique_id); it reaches the other ranks through the torch.distributed group the host
    already has (any backend -- 128 bytes as a tensor broadcast), or, with world == 1, not at all."""
    from .engine import comm_unique_id

    if world == 1:
        return engine.comm_create(comm_unique_id(), 0, 1)
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("init_comm with world > 1 needs an initialised torch.distributed group to hand the id over")
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", engine.device) if on_gpu else torch.device("cpu")
    # byte 0: rank 0 could draw the id.  A failure on rank 0 reaches every rank here, BEFORE any of them enters
    # ncclCommInitRank (which would wait for rank 0 for ever); all ranks then raise the same error.
    msg = bytearray(1 + 128)
    why = ""
    if rank == 0:
        try:
            msg[1:] = comm_unique_id()
            msg[0] = 1
        except Exception as ex:                                # noqa: BLE001 -- reported on every rank below
            why = f": {ex}"
    t = torch.frombuffer(msg, dtype=torch.uint8).to(dev) if rank == 0 else torch.zeros(129, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    if raw[0] != 1:
        raise RuntimeError("init_comm: rank 0 could not obtain an RCCL unique id (ssym_comm_unique_id)" + why)
    return engine.comm_create(raw[1:], rank, world)


def match_sharded(engine, comm, d, q, index_base, out_idx=None, out_cost=None, distance=None, prune=False):
    """One rank's part of a source-sharded match, collectives included: ssym_match_sharded.  Every rank
    returns the merged (idx [M], cost [M]); with torch CUDA tensors in out_idx / out_cost they stay on
    the device."""
    return engine.match_sharded(comm, d, q, distance=distance, index_base=index_base, out_idx=out_idx,
                                out_cost=out_cost, prune=prune)


def _all_ranks_ok(err, like, group=None):
    """One tiny all-reduce(MAX) of "this rank's local phase failed": every rank learns whether ANY rank failed, so that all
    of them leave the step together (the rule of ssym_match_sharded, spelled with torch.distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return err is None
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cpu" if on_host else like.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item()) == 0


class ShardedStepError(RuntimeError):
    """A rank's local phase of match_sharded_torch failed; raised on EVERY rank of the group (on the failing one with
    its own exception as the cause)."""


def match_sharded_torch(engine, d, q, index_base, out_idx, out_cost, bounds, group=None, distance=None, prune=False):
    """The same step with torch.distributed collectives around the two-phase C-ABI calls (gloo rehearsals):
    filter, agree on the per-target bound with the other ranks, select / re-score against it, gather every
    rank's winners and merge them.  All tensors are CUDA tensors on this rank's GPU; returns (idx [M], cost [M]).
    prune=True (plain nearest-neighbour search only): every rank scores one candidate pair per target
    first, the costs are reduced with MIN, and the filters abandon against them -- one more exchange of
    M f64 values, same results.
    A rank whose local work raises still takes part in every collective of the step (with +inf bounds / the fold start
    as its block) and ALL ranks raise ShardedStepError afterwards: no rank is left inside a collective, as in
    ssym_match_sharded."""
    import torch

    err = None
    try:
        if prune and distance is None:
            cand = torch.empty_like(bounds)
            try:
                engine.match_candidates(d, q, cand)
            except Exception as ex:                        # noqa: BLE001 -- reported by every rank below
                err = ex
                cand.fill_(float("inf"))
            reduce_bounds(cand, group)
            torch.cuda.current_stream().synchronize() if cand.is_cuda else None
            if err is None:
                engine.match_begin_pruned(d, q, bounds, cand, index_base=index_base)
        elif err is None:
            engine.match_begin(d, q, bounds, distance=distance, index_base=index_base)
    except Exception as ex:                                # noqa: BLE001
        err = err or ex
    if err is not None:
        bounds.fill_(float("inf"))
    reduce_bounds(bounds, group)
    if bounds.is_cuda:
        torch.cuda.current_stream().synchronize()      # the library runs on its own stream
    if err is None:
        try:
            engine.match_finish(bounds, out_idx, out_cost)
        except Exception as ex:                            # noqa: BLE001
            err = ex
    if err is not None:
        out_idx.fill_(0)
        out_cost.fill_(float("inf"))
    costs, idxs = gather_candidates(out_cost, out_idx, group)
    if not _all_ranks_ok(err, bounds, group):
        if err is not None:
            raise ShardedStepError(f"this rank's local phase failed: {err}") from err
        raise ShardedStepError("another rank's local phase failed; this rank's results are void")
    return merge_shards(engine, costs, idxs, distance)


def merge_shards(engine, costs, idx, distance=None):
    """Final G-way reduce on the GPU (ssym_merge_shards).  Inputs must be CUDA tensors."""
    import torch

    if not costs.is_cuda or not idx.is_cuda:
        raise RuntimeError("merge_shards runs the HIP kernel: tensors must be on the GPU")
    g, m = costs.shape
    # the gathered tensors were produced on torch's stream (RCCL enqueues and returns); the library
    # launches on its own stream, so the host orders the two
    torch.cuda.current_stream(costs.device).synchronize()
    out_idx = torch.empty(m, dtype=idx.dtype, device=idx.device)
    out_cost = torch.empty(m, dtype=torch.float64, device=costs.device)
    engine.merge_shards(costs.contiguous(), idx.contiguous(), out_idx, out_cost, distance)
    return out_idx, out_cost
