#!/usr/bin/env python3
"""bench.py -- segment-pairs/sec of the DTW matching hot path (BASELINE.json metric).

One step = one pass of the hot path over one batch: every (source, target) pair's DTW cost on the
f32 MFMA filter kernel, candidate selection, exact f64 re-scoring of the candidates and the
per-target argmin (ssym_match_queries), plus -- with more than one GPU -- the all-gather of the
per-target (cost, index) candidates and the merge kernel.  Features are resident in HBM before
the timed region.

N = 1: BASELINE.json configs[2], 4096 x 4096 segments, 128 frames x 13 dims, f32 ("the roofline
run"; the metric is quoted on it).  N > 1: source-axis sharding with 4096 sources per GPU and all
4096 targets on every GPU (weak scaling; 8 GPUs = 32768 x 4096).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SRC_PER_GPU = 4096
N_TGT = 4096
FRAMES = 128
DIM = 13
SEED = 0x5EED0003
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _oracle_sample(o, grid, gpu_idx, n_src, n_tgt, threads):
    tsel = np.arange(n_tgt)
    # keep each sampled target's planted source inside the sampled source block
    planted = grid.planted[tsel]
    rest = np.setdiff1d(np.arange(grid.sources.shape[0]), planted)[:max(0, n_src - planted.size)]
    ssel = np.sort(np.concatenate([planted, rest]))
    sf = np.ascontiguousarray(grid.sources[ssel], dtype=np.float64).reshape(-1)
    tf = np.ascontiguousarray(grid.targets[tsel], dtype=np.float64).reshape(-1)
    so = np.arange(ssel.size + 1, dtype=np.uint64) * grid.frames
    to = np.arange(tsel.size + 1, dtype=np.uint64) * grid.frames
    t0 = time.perf_counter()
    idx, _ = o.dtw_match_all(sf, so, tf, to, grid.dim, nthreads=threads)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(ssel[idx], gpu_idx[tsel])) if gpu_idx is not None else None
    return ssel.size * tsel.size, dt, ok


def cpu_baseline(grid, gpu_idx, budget_s=12.0):
    """The CPU oracle (a port: the reference is Rust and cannot be built here) on a bounded
    sample of the same workload, all host cores, OpenMP over targets.  A small probe sizes the
    sample so that it takes about `budget_s` seconds on whatever host this is."""
    import oracle
    o = oracle.load()
    threads = max(1, min(o.max_threads(), os.cpu_count() or 1))
    n_all_s, n_all_t = grid.sources.shape[0], grid.targets.shape[0]
    pairs, dt, _ = _oracle_sample(o, grid, None, 64, min(n_all_t, 4 * threads), threads)
    rate = pairs / max(dt, 1e-6)
    want = max(pairs, rate * budget_s)
    n_tgt = int(min(n_all_t, max(threads, 256)))
    n_src = int(min(n_all_s, max(n_tgt, want // n_tgt)))
    if n_src == n_all_s:
        n_tgt = int(min(n_all_t, max(n_tgt, want // n_src)))
    pairs, dt, ok = _oracle_sample(o, grid, gpu_idx, n_src, n_tgt, threads)
    return {
        "value": pairs / dt,
        "unit": "segment-pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_src}x{n_tgt} sub-grid of the workload ({pairs} pairs, {dt:.1f} s, f64 oracle, "
                  f"OpenMP over targets); indices equal the GPU's: {ok}",
    }


def secondary_metrics(device: int, with_cpu: bool) -> dict:
    """Outside the timed region: the reference's own metric (refcos, SURVEY.md 8(d) "also report")
    and the neighbouring rows (chain F4, MFCC F3), each one short measurement."""
    from soundsym_amd import Engine, synth
    out = {}
    n, f, dd = 4096, 128, 12
    g = synth.make_grid(n, n, f, dd, 0x5EED0103)
    off = np.arange(n + 1, dtype=np.uint64) * f
    e = Engine(metric="refcos", dtype="f64", device=device)
    d = e.dictionary(g.sources.astype(np.float64).reshape(-1), off, dd)
    q = e.queries(g.targets.astype(np.float64).reshape(-1), off, dd)
    e.match(d, q)
    t0 = time.perf_counter()
    for _ in range(5):
        e.match(d, q)
    dt = (time.perf_counter() - t0) / 5
    out["refcos"] = {"value": n * n / dt, "unit": "segment-pairs/s", "ms_per_step": dt * 1e3,
                     "workload": f"{n}x{n} segments, {f} frames x {dd} dims, f64, reference metric "
                                 "(cosine_sim + at_distance, bit-exact)"}
    dist_ = np.linspace(0.2, 1.2, 256)
    e.chain(d, g.targets[0].astype(np.float64).reshape(-1), dist_[:2])
    t0 = time.perf_counter()
    e.chain(d, g.targets[0].astype(np.float64).reshape(-1), dist_)
    out["chain"] = {"value": (time.perf_counter() - t0) / dist_.size * 1e6, "unit": "us/step",
                    "workload": f"from_distances, 256 steps on the {n}-entry dictionary (refcos)"}
    # the reference's own call pattern: at_distance / match_sound ONE query at a time (src/sound.rs:453-454), against a
    # 1024-entry dictionary of 5...40-frame segments (through the Python binding, a few us of ctypes included)
    from soundsym_amd.engine import pack_segments
    rsrc, rtgt = synth.make_ragged(1024, 32, 5, 40, dd, 0x5EED0700)
    sf1, so1 = pack_segments([s_.astype(np.float64) * 0.05 for s_ in rsrc], dd)
    d1 = e.dictionary(sf1, so1, dd)
    qs = [t_.astype(np.float64).reshape(-1) * 0.05 for t_ in rtgt]
    e.match_one(d1, qs[0], 1.0)
    t0 = time.perf_counter()
    for _ in range(8):
        for q1 in qs:
            e.match_one(d1, q1, 1.0)
    out["match_one"] = {"value": (time.perf_counter() - t0) / (8 * len(qs)) * 1e6, "unit": "us/query",
                        "workload": "ssym_match_one (refcos), 1024-entry dictionary, queries of 5...40 frames x 12 dims"}
    rate = 44100.0
    x = np.sin(2 * np.pi * 440.0 * np.arange(int(rate * 120)) / rate)
    e.mfcc(x[:44100], rate)
    t0 = time.perf_counter()
    fr = e.mfcc(x, rate).shape[0]
    dt = time.perf_counter() - t0
    out["mfcc"] = {"value": fr / dt, "unit": "frames/s", "workload": "120 s of 44.1 kHz audio, host to host"}
    if with_cpu:
        import oracle as oracle_pkg
        o = oracle_pkg.load()
        k = 192
        offk = np.arange(k + 1, dtype=np.uint64) * f
        sf = g.sources[:k].astype(np.float64).reshape(-1)
        tf = g.targets[:k].astype(np.float64).reshape(-1)
        t0 = time.perf_counter()
        o.refcos_match_all(sf, offk, tf, offk, dd)
        dt = time.perf_counter() - t0
        out["refcos"]["cpu_baseline"] = {"value": k * k / dt, "unit": "segment-pairs/s", "cores": 1, "kind": "port",
                                         "sample": f"{k}x{k} sub-grid, single thread as the reference runs it"}
    e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the refcos / chain / mfcc measurements reported beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--src-per-gpu", type=int, default=SRC_PER_GPU)
    ap.add_argument("--targets", type=int, default=N_TGT)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--band", type=int, default=-1, help="Sakoe-Chiba radius (configs[4]: 32), -1 = none")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from soundsym_amd import Engine, sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SSYM_BENCH_FORCE_DIST=1: run the collective code path (RCCL init, all-gather, merge) even with
    # one rank -- a smoke check of the N > 1 plumbing on a single-GPU box
    force_dist = os.environ.get("SSYM_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N > 1 must be launched with `python -m torch.distributed.run "
                 "--nproc-per-node N ... bench.py --gpus N` (one rank per GPU)")
    # SSYM_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
    backend = os.environ.get("SSYM_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1 or force_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = max(world, 1)
    n_src_total = args.src_per_gpu * n_gpus
    m = args.targets

    # synthetic workload (seeded; the generator is counter-based, so each rank materialises only its
    # own source shard of the one global grid -- the targets and planted indices are the same on all)
    DIM_ = args.dim
    lo, hi = sharding.shard_range(n_src_total, n_gpus, rank)
    if n_gpus > 1:
        grid = synth.make_grid(n_src_total, m, args.frames, DIM_, SEED, src_range=(lo, hi))
        shard = grid.sources
    else:
        grid = synth.make_grid(n_src_total, m, args.frames, DIM_, SEED)
        shard = grid.sources[lo:hi]
    eng = Engine(metric="dtw", dtype="f32", device=local_rank, band=args.band)
    src_dev = torch.from_numpy(np.ascontiguousarray(shard).reshape(-1)).cuda()
    tgt_dev = torch.from_numpy(np.ascontiguousarray(grid.targets).reshape(-1)).cuda()
    so = np.arange(hi - lo + 1, dtype=np.uint64) * args.frames
    to = np.arange(m + 1, dtype=np.uint64) * args.frames
    d = eng.dictionary(src_dev, so, DIM_)
    q = eng.queries(tgt_dev, to, DIM_)
    out_idx = torch.empty(m, dtype=torch.int32, device="cuda")
    out_cost = torch.empty(m, dtype=torch.float64, device="cuda")

    bounds = torch.empty(m, dtype=torch.float64, device="cuda")

    def step():
        if world > 1 or force_dist:
            # filter -> all-reduce(MIN) of the per-target bounds -> select / re-score -> all-gather + merge
            return sharding.match_sharded(eng, d, q, lo, out_idx, out_cost, bounds)
        eng.match(d, q, index_base=lo, out_idx=out_idx, out_cost=out_cost)
        return out_idx, out_cost

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    main_ms, total_ms, refined, last_tm = [], [], 0, {}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fin_idx, fin_cost = step()
        tm = eng.timings()
        main_ms.append(tm["main_ms"])
        total_ms.append(tm["total_ms"])
        refined = tm["n_refined"]
        last_tm = tm
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_per_step = n_src_total * m
    value = pairs_per_step * args.steps / elapsed
    idx_host = fin_idx.cpu().numpy().astype(np.int64)
    planted_ok = bool(np.array_equal(idx_host, grid.planted))

    # Beside the headline (which fills the whole cost matrix): the same search with early abandoning
    # (SSYM_DTW_PRUNE) -- identical indices and costs, but the time depends on the data, and
    # this planted grid is its best case; reported separately and never as `value`.
    early = None
    if args.band < 0 and not args.no_secondary:
        p_idx, p_cost = torch.empty_like(out_idx), torch.empty_like(out_cost)
        full_idx, full_cost = fin_idx.clone(), fin_cost.clone()

        def pstep():
            if world > 1 or force_dist:
                # candidates' costs all-reduced (MIN) first, then the sequence of step() with abandoning filters
                return sharding.match_sharded(eng, d, q, lo, p_idx, p_cost, bounds, prune=True)
            eng.match(d, q, index_base=lo, out_idx=p_idx, out_cost=p_cost, prune=True)
            return p_idx, p_cost

        for _ in range(args.warmup):
            pstep()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pf_idx, pf_cost = pstep()
        fence()
        p_elapsed = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([p_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_elapsed = float(t.item())
        ptm = eng.timings()
        full_cells = float(((hi - lo + 7) // 8 * 8)) * ((m + 31) // 32 * 32) * (16 * 4 * ((args.frames + 63) // 64)
                                                                                 if args.frames > 48 else 16 * ((args.frames + 15) // 16)) * args.frames
        early = {
            "value": pairs_per_step * args.steps / p_elapsed, "unit": "segment-pairs/s",
            "ms_per_step": p_elapsed / args.steps * 1e3,
            "identical_to_full_search": bool(torch.equal(pf_idx, full_idx) and torch.equal(pf_cost, full_cost)),
            "filter_cells_swept_frac": ptm["n_filter_cells"] / full_cells if ptm["pruned"] else None,
            "phase_ms": {k: round(float(v), 3) for k, v in ptm.items() if k.endswith("_ms")},
            "rank0_only": ["filter_cells_swept_frac", "phase_ms"] if world > 1 else [],
            "note": "one centroid-nearest candidate per target scored exactly, then the filter stops row passes and drops "
                    "64-pair tasks that are provably above it; planted grid = best case (no-close-pair data: +4 % over the full search)",
        }

    if rank == 0:
        # Roofline of the dominant kernel (dtw_filter_kernel).  Its duration is measured live with
        # HIP events on the library's own stream (ssym_get_timings).  Algorithmic work per pair is
        # SURVEY.md 8(d)'s: bytes 2*F*d*4 (per-pair operand-streaming model -- the model the
        # north star's ">= 60 % HBM roofline" is stated in), matrix flops 2*F^2*d, DP cells F^2.
        k_ms = float(np.mean(main_ms))
        k_s = k_ms * 1e-3
        pairs_launch = (hi - lo) * m
        f, dd = args.frames, DIM_
        r = args.band
        cells_pair = float(f) * f if r < 0 else float(f * (2 * r + 1) - r * (r + 1))
        stream_gbps = pairs_launch * 2 * f * dd * 4 / k_s / 1e9
        flops_tf = pairs_launch * 2.0 * cells_pair * dd / k_s / 1e12
        cells_per_s = pairs_launch * cells_pair / k_s
        traffic = mfma_busy = valu_busy = None
        prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench_1gpu.json")))
        if prof and n_gpus == 1 and (hi - lo, m, f, dd, r) == (SRC_PER_GPU, N_TGT, FRAMES, DIM, -1):
            pj = json.load(open(prof[-1]))           # rocprofv3 PMC passes over this same command (tools/profile_bench.sh)
            traffic = pj.get("hbm_traffic_bytes_per_launch")
            mfma_busy, valu_busy = pj.get("mfma_busy_fraction"), pj.get("valu_busy_fraction")
        # VALU floor measured on MI355X (profiles/): v_sqrt_f32 8 + v_min3_f32 4 + v_add_f32 4
        # cycles per wave-instruction = 16 cycles per 64 cells per SIMD, 1024 SIMDs
        valu_peak_cells = 1024 * 64 / 16.0 * 2.4e9
        line = {
            "metric": "segment-pairs/sec (DTW cost+argmin)",
            "value": value,
            "unit": "segment-pairs/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_src_total}x{m} segments, {f} frames x {dd} dims, f32, "
                            f"dtw (L2 local cost, {'full matrix' if r < 0 else f'Sakoe-Chiba r={r}'}), "
                            f"planted neighbours, seed 0x{SEED:X}",
                "sources_per_gpu": hi - lo,
                "parallelism": f"source-shard x{n_gpus}" if n_gpus > 1 else "single GPU",
                "indices_equal_planted": planted_ok,
                "pairs_refined_f64": int(refined),
                "phase_ms": {k: round(float(v), 3) for k, v in last_tm.items() if k.endswith("_ms")},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "dtw_filter_kernel" if r < 0 else "dtw_band_kernel",
                "model": "per-pair operand-streaming bytes 2*F*d*4 (SURVEY.md 8(d)); the kernel keeps "
                         "operands on chip, so measured HBM traffic is far below this and the "
                         "limiter is VALU issue, see 'valu' and DESIGN.md",
                "achieved": stream_gbps,
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": stream_gbps / PEAK_HBM_GBPS,
                "traffic": traffic,
                "traffic_note": "bytes per launch at the L2-fabric boundary (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                                "profiles/r01_bench_1gpu.md): the per-wave hand-off rows written and read once",
                "algorithmic_bytes_per_launch": pairs_launch * 2 * f * dd * 4,
                "kernel_ms": k_ms,
                "mfma": {"achieved": flops_tf, "unit": "TFLOP/s", "algorithmic_flops_per_pair": 2 * cells_pair * dd,
                         "peak_f32_mfma": PEAK_F32_MFMA_TFLOPS, "frac_of_f32_mfma_peak": flops_tf / PEAK_F32_MFMA_TFLOPS,
                         "pipe_busy_frac_pmc": mfma_busy,
                         "note": "cost block runs on the f16 matrix pipe (3 x 32x32x16 per 32x32 tile, "
                                 "two-piece operand split); algorithmic flops, not issued flops"},
                "valu": {"achieved": cells_per_s, "unit": "DP cells/s", "peak": valu_peak_cells,
                         "frac": cells_per_s / valu_peak_cells, "busy_frac_pmc": valu_busy,
                         "note": "16 VALU cycles per cell per SIMD at 2.4 GHz (measured issue costs)"},
            },
        }
        if early is not None:
            line["early_abandon"] = early
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(grid, idx_host)
        if n_gpus == 1 and not args.no_secondary:
            line["secondary"] = secondary_metrics(local_rank, not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
