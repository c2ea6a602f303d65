"""bench.py keeps its contract: ONE JSON line with the keys the driver reads, on a small workload so that the
check costs a second (the default workload is measured by the driver itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None, shape=("--src-per-gpu", "512", "--targets", "256", "--frames", "64")):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"] + list(shape) + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines          # stdout is the record and nothing else (library banners go to stderr)
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_contract_keys():
    line = _run(["--no-secondary"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["metric"] == "segment-pairs/sec (DTW cost+argmin)" and line["unit"] == "segment-pairs/s"
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "n/a" and line["vs_baseline"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and line["config"]["indices_equal_planted"] is True
    rl = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rl, key
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    assert line["value"] > 0 and abs(line["value"] - 512 * 256 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_bench_early_abandon_leg_is_identical_to_the_full_search():
    line = _run([])                                   # with the secondary legs (refcos, chain, mfcc, early abandoning)
    ea = line["early_abandon"]
    assert ea["identical_to_full_search"] is True and ea["value"] > 0
    assert 0 < ea["filter_cells_swept_frac"] <= 1
    assert set(line["secondary"]) >= {"refcos", "chain", "mfcc", "match_one", "host_batch"}
    hb = line["secondary"]["host_batch"]
    assert hb["indices_equal_planted"] is True and hb["pack_ms"] > 0 and hb["ms_per_call"] > 0
    ncp = ea["no_close_pair"]
    assert ncp["identical_to_full_search"] is True and ncp["pruned_ms_per_step"] > 0


def test_bench_sharded_path_through_the_library_collectives():
    # one rank, but the N > 1 code path: RCCL communicator behind the C ABI, ssym_match_sharded per step
    line = _run(["--no-secondary"], env={"SSYM_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29547"},
                shape=("--sources", "512", "--targets", "256", "--frames", "64"))
    assert line["config"]["indices_equal_planted"] is True and line["config"]["collective_ms"] > 0
    assert line["config"]["per_rank"][0]["attempts"] == 1
    assert line["scaling"] == "n/a" and line["n_gpus"] == 1       # (strong is reported from 2 GPUs on)


def test_bench_banded_workload_shape():
    line = _run(["--no-secondary", "--workload", "c5"], shape=("--sources", "256", "--targets", "256", "--frames", "96"))
    assert line["roofline"]["kernel"] == "dtw_band_kernel" and line["config"]["indices_equal_planted"] is True
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1 and "hbm_streaming_model" in line["roofline"]


def test_bench_two_ranks_rehearsal_over_gloo():
    # two ranks on the one GPU of the test box (RCCL cannot put two ranks on one device, so the exchange runs as
    # torch.distributed collectives over gloo around the two-phase C-ABI calls): the strong-scaling split, the
    # per-rank figures, the planted check and the one JSON line of rank 0
    env = dict(os.environ, SSYM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29561", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-secondary", "--sources", "512",
                          "--targets", "256", "--frames", "64"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["sources_per_gpu"] == 256
    assert line["config"]["indices_equal_planted"] is True and len(line["config"]["per_rank"]) == 2
    assert abs(line["value"] - 512 * 256 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_bench_gpus_2_starts_its_own_ranks():
    # the driver's command shape for one GPU, with --gpus 2 and NO launcher around it: bench.py starts the two ranks
    # itself (before any GPU call of its own), relays rank 0's one line and the launcher's exit code.  gloo here
    # because the box has one GPU; on a multi-GPU node the same command runs RCCL inside the library.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SSYM_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-secondary", "--sources", "512", "--targets", "256", "--frames", "64"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["sources_per_gpu"] == 256
    assert line["config"]["indices_equal_planted"] is True and len(line["config"]["per_rank"]) == 2


def test_bench_self_launch_relays_a_failing_rank():
    # a rank that dies makes the whole run exit non-zero and print no record (SSYM_BENCH_FAIL_RANK makes rank 1 exit
    # before it joins the process group; the launcher then ends rank 0 as well)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SSYM_BENCH_BACKEND"] = "gloo"
    env["SSYM_BENCH_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-secondary", "--sources", "128", "--targets", "64", "--frames", "16"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
