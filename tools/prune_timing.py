"""tools/prune_timing.py -- the early-abandoning filter (SSYM_DTW_PRUNE) beside the full one on the same
grid: results must be identical; prints both timings and the share of DP cells the pruned run swept."""
import os, sys, time
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soundsym_amd import Engine, synth


def run(n, m, f, d, seed, planted=True, reps=5, band=-1):
    g = synth.make_grid(n, m, f, d, seed)
    src = g.sources
    tgt = g.targets if planted else synth.make_grid(m, 1, f, d, seed + 77).sources
    e = Engine(metric="dtw", dtype="f32", device=0, band=band)
    sd = torch.from_numpy(np.ascontiguousarray(src).reshape(-1)).cuda()
    td = torch.from_numpy(np.ascontiguousarray(tgt).reshape(-1)).cuda()
    so = np.arange(n + 1, dtype=np.uint64) * f
    to = np.arange(m + 1, dtype=np.uint64) * f
    dd, q = e.dictionary(sd, so, d), e.queries(td, to, d)
    oi = torch.empty(m, dtype=torch.int32, device="cuda"); oc = torch.empty(m, dtype=torch.float64, device="cuda")
    pi = torch.empty(m, dtype=torch.int32, device="cuda"); pc = torch.empty(m, dtype=torch.float64, device="cuda")
    out = {}
    for prune in (False, True):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.match(dd, q, out_idx=pi if prune else oi, out_cost=pc if prune else oc, prune=prune)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        out[prune] = (best, e.timings())
    same = bool(torch.equal(oi, pi) and torch.equal(oc, pc))
    tm = out[True][1]
    full_cells = n * m * (f * f if band < 0 else f * (2 * band + 1))
    print(f"{n}x{m}x{f}f x{d}d band={band} planted={planted}: full {out[False][0]:.2f} ms  pruned {out[True][0]:.2f} ms "
          f"(thresholds {tm['prune_ms']:.2f}, filter {tm['main_ms']:.2f}, select {tm['select_ms']:.2f}, refine {tm['refine_ms']:.2f}, "
          f"refined {tm['n_refined']})  cells swept {tm['n_filter_cells'] / full_cells:.3f} of full  identical={same}", flush=True)
    assert same
    if planted:
        assert np.array_equal(pi.cpu().numpy(), g.planted)
    e.close()


if __name__ == "__main__":
    run(1024, 1024, 64, 13, 0x5EED0002)
    run(4096, 4096, 128, 13, 0x5EED0003)
    run(4096, 4096, 128, 13, 0x5EED0003, planted=False)
    run(2048, 2048, 256, 13, 0x5EED0013)
    run(4096, 4096, 40, 13, 0x5EED0023)
    run(1024, 1024, 512, 13, 0x5EED0033, reps=3)
    run(4096, 4096, 256, 40, 0x5EED0005, band=32, reps=3)
    run(4096, 4096, 256, 40, 0x5EED0005, band=32, reps=3, planted=False)
