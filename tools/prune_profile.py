"""tools/prune_profile.py -- the pruned search (SSYM_DTW_PRUNE) on the bench workload, for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prune_prof -- python3 tools/prune_profile.py"""
import os, sys
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundsym_amd import Engine, synth

n = m = 4096
f, d = 128, 13
g = synth.make_grid(n, m, f, d, 0x5EED0003)
e = Engine(metric="dtw", dtype="f32", device=0)
off = np.arange(n + 1, dtype=np.uint64) * f
dd = e.dictionary(torch.from_numpy(g.sources.reshape(-1)).cuda(), off, d)
q = e.queries(torch.from_numpy(g.targets.reshape(-1)).cuda(), off, d)
oi = torch.empty(m, dtype=torch.int32, device="cuda")
oc = torch.empty(m, dtype=torch.float64, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    e.match(dd, q, out_idx=oi, out_cost=oc, prune=True)
torch.cuda.synchronize()
assert np.array_equal(oi.cpu().numpy(), g.planted)
print({k: round(float(v), 3) for k, v in e.timings().items()})
e.close()
