#!/usr/bin/env python3
"""tools/rccl_beside_torch.py -- the process shape of a bench.py rank at N > 1, on one GPU: torch.distributed's "nccl"
process group initialised and used FIRST (torch loads the RCCL bundled in its wheel), then the library's own
communicator (csrc/comm.hip binds RCCL at run time) and sharded steps through it, then torch's group again.
Prints which librccl files the process has mapped and whether the sharded step equals the unsharded match.
World size 1 (RCCL cannot put two ranks on one device): what this checks is that the two users of RCCL live
in one process, not the transport."""
import json
import os
os.environ.setdefault("SSYM_TEST_HOOKS", "1")      # the library reads its measurement knobs only when asked to
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from soundsym_amd import Engine, sharding, synth
from soundsym_amd.engine import comm_available


def rccl_maps():
    out = set()
    with open("/proc/self/maps") as fh:
        for ln in fh:
            if "rccl" in ln or "nccl" in ln:
                out.add(ln.split()[-1])
    return sorted(out)


torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1024, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
before = rccl_maps()

n, m, f, d = 256, 128, 48, 13
g = synth.make_grid(n, m, f, d, 0x5EED0077)
e = Engine(metric="dtw", dtype="f32")
dd = e.dictionary(torch.from_numpy(g.sources.reshape(-1)).cuda(), np.arange(n + 1, dtype=np.uint64) * f, d)
q = e.queries(torch.from_numpy(g.targets.reshape(-1)).cuda(), np.arange(m + 1, dtype=np.uint64) * f, d)
ok_bind = bool(comm_available())
comm = sharding.init_comm(e, 0, 1)
oi = torch.empty(m, dtype=torch.int32, device="cuda"); oc = torch.empty(m, dtype=torch.float64, device="cuda")
ri = torch.empty_like(oi); rc = torch.empty_like(oc)
e.match(dd, q, out_idx=ri, out_cost=rc)
same = True
for _ in range(3):
    sharding.match_sharded(e, comm, dd, q, 0, out_idx=oi, out_cost=oc)
    dist.all_reduce(t)                       # torch's communicator between the library's steps
    torch.cuda.synchronize()
    same = same and bool(torch.equal(oi, ri)) and bool(torch.equal(oc, rc))
after = rccl_maps()
planted = bool(np.array_equal(oi.cpu().numpy().view(np.uint32).astype(np.int64), g.planted))
comm.close()
dist.destroy_process_group()
print(json.dumps({"comm_available": ok_bind, "sharded_equals_match": same, "planted": planted,
                  "torch_allreduce": float(t[0].item()), "rccl_mapped_before": before, "rccl_mapped_after": after}))
