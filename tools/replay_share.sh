#!/bin/bash
# tools/replay_share.sh <tag> -- the per-rank step of a G-GPU strong-scaling run, MEASURED on one GPU: bench.py --replay-world
# G for G = 2, 4, 8 on configs[2], [3], [4] (rank 0's shard, the full dictionary's bounds replayed into the bound exchange),
# one JSON line each under gpurun_out/<tag>/, plus the plain one-GPU step of each workload for the ratio.
root=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-share}; out=$root/gpurun_out/$tag; mkdir -p $out
for wl in c3 c4 c5; do
  python3 $root/bench.py --workload $wl --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $out/${wl}_g1.json 2> $out/${wl}_g1.err || echo "$wl g1 failed"
  for g in 2 4 8; do
    python3 $root/bench.py --workload $wl --steps 40 --warmup 5 --replay-world $g > $out/${wl}_g$g.json 2> $out/${wl}_g$g.err || echo "$wl g$g failed"
  done
done
python3 - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for wl in ("c3", "c4", "c5"):
    try:
        one = json.load(open(os.path.join(out, wl + "_g1.json")))
    except Exception as e:
        print(wl, "g1 missing", e); continue
    print(f"{wl}: one GPU {one['ms_per_step']:.3f} ms per step")
    for g in (2, 4, 8):
        try:
            j = json.load(open(os.path.join(out, f"{wl}_g{g}.json")))
        except Exception as e:
            print(wl, g, "missing", e); continue
        pr = j["config"]["per_rank"][0]
        print(f"  G={g}: rank step {j['ms_per_step']:.3f} ms (filter {pr['main_ms']:.3f}, select {pr['select_ms']:.3f}, refine {pr['refine_ms']:.3f}, "
              f"collectives {pr['collective_ms']:.3f}, n_refined {pr['n_refined']}), ideal {one['ms_per_step'] / g:.3f} -> predicted efficiency "
              f"{one['ms_per_step'] / g / j['ms_per_step']:.3f}; planted ok {j['config']['indices_equal_planted']}")
PY
