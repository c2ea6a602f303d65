#!/bin/bash
# tools/ragged_traffic.sh <tag> -- fabric traffic (FETCH_SIZE / WRITE_SIZE passes) and kernel times of the ragged search's filter
# launches; run twice to compare task orders (SSYM_SP_PAIRBLOCK=0: all source pairs under one target group).
root=${GRAFT_REPO_ROOT:-$(pwd)}; tag=${1:-traffic}; out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
py=$root/tools/ragged_profile_cmd.py
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $py 4096 5 40 12 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 $py 4096 5 40 12 > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $py 4096 5 40 12 > $out/trace.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"dtw_filter_sp_kernelILi(\d)", r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0.0
for c in sorted(agg):
    v = {k: sum(x) / len(x) for k, x in agg[c].items()}
    mb = (v.get("FETCH_SIZE", 0) * 2 + v.get("WRITE_SIZE", 0)) * 1024 / 1e6
    tot += mb
    hit = v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1)
    print("class %s tiles: read %.1f MB + write %.1f MB = %.1f MB per launch, L2 hit %.3f" % (c, v.get("FETCH_SIZE", 0) * 2048 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6, mb, hit))
print("per search: %.1f MB" % tot)
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dtw_filter_sp" in r["Name"]:
            print("  %s avg %.4f ms" % (re.search(r"ILi\d", r["Name"]).group(0), float(r["AverageNs"]) / 1e6))
PY
