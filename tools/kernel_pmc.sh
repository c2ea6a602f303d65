#!/bin/bash
# tools/kernel_pmc.sh <tag> <kernel-substring> <script.py> [args...] -- two rocprofv3 --pmc passes (SQ issue / wait
# counters with GRBM_GUI_ACTIVE, LDS counters) over `python3 <script> <args>` on the GPU box; the means per launch of the
# kernels whose name contains <kernel-substring> are printed and kept in gpurun_out/<tag>/summary.txt.  SSYM_LIB and the
# library's measurement knobs pass through from the environment.
root=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; kern=$2; shift 2
out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
py=$root/$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $out/pmc_sq -- python3 $py "$@" > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_lds -- python3 $py "$@" > $out/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $py "$@" > $out/trace.log 2>&1
python3 - "$out" "$kern" "$tag" <<'PY' | tee $out/summary.txt
import csv, glob, collections, sys
out, kern, tag = sys.argv[1:4]
vals = {}
for sub in ("pmc_sq", "pmc_lds"):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        vals[k] = sum(v) / len(v)
        print(tag, kern, k, "%.5g" % vals[k], "(mean of %d launches)" % len(v))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            print(tag, kern, "avg_ms %.4f over %s calls" % (float(r["AverageNs"]) / 1e6, r["Calls"]))
if "GRBM_GUI_ACTIVE" in vals and "SQ_ACTIVE_INST_VALU" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    print(tag, kern, "VALU busy per SIMD %.3f" % (4.0 * vals["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0)))
if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE"):
    print(tag, kern, "LDS bank conflict / idx active %.3f" % (vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
PY
