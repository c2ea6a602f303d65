#!/bin/bash
# quick LDS-conflict PMC of the refcos main kernel for a given library
root=$GRAFT_REPO_ROOT; tag=$1; lib=$2
out=$root/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
[ -n "$lib" ] && export SSYM_LIB=$lib
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_lds -- python3 $root/tools/refcos_profile_cmd.py 4 > $out/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/pmc_sq -- python3 $root/tools/refcos_profile_cmd.py 4 > $out/pmc_sq.log 2>&1
python3 - <<PY
import csv,glob,collections
for sub in ("pmc_lds","pmc_sq"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "refcos_mfma_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print("$tag", k, "%.4g"%(sum(v)/len(v)))
PY
