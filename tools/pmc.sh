#!/bin/bash
# usage: tools/pmc.sh <out-subdir> <binary> <args...> : three PMC passes + one kernel-trace pass
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
bin=$(realpath "$1"); shift
set -- "$bin" "$@"
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc1 -- "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- "$@" > $out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -- "$@" > $out/pmc4.log 2>&1
find $out -name "*.csv" | head -20
