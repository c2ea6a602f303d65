#!/bin/bash
# tools/profile_bench.sh <tag> [bench.py args...] -- rocprofv3 passes over `python3 bench.py <args>` on the GPU box:
# one kernel-trace/stats pass and separate PMC passes (counters never share a run with tracing), raw CSVs under
# gpurun_out/<tag>/, digested by tools/summarize_profile.py into profiles/.  Default args = the headline command.
# SSYM_PROFILE_PY=<script> profiles another script of this repository instead of bench.py.
set -o pipefail
tag=${1:-prof}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- --steps 5 --warmup 2 --no-cpu-baseline --no-secondary; fi
py=$root/${SSYM_PROFILE_PY:-bench.py}
echo "python3 ${SSYM_PROFILE_PY:-bench.py} $*" > $out/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $py "$@" > $out/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $py "$@" > $out/pmc_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $py "$@" > $out/pmc_write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/pmc_sq -- python3 $py "$@" > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_lds -- python3 $py "$@" > $out/pmc_lds.log 2>&1 || echo "lds pass failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_misc -- python3 $py "$@" > $out/pmc_misc.log 2>&1 || echo "misc pass failed"
tail -1 $out/trace.log
echo "profile passes done: $out"
