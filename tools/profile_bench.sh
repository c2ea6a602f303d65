#!/bin/bash
# tools/profile_bench.sh <tag> -- rocprofv3 passes over `python3 bench.py` on the GPU box:
# one kernel-trace/stats pass and separate PMC passes (counters never share a run with tracing),
# raw CSVs under gpurun_out/<tag>/, digest by tools/summarize_profile.py into profiles/.
set -o pipefail
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
cmd="python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $cmd > $out/pmc_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $cmd > $out/pmc_write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/pmc_sq -- $cmd > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_misc -- $cmd > $out/pmc_misc.log 2>&1 || echo "misc pass failed"
tail -1 $out/trace.log
echo "profile passes done: $out"
