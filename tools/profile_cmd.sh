#!/bin/bash
# tools/profile_cmd.sh <tag> <python args...> -- rocprofv3 kernel trace of `python3 <args>` on the GPU box
# (raw CSVs under gpurun_out/<tag>/trace; top kernels printed)
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
args=()
for a in "$@"; do case "$a" in *.py) args+=("$root/$a");; *) args+=("$a");; esac; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "${args[@]}" > $out/trace.log 2>&1 || echo "trace pass failed"
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200
