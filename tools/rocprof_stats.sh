#!/bin/bash
# Kernel-trace statistics of one command on the GPU box, written under gpurun_out/<name>/ (copy the summary into profiles/).
#   usage: tools/rocprof_stats.sh <name> <python args ...>        e.g.  tools/rocprof_stats.sh r2_vae tests/perf/vae_profile.py 17002 6147 1024
# rocprofv3 needs a writable cwd and TMPDIR; the program itself comes straight after `--` (no env / bash -c hop); the whole
# thing runs under `timeout` because a profiler that does not exit must not eat the GPU budget.
set -uo pipefail
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout ${ROCPROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- python3 "$root/$1" "${@:2}" > "$out/stdout.log" 2>&1
echo "rocprofv3 exit $?"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-180 "$f" | head -${ROCPROF_ROWS:-14}
