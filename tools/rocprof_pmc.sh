#!/bin/bash
# One PMC pass of one command on the GPU box: per-dispatch counters -> per-kernel sums under gpurun_out/<name>/.
#   usage: tools/rocprof_pmc.sh <name> "<counters>" <kernel-regex> <python script> [args ...]
# Counters go in their own run (kernel trace only - never with sys/runtime/hip/hsa tracing); SQ has 8 slots, TCC 4
# (FETCH_SIZE costs 3, WRITE_SIZE 2: separate passes), GRBM 2.  The program itself follows `--`; everything runs under `timeout`.
set -uo pipefail
name=$1; counters=$2; kre=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout ${ROCPROF_TIMEOUT:-300} rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$out" -o "$name" -- python3 "$root/$1" "${@:2}" > "$out/stdout.log" 2>&1
echo "rocprofv3 exit $?"
f=$(find "$out" -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { tail -5 "$out/stdout.log"; exit 1; }
python3 - "$f" "$kre" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
kre = re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if not kre.search(k): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    n = len(calls[k])
    print(k[:90], "dispatches", n)
    for cn, v in sorted(c.items()):
        print(f"   {cn:32s} total {v:.6g}   per dispatch {v / n:.6g}")
PY
