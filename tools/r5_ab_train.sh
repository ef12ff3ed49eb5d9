#!/bin/bash
# Interleaved A/B of library variants on the training step (round 5): tools/r5_ab_train.sh <reps> <cells> <precision> name [name ...]
# ("base" = the in-tree library, anything else = build/libx_<name>.so); prints ms/step per run and the backward / wgrad kernel means
# of one rocprofv3 --kernel-trace --stats pass per variant.
reps=$1; cells=$2; prec=$3; shift 3
root=$PWD
lib() { if [ "$1" = base ]; then echo $root/scldm_amd/libscldm_hip.so; else echo $root/build/libx_$1.so; fi; }
for r in $(seq $reps); do
  for n in "$@"; do
    printf "%-14s " $n; SCLDM_LIB=$(lib $n) python tools/train_bench.py $cells 30 $prec 2>&1 | grep "ms/step" | sed -e 's/, loss.*//'
  done
done
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/prof_$n
  SCLDM_LIB=$(lib $n) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python3 $root/tools/train_bench.py $cells 25 $prec > /dev/null 2>&1
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
  printf "%-14s " $n; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda key: next((float(r["AverageNs"]) / 1e3 for r in rows if key in r["Name"]), float("nan"))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 28 / 1e3   # 3 warm-up + 25 timed steps
print("bwd %.1f us  wgrad %.1f us  fwdREC %.1f us  pack %.1f us  | all kernels %.0f us/step" % (pick("dit_backward_kernel"), pick("wgrad_bf16_kernel"), pick("dit_forward_kernel"), pick("pack_jobs"), tot))
PY
done
