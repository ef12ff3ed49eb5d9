#!/bin/bash
# A/B of library builds on the GPU box: for every scldm_amd/libx_*.so (and the default library) the default bench workload,
# `reps` times each, interleaved (same process conditions); prints cells/s and the fused kernel's mean launch time.
#   usage: tools/exp_libs.sh [reps] [extra bench args...]
reps=${1:-2}; shift || true
libs="scldm_amd/libscldm_hip.so $(ls scldm_amd/libx_*.so 2>/dev/null)"
for r in $(seq $reps); do
  for lib in $libs; do
    SCLDM_LIB=$PWD/$lib timeout 300 python bench.py --steps 2 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {})
print('%-34s %s cells/s %7.0f  frac %.4f  launch_us %7.1f' % ('$lib', j['dtype'], j['value'], r.get('frac', 0), r.get('avg_launch_us', 0)))"
  done
done
