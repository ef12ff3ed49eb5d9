#!/bin/bash
# Interleaved A/B of library builds / run-time knobs on the GPU box (round 4).  Each spec is "label|library|ENV=V ENV2=V2" (library
# relative to the repo root, environment optional); every spec runs every workload `reps` times, interleaved.
#   usage: tools/r4_ab.sh <reps> "<workload> [bench args]" spec [spec ...]
reps=$1; shift
wl=$1; shift
for r in $(seq $reps); do
  for spec in "$@"; do
    IFS='|' read -r label lib envs <<< "$spec"
    env $envs SCLDM_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {})
print('%-28s %-34s %s cells/s %8.0f  ms/step %8.2f  frac %.4f  launch_us %7.1f lpl %s' % ('$label', '$wl', j['dtype'], j['value'], j['ms_per_step'], r.get('frac', 0), r.get('avg_launch_us', 0), r.get('layers_per_launch')))"
  done
done
