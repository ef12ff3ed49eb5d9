#!/bin/bash
# run-time knob sweep of the fused DiT kernel on the default workload (same box, back to back)
B="timeout 300 python bench.py --steps 2 --no-extra --no-cpu-baseline"
run() { printf "%-28s" "$1"; shift; env "$@" $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {})
print(j['dtype'], 'cells/s %.0f' % j['value'], 'frac %.4f' % r.get('frac', 0), 'launch_us %.1f' % r.get('avg_launch_us', 0), 'lpl', r.get('layers_per_launch'))"; }
run default X=1
run LPL2 SCLDM_LPL=2
run LPL3 SCLDM_LPL=3
run GROUPS2 SCLDM_GROUPS=2
run GROUPS2_LPL2 SCLDM_GROUPS=2 SCLDM_LPL=2
run PAD128 SCLDM_PAD128=1
run default X=1
