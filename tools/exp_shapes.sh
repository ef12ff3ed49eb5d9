mkdir -p gpurun_out
B="timeout 300 python bench.py --steps 2 --no-extra --no-cpu-baseline"
run() { echo "=== $1"; shift; env "$@" $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j.get('roofline', {})
print(j['dtype'], 'cells/s %.0f' % j['value'], 'frac %.4f' % r.get('frac', 0), 'avg_launch_us %.1f' % r.get('avg_launch_us', 0), 'lpl', r.get('layers_per_launch'))
"; }
run "bf16 default" X=1
run "bf16 default again" X=1
run "bf16 groups2" SCLDM_GROUPS=2
run "bf16 groups3" SCLDM_GROUPS=3
run "bf16 FT1 NTT2" SCLDM_FT=1
run "bf16 LPL2" SCLDM_LPL=2
B="timeout 300 python bench.py --steps 2 --no-extra --no-cpu-baseline --precision bf16x3"
run "x3 default" X=1
run "x3 FT1" SCLDM_X3_FT=1
run "x3 NTT1" SCLDM_X3_NTT=1
run "x3 groups2" SCLDM_GROUPS=2
run "x3 LPL2" SCLDM_LPL=2
B="timeout 300 python bench.py --steps 1 --no-extra --no-cpu-baseline --precision fp32"
run "fp32" X=1
