B="timeout 300 python bench.py --steps 2 --no-extra --no-cpu-baseline --precision bf16x3"
run() { echo "=== $1"; shift; env "$@" $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j.get('roofline', {})
print(j['dtype'], 'cells/s %.0f' % j['value'], 'frac %.4f' % r.get('frac', 0), 'avg_launch_us %.1f' % r.get('avg_launch_us', 0))"; }
run "x3 NTT2 default" X=1
run "x3 NTT1 2WG" SCLDM_X3_NTT=1
run "x3 NTT2 default" X=1
run "x3 NTT1 2WG" SCLDM_X3_NTT=1
SCLDM_X3_NTT=1 timeout 600 python -m pytest tests/test_gpu_dit.py -m gpu -q -x -k "bf16x3 and (golden or ragged or sampler)" 2>&1 | tail -2
