cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pz in 0 1 0 1; do
  SCLDM_BGEMM_PERSIST=$pz timeout 300 python bench.py --workload replogle_train_ditl_b1024 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PERSIST=$pz b1024', round(j['ms_per_step'],2), 'ms', round(j['train_tflops_per_gpu'],1), 'TF')"
done > gpurun_out/r3h_persist_ab.txt 2>&1
cat gpurun_out/r3h_persist_ab.txt
SCLDM_BGEMM_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_train.py -q -k "wider or full_depth or batched" 2>&1 | tail -3
