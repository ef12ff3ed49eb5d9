python -m pytest tests/test_gpu_vae_train.py -x -q 2>&1 | tail -5
for v in 1 2 1 2; do echo "== GENE_MFMA=$v"; SCLDM_VAE_GENE_MFMA=$v python tests/perf/vae_train_bench.py 512 2>&1 | grep "^B="; done
for v in 1 2; do echo "== GENE_MFMA=$v kernel"; SCLDM_VAE_GENE_MFMA=$v ROCPROF_ROWS=4 tools/rocprof_stats.sh vg_$v tests/perf/vae_train_bench.py 512 | cut -d, -f1-4; done
