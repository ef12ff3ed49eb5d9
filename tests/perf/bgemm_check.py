"""DiT-L training step (generic path) with bf16 arrays + bgemm_kernel vs fp32 arrays + hgemm_kernel: step time A/B."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
if len(sys.argv) > 1:
    import torch, bench
    dev = torch.device("cuda:0")
    wl = dict(bench.TRAIN_WORKLOADS[sys.argv[1]])
    dt, loss = bench.time_training(wl, "bf16", dev, 6, 2, False, 1)
    print(f"{sys.argv[1]} SCLDM_TRAIN_BF16_SOURCES={os.environ.get('SCLDM_TRAIN_BF16_SOURCES', '1')}: ms/step {1e3 * dt / 6:.2f}  cells/s {wl['B'] / (dt / 6):.0f}  loss {loss:.4f}")
else:
    for src in ("1", "0", "1"):
        subprocess.run([sys.executable, __file__, "replogle_train_ditl_b256"], env=dict(os.environ, SCLDM_TRAIN_BF16_SOURCES=src))
