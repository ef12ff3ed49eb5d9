"""DiT-L training step (generic path): bf16 arrays + bgemm kernels (256-tile kernel on / off) vs fp32 arrays + hgemm_kernel."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
if len(sys.argv) > 2 and sys.argv[1] == "run":
    import torch, bench
    dev = torch.device("cuda:0")
    wl = dict(bench.TRAIN_WORKLOADS["replogle_train_ditl_b256"])
    wl["B"] = int(sys.argv[2])
    dt, loss = bench.time_training(wl, "bf16", dev, 6, 2, False, 1)
    print(f"B={wl['B']} BF16_SOURCES={os.environ.get('SCLDM_TRAIN_BF16_SOURCES', '1')} BGEMM256={os.environ.get('SCLDM_BGEMM256', '1')}: "
          f"ms/step {1e3 * dt / 6:.2f}  cells/s {wl['B'] / (dt / 6):.0f}  loss {loss:.4f}")
else:
    for B in (sys.argv[1:] or ["256"]):
        for src, big in (("1", "1"), ("1", "0"), ("0", "1"), ("1", "1")):
            subprocess.run([sys.executable, __file__, "run", B], env=dict(os.environ, SCLDM_TRAIN_BF16_SOURCES=src, SCLDM_BGEMM256=big))
