"""Training step wall time vs batch size (host-bound if the time does not grow with the batch).
   python tests/perf/train_scale.py            one process, models built one after the other, the previous one released first
   python tests/perf/train_scale.py nogc       the round-2 behaviour: no explicit release / collection between sizes (see HISTORY 4.4a:
                                               Python's cyclic collector then frees the previous model - ~30 hipFree, each a device
                                               synchronisation - in the middle of the next size's timed loop)"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from scldm_amd.training import train_step
from scldm_amd.transport import create_transport
dev = torch.device("cuda:0")
m = opt = None
for B in (256, 512, 1024, 2048, 4096):
    if "nogc" not in sys.argv:
        m = opt = None            # torch optimizers sit in reference cycles: drop the previous model + optimizer NOW, not whenever
        gc.collect()              # the cyclic collector runs (its __del__ destroys the native handle: ~30 synchronising hipFree)
        torch.cuda.synchronize()
    wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"]); wl["B"] = B
    m = bench.make_model(wl, "bf16", dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(B, 16, 16, generator=g).to(dev)
    cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
    for _ in range(5):
        train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B:5d}  {1e3 * dt:.3f} ms/step  {B / dt / 1e3:.1f} k cells/s")

# activation record actually allocated per cell per layer (fused bf16 vs generic)
from scldm_amd import _lib
m = bench.make_model(dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"]), "bf16", dev).train()
L_, h_ = m._native_handle()
for prec, name in ((1, "bf16 (fused)"), (0, "fp32 (generic)")):
    sb, wb = L_.scldm_dit_train_saved_bytes_for(h_, 1024, prec), L_.scldm_dit_train_workspace_bytes_for(h_, 1024, prec)
    print(f"{name}: saved {sb / 2**20:.0f} MiB = {sb / 1024 / 8 / 1024:.1f} KiB per cell per layer, workspace {wb / 2**20:.0f} MiB")
