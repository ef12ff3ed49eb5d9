"""Host-side vs device time of one training step (bench workload replogle_train_b1024)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from scldm_amd.training import train_step  # noqa: E402
from scldm_amd.transport import create_transport  # noqa: E402

wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"])
dev = torch.device("cuda:0")
m = bench.make_model(wl, "bf16", dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(wl["B"], 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(dev) for k, v in wl["vocab"].items()}
for _ in range(5):
    train_step(m, tr, opt, x1, cond)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    train_step(m, tr, opt, x1, cond)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per step: host enqueue {1e3 * t_host / N:.3f} ms, wall {1e3 * t_all / N:.3f} ms")
if len(sys.argv) > 1:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# ---- host-time split of one step (no device sync inside: pure enqueue cost) ----
import collections
from scldm_amd import nnets as _nn
seg = collections.Counter()
_orig_fwd, _orig_bwd = _nn._DiTTrainFn.forward, _nn._DiTTrainFn.backward
L_, h_ = m._native_handle()
_cf, _cb = L_.scldm_dit_train_forward, L_.scldm_dit_train_backward


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        seg[name] += time.perf_counter() - t
        return r
    return w


class _LibProxy:
    def __init__(self, lib):
        self._lib = lib
        self.scldm_dit_train_forward = timed("C forward", lib.scldm_dit_train_forward)
        self.scldm_dit_train_backward = timed("C backward", lib.scldm_dit_train_backward)

    def __getattr__(self, k):
        return getattr(self._lib, k)


proxy = _LibProxy(L_)
m._native_handle = lambda: (proxy, h_)
torch.cuda.synchronize()
for _ in range(N):
    t = time.perf_counter(); opt.zero_grad(set_to_none=True); seg["zero_grad"] += time.perf_counter() - t
    t = time.perf_counter(); loss = tr.training_losses(m, x1, {"condition": cond})["loss"].mean(); seg["training_losses (incl. fwd)"] += time.perf_counter() - t
    t = time.perf_counter(); loss.backward(); seg["backward (incl. C)"] += time.perf_counter() - t
    t = time.perf_counter(); opt.step(); seg["opt.step"] += time.perf_counter() - t
torch.cuda.synchronize()
for k, v in seg.items():
    print(f"{k:32s} {1e3 * v / N:.3f} ms/step (host)")
