"""Debug / A-B script for the fused bf16 training path: per-tensor gradient error vs the fp32 CPU oracle with the fused path
on and off (SCLDM_TRAIN_FUSED), and step timing at the bench batch size.  Test infrastructure (imports oracle/)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from test_gpu_train import build, hip_training_step  # noqa: E402
from oracle.train import FROZEN, training_grads  # noqa: E402


def errors(fused: bool, n=48, n_layer=8):
    os.environ["SCLDM_TRAIN_FUSED"] = "1" if fused else "0"
    vocab = {"cell_line": 4, "gene": 2024}
    m, sd, cfg = build(vocab, "joint", n_layer, 81)
    m.precision = "bf16"
    gen = torch.Generator().manual_seed(9)
    x1, x0 = torch.randn(n, 16, 16, generator=gen), torch.randn(n, 16, 16, generator=gen)
    t = torch.rand(n, generator=gen)
    cond = {k: torch.randint(0, v + 1, (n,), generator=gen) for k, v in vocab.items()}
    terms = hip_training_step(m, x1, x0, t, cond)
    loss, pred, grads, _ = training_grads(sd, cfg, x1, x0, t, cond)
    out = {"pred": float((terms["pred"].detach().cpu() - pred).abs().max() / pred.abs().max())}
    for name, p in m.named_parameters():
        if name in FROZEN:
            continue
        ref = grads[name].double()
        out[name] = float((p.grad.cpu().double() - ref).norm() / ref.norm())
    return out


def timing(fused: bool, n=1024, steps=10):
    os.environ["SCLDM_TRAIN_FUSED"] = "1" if fused else "0"
    vocab = {"cell_line": 4, "gene": 2024}
    m, sd, cfg = build(vocab, "joint", 8, 81)
    m.precision = "bf16"
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, 16, 16, device="cuda", generator=gen)
    t = torch.rand(n, device="cuda", generator=gen)
    cond = {k: torch.randint(0, v, (n,), device="cuda", generator=gen) for k, v in vocab.items()}
    tgt = torch.randn(n, 16, 16, device="cuda", generator=gen)

    def step():
        for p in m.parameters():
            p.grad = None
        out = m(x, t, cond, force_drop_ids=False)
        ((out - tgt) ** 2).mean().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "err"):
        nl = int(os.environ.get("CHECK_LAYERS", "8"))
        e1, e0 = errors(True, n_layer=nl), errors(False, n_layer=nl)
        for k in e1:
            flag = "  <<<" if e1[k] > 3e-2 else ""
            print(f"{k:50s} fused {e1[k]:.3e}   generic {e0[k]:.3e}{flag}")
    if what in ("all", "time"):
        print(f"step ms: fused {timing(True):.3f}   generic {timing(False):.3f}")
    if what == "fused":     # profile target: only the fused path
        print(f"step ms: fused {timing(True, steps=20):.3f}")
