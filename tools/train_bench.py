#!/usr/bin/env python3
"""Time one flow-matching training step (Transport.training_losses -> backward -> AdamW) of the base DiT on synthetic latents.
usage: python tools/train_bench.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scldm_amd.nnets import DiT
from scldm_amd.transport import create_transport

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
torch.manual_seed(0)
m = DiT(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4,
        layernorm_eps=1e-8, class_vocab_sizes={"cell_line": 4, "gene": 2024}, cfg_dropout_prob=0.8, condition_strategy="joint").cuda().train()
for p in m.parameters():   # adaLN-Zero init would zero the whole backward: use non-degenerate weights
    if p.requires_grad and float(p.detach().abs().sum()) == 0:
        torch.nn.init.normal_(p, std=0.02)
m.precision = precision
if os.environ.get("TRAIN_BENCH_TORCH_ADAMW") == "1":
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
else:
    from scldm_amd.optim import AdamW
    opt = AdamW(m.parameters(), lr=1e-4)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
x1 = torch.randn(B, 16, 16, device="cuda")
cond = {"cell_line": torch.randint(0, 4, (B,), device="cuda"), "gene": torch.randint(0, 2024, (B,), device="cuda")}

def step():
    opt.zero_grad(set_to_none=True)
    loss = tr.training_losses(m, x1, {"condition": cond})["loss"].mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flops = 3 * 210_763_776 * B
print(f"{precision} batch {B}: {dt*1e3:.2f} ms/step, {B/dt:.0f} cells/s, {flops/dt/1e12:.1f} TFLOP/s (3x fwd FLOPs), loss {float(loss.detach()):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
