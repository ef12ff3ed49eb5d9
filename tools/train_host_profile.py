#!/usr/bin/env python3
"""Where the HOST time of a small-batch training step goes (cProfile over 100 steps after warm-up).  usage: tools/train_host_profile.py [cells]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scldm_amd.nnets import DiT
from scldm_amd.training import train_step
from scldm_amd.transport import create_transport

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = DiT(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4,
        layernorm_eps=1e-8, class_vocab_sizes={"cell_line": 4, "gene": 2024}, cfg_dropout_prob=0.8, condition_strategy="joint").cuda().train()
for p in m.parameters():
    if p.requires_grad and float(p.detach().abs().sum()) == 0:
        torch.nn.init.normal_(p, std=0.02)
m.precision = "bf16"
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
x1 = torch.randn(B, 16, 16, device="cuda")
cond = {"cell_line": torch.randint(0, 4, (B,), device="cuda"), "gene": torch.randint(0, 2024, (B,), device="cuda")}
for _ in range(10):
    train_step(m, tr, opt, x1, cond)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    train_step(m, tr, opt, x1, cond)
t_enq = (time.perf_counter() - t0) / 100
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 100
print(f"{B} cells: host enqueue {t_enq*1e3:.3f} ms/step, with final sync {t_all*1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    train_step(m, tr, opt, x1, cond)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
