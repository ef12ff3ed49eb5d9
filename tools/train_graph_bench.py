import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from scldm_amd.nnets import DiT
from scldm_amd.training import GraphedTrainStep, train_step
from scldm_amd.transport import create_transport
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
torch.manual_seed(0)
m = DiT(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4,
        layernorm_eps=1e-8, class_vocab_sizes={"cell_line": 4, "gene": 2024}, cfg_dropout_prob=0.8, condition_strategy="joint").cuda().train()
for p in m.parameters():
    if p.requires_grad and float(p.detach().abs().sum()) == 0:
        torch.nn.init.normal_(p, std=0.02)
m.precision = prec
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True, capturable=True)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
x1 = torch.randn(B, 16, 16, device="cuda")
cond = {"cell_line": torch.randint(0, 4, (B,), device="cuda"), "gene": torch.randint(0, 2024, (B,), device="cuda")}
for _ in range(5):
    train_step(m, tr, opt, x1, cond)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    l = train_step(m, tr, opt, x1, cond)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
g = GraphedTrainStep(m, tr, opt, x1, cond)
for _ in range(5):
    g(x1, cond)
torch.cuda.synchronize(); t0 = time.perf_counter()
losses = []
for _ in range(50):
    losses.append(g(x1, cond).clone())
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
print(f"{B} cells {prec}: eager {te*1e3:.3f} ms/step, graphed {tg*1e3:.3f} ms/step; loss eager {float(l):.4f} graphed first {float(losses[0]):.4f} last {float(losses[-1]):.4f}")
