"""Child process of test_gpu_train.py::test_lds_dma_gemm_is_bit_identical_to_the_register_staged_gemm: one bf16 training step of a
DiT-L-wide model with the GEMM route knobs of the environment (they are read when the library is loaded), results to argv[1]."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from test_gpu_train import build, hip_training_step

out, n_embed, n_head, n_layer, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
vocab = {"cell_line": 4, "gene": 2024}
gen = torch.Generator().manual_seed(1000 + n)
x1, x0 = torch.randn(n, 16, 16, generator=gen), torch.randn(n, 16, 16, generator=gen)
t = torch.rand(n, generator=gen)
cond = {k: torch.randint(0, v + 1, (n,), generator=gen) for k, v in vocab.items()}
m, sd, cfg = build(vocab, "joint", n_layer, 55, n_embed=n_embed, n_head=n_head)
sd = {k: (v * (256.0 / n_embed) ** 0.5 if v.dim() == 2 else v) for k, v in sd.items()}
m.load_state_dict(sd, strict=True)
m = m.cuda()
m.precision = "bf16"
res = {}
for r in range(reps):
    terms = hip_training_step(m, x1, x0, t, cond)
    cur = {"pred": terms["pred"].detach().cpu()}
    cur.update({k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
    if r == 0:
        res = cur
    else:   # race screen: every repetition reproduces the first bit for bit (label tables: atomics, compared loosely by the parent)
        for k, v in cur.items():
            if k.startswith("class_embeddings"):
                continue
            assert torch.equal(v, res[k]), f"repetition {r}: {k} differs from the first run"
torch.save(res, out)
