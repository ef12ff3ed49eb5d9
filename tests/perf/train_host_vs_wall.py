import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from scldm_amd.training import train_step
from scldm_amd.transport import create_transport
dev = torch.device("cuda:0")
for n_embed, n_layer, n_head, B in ((1024, 24, 16, 256), (1024, 24, 16, 128), (512, 12, 8, 128), (512, 12, 8, 512), (1024, 24, 16, 1024)):
    wl = dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=B, shape=dict(n_embed=n_embed, n_layer=n_layer, n_head=n_head))
    m = bench.make_model(wl, "bf16", dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(B, 16, 16, generator=g).to(dev)
    cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
    for _ in range(3):
        train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize()
    N = 8
    t0 = time.perf_counter()
    for _ in range(N):
        train_step(m, tr, opt, x1, cond)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{n_embed} x {n_layer}, {B} cells: host enqueue {1e3 * t_host / N:.2f} ms per step, wall {1e3 * t_all / N:.2f} ms")
    del m, opt
