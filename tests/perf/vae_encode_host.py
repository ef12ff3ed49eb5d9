"""Host (enqueue) time against wall time of TransformerVAE.encode at the dentate shape.  usage: vae_encode_host.py [B] [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_abi_cpu import _build_vae
G, S = 17002, 6147
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vae = _build_vae(G).cuda().eval()
vae.precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
gen = torch.Generator(device="cuda").manual_seed(B)
counts = torch.poisson(torch.full((B, S), 1.5, device="cuda"), generator=gen) + 1
genes = (torch.arange(S, device="cuda").unsqueeze(0) * 2 + torch.arange(B, device="cuda").unsqueeze(1)) % G + 1
with torch.no_grad():
    for _ in range(5): vae.encode(counts, genes)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): vae.encode(counts, genes)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tn = time.perf_counter()
    for _ in range(n): vae._native()
    tn = (time.perf_counter() - tn) / n
    torch.cuda.synchronize()
print(f"B={B} {vae.precision}: host enqueue {1e3*(t1-t0)/n:.3f} ms/call, wall {1e3*(t2-t0)/n:.3f} ms/call ({B*n/(t2-t0):.0f} cells/s); _native() alone {1e3*tn:.3f} ms")
