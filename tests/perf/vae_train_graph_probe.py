"""Does the VAE optimisation step capture into a HIP graph, and what does a replay cost against the eager step?
usage: vae_train_graph_probe.py [cells=32] [n_genes=17002] [S=6147]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from scldm_amd.distributions import log_nb_positive

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 17002
S = int(sys.argv[3]) if len(sys.argv) > 3 else 6147
STAGE = sys.argv[4] if len(sys.argv) > 4 else "all"      # fwd | loss | bwd | all : how much of the step is inside the graph
dev = torch.device("cuda:0")
vae = bench.make_vae(G, dev).train()
g = torch.Generator().manual_seed(5)
counts = torch.poisson(torch.full((B, G), 0.5), generator=g).to(dev)
genes = torch.arange(G, device=dev).repeat(B, 1)
gs = torch.stack([torch.sort(torch.randperm(G, generator=g)[:S]).values for _ in range(B)]).to(dev)
cs = counts.gather(1, gs)
lib = counts.sum(1, keepdim=True)
opt = bench.make_optimizer(vae.parameters(), 1e-3)

def body():
    opt.zero_grad(set_to_none=True)
    params, _ = vae(counts, genes, lib, cs, gs)
    if STAGE == "fwd":
        return params["mu"].detach().sum()
    loss = (-log_nb_positive(counts, params["mu"], params["theta"])).sum(1).mean()
    if STAGE == "loss":
        return loss.detach()
    loss.backward()
    if STAGE == "bwd":
        return loss.detach()
    opt.step()
    return loss.detach()

def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

# (everything on a non-default stream: gradient accumulators created on the legacy default stream break a later capture)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    print(f"{B} cells eager ({STAGE}): {timeit(body):.3f} ms/step", flush=True)
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    loss = body()
print("captured", flush=True)
def replay():
    opt.refresh_hyper() if hasattr(opt, "refresh_hyper") else None
    graph.replay()
print(f"{B} cells graph replay: {timeit(replay):.3f} ms/step   loss {float(loss):.2f}")
