"""A/B of the two-stream prediction loop (scldm_amd.sampling.generate_cells_stream): serial chain vs the pipeline.  (Round 6 also ran the
second stream at the lowest HIP priority - hipStreamCreateWithPriority through ctypes: slower than serial, profiles/r6_gen_stream_ab.txt.)  usage: python tests/perf/gen_stream_ab.py [workload] [n_genes]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from scldm_amd.datamodule import dense_to_csr, to_host
from scldm_amd.sampling import SizeFactorSampler, generate_cells_stream, sample_latents

wl_name = sys.argv[1] if len(sys.argv) > 1 else "dentate_b512_euler50"
n_genes = int(sys.argv[2]) if len(sys.argv) > 2 else 17002
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS[wl_name])
B = wl["B"]
m = bench.make_model(wl, "bf16", dev)
vae = bench.make_vae(n_genes, dev)
vae.precision = "fp16"
smp = SizeFactorSampler(bench.synthetic_vocabulary_encoder(wl["vocab"], wl["strategy"]), wl["strategy"], dev)
g = torch.Generator().manual_seed(21)
cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
scales = {k: wl["scale"] for k in wl["vocab"]}
genes = torch.arange(n_genes, device=dev).repeat(B, 1)
genes2 = torch.cat([genes, genes])
steps = wl["evals"] + 1 if wl["method"] == "euler" else wl["evals"] // 2 + 1
K = 8

def serial():
    for _ in range(K):
        sf = smp.sample(cond, B)
        z = sample_latents(m, torch.randn((B, 16, 16), device=dev), cond, scales, steps, wl["method"])
        lib = torch.exp(sf).view(-1, 1)
        to_host(*dense_to_csr(vae.decode_sample(z, genes2, torch.cat([lib, lib]))), z)

def piped(merge=1):
    for _ in generate_cells_stream(m, vae, [cond] * K, scales, genes, steps, wl["method"], size_factor_sampler=smp, merge_batches=merge):
        pass

for name, fn in (("serial", serial), ("pipeline", piped), ("pipeline, 2 batches per solve", lambda: piped(2)), ("pipeline, 4 batches per solve", lambda: piped(4)),
                 ("serial", serial), ("pipeline", piped), ("pipeline, 2 batches per solve", lambda: piped(2)), ("pipeline, 4 batches per solve", lambda: piped(4))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K)
    dt = statistics.median(ts)
    print(f"{wl_name} {name:32s} {1e3 * dt:7.3f} ms per batch  {B / dt:9.0f} cells/s   ({', '.join(f'{1e3 * t:.2f}' for t in ts)})")
