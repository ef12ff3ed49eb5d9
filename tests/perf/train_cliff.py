"""Per-batch-size training-step timing with a per-phase breakdown (forward / backward / optimizer) from HIP events and the host's
enqueue time, to localise the step-time cliff at 512 cells.   usage: python tests/perf/train_cliff.py B [B2 ...]   (several sizes = one process, models built one after the other)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from scldm_amd.transport import create_transport
dev = torch.device("cuda:0")
steps = 20
def run(B):
  global host
  wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"]); wl["B"] = B
  m = bench.make_model(wl, "bf16", dev).train()
  opt = bench.make_optimizer(m.parameters(), 1e-4)
  tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
  g = torch.Generator().manual_seed(3)
  x1 = torch.randn(B, 16, 16, generator=g).to(dev)
  cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
  host = []
  def step(e=None):
      h0 = time.perf_counter()
      opt.zero_grad(set_to_none=True)
      if e: e[0].record()
      loss = tr.training_losses(m, x1, {"condition": cond})["loss"].mean()
      if e: e[1].record()
      h1 = time.perf_counter()
      loss.backward()
      if e: e[2].record()
      h2 = time.perf_counter()
      opt.step()
      if e: e[3].record()
      h3 = time.perf_counter()
      host.append((h1 - h0, h2 - h1, h3 - h2))
  for _ in range(5):
      step()
  torch.cuda.synchronize(); host.clear()
  t0 = time.perf_counter()
  for i in range(steps):
      step(ev[i])
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  f = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
  b = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
  o = sum(e[2].elapsed_time(e[3]) for e in ev) / steps
  hf, hb, ho = (1e3 * sum(h[i] for h in host) / steps for i in range(3))
  print(f"B={B:5d} {1e3*dt:.3f} ms/step | device: fwd {f:.3f} bwd {b:.3f} opt {o:.3f} | host enqueue: fwd {hf:.3f} bwd {hb:.3f} opt {ho:.3f}")

  ms = torch.cuda.memory_stats()
  print(f"        segments allocated so far {ms['segment.all.allocated']}, freed {ms['segment.all.freed']}, reserved {ms['reserved_bytes.all.current'] >> 20} MiB")
  return m

keep = []
for b in sys.argv[1:]:
    if b == "keep":          # keep the earlier models alive (separates 'old handle destroyed' from allocator effects)
        keep.append(None); continue
    mm = run(int(b))
    if keep: keep.append(mm)
