"""Is the fused DiT kernel power-limited?  Same instruction stream, three data fills: the bench's random weights / latents, all-zero
weights (every MFMA multiplies zeros: minimal switching), and random weights scaled by 1e-3.  Identical cycle counts; any difference
in wall time is clock (DVFS).  usage: power_probe.py [cells] [evals]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from bench import WORKLOADS, make_model, make_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wl = dict(WORKLOADS["dentate_b4096_euler100"]); wl["B"] = B; wl["evals"] = evals
dev = torch.device("cuda")
precs = os.environ.get("PROBE_PRECS", "bf16,fp16").split(",")
fills = os.environ.get("PROBE_FILLS", "random,zero,random_x1e-3,random").split(",")
for prec in precs:
    for fill in fills:
        m = make_model(wl, prec, dev)
        with torch.no_grad():
            for p in m.parameters():
                if fill == "zero": p.zero_()
                elif fill == "random_x1e-3": p.mul_(1e-3)
        z2, cond2, scales = make_inputs(wl, B, dev, 1)
        if fill == "zero": z2.zero_()
        m.sample_ode_cfg(z2, cond2, scales, evals + 1, "euler"); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); m.sample_ode_cfg(z2, cond2, scales, evals + 1, "euler"); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[1]
        print(f"{prec:5s} fill {fill:13s}: {1e3 * dt:8.2f} ms per solve  {B / dt:9.0f} cells/s  (x{(3 * B * 210.76e6 * evals / dt) / 2.5e15:.3f} of 2.5 PF)", flush=True)
        del m
