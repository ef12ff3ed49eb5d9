#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused DiT block kernel (debug build with s_memtime stamps).

  SCLDM_HIPCC_FLAGS=-DSCLDM_PHASE_TIMING OUT=scldm_amd/libscldm_hip_dbg.so ./build.sh
  SCLDM_LIB=$PWD/scldm_amd/libscldm_hip_dbg.so python tools/phase_timing.py [n_fwd] [precision]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scldm_amd import _lib  # noqa: E402
from __graft_entry__ import _random_dit  # noqa: E402

NAMES = ["LN1", "Q pass", "K pass", "scores+softmax", "V pass+PV", "barrier(AO)", "proj+residual", "LN2",
         "MLP (all chunks)", "gated residual", "| chunk0 W12+silu", "| chunk0 barrier", "| chunk0 c_proj", "LAYER total"]
PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (0, 10)]

n_fwd = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n_layer = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = _random_dit(n_layer=n_layer).cuda()
m.precision = prec
L, h = m._native()
n_blocks = (n_fwd * 16 + 31) // 32 + 2  # upper bound: small launches run 32-token tiles (NTT = 1), i.e. twice the workgroups of NTT = 2
# (round 6: sized for NTT = 2 only, the stamps of a <= 512-sample-forward launch ran past the buffer - a GPU memory fault)
buf = torch.zeros(n_blocks * 8 * 32, dtype=torch.int64, device="cuda")
_lib.check(L.scldm_dit_set_debug_buffer(h, buf.data_ptr()), "set_debug_buffer")
x = torch.randn(n_fwd, 16, 16, device="cuda")
t = torch.rand(n_fwd, device="cuda")
lab = torch.randint(0, 14, (n_fwd,), device="cuda")
for _ in range(2):
    m(x, t, {"clusters": lab})
torch.cuda.synchronize()
st = buf.view(n_blocks * 8, 32).cpu()
st = st[st[:, 14] > 0]
print(f"n_fwd={n_fwd} precision={prec} layers={n_layer} waves recorded={st.shape[0]}  (s_memtime ticks; phases are those of layer $SCLDM_DBG_LAYER, default the middle layer)")
whole = (st[:, 14] - st[:, 0]).double()
print(f"{'start -> kernel end':24s} mean {whole.mean():10.0f}  min {whole.min():10.0f}  max {whole.max():10.0f}")
tot = (st[:, 10] - st[:, 0]).double()
for name, (a, b) in zip(NAMES, PAIRS):
    d = (st[:, b] - st[:, a]).double()
    print(f"{name:24s} mean {d.mean():10.0f}  ({100 * d.mean() / tot.mean():5.1f}%)  min {d.min():8.0f} max {d.max():8.0f}")
print("LN2 detail: proj-done->stats written", int((st[:, 16] - st[:, 7]).double().mean()), " stats barrier(s)+combine", int((st[:, 19] - st[:, 16]).double().mean()),
      " modulate+store", int((st[:, 20] - st[:, 19]).double().mean()), " final barrier", int((st[:, 8] - st[:, 20]).double().mean()))
print("LN1 detail: start->prologue issued (x/mod/ring loads)", int((st[:, 21] - st[:, 0]).double().mean()), " ->stats written (incl. load wait + MOD publish)", int((st[:, 22] - st[:, 21]).double().mean()),
      " barrier+combine", int((st[:, 25] - st[:, 22]).double().mean()), " modulate+store", int((st[:, 26] - st[:, 25]).double().mean()), " final barrier", int((st[:, 1] - st[:, 26]).double().mean()))
span = (st[:, 14].max() - st[:, 0].min())
print("launch span (first start -> last end):", int(span))

# within-workgroup skew: spread (max - min over the 4 waves) of each stamp, averaged over workgroups
full = buf.view(n_blocks, 8, 32).cpu()
nw = int((full[0, :, 14] > 0).sum())
full = full[:, :nw]
ok = (full[:, :, 14] > 0).all(dim=1)
full = full[ok].double()
print("within-workgroup spread (max-min over the 4 waves) per stamp:")
labels = {0: "start", 1: "LN1 done", 2: "Q done", 3: "K done", 4: "softmax done", 5: "V+PV done", 6: "after AO barrier", 7: "proj done",
          8: "LN2 done", 11: "chunk0 W12 done", 12: "chunk0 barrier", 13: "chunk0 cproj done", 9: "MLP done", 10: "residual done"}
for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 9, 10):
    sp = full[:, :, k].max(dim=1).values - full[:, :, k].min(dim=1).values
    print(f"  stamp {k:2d} {labels[k]:20s} mean spread {sp.mean():9.0f}  p90 {sp.quantile(0.9):9.0f}")
