"""AdamW for scldm_amd models: one HIP launch per step over every parameter tensor (csrc/optim.hip).

Drop-in for `torch.optim.AdamW` in the reference's trainer (src/scldm/models.py configure_optimizers; Hydra `_target_`): same
hyper-parameters, same arithmetic as torch's fused implementation, `state_dict()` with torch's keys (`step`, `exp_avg`,
`exp_avg_sq`), so checkpoints move between the two.  Why: torch's fused multi-tensor AdamW covers the base DiT's 84 tensors with four
launches of ~150 workgroups - ~200 us of a 2.2 ms training step on a 256-CU part; here every 4 096-element chunk is its own workgroup.
The step count lives on the device (capturable in a HIP graph: `scldm_amd.training.GraphedTrainStep`), and `found_inf` (GradScaler's
protocol, the fp16 backward's overflow flag) skips an update without a host read.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True      # torch.cuda.amp.GradScaler / scldm_amd.training.train_step hand over `found_inf`

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, amsgrad: bool = False,
                 *, maximize: bool = False, capturable: bool = True, fused: bool | None = True, foreach: bool | None = None,
                 differentiable: bool = False):
        if amsgrad or differentiable:
            raise NotImplementedError("scldm_amd.optim.AdamW: amsgrad / differentiable are not provided (no caller in the reference)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize,
                                      capturable=True, fused=True, foreach=None, differentiable=False,
                                      decoupled_weight_decay=True))   # (torch >= 2.6: AdamW is Adam with this flag; kept so a state_dict loads there as AdamW)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, "found_inf", None)
        grad_scale = getattr(self, "grad_scale", None)
        if grad_scale is not None:
            raise NotImplementedError("scldm_amd.optim.AdamW takes found_inf only: un-scale the gradients before the step (the fp16 backward does)")
        L = _lib.lib()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            step_t = group.get("_step_t")
            if step_t is None or step_t.device != dev:
                # one device counter per group; restored from a loaded state_dict's per-parameter `step` if there is one
                prev = next((self.state[p]["step"] for p in ps if "step" in self.state.get(p, {})), None)
                step_t = group["_step_t"] = torch.zeros((), dtype=torch.float32, device=dev) if prev is None else \
                    torch.as_tensor(float(prev), dtype=torch.float32, device=dev).clone()
            # the pointer table is rebuilt only when a parameter, gradient or state tensor moved (the HIP backward's flat gradient buffer
            # usually comes back at the same address every step): one pass of data_ptr() calls instead of 84 x the full checks
            # (an exact, order-sensitive key: a checksum can collide when the allocator hands equal-sized blocks back in another order)
            key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
            cached = group.get("_table")
            if cached is not None and cached[0] == key and all(
                    self.state[p]["exp_avg"].data_ptr() == cached[2][i][0] and self.state[p]["exp_avg_sq"].data_ptr() == cached[2][i][1]
                    for i, p in enumerate(ps)):
                ent = cached[1]
                ps_build = ()
            else:
                ent = (_lib.AdamwEntry * len(ps))()
                ps_build = ps
            keep = []
            for i, p in enumerate(ps_build):
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("scldm_amd.optim.AdamW: fp32 CUDA (ROCm) parameters with dense fp32 gradients only; there is no CPU path")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = step_t          # (shared tensor: torch's per-parameter key, one counter)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous() or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                    raise RuntimeError("scldm_amd.optim.AdamW: parameters and optimizer state must be contiguous")
                if g is not p.grad:
                    key = None                       # a temporary contiguous copy: never cache its address
                ent[i].p, ent[i].g, ent[i].m, ent[i].v, ent[i].n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                keep.append(g)
            if ps_build:
                group["_table"] = (key, ent, [(self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps]) \
                    if key is not None else None
            lr = float(group["lr"])
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):
                _lib.check(L.scldm_adamw_step(ent, len(ps), step_t.data_ptr(), None if found_inf is None else found_inf.data_ptr(), lr, float(b1), float(b2),
                                              float(group["eps"]), float(group["weight_decay"]), int(bool(group["maximize"])),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "scldm_adamw_step")
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:          # the state tensors were replaced: rebuild the pointer table and the shared step count
            g.pop("_table", None)
            g.pop("_step_t", None)

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_step_t", None)
            g.pop("_table", None)
        # every parameter gets its OWN copy of the step count: torch's optimizers increment the `step` tensor of each parameter, so a
        # shared tensor loaded there would advance once per parameter per step
        sd["state"] = {k: {kk: (vv.clone() if kk == "step" and torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
        return sd
