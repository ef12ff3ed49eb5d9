"""Plain restatement of ema-pytorch's `EMA.update()` schedule as `LatentDiffusion` uses it.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference imports `EMA` from the third-party package ema-pytorch, pinned `ema-pytorch==0.7.7`
(/root/reference/pyproject.toml:30; call sites src/scldm/models.py:9,83-87,446-453; hyper-parameters
experiments/configs/model/ldm_base.yaml:51-55: beta 0.9999, update_every 10, update_after_step 10 000).  The package is neither
vendored under /root/reference nor installed in this image and there is no network, so this file restates its published algorithm
(ema_pytorch/ema_pytorch.py: `update`, `get_current_decay`, `update_moving_average`) instead of being checked against it:

    def update(self):
        step = self.step.item(); self.step += 1
        if not self.initted.item():                      # first call: copy, mark initialised, return
            self.copy_params_from_model_to_ema(); self.initted.data.copy_(True); return
        should_update = step % self.update_every == 0
        if should_update and step <= self.update_after_step:
            self.copy_params_from_model_to_ema(); return
        if should_update:
            self.update_moving_average(self.ema_model, self.model)

    def get_current_decay(self):                         # self.step already incremented
        epoch = (self.step - self.update_after_step - 1).clamp(min=0.)
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        if epoch.item() <= 0: return 0.
        return value.clamp(min=self.min_value, max=self.beta).item()

    update_moving_average: for every parameter and buffer  ma.lerp_(current, 1. - current_decay)   (non-float: copy)
"""
from __future__ import annotations

import torch


def schedule(n_steps: int, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3, min_value=0.0):
    """[(action, lerp_weight)] of update() calls 0 .. n_steps-1: action "copy" | "lerp" | "none"."""
    out, initted = [], False
    for step in range(n_steps):
        step_t = step + 1                      # self.step after the increment
        if not initted:
            initted = True
            out.append(("copy", 1.0))
            continue
        if step % update_every != 0:
            out.append(("none", 0.0))
            continue
        if step <= update_after_step:
            out.append(("copy", 1.0))
            continue
        epoch = torch.tensor(float(step_t - update_after_step - 1)).clamp(min=0.0)
        if float(epoch) <= 0:
            decay = 0.0
        else:
            decay = float((1 - (1 + epoch / inv_gamma) ** -power).clamp(min=min_value, max=beta))
        out.append(("lerp", 1.0 - decay))
    return out


def run(online_trajectory, **kw):
    """EMA tensors after each update() for a list of online tensors (one per step, taken AFTER that step's optimizer update)."""
    ema, outs = None, []
    for (action, w), p in zip(schedule(len(online_trajectory), **kw), online_trajectory):
        if action == "copy":
            ema = p.clone()
        elif action == "lerp":
            ema = ema.clone().lerp_(p, w)
        outs.append(ema.clone())
    return outs
