"""Exponential moving average of the diffusion model with the API `LatentDiffusion` uses (src/scldm/models.py:446-453):
`EMA(model=..., beta=, update_every=, allow_different_devices=, use_foreach=, update_after_step=)`, `.update()` after every optimizer
step (models.py:83-87), calling the object runs the averaged model (models.py:690), `state_dict()` keys `initted`, `step`,
`ema_model.*`, `online_model.*`.

The reference imports this class from the third-party package ema-pytorch (pinned `ema-pytorch==0.7.7`, pyproject.toml:30), which is
neither vendored under /root/reference nor installed in this image: the schedule below RESTATES that package's published algorithm
(`EMA.update`, `EMA.get_current_decay`, `EMA.update_moving_average`) - parity unpinned, see oracle/ema.py for the plain restatement
the tests compare against:

    step s (0-based count of update() calls):
      first call                                   -> copy online -> ema, initted = True
      s % update_every == 0 and s <= update_after  -> copy
      s % update_every == 0 otherwise              -> ema.lerp_(online, 1 - decay),
           decay = clamp(1 - (1 + epoch / inv_gamma) ** -power, min_value, beta), epoch = max(s - update_after_step, 0)   [float32]
           (decay = 0 when epoch <= 0)

MI355X path: `scldm_amd.optim.AdamW.attach_ema(ema)` folds the action of a step into the AdamW launch (csrc/optim.hip,
adamw_table_kernel: the new parameter value goes from the update's registers into the EMA tensor, torch.lerp's arithmetic bit for
bit); the action code and lerp weight of the step travel in the optimizer's 16-byte device `hyper` vector, so a captured HIP graph
follows the schedule.  `update()` is then a no-op that only checks it was applied.  Without an attached optimizer `update()` applies
the action itself with torch's foreach ops on the device (plumbing, not a hot path: 1 of 10 steps, 39 MB).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

EMA_NONE, EMA_COPY, EMA_LERP = 0, 1, 2


def decay_at(step_after_increment: int, update_after_step: int, inv_gamma: float, power: float, min_value: float, beta: float) -> float:
    """ema_pytorch's get_current_decay() evaluated with `self.step` == step_after_increment, in float32 tensor arithmetic like the
    package (`(self.step - update_after_step - 1).clamp(min=0.)` is a float32 tensor; the result is `.item()`-ed to a Python float)."""
    epoch = torch.tensor(float(step_after_increment - update_after_step - 1), dtype=torch.float32).clamp(min=0.0)
    if float(epoch) <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return float(value.clamp(min=min_value, max=beta))


class EMA(nn.Module):
    def __init__(self, model: nn.Module, ema_model: nn.Module | None = None, beta: float = 0.9999, update_after_step: int = 100,
                 update_every: int = 10, inv_gamma: float = 1.0, power: float = 2 / 3, min_value: float = 0.0,
                 include_online_model: bool = True, allow_different_devices: bool = False, use_foreach: bool = False, **unsupported):
        super().__init__()
        if unsupported:
            raise NotImplementedError(f"scldm_amd.ema.EMA: options {sorted(unsupported)} have no caller in the reference (models.py:446-453)")
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.allow_different_devices, self.use_foreach = allow_different_devices, use_foreach
        self.include_online_model = include_online_model
        if include_online_model:
            self.online_model = model
        else:
            self.__dict__["_online"] = [model]          # (hidden from the module tree, like the package's list wrapper)
        self.ema_model = ema_model if ema_model is not None else copy.deepcopy(model)
        for p in self.ema_model.parameters():
            p.detach_()
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))
        self._host_step = 0            # the schedule runs on host integers: no device read per step
        self._host_initted = False
        self._fused_by = None          # the optimizer that applies the actions (AdamW.attach_ema)
        self._pending = 0              # actions the optimizer applied that update() has not acknowledged yet

    # ---- the pieces of the package's API the reference touches
    @property
    def model(self) -> nn.Module:
        return self.online_model if self.include_online_model else self.__dict__["_online"][0]

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)

    def get_current_decay(self) -> float:
        return decay_at(self._host_step, self.update_after_step, self.inv_gamma, self.power, self.min_value, self.beta)

    def next_action(self) -> tuple[int, float]:
        """(action, lerp weight) of the NEXT update() call and advance the schedule: what `update()` does, as data."""
        step = self._host_step
        self._host_step += 1
        if not self._host_initted:
            self._host_initted = True
            return EMA_COPY, 1.0
        if step % self.update_every != 0:
            return EMA_NONE, 0.0
        if step <= self.update_after_step:
            return EMA_COPY, 1.0
        return EMA_LERP, 1.0 - self.get_current_decay()

    def _pairs(self):
        on, av = dict(self.model.named_parameters()), dict(self.ema_model.named_parameters())
        ob, ab = dict(self.model.named_buffers()), dict(self.ema_model.named_buffers())
        return [(av[k], on[k]) for k in on] + [(ab[k], ob[k]) for k in ob]

    @torch.no_grad()
    def copy_params_from_model_to_ema(self) -> None:
        for dst, src in self._pairs():
            dst.data.copy_(src.data)

    @torch.no_grad()
    def update_moving_average(self, ma_model=None, current_model=None, weight: float | None = None) -> None:
        w = 1.0 - self.get_current_decay() if weight is None else weight
        fl = [(d.data, s.data) for d, s in self._pairs() if d.is_floating_point()]
        for d, s in self._pairs():
            if not d.is_floating_point():
                d.data.copy_(s.data)
        if fl:
            torch._foreach_lerp_([d for d, _ in fl], [s for _, s in fl], w)

    @torch.no_grad()
    def update(self) -> None:
        if self._fused_by is not None:
            # the optimizer's launch already applied this step's action (attach_ema); one update() per optimizer step is expected
            if self._pending <= 0:
                raise RuntimeError("EMA.update(): the attached optimizer has not stepped since the last update() - the averaged model "
                                   "is advanced by optimizer.step() on this path (one update() per step, after it, as models.py:83-87 does)")
            self._pending -= 1
            return
        action, w = self.next_action()
        if action == EMA_COPY:
            self.copy_params_from_model_to_ema()
        elif action == EMA_LERP:
            self.update_moving_average(weight=w)

    # ---- checkpoints: the package's two buffers mirror the host counters
    def _sync_buffers(self) -> None:
        self.initted.fill_(bool(self._host_initted))
        self.step.fill_(int(self._host_step))

    def state_dict(self, *args, **kwargs):
        self._sync_buffers()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._host_step, self._host_initted = int(self.step), bool(self.initted)
        self._pending = 0
        return out
