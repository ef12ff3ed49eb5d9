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
                 differentiable: bool = False, max_grad_norm: float | None = None):
        if amsgrad or differentiable:
            raise NotImplementedError("scldm_amd.optim.AdamW: amsgrad / differentiable are not provided (no caller in the reference)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid AdamW hyper-parameter")
        if max_grad_norm is not None and not max_grad_norm > 0.0:
            raise ValueError("max_grad_norm must be positive (or None: no clipping)")
        # max_grad_norm (an extension of torch's signature): the reference's trainer clips the global gradient norm between backward and
        # optimizer.step (experiments/configs/training/default.yaml:15-16 -> Lightning -> torch.nn.utils.clip_grad_norm_(parameters, 10.0));
        # here the norm is one more launch over the same table and the update reads g * clip_coef - no pass that rewrites the gradients.
        # `.grad` itself is left unscaled.  `last_grad_norm` is the device scalar clip_grad_norm_ would have returned.
        self.max_grad_norm = None if max_grad_norm is None else float(max_grad_norm)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize,
                                      capturable=True, fused=True, foreach=None, differentiable=False,
                                      decoupled_weight_decay=True))   # (torch >= 2.6: AdamW is Adam with this flag; kept so a state_dict loads there as AdamW)
        # bumped whenever the device launch table / step counter / hyper vector of a group are dropped (load_state_dict, attach_ema):
        # holders of raw pointers into them (FusedTrainStep's launch struct, a captured HIP graph) compare it before every step
        self._generation = 0

    # ---- EMA of the reference's trainer in the same launch (scldm_amd.ema.EMA; models.py:83-87,446-453)
    def attach_ema(self, ema) -> None:
        """Fold `ema.update()` into this optimizer's launch: every step() applies the EMA action of that step (copy / lerp / none,
        ema.next_action()) to the averaged copy of each parameter it updates.  Parameters outside the optimizer (frozen ones: equal to
        the deep copy EMA made, and lerp(a, a, w) == a) and the model's buffers are left to the host-side copy EMA made at construction."""
        by_obj = {id(p): n for n, p in ema.model.named_parameters()}
        avg = dict(ema.ema_model.named_parameters())
        self._ema, self._ema_of = ema, {}
        for group in self.param_groups:
            for p in group["params"]:
                n = by_obj.get(id(p))
                if n is None:
                    raise ValueError("attach_ema: the optimizer holds a parameter that is not one of ema.model's")
                self._ema_of[id(p)] = avg[n]
            group.pop("_table", None)
        ema._fused_by = self
        self._generation = getattr(self, "_generation", 0) + 1

    def _hyper(self, group, dev):
        h = group.get("_hyper")
        if h is None or h[0].device != dev:
            h = group["_hyper"] = (torch.zeros(4, dtype=torch.float32, device=dev), None)
        return h

    def _clip_args(self, group, n_blocks: int, dev, capturing: bool = False):
        """(max_grad_norm, device workspace pointer) of a group's launch: (0.0, None) without clipping."""
        if self.max_grad_norm is None:
            return 0.0, None
        if len(self.param_groups) != 1:
            raise NotImplementedError("scldm_amd.optim.AdamW: max_grad_norm is the GLOBAL norm over every parameter - one parameter group only")
        need = _lib.lib().scldm_adamw_clip_workspace_bytes(n_blocks) // 4
        ws = group.get("_clip_ws")
        if ws is None or ws.numel() < need or ws.device != dev:
            if capturing:
                raise RuntimeError("scldm_amd.optim.AdamW: run at least one ordinary step() before capturing it in a HIP graph")
            ws = group["_clip_ws"] = torch.zeros(need, dtype=torch.float32, device=dev)
        return self.max_grad_norm, ws.data_ptr()

    @property
    def last_grad_norm(self):
        """Total gradient norm of the last step() (before clipping) as a device scalar - what clip_grad_norm_ returns; None without
        max_grad_norm or before the first step."""
        ws = self.param_groups[0].get("_clip_ws")
        return None if ws is None else ws[0]

    def refresh_hyper(self) -> None:
        """Stage this step's learning rate, weight decay and EMA action in the device `hyper` vector of every group (asynchronous 16-byte
        copy on the current stream).  step() calls it; `GraphedTrainStep` calls it before each replay (inside a capture it is skipped)."""
        ema = getattr(self, "_ema", None)
        live = any(p.grad is not None for g in self.param_groups for p in g["params"])
        # (a step() that will launch nothing - no parameter has a gradient - must not consume the EMA action of a step)
        action, w = ema.next_action() if (ema is not None and live) else (0, 0.0)
        if ema is not None and live:
            ema._pending += 1
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None] or list(group["params"])
            if not ps:
                continue
            dev_t, _ = self._hyper(group, ps[0].device)
            # The host runs several steps ahead of the device (always, when the step is a graph replay), so ONE reused staging buffer would
            # be overwritten before the queued copy of an earlier step has read it (found as a replay-vs-eager mismatch), and a fresh
            # pinned tensor per step makes the pinned allocator call hipHostMalloc whenever its cached blocks are still owned by copies
            # in flight - milliseconds each (found as 3.3 ms steps in a 20-step bench window).  A ring of 64 pinned slots, each guarded
            # by the event of the copy that last read it, costs neither.
            ring = group.get("_hyper_ring")
            if ring is None:
                ring = group["_hyper_ring"] = {"buf": torch.zeros(64, 4, dtype=torch.float32).pin_memory(), "ev": [None] * 64, "i": 0}
            slot = ring["i"] % 64
            ring["i"] += 1
            if ring["ev"][slot] is not None:
                ring["ev"][slot].synchronize()        # 64 steps old: long complete
            else:
                ring["ev"][slot] = torch.cuda.Event()
            host_t = ring["buf"][slot]
            host_t[0], host_t[1], host_t[2], host_t[3] = float(group["lr"]), float(group["weight_decay"]), float(action), float(w)
            with torch.cuda.device(dev_t.device):
                dev_t.copy_(host_t, non_blocking=True)
                ring["ev"][slot].record(torch.cuda.current_stream(dev_t.device))

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
        first = next((p for g in self.param_groups for p in g["params"] if p.grad is not None), None)
        if first is not None and not first.is_cuda:
            raise RuntimeError("scldm_amd.optim.AdamW: fp32 CUDA (ROCm) parameters with dense fp32 gradients only; there is no CPU path")
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.refresh_hyper()
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
            # the launch table is rebuilt (host build + one host-to-device copy) only when a parameter, gradient, state or EMA tensor
            # moved - the HIP backward's flat gradient buffer usually comes back at the same address every step.  An exact,
            # order-sensitive key: a checksum can collide when the allocator hands equal-sized blocks back in another order.
            ema_of = getattr(self, "_ema_of", None)
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    if not p.is_cuda or p.dtype != torch.float32:
                        raise RuntimeError("scldm_amd.optim.AdamW: fp32 CUDA (ROCm) parameters with dense fp32 gradients only; there is no CPU path")
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = step_t          # (shared tensor: torch's per-parameter key, one counter)
            key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                         ema_of[id(p)].data_ptr() if ema_of else 0) for p in ps)
            cached = group.get("_table")
            if cached is None or cached[0] != key:
                # (inside a HIP-graph capture - GraphedTrainStep: autograd allocates the flat gradient buffer from the graph's pool, at another
                # address than in the warm-up steps - the upload below becomes a copy node that re-reads the pinned table at every replay:
                # the pinned buffer is kept alive and unchanged in the cache entry)
                ent = (_lib.AdamwEntry * len(ps))()
                emas = (C.c_void_p * len(ps))()
                keep = []
                cacheable = True
                for i, p in enumerate(ps):
                    st = self.state[p]
                    if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                        raise RuntimeError("scldm_amd.optim.AdamW: fp32 CUDA (ROCm) parameters with dense fp32 gradients only; there is no CPU path")
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if not p.is_contiguous() or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                        raise RuntimeError("scldm_amd.optim.AdamW: parameters and optimizer state must be contiguous")
                    if g is not p.grad:
                        cacheable = False               # a temporary contiguous copy: never cache its address
                    ent[i].p, ent[i].g, ent[i].m, ent[i].v, ent[i].n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                    if ema_of:
                        e = ema_of[id(p)]
                        if e.device != p.device or e.dtype != torch.float32 or not e.is_contiguous() or e.shape != p.shape:
                            raise RuntimeError("attach_ema: the averaged copy of a parameter must be a contiguous fp32 tensor on the parameter's device")
                        emas[i] = e.data_ptr()
                    keep.append(g)
                ema_arg = C.cast(emas, C.POINTER(C.c_void_p)) if ema_of else None
                sizes = tuple(p.numel() for p in ps)
                if cached is not None and cached[5] == sizes:
                    # same tensors at other addresses (autograd handed out another flat gradient buffer - every step on the generic route,
                    # where a full rebuild of the DiT-L shape's 0.9 MB workgroup map cost 13 ms of a 20 ms step): only the records change.
                    # Staged through a ring of pinned slots like refresh_hyper's - except while capturing, where the pinned buffer of the
                    # initial build is rewritten (the device was synchronised before the capture began) and the upload becomes a copy node
                    # that re-reads it at every replay: that buffer is never written again afterwards.
                    rbytes = L.scldm_adamw_table_records_bytes(len(ps))
                    if capturing:
                        stage = cached[3]
                        if stage.numel() < rbytes:
                            raise RuntimeError("scldm_amd.optim.AdamW: run at least one ordinary step() before capturing it in a HIP graph")
                    else:      # a ring of 8 pinned staging slots guarded by events (see refresh_hyper: no pinned allocation per step)
                        ring = group.get("_rec_ring")
                        if ring is None or ring["buf"].shape[1] < rbytes:
                            ring = group["_rec_ring"] = {"buf": torch.zeros(8, rbytes, dtype=torch.uint8).pin_memory(), "ev": [None] * 8, "i": 0}
                        slot = ring["i"] % 8
                        ring["i"] += 1
                        if ring["ev"][slot] is not None:
                            ring["ev"][slot].synchronize()
                        else:
                            ring["ev"][slot] = torch.cuda.Event()
                        stage = ring["buf"][slot]
                    _lib.check(L.scldm_adamw_table_update(ent, ema_arg, len(ps), stage.data_ptr(), rbytes), "scldm_adamw_table_update")
                    with torch.cuda.device(dev):
                        cached[1][:rbytes].copy_(stage[:rbytes], non_blocking=True)
                        if not capturing:
                            ring["ev"][slot].record(torch.cuda.current_stream(dev))
                    cached = (key if cacheable else None, cached[1], cached[2], cached[3] if not capturing else stage, keep, sizes)
                else:
                    if capturing:
                        raise RuntimeError("scldm_amd.optim.AdamW: run at least one ordinary step() before capturing it in a HIP graph")
                    nbytes = L.scldm_adamw_table_bytes(ent, len(ps))
                    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
                    nblk = C.c_int(0)
                    _lib.check(L.scldm_adamw_table_build(ent, ema_arg, len(ps), host.data_ptr(), nbytes, C.byref(nblk)), "scldm_adamw_table_build")
                    table = host.to(dev, non_blocking=True)       # (kept alive by the cache entry, like the pinned source)
                    cached = (key if cacheable else None, table, nblk.value, host, keep, sizes)
                group["_table"] = cached
            table, nblk = cached[1], cached[2]
            dev_hyper, _ = self._hyper(group, dev)
            b1, b2 = group["betas"]
            max_norm, clip_ws = self._clip_args(group, nblk, dev, capturing)
            launch = _lib.AdamwLaunch(table=table.data_ptr(), count=len(ps), n_blocks=nblk, step=step_t.data_ptr(),
                                      found_inf=None if found_inf is None else found_inf.data_ptr(), hyper=dev_hyper.data_ptr(),
                                      lr=float(group["lr"]) if not torch.is_tensor(group["lr"]) else 0.0, beta1=float(b1), beta2=float(b2),
                                      eps=float(group["eps"]), weight_decay=float(group["weight_decay"]), maximize=int(bool(group["maximize"])),
                                      max_grad_norm=max_norm, clip_ws=clip_ws)
            with torch.cuda.device(dev):
                _lib.check(L.scldm_adamw_table_step(C.byref(launch), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "scldm_adamw_table_step")
        return loss

    def launch_struct(self, group_index: int = 0):
        """The scldm_adamw_launch of a group whose table is current (after one step()): what `scldm_dit_train_step` takes to run the
        optimizer inside the fused step (scldm_amd.training.FusedTrainStep)."""
        group = self.param_groups[group_index]
        cached, step_t = group.get("_table"), group.get("_step_t")
        if cached is None or cached[0] is None or step_t is None:
            raise RuntimeError("AdamW.launch_struct: take one ordinary step() first (it builds the device launch table)")
        dev_hyper, _ = self._hyper(group, step_t.device)
        b1, b2 = group["betas"]
        found_inf = getattr(self, "found_inf", None)
        max_norm, clip_ws = self._clip_args(group, cached[2], step_t.device)
        return _lib.AdamwLaunch(table=cached[1].data_ptr(), count=len(cached[0]), n_blocks=cached[2], step=step_t.data_ptr(),
                                found_inf=None if found_inf is None else found_inf.data_ptr(), hyper=dev_hyper.data_ptr(), lr=0.0,
                                beta1=float(b1), beta2=float(b2), eps=float(group["eps"]), weight_decay=float(group["weight_decay"]),
                                maximize=int(bool(group["maximize"])), max_grad_norm=max_norm, clip_ws=clip_ws)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._generation = getattr(self, "_generation", 0) + 1
        for g in self.param_groups:          # the state tensors were replaced: rebuild the pointer table and the shared step count
            g.pop("_table", None)
            g.pop("_step_t", None)
            g.pop("_hyper", None)
            g.pop("_hyper_ring", None)
            g.pop("_rec_ring", None)
            g.pop("_clip_ws", None)
            steps = {float(self.state[p]["step"]) for p in g["params"] if "step" in self.state.get(p, {})}
            if len(steps) > 1:   #  one counter per group here: a checkpoint whose parameters took different numbers of steps does not fit
                raise ValueError(f"scldm_amd.optim.AdamW keeps ONE step count per parameter group; the loaded state has {sorted(steps)}")

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_step_t", None)
            g.pop("_table", None)
            g.pop("_hyper", None)
            g.pop("_hyper_ring", None)
            g.pop("_rec_ring", None)
            g.pop("_clip_ws", None)
        # every parameter gets its OWN copy of the step count: torch's optimizers increment the `step` tensor of each parameter, so a
        # shared tensor loaded there would advance once per parameter per step
        sd["state"] = {k: {kk: (vv.clone() if kk == "step" and torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
        return sd
