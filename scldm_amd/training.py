"""Data-parallel flow-matching training step for the scldm_amd DiT (SURVEY.md section 8a row T1, 8e training row).

The reference trains through Lightning + DDP (experiments/scripts/train_ldm.py:101; src/scldm/models.py:443-470): per-rank
mini-batches, `Transport.training_losses`, `loss.mean().backward()`, gradient all-reduce in buckets that overlap the rest of the
backward, AdamW.  Here the backward is one C call (scldm_dit_train_backward) that only ENQUEUES kernels; it writes every gradient
into ONE flat fp32 buffer laid out in the order the gradients are completed (DiT.grad_segments: last layer first, then the adaLN
projections, then the ends) and records a HIP event per bucket at the point of the stream where that bucket is final
(scldm_dit_train_set_grad_events).  `OverlappedGradSync` queues one in-place all-reduce per bucket on a side stream behind its
event: contiguous slices of the buffer, no gather / scatter copies, bucket sizes chosen for xGMI's point-to-point links (few,
large collectives: 64-128 MB), overlapped with the layers still being differentiated.  The base DiT (39 MB of gradients, a fused
backward that finishes everything at once) degenerates to one collective, the DiT-L shape (1.84 GB) to ~20.
`torch.nn.parallel.DistributedDataParallel` also works on the module unchanged (its hooks see ordinary .grad tensors).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def grad_buckets(params, bucket_bytes: int = 256 << 20):
    """Deterministic partition of the trainable parameters into buckets of at most bucket_bytes (at least one each)."""
    buckets, cur, size = [], [], 0
    for p in params:
        if not p.requires_grad:
            continue
        nbytes = p.numel() * p.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def _shared_flat_grad(params):
    """One 1-D tensor spanning every .grad when all of them are contiguous views of the same storage (what
    scldm_amd.nnets._DiTTrainFn.backward produces), else None."""
    grads = [p.grad for p in params if p.requires_grad]
    if not grads or any(g is None or not g.is_contiguous() for g in grads):
        return None
    g0 = grads[0]
    st = g0.untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() or g.dtype != g0.dtype or g.device != g0.device for g in grads):
        return None
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    if sum(g.numel() for g in grads) < 0.9 * (hi - lo):     # mostly gaps: not the flat layout this shortcut is for
        return None
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(st, lo, (hi - lo,))


@torch.no_grad()
def allreduce_gradients(params, group=None, bucket_bytes: int = 128 << 20, average: bool = True) -> int:
    """Sum (or average) .grad over the ranks of `group` AFTER the backward (no overlap; `OverlappedGradSync` is the overlapped
    form).  Gradients that are views of one flat buffer (the HIP backward's) are reduced IN PLACE, in contiguous slices of at most
    bucket_bytes - no gather / scatter copies whatever the model size; anything else goes through per-bucket flat copies.  A
    parameter whose .grad is None on this rank contributes zeros (and receives the reduced value).  Returns the number of
    collectives issued."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    calls = 0
    params = list(params)
    flat = _shared_flat_grad(params)
    if flat is not None:
        # (the alignment gaps between the views are reduced along with them and never read)
        step = max(1, bucket_bytes // flat.element_size())
        for a in range(0, flat.numel(), step):
            piece = flat[a:a + step]
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
            if average:
                piece.div_(world)
            calls += 1
        return calls
    for bucket in grad_buckets(params, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        calls += 1
    return calls


class OverlappedGradSync:
    """Bucketed, in-place gradient all-reduce overlapped with the HIP backward of one scldm_amd DiT.

    attach(dit) -> the module's next backward (a) asks the C side to record one event per bucket of `dit.grad_bucket_plan()`
    and (b) right after the backward call returned - its kernels are only queued - queues, on a side stream, one all-reduce per
    bucket behind that bucket's event.  finish() makes the current stream wait for them (and scales by 1 / world).  Every bucket
    is a contiguous slice of the backward's flat gradient buffer: there is no copy on either side of a collective.
    On CPU tensors (the gloo tests) there are no streams / events: buckets are reduced in plan order at after_backward().
    Counters: `collectives` (issued by the last step), `copies` (always 0 on this path; the tests assert it)."""

    def __init__(self, group=None, bucket_bytes: int = 128 << 20, average: bool = True):
        self.group, self.bucket_bytes, self.average = group, bucket_bytes, average
        self.collectives = 0
        self.copies = 0
        self._works, self._flat, self._plan, self._events, self._side = [], None, None, None, None
        self._params, self._offs = None, None
        self.handled = False

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def attach(self, dit) -> None:
        dit.__dict__["_grad_sync"] = self
        self.handled = False
        self.collectives = 0
        self.copies = 0

    @staticmethod
    def detach(dit) -> None:
        dit.__dict__.pop("_grad_sync", None)

    # ---- called by scldm_amd.nnets._DiTTrainFn.backward -----------------------------------------------------------------------
    def before_backward(self, dit, L, h) -> None:
        if self._works or self._flat is not None:
            raise RuntimeError("OverlappedGradSync: a second backward started before finish() consumed the first one's all-reduces "
                               "(one backward per step; call finish() - train_step does - before the next)")
        self._plan = dit.grad_bucket_plan(self.bucket_bytes)
        # the in-place exchange relies on autograd ADOPTING the views of the flat buffer as .grad (no clone, no accumulation into an
        # older .grad while the side stream is still reducing the buffer): gradients must be None when the backward starts
        self._offs = dit.__dict__["_grad_offsets"]
        self._params = [p for p in dit.parameters() if id(p) in self._offs]
        stale = sum(p.grad is not None for p in self._params)
        if stale:
            raise RuntimeError(f"OverlappedGradSync: {stale} parameter(s) still hold a .grad from an earlier backward; gradient "
                               "accumulation is not supported on the overlapped path - call optimizer.zero_grad(set_to_none=True) "
                               "first, or use allreduce_gradients() after the last backward")
        dev = dit.pos_embed.device
        if dev.type != "cuda":
            self._events = None
            return
        kinds = {"layer": 0, "ada": 1, "end": 2}
        self._events = [torch.cuda.Event() for _ in self._plan]
        cur = torch.cuda.current_stream(dev)
        for e in self._events:
            e.record(cur)          # (creates the underlying hipEvent_t; re-recorded by the C side at the bucket's completion point)
        n = len(self._plan)
        ev = (C.c_void_p * n)(*[e.cuda_event for e in self._events])
        kd = (C.c_int * n)(*[kinds[b[2]] for b in self._plan])
        ly = (C.c_int * n)(*[b[3] for b in self._plan])
        from . import _lib
        _lib.check(L.scldm_dit_train_set_grad_events(h, C.cast(ev, _lib.c_void_pp), kd, ly, n), "scldm_dit_train_set_grad_events")

    def after_backward(self, flat: torch.Tensor) -> None:
        if self._works or self._flat is not None:
            raise RuntimeError("OverlappedGradSync.after_backward called twice before finish()")
        self._flat = flat
        self._works = []
        op = dist.ReduceOp.SUM
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(flat.device)
            for (a, b, _, _), ev in zip(self._plan, self._events):
                self._side.wait_event(ev)                      # the bucket is final from this point of the backward's stream on
                with torch.cuda.stream(self._side):
                    self._works.append(dist.all_reduce(flat[a:b], op=op, group=self.group, async_op=True))
                self.collectives += 1
            flat.record_stream(self._side)
        else:
            for a, b, _, _ in self._plan:
                self._works.append(dist.all_reduce(flat[a:b], op=op, group=self.group, async_op=True))
                self.collectives += 1
        self.handled = True

    def finish(self) -> None:
        """The current stream waits for every bucket; gradients become the mean over ranks.  Every .grad must be the view of the
        flat buffer the backward returned for it: a parameter whose .grad is something else (autograd cloned the view instead of
        adopting it) gets the reduced slice copied in (`copies` counts them; the clone may have been taken while the side stream
        was reducing the buffer, so its contents are not trusted)."""
        for w in self._works:
            w.wait()
        flat = self._flat
        if flat is not None and self.average:
            flat.div_(dist.get_world_size(self.group))
        if flat is not None and self._params is not None:
            base, es = flat.data_ptr(), flat.element_size()
            for p in self._params:
                g = p.grad
                if g is None:
                    continue
                o = self._offs[id(p)]
                if g.data_ptr() != base + es * o or not g.is_contiguous():
                    with torch.no_grad():
                        g.copy_(flat[o:o + p.numel()].view(p.shape))
                    self.copies += 1
            if self.copies:
                import warnings
                warnings.warn(f"OverlappedGradSync: autograd did not adopt {self.copies} gradient view(s) of the flat buffer; the reduced "
                              "values were copied into .grad (slower; results are correct)", RuntimeWarning, stacklevel=2)
        self._works, self._flat, self._events, self._params, self._offs = [], None, None, None, None


def train_step(dit, transport, optimizer, x1: torch.Tensor, condition: dict[str, torch.Tensor], group=None,
               bucket_bytes: int = 128 << 20, sync: OverlappedGradSync | None = None) -> torch.Tensor:
    """One optimisation step on this rank's mini-batch: loss = mean_b training_losses(...)["loss"] (models.py:443-470),
    backward through the HIP kernels with the bucketed gradient all-reduce (mean over ranks) overlapped with it, then
    optimizer.step().  Returns the local loss."""
    optimizer.zero_grad(set_to_none=True)
    # fp16 training: the backward's overflow flag (non-finite gradients after the loss-scaled backward) makes the optimizer skip the
    # step - on device for optimizers that take GradScaler's `found_inf` (torch's fused Adam / AdamW), by one host read otherwise
    found_inf = dit.found_inf_flag() if getattr(dit, "precision", None) == "fp16" and hasattr(dit, "found_inf_flag") else None
    loss = transport.training_losses(dit, x1, {"condition": condition})["loss"].mean()
    if sync is None:
        sync = dit.__dict__.get("_train_step_sync")
        if sync is None or sync.group is not group or sync.bucket_bytes != bucket_bytes:
            sync = dit.__dict__["_train_step_sync"] = OverlappedGradSync(group, bucket_bytes)
    overlapped = sync.active() and hasattr(dit, "grad_bucket_plan")
    if overlapped:
        sync.attach(dit)
    try:
        loss.backward()
    finally:
        if overlapped:
            OverlappedGradSync.detach(dit)
    if overlapped and sync.handled:
        sync.finish()
        extra = [p for p in dit.parameters() if p.requires_grad and p.grad is not None and p is getattr(dit, "pos_embed", None)]
        if extra:          # (pos_embed is frozen in the reference; if unfrozen its gradient lives outside the flat buffer)
            allreduce_gradients(extra, group, bucket_bytes)
    else:
        allreduce_gradients(dit.parameters(), group, bucket_bytes)
    if found_inf is None:
        optimizer.step()
    else:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(found_inf, op=dist.ReduceOp.MAX, group=group)     # every rank skips the same steps
        if getattr(optimizer, "_step_supports_amp_scaling", False):
            optimizer.found_inf, optimizer.grad_scale = found_inf, None
            try:
                optimizer.step()
            finally:
                del optimizer.found_inf, optimizer.grad_scale
        elif float(found_inf) == 0.0:
            optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """The whole optimisation step - Transport.training_losses -> loss.mean().backward() -> optimizer.step() (models.py:443-470) -
    captured ONCE in a HIP graph and replayed per mini-batch: one graph launch instead of ~100 kernel launches, the autograd walk,
    the optimizer's Python and ~2 ms of host time per step (at <= 512 cells per GPU the eager step is host-bound).

    The library's side streams (weight re-pack, the backward's independent tails) fork from and join the capturing stream through
    events, so they become branches of the graph.  Requirements: a CUDA model in training mode, fixed batch size and label keys, an
    optimizer whose step is capturable (torch.optim.Adam / AdamW with `fused=True, capturable=True`), single process (the
    gradient all-reduce of data-parallel training is not captured: use `train_step`).  `t` is drawn on the device inside the
    graph (Transport.sample), x0 as always.  `__call__(x1, condition)` copies the batch into the graph's static inputs, replays
    and returns the (static) loss tensor."""

    def __init__(self, dit, transport, optimizer, x1: torch.Tensor, condition: dict[str, torch.Tensor], warmup: int = 3):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise NotImplementedError("GraphedTrainStep captures a single-process step; data-parallel training uses train_step")
        if not x1.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA (ROCm) tensors")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("the optimizer step is captured in a HIP graph: construct it with capturable=True "
                                 "(e.g. torch.optim.AdamW(params, fused=True, capturable=True))")
        # Host-side decisions a capture would freeze: (1) a multi-class mutually_exclusive model draws the class of a
        # forward with a CPU randint (nnets.py:395; DiT._label_ptrs) - every replay would train the class drawn at capture;
        # (2) fp16 with an optimizer that cannot take GradScaler's found_inf would apply an overflowed step (the eager train_step
        # reads the flag on the host instead).  `FusedTrainStep` has neither restriction (class draw and schedules live on the device).
        if getattr(dit, "condition_strategy", None) == "mutually_exclusive" and len([c for c in dit._class_names if c in condition]) > 1:
            raise NotImplementedError("GraphedTrainStep: a mutually_exclusive model with several condition classes draws its class on the host "
                                      "per forward; a captured graph would freeze it - use FusedTrainStep (device-side draw) or train_step")
        if getattr(dit, "precision", None) == "fp16" and not getattr(optimizer, "_step_supports_amp_scaling", False):
            raise NotImplementedError("GraphedTrainStep: fp16 training needs an optimizer that takes GradScaler's found_inf "
                                      "(scldm_amd.optim.AdamW, torch's fused Adam / AdamW): an overflowed step cannot be skipped on the host inside a graph")
        if not hasattr(optimizer, "refresh_hyper") and any(torch.is_tensor(g["lr"]) is False for g in optimizer.param_groups):
            import warnings
            warnings.warn("GraphedTrainStep: this optimizer passes its learning rate by value into the captured kernels - a per-step LR "
                          "schedule (models.py:603-605) has no effect on replays; use scldm_amd.optim.AdamW (device-resident hyper-parameters) "
                          "or a tensor lr", RuntimeWarning, stacklevel=2)
        self.dit, self.transport, self.optimizer = dit, transport, optimizer
        self.x1 = x1.detach().clone()
        self.condition = {k: v.detach().clone() for k, v in condition.items()}
        self.found_inf = dit.found_inf_flag() if getattr(dit, "precision", None) == "fp16" and hasattr(dit, "found_inf_flag") else None
        side = torch.cuda.Stream(device=x1.device)
        side.wait_stream(torch.cuda.current_stream(x1.device))
        with torch.cuda.stream(side):          # warm-up on a side stream: allocations, lazy handles, pack tables, workspaces
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream(x1.device).wait_stream(side)
        torch.cuda.synchronize(x1.device)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        self.replays = 0
        self._generation = getattr(optimizer, "_generation", None)

    def _body(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.transport.training_losses(self.dit, self.x1, {"condition": self.condition})["loss"].mean()
        loss.backward()
        if self.found_inf is not None and getattr(self.optimizer, "_step_supports_amp_scaling", False):
            self.optimizer.found_inf, self.optimizer.grad_scale = self.found_inf, None
            try:
                self.optimizer.step()
            finally:
                del self.optimizer.found_inf, self.optimizer.grad_scale
        else:
            self.optimizer.step()
        return loss.detach()

    def __call__(self, x1: torch.Tensor, condition: dict[str, torch.Tensor]) -> torch.Tensor:
        if x1.shape != self.x1.shape or set(condition) != set(self.condition):
            raise ValueError(f"GraphedTrainStep was captured for x1 {tuple(self.x1.shape)} and labels {sorted(self.condition)}")
        if getattr(self.optimizer, "_generation", None) != self._generation:
            raise RuntimeError("GraphedTrainStep: the optimizer's device state was replaced after the capture (optimizer.load_state_dict / attach_ema): "
                               "the graph holds the old tensors' addresses - load checkpoints first, then build GraphedTrainStep")
        self.x1.copy_(x1, non_blocking=True)
        for k, v in condition.items():
            self.condition[k].copy_(v, non_blocking=True)
        if hasattr(self.optimizer, "refresh_hyper"):
            self.optimizer.refresh_hyper()       # this step's learning rate / weight decay / EMA action -> the device vector the captured kernel reads
        self.graph.replay()
        self.replays += 1
        return self.loss


class FusedTrainStep:
    """One optimisation step of `LatentDiffusion.training_step` (src/scldm/models.py:628-663) + Lightning's backward, optimizer step
    and EMA hook (models.py:83-87) as ONE C call, `scldm_dit_train_step` (include/scldm_hip.h; csrc/train_step.hip):

        [frozen vae.encode of the tokenised batch, models.py:641]  ->  batch preparation (t, x0, xt, ut, label dropout, class draw: one
        kernel, Philox state on the device)  ->  DiT forward with the training record  ->  loss + d loss (one kernel)  ->  DiT backward
        ->  AdamW + EMA (one launch, scldm_amd.optim.AdamW with attach_ema)

    No autograd, no Python between the kernels, no decision on the host: the call is captured once in a HIP graph (`graph=True`) and
    replayed - per replay the host stages 16 bytes (learning rate, weight decay, EMA action) and copies the batch.  Gradients live in
    one flat fp32 buffer (`.grad` of every parameter is a view), so a data-parallel group reduces it in place between backward and
    optimizer (`group=`; that variant is not graphed).  `grad_clip_norm=` = the trainer's `gradient_clip_val` (global-norm clipping inside
    the optimizer's launch; `optimizer.last_grad_norm` is the step's norm).  Requires the fused training route (base DiT shape, precision bf16 | fp16), the
    Linear path with velocity prediction (ldm_base.yaml:30-35) and `scldm_amd.optim.AdamW`.
    `__call__(x1, condition)` or `__call__(condition=..., counts=, genes=, counts_subset=, genes_subset=)` with a frozen `vae`; returns the
    step's loss as a STATIC device scalar (overwritten by the next step: `.clone()` or `float()` it to keep a value)."""

    def __init__(self, dit, transport, optimizer, batch_size: int, condition_keys, ema=None, vae=None, seed: int | None = None,
                 graph: bool = True, group=None, encode_shape: tuple[int, int] | None = None, grad_clip_norm: float | None = None):
        from . import _lib
        from .optim import AdamW
        if not isinstance(optimizer, AdamW):
            raise TypeError("FusedTrainStep runs the optimizer inside the step: it needs scldm_amd.optim.AdamW")
        if len(optimizer.param_groups) != 1:
            raise NotImplementedError("FusedTrainStep: one parameter group (the reference's configure_optimizers builds one, models.py:598-601)")
        if grad_clip_norm is not None:
            # the trainer's gradient_clip_val (experiments/configs/training/default.yaml:15-16: 10.0, algorithm "norm"): inside the optimizer's
            # launch (AdamW.max_grad_norm), between the backward and the update like Lightning's clip_grad_norm_ call
            if not grad_clip_norm > 0.0:
                raise ValueError("grad_clip_norm must be positive (or None: no clipping)")
            optimizer.max_grad_norm = float(grad_clip_norm)
        dev = dit.pos_embed.device
        if dev.type != "cuda":
            raise RuntimeError("FusedTrainStep needs the model on a CUDA (ROCm) device; there is no CPU path")
        self.dit, self.transport, self.optimizer, self.ema, self.vae = dit, transport, optimizer, ema, vae
        self.n, self.group, self.dev = int(batch_size), group, dev
        self.keys = [c for c in dit._class_names if c in set(condition_keys)]
        if not self.keys:
            raise ValueError("condition_keys holds none of the model's classes")
        if dit.condition_strategy == "joint" and len(self.keys) != len(dit._class_names):
            raise KeyError(next(c for c in dit._class_names if c not in self.keys))      # (nnets.py:449 indexes every class)
        dit._need_null_row("label dropout")
        if ema is not None and getattr(optimizer, "_ema", None) is not ema:
            optimizer.attach_ema(ema)
        L, h = dit._native_handle()
        self._L, self._h = L, h
        prec = dit._prec()
        if prec not in (_lib.PRECISIONS["bf16"], _lib.PRECISIONS["fp16"]) or not dit.fused_shape:
            raise NotImplementedError("FusedTrainStep serves the fused training route: the reference's DiT shape at precision 'bf16' or 'fp16'")
        self.prec = prec
        n, e = self.n, dit.seq_len * dit.n_embed_input
        f32 = dict(dtype=torch.float32, device=dev)
        self.x1 = torch.zeros(n, dit.seq_len, dit.n_embed_input, **f32)
        self.labels_in = {k: torch.zeros(n, dtype=torch.long, device=dev) for k in self.keys}
        self.t, self.loss_rows, self.loss = torch.empty(n, **f32), torch.empty(n, **f32), torch.zeros((), **f32)
        self.xt, self.ut, self.pred, self.dpred = (torch.empty(n, e, **f32) for _ in range(4))
        self.labels = torch.empty(len(dit._class_names), n, dtype=torch.long, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        self.rng = torch.tensor([seed, 0], dtype=torch.int64, device=dev)         # [0] Philox key, [1] step counter (advanced on the device)
        self.saved = torch.empty(L.scldm_dit_train_saved_bytes_for(h, n, prec), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(L.scldm_dit_train_workspace_bytes_for(h, n, prec), dtype=torch.uint8, device=dev)
        self.params = tuple(dit.parameters())
        self._w, self._keep_w = dit._weights_struct(self.params)
        with torch.cuda.device(dev):
            _lib.check(L.scldm_dit_train_prepare(h, C.byref(self._w), n, prec, torch.cuda.current_stream(dev).cuda_stream), "scldm_dit_train_prepare")
        dit.__dict__["_prepared_key"] = (id(self._w), n, prec)
        offs = dit._grad_offsets
        self.flat = torch.zeros(dit._grad_numel, **f32)
        base = self.flat.data_ptr()
        self._gpos = torch.zeros_like(dit.pos_embed) if dit.pos_embed.requires_grad else None
        self._g, self._keep_g = dit._param_struct(lambda p: (self._gpos.data_ptr() if self._gpos is not None else None) if p is dit.pos_embed
                                                  else base + 4 * offs[id(p)])
        names = dit._class_names
        self._lab_ptrs = _lib.ptr_array([self.labels_in[c].data_ptr() if c in self.labels_in else None for c in names])
        self._nulls = (C.c_int * len(names))(*[int(dit.class_vocab_sizes[c]) for c in names])
        self._buf = _lib.TrainStepBuffers(row_elems=e, t=self.t.data_ptr(), x0=None, xt=self.xt.data_ptr(), ut=self.ut.data_ptr(),
                                          pred=self.pred.data_ptr(), dpred=self.dpred.data_ptr(), labels=self.labels.data_ptr(),
                                          loss_rows=self.loss_rows.data_ptr(), loss_mean=self.loss.data_ptr(), ticket=self.ticket.data_ptr(),
                                          saved=self.saved.data_ptr(), ws=self.ws.data_ptr())
        self.found_inf = dit.found_inf_flag() if dit.precision == "fp16" else None
        # (an explicit group selects the exchange path even with one rank: the collectives are identities, the plumbing is the multi-GPU one)
        self.distributed = dist.is_available() and dist.is_initialized() and (group is not None or dist.get_world_size(group) > 1)
        self.steps = 0
        # the encode of the tokenised batch (frozen VAE as tokenizer, models.py:641): static inputs when graphed
        self.enc = None
        self._vae_frozen = vae is not None and not any(p.requires_grad for p in vae.parameters())
        if vae is not None and encode_shape is not None:
            S = int(encode_shape[1])
            self.enc = (torch.zeros(n, S, **f32), torch.zeros(n, S, dtype=torch.long, device=dev))
        self._want_graph = bool(graph) and not self.distributed
        self._grad_probe = next(p for p in self.params if p.requires_grad)
        self._bind_optimizer()

    def _attach_grads(self) -> None:
        offs = self.dit._grad_offsets
        for p in self.params:
            if p.requires_grad:
                p.grad = self._gpos if p is self.dit.pos_embed else self.flat[offs[id(p)]:offs[id(p)] + p.numel()].view(p.shape)

    def _bind_optimizer(self) -> None:
        """Everything that holds raw addresses of the optimizer's device state - the launch struct and the captured graph.  Runs at
        construction and again when the optimizer dropped that state (`optimizer.load_state_dict` on resume, `attach_ema`:
        AdamW._generation) - a replay of the old graph would update freed moment tensors."""
        optimizer = self.optimizer
        self.graph = None
        self._attach_grads()                      # (optimizer.zero_grad(set_to_none=True) by a caller detaches the views)
        # one ordinary optimizer step builds the device launch table against the flat gradient buffer (zero gradients: a no-op update
        # that still counts as step 1 would shift the bias correction, so its effects are rolled back)
        self._prime_optimizer()
        if self.distributed:
            self._opt = None
        else:
            if self.found_inf is not None:
                optimizer.found_inf = self.found_inf          # (launch_struct reads it: the fp16 backward's overflow flag skips the update on device)
            try:
                self._opt = optimizer.launch_struct()
            finally:
                if self.found_inf is not None:
                    del optimizer.found_inf
        self._bound_generation = optimizer._generation
        self._bound_clip = optimizer.max_grad_norm
        if self._want_graph:
            self._capture()

    def _prime_optimizer(self) -> None:
        """One ordinary optimizer.step() on the (zero) flat gradient buffer builds the device launch table; everything it changed -
        parameters (weight decay), moments, step count, EMA - is put back, so an optimizer that already trained keeps its state."""
        opt = self.optimizer
        g = opt.param_groups[0]
        had_state = {id(p) for p in g["params"] if "exp_avg" in (opt.state.get(p) or {})}
        step_before = g["_step_t"].clone() if g.get("_step_t") is not None else None
        if step_before is None:    # after optimizer.load_state_dict the counter lives only in the per-parameter state: step() re-creates it from there
            prev = next((opt.state[p]["step"] for p in g["params"] if "step" in (opt.state.get(p) or {})), None)
            step_before = None if prev is None else torch.as_tensor(float(prev), dtype=torch.float32, device=self.dev)
        state = self._snapshot(with_state=False)
        moments = {id(p): (opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in g["params"] if id(p) in had_state}
        self.flat.zero_()
        opt.step()
        self._restore(state, with_state=False)
        with torch.no_grad():
            if step_before is not None:
                g["_step_t"].copy_(step_before)
            else:
                g["_step_t"].zero_()
            for p in g["params"]:
                st = opt.state.get(p)
                if not st:
                    continue
                if id(p) in moments:
                    st["exp_avg"].copy_(moments[id(p)][0])
                    st["exp_avg_sq"].copy_(moments[id(p)][1])
                else:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()

    def _launch(self) -> None:
        from . import _lib
        dit = self.dit
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            if self.enc is not None:
                # a frozen tokenizer (models.py:432-435: requires_grad False on every VAE parameter): its packed weight copies are checked
                # through the version counters only - the device-side fingerprint pass (five small launches per encode, there for EMA-style
                # `.data` updates) is skipped inside the step
                keep = self.vae.check_weight_fingerprint
                if self._vae_frozen:
                    self.vae.check_weight_fingerprint = False
                try:
                    z = self.vae.encode(self.enc[0], self.enc[1])
                finally:
                    self.vae.check_weight_fingerprint = keep
                self.x1.copy_(z.view_as(self.x1))
            _lib.check(self._L.scldm_dit_train_step(self._h, C.byref(self._w), C.byref(self._g), self.x1.data_ptr(), C.cast(self._lab_ptrs, _lib.c_void_pp),
                                                    self._nulls, len(dit._class_names), 1 if dit.condition_strategy == "joint" else 0,
                                                    float(dit.cfg_dropout_prob), self.rng.data_ptr(), self.n, self.prec, C.byref(self._buf),
                                                    C.byref(self._opt) if self._opt is not None else None, st), "scldm_dit_train_step")

    def profile_stages(self) -> dict:
        """Device time of each stage of ONE (real, eager) step, in ms: the sub-entry points scldm_dit_train_step itself calls, issued one by
        one with HIP events between them on the current stream (bench.py's `train_e2e` record)."""
        from . import _lib
        dit, L, h = self.dit, self._L, self._h
        names = dit._class_names
        lab = _lib.ptr_array([self.labels.data_ptr() + 8 * self.n * i for i in range(len(names))])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        self.optimizer.refresh_hyper()
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            b = self._buf
            ev[0].record()
            if self.enc is not None:
                self.x1.copy_(self.vae.encode(self.enc[0], self.enc[1]).view_as(self.x1))
            ev[1].record()
            _lib.check(L.scldm_fm_prepare(self.x1.data_ptr(), C.cast(self._lab_ptrs, _lib.c_void_pp), self._nulls, len(names),
                                          1 if dit.condition_strategy == "joint" else 0, 1, float(dit.cfg_dropout_prob), self.rng.data_ptr(), self.n,
                                          b.row_elems, b.t, b.x0, b.xt, b.ut, b.labels, st), "scldm_fm_prepare")
            ev[2].record()
            _lib.check(L.scldm_dit_train_forward(h, C.byref(self._w), b.xt, b.t, C.cast(lab, _lib.c_void_pp), self.n, b.pred, self.prec, b.saved, b.ws, st),
                       "scldm_dit_train_forward")
            ev[3].record()
            _lib.check(L.scldm_fm_loss_grad(b.pred, b.ut, self.n, b.row_elems, b.loss_rows, b.loss_mean, b.dpred, b.ticket, self.rng.data_ptr(), st),
                       "scldm_fm_loss_grad")
            ev[4].record()
            _lib.check(L.scldm_dit_train_backward(h, C.byref(self._w), C.byref(self._g), b.xt, C.cast(lab, _lib.c_void_pp), b.dpred, self.n, None, self.prec,
                                                  b.saved, b.ws, st), "scldm_dit_train_backward")
            ev[5].record()
            if self._opt is not None:
                _lib.check(L.scldm_adamw_table_step(C.byref(self._opt), st), "scldm_adamw_table_step")
            ev[6].record()
        torch.cuda.synchronize(self.dev)
        keys = ("vae_encode", "prepare_batch", "forward_with_record", "loss_and_grad", "backward", "adamw_and_ema")
        return {k: ev[i].elapsed_time(ev[i + 1]) for i, k in enumerate(keys)}

    def _capture(self) -> None:
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        state = self._snapshot()
        with torch.cuda.stream(side):            # warm-up: lazy allocations, side streams, pack tables
            self.optimizer.refresh_hyper()
            self._launch()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self._restore(state)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch()

    def _snapshot(self, with_state: bool = True):
        opt, ema = self.optimizer, getattr(self.optimizer, "_ema", None)
        g = opt.param_groups[0]
        return ([p.detach().clone() for p in self.params], g["_step_t"].clone() if with_state else None,
                [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in g["params"] if p in opt.state] if with_state else None,
                self.rng.clone(),
                None if ema is None else ([p.detach().clone() for p in ema.ema_model.parameters()], ema._host_step, ema._host_initted, ema._pending))

    def _restore(self, state, with_state: bool = True) -> None:
        opt, ema = self.optimizer, getattr(self.optimizer, "_ema", None)
        g = opt.param_groups[0]
        with torch.no_grad():
            for p, s0 in zip(self.params, state[0]):
                p.copy_(s0)
            if with_state:
                g["_step_t"].copy_(state[1])
                for p, (m, v) in zip([p for p in g["params"] if p in opt.state], state[2]):
                    opt.state[p]["exp_avg"].copy_(m)
                    opt.state[p]["exp_avg_sq"].copy_(v)
            self.rng.copy_(state[3])
            if ema is not None:
                for p, s0 in zip(ema.ema_model.parameters(), state[4][0]):
                    p.copy_(s0)
                ema._host_step, ema._host_initted, ema._pending = state[4][1], state[4][2], state[4][3]

    @torch.no_grad()
    def __call__(self, x1: torch.Tensor | None = None, condition: dict[str, torch.Tensor] | None = None, *, counts=None, genes=None,
                 counts_subset=None, genes_subset=None) -> torch.Tensor:
        if condition is None or any(k not in condition for k in self.keys):
            raise KeyError(f"FusedTrainStep was built for the condition classes {self.keys}")
        if x1 is None:
            if self.vae is None:
                raise ValueError("FusedTrainStep: pass latents x1, or build it with a frozen vae and pass the tokenised batch")
            c = counts_subset if counts_subset is not None else counts
            gs = genes_subset if genes_subset is not None else genes
            if self.enc is not None:
                self.enc[0].copy_(c, non_blocking=True)
                self.enc[1].copy_(gs, non_blocking=True)
            else:
                x1 = self.vae.encode(counts, genes, counts_subset, genes_subset)
        if x1 is not None:
            if tuple(x1.shape) != tuple(self.x1.shape):
                raise ValueError(f"FusedTrainStep was built for latents {tuple(self.x1.shape)}, got {tuple(x1.shape)}")
            self.x1.copy_(x1, non_blocking=True)
        for k in self.keys:
            self.labels_in[k].copy_(condition[k], non_blocking=True)
        if self.optimizer._generation != self._bound_generation or self.optimizer.max_grad_norm != self._bound_clip:
            self._bind_optimizer()                                 # the optimizer's device state was replaced (checkpoint resume)
        elif self._grad_probe.grad is None:
            self._attach_grads()                                   # zero_grad(set_to_none=True) by the caller
        if self.distributed:
            self._launch()                                         # backward only (opt = NULL): gradients in self.flat
            # in-place, in slices of <= 128 MB of the flat buffer the .grad views share (the path the gloo tests exercise)
            allreduce_gradients([p for p in self.params if p.requires_grad and p is not self.dit.pos_embed], self.group)
            if self._gpos is not None:
                allreduce_gradients([self.dit.pos_embed], self.group)
            if self.found_inf is not None:
                dist.all_reduce(self.found_inf, op=dist.ReduceOp.MAX, group=self.group)
                self.optimizer.found_inf, self.optimizer.grad_scale = self.found_inf, None
            try:
                self.optimizer.step()
            finally:
                if self.found_inf is not None:
                    del self.optimizer.found_inf, self.optimizer.grad_scale
        else:
            self.optimizer.refresh_hyper()
            if self.graph is not None:
                self.graph.replay()
            else:
                self._launch()
        self.steps += 1
        self.optimizer._opt_called = True      # (torch's LR schedulers warn when scheduler.step() is not preceded by an optimizer.step(): the fused call is that step)
        return self.loss
