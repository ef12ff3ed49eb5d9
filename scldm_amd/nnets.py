"""Networks with the reference's API (src/scldm/nnets.py) whose forward runs on MI355X HIP kernels.

`DiT` keeps `scldm.nnets.DiT`'s constructor kwargs, attributes, method signatures and state_dict
keys (nnets.py:216-297,336-378), so `hydra.utils.instantiate` with `_target_: scldm_amd.nnets.DiT`
and `load_state_dict(strict=True)` of reference checkpoints work unchanged.  All arithmetic goes
through libscldm_hip.so (include/scldm_hip.h); there is no eager/CPU fallback.
"""
from __future__ import annotations

import contextlib
import threading

import ctypes as C
import os
from typing import Literal

import torch
import torch.nn as nn

from . import _lib
from .layers import Block, CrossAttentionBlock, FinalLayerDit, TimestepEmbedder, sincos_pos_embed, swiglu_hidden


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: scldm_amd runs only on the MI355X HIP path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Encoder(nn.Module):
    """MCAB pooling encoder - parameter tree of scldm.nnets.Encoder (nnets.py:81-135).  Its arithmetic runs inside
    scldm_amd.vae.TransformerVAE.encode (scldm_vae_encode); calling it on its own is not supported."""

    def __init__(self, n_layer: int, n_inducing_points: int, n_embed: int, n_embed_latent: int, n_head: int, n_head_cross: int,
                 dropout: float, bias: bool, multiple_of: int, layernorm_eps: float, norm_layer: str,
                 positional_encoding: bool = False):
        super().__init__()
        if bias:
            raise NotImplementedError("the MCAB kernels expect bias=False (vae_base.yaml:15)")
        self.latent_embedding = n_embed_latent
        self.latent_dim = n_inducing_points
        self.n_embed, self.n_head, self.n_head_cross, self.n_layer = n_embed, n_head, n_head_cross, n_layer
        self.multiple_of, self.layernorm_eps = multiple_of, layernorm_eps
        self.pos_embed = nn.Parameter(torch.zeros(1, n_inducing_points, n_embed), requires_grad=False) if positional_encoding else None
        self.ca_layer = CrossAttentionBlock(n_embed=n_embed, n_inducing_points=n_inducing_points, n_head=n_head_cross, dropout=dropout,
                                            bias=bias, norm_layer=norm_layer, multiple_of=multiple_of, layernorm_eps=layernorm_eps)
        self.encoder_layers = nn.ModuleList([
            Block(n_embed=n_embed, n_head=n_head, dropout=dropout, bias=bias, norm_layer=norm_layer, multiple_of=multiple_of,
                  layernorm_eps=layernorm_eps) for _ in range(n_layer)])
        self.encoder_latent_input = nn.Sequential(nn.Linear(n_embed, n_embed_latent, bias=bias),
                                                  nn.LayerNorm(n_embed_latent, eps=layernorm_eps, elementwise_affine=False))

    def forward(self, x):  # pragma: no cover - guard only
        raise RuntimeError("Encoder is fused into scldm_amd.vae.TransformerVAE.encode (gene-embedding gather included); "
                           "call TransformerVAE.encode")


class Decoder(nn.Module):
    """MCAB unpooling decoder - parameter tree of scldm.nnets.Decoder (nnets.py:147-198), shared_embedding only."""

    def __init__(self, n_genes: int, n_embed: int, n_embed_latent: int, n_head: int, n_head_cross: int, n_layer: int,
                 n_inducing_points: int, dropout: float, bias: bool, multiple_of: int, layernorm_eps: float, norm_layer: str,
                 shared_embedding: bool, use_adaln: bool = False):
        super().__init__()
        if bias or use_adaln or not shared_embedding:
            raise NotImplementedError("the MCAB kernels cover the reference configuration: bias=False, use_adaln=False, "
                                      "shared_embedding=True (vae_base.yaml:15,36-37)")
        self.gene_embedding = nn.Identity()
        self.n_genes = n_genes
        self.decoder_latent_input = nn.Sequential(nn.LayerNorm(n_embed_latent, eps=layernorm_eps, elementwise_affine=False),
                                                  nn.Linear(n_embed_latent, n_embed, bias=bias))
        self.decoder_layers = nn.ModuleList([
            Block(n_embed=n_embed, n_head=n_head, dropout=dropout, bias=bias, norm_layer=norm_layer, multiple_of=multiple_of,
                  layernorm_eps=layernorm_eps, use_adaln=use_adaln) for _ in range(n_layer)])
        self.decoder_cross_attention = CrossAttentionBlock(n_embed=n_embed, n_inducing_points=0, n_head=n_head_cross, dropout=dropout,
                                                           bias=bias, norm_layer=norm_layer, multiple_of=multiple_of,
                                                           layernorm_eps=layernorm_eps, use_adaln=use_adaln)

    def forward(self, x, genes, condition=None):  # pragma: no cover - guard only
        raise RuntimeError("Decoder is fused into scldm_amd.vae.TransformerVAE.decode; call TransformerVAE.decode")


class _DiTTrainFn(torch.autograd.Function):
    """DiT.forward with a HIP backward: scldm_dit_train_forward / scldm_dit_train_backward (include/scldm_hip.h).
    Replaces torch autograd over scldm.nnets.DiT.forward (nnets.py:273-297) inside Transport.training_losses.

    Host-side cost matters here (a base-DiT step at 1 024 cells is ~2 ms of device time): the parameter list, the pointer
    struct of the weights and the layout of the flat gradient buffer are cached on the module; every gradient is a view of ONE
    allocation."""

    @staticmethod
    def forward(ctx, module, x, t, labels, label_keep, *params):
        L, h = module._native_handle()
        n = x.shape[0]
        dev = x.device
        x = x.detach().contiguous()
        prec = module._prec()
        saved = torch.empty(L.scldm_dit_train_saved_bytes_for(h, n, prec), dtype=torch.uint8, device=dev)
        ws = torch.empty(L.scldm_dit_train_workspace_bytes_for(h, n, prec), dtype=torch.uint8, device=dev)
        w, _ = module._weights_struct(params)
        pkey = (id(w), n, prec)
        if module.__dict__.get("_prepared_key") != pkey:
            # one-time set-up for these parameter tensors (weight mirrors, job tables, side streams): the calls below then only
            # launch kernels.  Re-run when the parameters' storage moved (a new pointer struct), the batch size or precision changes.
            with torch.cuda.device(dev):
                _lib.check(L.scldm_dit_train_prepare(h, C.byref(w), n, prec, _stream_ptr()), "scldm_dit_train_prepare")
            module.__dict__["_prepared_key"] = pkey
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.check(L.scldm_dit_train_forward(h, C.byref(w), x.data_ptr(), t.data_ptr(), C.cast(labels, _lib.c_void_pp), n,
                                                 out.data_ptr(), prec, saved.data_ptr(), ws.data_ptr(), _stream_ptr()),
                       "scldm_dit_train_forward")
        ctx.module, ctx.saved, ctx.ws, ctx.x, ctx.n, ctx.prec = module, saved, ws, x, n, prec
        ctx.labels, ctx.label_keep = labels, label_keep
        ctx.params = params
        ctx.param_versions = [p._version for p in params]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        module, n, params = ctx.module, ctx.n, ctx.params
        L, h = module._native_handle()
        if [p._version for p in params] != ctx.param_versions:
            raise RuntimeError("DiT parameters were modified between forward and backward (the HIP backward reads them live)")
        dout = dout.contiguous().float()
        w, _ = module._weights_struct(params)
        pos_i = module._pos_index(params)
        need_pos = ctx.needs_input_grad[5 + pos_i]
        flat = torch.empty(module._grad_numel, dtype=torch.float32, device=dout.device)   # every gradient is a view of this buffer
        base = flat.data_ptr()
        offs = module._grad_offsets
        gpos = torch.empty_like(module.pos_embed) if need_pos else None
        g, keep_g = module._param_struct(lambda p: (gpos.data_ptr() if gpos is not None else None) if p is module.pos_embed
                                         else base + 4 * offs[id(p)])
        dx = torch.empty_like(ctx.x) if ctx.needs_input_grad[1] else None
        sync = module.__dict__.get("_grad_sync")      # scldm_amd.training.OverlappedGradSync while a data-parallel step runs
        if sync is not None:
            sync.before_backward(module, L, h)
        with torch.cuda.device(dout.device):
            _lib.check(L.scldm_dit_train_backward(h, C.byref(w), C.byref(g), ctx.x.data_ptr(), C.cast(ctx.labels, _lib.c_void_pp),
                                                  dout.data_ptr(), n, dx.data_ptr() if dx is not None else None,
                                                  ctx.prec, ctx.saved.data_ptr(), ctx.ws.data_ptr(), _stream_ptr()),
                       "scldm_dit_train_backward")
        del keep_g
        ctx.saved = ctx.ws = None
        if sync is not None:
            sync.after_backward(flat)        # the call above only ENQUEUED the kernels: the bucket all-reduces are queued behind their events
        out = []
        for i, p in enumerate(params):
            if not ctx.needs_input_grad[5 + i]:
                out.append(None)
            elif p is module.pos_embed:
                out.append(gpos)
            else:
                o = offs[id(p)]
                out.append(flat[o:o + p.numel()].view(p.shape))
        return (None, dx, None, None, None, *out)


class _PerThreadDepth(threading.local):
    depth = 0


_ASSUME_WEIGHTS_UNCHANGED = _PerThreadDepth()      # (per thread: another thread's sampler loop says nothing about this thread's updates)


@contextlib.contextmanager
def weights_unchanged():
    """Inside this block a DiT trusts its packed weight copies whenever parameter storages and torch version counters are unchanged: the
    per-call device-side fingerprint pass (there for in-place `.data` updates, which those counters do not see) is skipped.  For loops that
    call the model many times with no parameter update in between - `scldm_amd.transport.Sampler`'s evaluation loops use it after their
    first evaluation (the reference's default dopri5 solve: 110 calls of forward_with_cfg)."""
    _ASSUME_WEIGHTS_UNCHANGED.depth += 1
    try:
        yield
    finally:
        _ASSUME_WEIGHTS_UNCHANGED.depth -= 1


class DiT(nn.Module):
    """Diffusion Transformer (adaLN-Zero) - drop-in for scldm.nnets.DiT.

    Extra (non-reference) knobs: `precision` - "fp32" (exact-fp32 MFMA), "bf16x3" (split-bf16, three bf16 MFMAs per
    product sum; both inside the 1e-4 parity gate vs exact fp32), "fp16" (fp16 operands = TF32's 10 mantissa bits, the
    arithmetic the reference itself runs under `set_float32_matmul_precision("high")`, inference.py:26 - ~1e-3 vs exact fp32,
    like the reference; weights are range-checked against +-65 504 when packed) or "bf16" (8 bits: throughput path); default
    from $SCLDM_PRECISION, else "fp32".  Training: "bf16" and - since round 4 - "fp16" (the reference's own training arithmetic class,
    train_ldm.py:18) run the fused route of the base shape at the matrix-core rate (fp16: the backward is loss-scaled on device with an
    overflow guard - `fp16_train_state()`, `found_inf_flag()`; the small conditioning / adaLN GEMMs around the fused layers use fp16 operands
    as well, SCLDM_TRAIN_FP16_EXACT_SMALL=1 keeps those exact fp32); "bf16x3", and "fp16" on shapes outside the fused family (e.g. a
    DiT-L), are served by the exact-fp32 GEMM route.

    The fused inference kernels read PACKED copies of the parameters.  They are refreshed automatically when a parameter's
    storage or torch version counter changes, and an on-device fingerprint of the parameters is re-checked at every call so
    that in-place updates through `.data` (ema_pytorch, reference models.py:446,690) are picked up as well;
    `invalidate_weights()` forces a re-pack.  `copy.deepcopy` / pickling drop the native handle (the copy builds its own).
    """

    _NATIVE_STATE = ("_handle", "_weights_key", "_ws", "_dedup_cache")

    def __init__(
        self,
        n_embed: int,
        n_embed_input: int,
        n_layer: int,
        n_head: int,
        seq_len: int,
        dropout: float,
        bias: bool,
        norm_layer: str,
        multiple_of: int,
        layernorm_eps: float,
        class_vocab_sizes: dict[str, int],
        cfg_dropout_prob: float = 0.1,
        condition_strategy: Literal["mutually_exclusive", "joint"] = "mutually_exclusive",
    ):
        super().__init__()
        if not bias:
            raise NotImplementedError("the fused DiT block expects bias=True (ldm_base.yaml:22)")
        self.class_vocab_sizes = dict(class_vocab_sizes)
        self.cfg_dropout_prob = cfg_dropout_prob
        self.condition_strategy = condition_strategy
        self.class_embeddings = nn.ModuleDict()
        for name, vocab in self.class_vocab_sizes.items():
            self.class_embeddings[name] = nn.Embedding(vocab + int(cfg_dropout_prob > 0), n_embed)
        self.t_embedder = TimestepEmbedder(n_embed)
        self.pos_embed = nn.Parameter(torch.zeros(1, seq_len, n_embed), requires_grad=False)
        self.blocks = nn.ModuleList([
            Block(n_embed=n_embed, n_head=n_head, dropout=dropout, bias=bias, norm_layer=norm_layer, multiple_of=multiple_of,
                  layernorm_eps=layernorm_eps, use_adaln=True, elementwise_affine=False) for _ in range(n_layer)
        ])
        self.n_embed, self.seq_len = n_embed, seq_len
        self.n_embed_input, self.n_layer, self.n_head = n_embed_input, n_layer, n_head
        self.layernorm_eps, self.multiple_of = layernorm_eps, multiple_of
        self.input_proj = nn.Linear(n_embed_input, n_embed, bias=bias)
        self.final_layer = FinalLayerDit(n_embed, n_embed_input, bias, layernorm_eps)
        self.precision = os.environ.get("SCLDM_PRECISION", "fp32")
        self.detect_uniform_t = True   # forward_with_cfg: test a dense t for uniformity on device (no host sync); see there
        # Non-reference knob, off by default: with ONE conditional pass at guidance scale exactly 1.0 the guided half
        # u2 + 1.0 * (c2 - u2) (nnets.py:368,376) equals the conditional output c2 up to one fp32 rounding, so the fused sampler /
        # scalar-t forward_with_cfg skip the unconditional forward of the second half (2B instead of 3B sample-forwards per
        # evaluation: 1.5 x the cells/s at the reference's default guidance of dentate_gyrus / parse1m).
        self.guidance1_direct = False
        self.tail_split = None          # SCLDM_OPT_TAIL_SPLIT (None: the library default, off - measured slower): 32-token tiles for a partial last round
        self._handle = None
        self._weights_key = None
        self._ws = None
        self._dedup_cache = {}
        # True: the CFG plan of the samplers never waits for the host (dense label-tuple rows instead of torch.unique); out-of-range labels
        # are then clamped and reported by check_labels() instead of raising at once.  scldm_amd.sampling.generate_cells_stream sets it
        # for the duration of its loop.
        self.deferred_label_check = False
        self.initialize_weights()

    # ------------------------------------------------------------------ init (nnets.py:458-492)
    def initialize_weights(self):
        def _basic(m):
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

        self.apply(_basic)
        self.pos_embed.data.copy_(torch.from_numpy(sincos_pos_embed(self.n_embed, self.seq_len)).float().unsqueeze(0))
        for emb in self.class_embeddings.values():
            nn.init.normal_(emb.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for blk in self.blocks:  # adaLN-Zero
            nn.init.zeros_(blk.adaln_modulation[-1].weight)
            nn.init.zeros_(blk.adaln_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.adaln_modulation[-1].weight)
        nn.init.zeros_(self.final_layer.adaln_modulation[-1].bias)
        nn.init.zeros_(self.final_layer.linear.weight)
        nn.init.zeros_(self.final_layer.linear.bias)

    # ------------------------------------------------------------------ native handle management
    @property
    def _class_names(self) -> list[str]:
        return sorted(self.class_vocab_sizes.keys())

    @property
    def _has_null_row(self) -> bool:
        """The class tables carry a null-token row only when the model was built with cfg_dropout_prob > 0 (nnets.py:241-243)."""
        names = self._class_names
        return (not names) or self.class_embeddings[names[0]].num_embeddings == self.class_vocab_sizes[names[0]] + 1

    def _need_null_row(self, what: str):
        if not self._has_null_row:   # the reference indexes row `vocab` of a `vocab`-row table here: IndexError
            raise IndexError(f"{what} needs the null-token rows of the class embeddings, but this DiT was built with "
                             "cfg_dropout_prob == 0 (index out of range in self)")

    def _native_handle(self):
        """The C handle without touching the packed inference weights (the training path reads parameters live)."""
        L = _lib.lib()
        dev = self.pos_embed.device
        if dev.type != "cuda":
            raise RuntimeError("DiT parameters must live on a CUDA (ROCm) device; call .cuda() first. There is no CPU path.")
        if self._handle is None:
            names = self._class_names
            if len(names) > _lib.MAX_CLASSES:
                raise ValueError(f"at most {_lib.MAX_CLASSES} condition classes are supported")
            cfg = _lib.DitConfig(n_embed=self.n_embed, n_embed_input=self.n_embed_input, n_layer=self.n_layer, n_head=self.n_head,
                                 seq_len=self.seq_len, hidden_dim=swiglu_hidden(self.n_embed, self.multiple_of),
                                 layernorm_eps=self.layernorm_eps, n_classes=len(names), has_null_row=int(self._has_null_row))
            for i, n in enumerate(names):
                cfg.class_vocab[i] = self.class_vocab_sizes[n]
                if self.class_embeddings[n].num_embeddings != self.class_vocab_sizes[n] + int(self._has_null_row):
                    raise ValueError(f"class_embeddings[{n!r}] has {self.class_embeddings[n].num_embeddings} rows, expected "
                                     f"{self.class_vocab_sizes[n] + int(self._has_null_row)}")
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(L.scldm_dit_create(C.byref(cfg), C.byref(h)), "scldm_dit_create")
            self._handle = h
        return L, self._handle

    def found_inf_flag(self) -> torch.Tensor:
        """Device float the fp16 training backward sets to 1.0 when it produced non-finite gradients (reset to 0.0 at the start of every
        fp16 backward): hand it to an optimizer that understands GradScaler's `found_inf` (torch's fused Adam / AdamW skip the step on
        device, no host read) - `scldm_amd.training.train_step` does.  The fp16 counterpart of torch.cuda.amp.GradScaler for the
        reference's trainer (experiments/scripts/train_ldm.py)."""
        L, h = self._native_handle()
        flag = self.__dict__.get("_found_inf")
        if flag is None or flag.device != self.pos_embed.device:
            flag = self.__dict__["_found_inf"] = torch.zeros((), dtype=torch.float32, device=self.pos_embed.device)
            self.__dict__["_found_inf_handle"] = None
        if self.__dict__.get("_found_inf_handle") != h.value:
            # (re-)register with THIS handle: a copy / unpickled module, or a handle re-created after .to(device), starts with none
            # (a stale cached flag left the copy's overflow guard silently off)
            _lib.check(L.scldm_dit_train_set_found_inf(h, flag.data_ptr()), "scldm_dit_train_set_found_inf")
            self.__dict__["_found_inf_handle"] = h.value
        return flag

    def fp16_train_state(self) -> dict:
        """Loss-scale state of the fp16 backward (synchronises the current stream): the scale S of the last backward, the number of
        non-finite gradient values it produced, the headroom (powers of two below the nominal scale, <= 0) and the count of
        backwards that overflowed so far."""
        L, h = self._native_handle()
        S, bad, head, steps = C.c_float(), C.c_longlong(), C.c_int(), C.c_longlong()
        with torch.cuda.device(self.pos_embed.device):
            _lib.check(L.scldm_dit_train_fp16_state(h, C.byref(S), C.byref(bad), C.byref(head), C.byref(steps), _stream_ptr()), "scldm_dit_train_fp16_state")
        return {"scale": S.value, "nonfinite_last": bad.value, "headroom": head.value, "overflow_steps": steps.value}

    def _native(self):
        L, _ = self._native_handle()
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key != self._weights_key:
            self._load_weights(L)
            self._weights_key = key
        elif not _ASSUME_WEIGHTS_UNCHANGED.depth:
            # same storages and version counters: `.data` updates (EMA) are invisible to both, so the C side compares a
            # device-side fingerprint of the parameters and re-packs in stream order if it moved (no host synchronisation).
            # Skipped inside `weights_unchanged()` (the samplers' evaluation loops: 25 + 5 + 10 us of launches per call).
            with torch.cuda.device(self.pos_embed.device):
                _lib.check(L.scldm_dit_refresh_weights(self._handle, _stream_ptr()), "scldm_dit_refresh_weights")
        _lib.check(L.scldm_dit_set_option(self._handle, _lib.OPT_CFG1_DIRECT, int(bool(getattr(self, "guidance1_direct", False)))), "scldm_dit_set_option")
        if getattr(self, "tail_split", None) is not None:
            _lib.check(L.scldm_dit_set_option(self._handle, _lib.OPT_TAIL_SPLIT, int(bool(self.tail_split))), "scldm_dit_set_option")
        if self.precision == "fp16" and self.__dict__.get("_fp16_checked") != self._weights_key:
            # once per (re)load: the fp16 stream's range report (one stream synchronisation).  `.data` updates that are picked up
            # by the fingerprint re-pack are not re-checked: call fp16_weight_report() after such an update if in doubt.
            self.fp16_weight_report(strict=True)
            self.__dict__["_fp16_checked"] = self._weights_key
        return L, self._handle

    def fp16_weight_report(self, strict: bool = False) -> dict:
        """Range report of the packed fp16 weight stream (precision "fp16"): `overflow` values beyond +-65 504 (stored as inf),
        `subnormal` non-zero values below 6.1e-5 (fewer than 10 mantissa bits), `nonzero` values packed.  strict: raise on any
        overflow, warn when more than 1 % of the non-zero weights are subnormal.  Synchronises the stream."""
        if self._handle is None:
            self._native_handle()
        o, sn, nz = C.c_longlong(), C.c_longlong(), C.c_longlong()
        with torch.cuda.device(self.pos_embed.device):
            _lib.check(_lib.lib().scldm_dit_fp16_stats(self._handle, C.byref(o), C.byref(sn), C.byref(nz), _stream_ptr()), "scldm_dit_fp16_stats")
        rep = {"overflow": o.value, "subnormal": sn.value, "nonzero": nz.value}
        if strict:
            if rep["overflow"]:
                raise ValueError(f"precision='fp16': {rep['overflow']} weight(s) exceed the fp16 range (|w| > 65504); use 'bf16x3' or 'fp32'")
            if rep["nonzero"] and rep["subnormal"] > 0.01 * rep["nonzero"]:
                import warnings
                warnings.warn(f"precision='fp16': {rep['subnormal']} of {rep['nonzero']} weights are below the smallest fp16 normal (6.1e-5) "
                              "and lose mantissa bits", RuntimeWarning, stacklevel=3)
        return rep

    def invalidate_weights(self) -> None:
        """Force the next call to re-pack the fused kernels' weight copies from the parameters."""
        self._weights_key = None

    def check_labels(self) -> int:
        """Number of out-of-range condition labels the kernels clamped since the last call (synchronises the stream).
        Raises IndexError like the reference's nn.Embedding (nnets.py:420,453) if there were any."""
        if self._handle is None:
            return 0
        n = C.c_int(0)
        with torch.cuda.device(self.pos_embed.device):
            _lib.check(_lib.lib().scldm_dit_label_errors(self._handle, C.byref(n), _stream_ptr()), "scldm_dit_label_errors")
        deferred = self.__dict__.pop("_label_bad", None)        # (labels clamped by the sync-free CFG plan, `deferred_label_check`)
        total = n.value + (int(deferred) if deferred is not None else 0)
        if total:
            raise IndexError(f"{total} condition label(s) were outside their class vocabulary (index out of range in self)")
        return 0

    # ------------------------------------------------------------------ copies / pickling: the native handle never travels
    def __getstate__(self):
        state = self.__dict__.copy()
        state.update(_handle=None, _weights_key=None, _ws=None, _dedup_cache={})
        for k in ("_wstruct_cache", "_grad_offsets", "_grad_numel", "_grad_segs", "_pos_idx", "_param_list", "_prepared_key", "_grad_sync",
                  "_train_step_sync", "_fp16_checked", "_found_inf", "_found_inf_handle", "_dense_rows", "_label_bad"):   # (_found_inf is registered with the handle that does not travel)
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        for k, v in (("_handle", None), ("_weights_key", None), ("_ws", None), ("_dedup_cache", {}), ("deferred_label_check", False)):
            self.__dict__.setdefault(k, v)

    def _param_struct(self, dp):
        """scldm_dit_weights / scldm_dit_grads (same field order) filled with dp(parameter) -> device pointer or None.
        Returns (struct, keepalive)."""
        blocks = list(self.blocks)
        keep = [
            _lib.ptr_array([dp(self.class_embeddings[n].weight) for n in self._class_names]),
            _lib.ptr_array([dp(b.attn.c_attn.weight) for b in blocks]), _lib.ptr_array([dp(b.attn.c_attn.bias) for b in blocks]),
            _lib.ptr_array([dp(b.attn.c_proj.weight) for b in blocks]), _lib.ptr_array([dp(b.attn.c_proj.bias) for b in blocks]),
            _lib.ptr_array([dp(b.mlp.w1.weight) for b in blocks]), _lib.ptr_array([dp(b.mlp.w2.weight) for b in blocks]),
            _lib.ptr_array([dp(b.mlp.c_proj.weight) for b in blocks]),
            _lib.ptr_array([dp(b.adaln_modulation[1].weight) for b in blocks]),
            _lib.ptr_array([dp(b.adaln_modulation[1].bias) for b in blocks]),
        ]
        cast = lambda a: C.cast(a, _lib.c_void_pp)
        w = _lib.DitWeights(
            pos_embed=dp(self.pos_embed), t_w0=dp(self.t_embedder.mlp[0].weight), t_b0=dp(self.t_embedder.mlp[0].bias),
            t_w2=dp(self.t_embedder.mlp[2].weight), t_b2=dp(self.t_embedder.mlp[2].bias), in_w=dp(self.input_proj.weight),
            in_b=dp(self.input_proj.bias), fin_w=dp(self.final_layer.linear.weight), fin_b=dp(self.final_layer.linear.bias),
            fin_ada_w=dp(self.final_layer.adaln_modulation[1].weight), fin_ada_b=dp(self.final_layer.adaln_modulation[1].bias),
            class_emb=cast(keep[0]), attn_w=cast(keep[1]), attn_b=cast(keep[2]), proj_w=cast(keep[3]), proj_b=cast(keep[4]),
            w1=cast(keep[5]), w2=cast(keep[6]), cproj=cast(keep[7]), ada_w=cast(keep[8]), ada_b=cast(keep[9]))
        return w, keep

    def _weights_struct(self, params):
        """scldm_dit_weights of the live parameters, rebuilt only when a parameter's storage moved (the optimizer updates in place)."""
        key = tuple(p.data_ptr() for p in params)
        c = self.__dict__.get("_wstruct_cache")
        if c is None or c[0] != key:
            self._check_params()
            w, keep = self._param_struct(lambda t: t.data_ptr())
            # the flat gradient buffer is laid out in the order the backward COMPLETES the gradients (grad_segments()), so that a
            # data-parallel caller can all-reduce contiguous slices of it in place while later slices are still being computed
            offs, total = {}, 0
            segs = []
            for kind, layer, plist in self.grad_segments():
                start = total
                for p in plist:
                    offs[id(p)] = total
                    total += (p.numel() + 63) // 64 * 64       # 256-byte aligned views
                segs.append((kind, layer, start, total))
            missing = [p for p in params if p is not self.pos_embed and id(p) not in offs]
            assert not missing, "grad_segments() must cover every parameter"
            self.__dict__["_wstruct_cache"] = c = (key, w, keep)
            self.__dict__["_grad_offsets"], self.__dict__["_grad_numel"], self.__dict__["_grad_segs"] = offs, total, segs
            self.__dict__["_pos_idx"] = next(i for i, p in enumerate(params) if p is self.pos_embed)
        return c[1], c[2]

    def grad_segments(self):
        """[(kind, layer, [parameters])] in the order scldm_dit_train_backward completes their gradients (include/scldm_hip.h,
        SCLDM_GRAD_*): the main weights of the layers from LAST to first ("layer", l), then the adaLN projections ("ada", n_layer: weights
        in layer order, then biases), then everything else ("end")."""
        segs = []
        for l in range(self.n_layer - 1, -1, -1):
            b = self.blocks[l]
            segs.append(("layer", l, [b.attn.c_attn.weight, b.attn.c_attn.bias, b.attn.c_proj.weight, b.attn.c_proj.bias,
                                      b.mlp.w1.weight, b.mlp.w2.weight, b.mlp.c_proj.weight]))
        # adaLN projections: every weight in layer order (the final layer's last), then every bias - the layout of the stacked matrix,
        # so that the bf16 route writes all of them with ONE weight-gradient product; they complete together (trigger: ada, n_layer)
        ada = [self.blocks[l].adaln_modulation[1] for l in range(self.n_layer)] + [self.final_layer.adaln_modulation[1]]
        for m in ada:
            segs.append(("ada", self.n_layer, [m.weight]))
        segs.append(("ada", self.n_layer, [m.bias for m in ada]))
        rest = [self.class_embeddings[n].weight for n in self._class_names]
        rest += [self.t_embedder.mlp[0].weight, self.t_embedder.mlp[0].bias, self.t_embedder.mlp[2].weight, self.t_embedder.mlp[2].bias,
                 self.input_proj.weight, self.input_proj.bias, self.final_layer.linear.weight, self.final_layer.linear.bias]
        segs.append(("end", 0, rest))
        return segs

    def grad_bucket_plan(self, bucket_bytes: int = 128 << 20):
        """Contiguous slices [(start, end, kind, layer)] (element offsets into the flat gradient buffer of the HIP backward) of at
        most ~bucket_bytes each, in completion order, each with the SCLDM_GRAD_* trigger after which the whole slice is final.
        Slices never mix kinds; a single segment larger than bucket_bytes is its own slice."""
        self._weights_struct(tuple(self.parameters()))
        plan, cur = [], None
        for kind, layer, a, b in self.__dict__["_grad_segs"]:
            if b == a:
                continue
            if cur is not None and cur[2] == kind and 4 * (b - cur[0]) <= bucket_bytes:
                cur = [cur[0], b, kind, layer]          # the trigger of the LAST segment of a run covers the earlier ones
            else:
                if cur is not None:
                    plan.append(tuple(cur))
                cur = [a, b, kind, layer]
        if cur is not None:
            plan.append(tuple(cur))
        return plan

    def _pos_index(self, params):
        return self.__dict__["_pos_idx"]

    def _check_params(self):
        for p in self.parameters():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("DiT parameters must be contiguous fp32 (master weights); bf16 copies are derived internally")

    def _load_weights(self, L):
        self._check_params()
        w, keep = self._param_struct(lambda t: t.data_ptr())
        with torch.cuda.device(self.pos_embed.device):
            _lib.check(L.scldm_dit_load_weights(self._handle, C.byref(w), _stream_ptr()), "scldm_dit_load_weights")
        del keep

    def _workspace(self, L, n_fwd: int, n_rows: int, n_state: int) -> int:
        need = L.scldm_dit_workspace_bytes(self._handle, n_fwd, n_rows, n_state)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.pos_embed.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.pos_embed.device)
        return self._ws.data_ptr()

    def _prec(self) -> int:
        try:
            return _lib.PRECISIONS[self.precision]
        except KeyError:
            raise ValueError(f"precision must be one of {list(_lib.PRECISIONS)}") from None

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().scldm_dit_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ shapes outside the fused family
    @property
    def fused_shape(self) -> bool:
        """True for the reference's DiT shape family (256 wide, 8 heads, 16 tokens), served by the fused layer kernel.
        Other shapes (n_embed % 256 == 0, head_dim 32/64, e.g. a DiT-L) run on the generic GEMM-based HIP path: the
        training kernels, also used for their inference."""
        return self.n_embed == 256 and self.n_head == 8 and self.seq_len == 16 and self.n_embed_input <= 32

    def _generic_forward(self, x: torch.Tensor, t: torch.Tensor, labels) -> torch.Tensor:
        """DiT.forward through scldm_dit_train_forward without keeping the activation record (inference on non-fused shapes)."""
        L, h = self._native_handle()
        self._check_params()
        n = x.shape[0]
        saved = torch.empty(L.scldm_dit_train_saved_bytes(h, n), dtype=torch.uint8, device=x.device)
        ws = torch.empty(L.scldm_dit_train_workspace_bytes(h, n), dtype=torch.uint8, device=x.device)
        w, keep = self._param_struct(lambda p: p.data_ptr())
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(L.scldm_dit_train_forward(h, C.byref(w), x.data_ptr(), t.data_ptr(), C.cast(labels, _lib.c_void_pp), n,
                                                 out.data_ptr(), self._prec(), saved.data_ptr(), ws.data_ptr(), _stream_ptr()),
                       "scldm_dit_train_forward")
        del keep
        return out

    def _generic_forward_with_cfg(self, x, t, condition, cfg_scale):
        """forward_with_cfg composed from forward passes exactly as the reference does (nnets.py:336-378)."""
        n = x.shape[0]
        B = n // 2
        names = self._class_names
        null = _lib.ptr_array([None] * len(names))
        u = self._generic_forward(x, t, null)                                    # all-null conditioning, :352-356
        if condition is None or cfg_scale is None:
            return u
        x2, t2, u2 = x[B:].contiguous(), t[B:].contiguous(), u[B:]
        half = {k: v[B:].to(device=x.device, dtype=torch.long).contiguous() for k, v in condition.items()}

        def cond_pass(keys):
            ptrs = _lib.ptr_array([half[c].data_ptr() if c in keys else None for c in names])
            return self._generic_forward(x2, t2, ptrs)

        if self.condition_strategy == "joint":                                   # :364-369
            for c in names:
                if c not in half:
                    raise KeyError(c)
            g = u2 + (sum(cfg_scale.values()) / len(cfg_scale)) * (cond_pass(set(names)) - u2)
        else:                                                                    # :372-376
            g = u2.clone()
            for c, s in cfg_scale.items():
                if c not in half or c not in names:
                    raise KeyError(c)
                g = g + float(s) * (cond_pass({c}) - u2)
        return torch.cat([u[:B], g], dim=0)

    # ------------------------------------------------------------------ label handling (nnets.py:380-456)
    def _label_ptrs(self, condition: dict[str, torch.Tensor], n: int, force_drop_ids: bool):
        """Device label pointers per class (sorted-name order); None -> the class uses its null token.

        Eval: mutually_exclusive keeps the class drawn by torch.randint among the available ones (the reference draws it even
        in eval, nnets.py:395), joint keeps every class (nnets.py:428-456).  Training / force_drop_ids additionally replaces
        labels by the null token with probability cfg_dropout_prob (one mask per batch row, nnets.py:401-402,440-443)."""
        names = self._class_names
        available = [c for c in names if c in condition]
        if not available:
            raise ValueError("condition must contain at least one known class (reference raises StopIteration/KeyError here)")
        keep = []
        dev = self.pos_embed.device
        if self.condition_strategy == "joint":
            missing = [c for c in names if c not in condition]
            if missing:
                raise KeyError(missing[0])  # the reference indexes condition[class_name] for every class (nnets.py:449)
            chosen = set(names)
            drop = self.training  # joint: dropout only in training mode (nnets.py:440-445)
        else:
            sel = int(torch.randint(0, len(available), ()).item()) if len(available) > 1 else 0
            chosen = {available[sel]}
            drop = force_drop_ids
        drop_mask = (torch.rand(n, device=dev) < self.cfg_dropout_prob) if drop else None
        if drop_mask is not None or any(c not in chosen for c in names):
            self._need_null_row("label dropout / an unselected condition class")
        ptrs = []
        for c in names:
            if c in chosen:
                lab = condition[c]
                if lab.shape[0] != n:
                    raise ValueError(f"Condition '{c}' length ({lab.shape[0]}) must match batch size ({n})")
                lab = lab.to(device=dev, dtype=torch.long)
                if drop_mask is not None:
                    lab = torch.where(drop_mask, self.class_vocab_sizes[c], lab)   # null token where dropped (ONE launch: masked_fill clones first)
                lab = lab.contiguous()
                keep.append(lab)
                ptrs.append(lab.data_ptr())
            else:
                ptrs.append(None)
        return _lib.ptr_array(ptrs), keep

    # ------------------------------------------------------------------ forward (nnets.py:273-297)
    def forward(self, x: torch.Tensor, t: torch.Tensor, condition: dict[str, torch.Tensor],
                force_drop_ids: bool | None = None) -> torch.Tensor:
        if force_drop_ids is None:
            force_drop_ids = self.training
        if not self.training:
            assert not force_drop_ids, "force_drop_ids must be False when not training"
        x = _require_cuda_f32("x", x)
        t = _require_cuda_f32("t", t)
        n = x.shape[0]
        if x.shape[1:] != (self.seq_len, self.n_embed_input) or t.shape != (n,):
            raise ValueError(f"expected x (B,{self.seq_len},{self.n_embed_input}) and t (B,), got {tuple(x.shape)}, {tuple(t.shape)}")
        labels, keep = self._label_ptrs(condition, n, force_drop_ids)
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            # differentiable path (Transport.training_losses -> loss.backward()): forward with saved activations + HIP backward.
            # In eval mode the fused inference kernel runs unless the input itself requires grad: an eval-mode output is not
            # differentiable w.r.t. the parameters (loss.backward() then fails loudly, it never returns silent zeros).
            params = tuple(self.parameters())     # one traversal per step: the identity of every Parameter object is re-checked
            cached = self.__dict__.get("_param_list")
            if cached is None or len(cached) != len(params) or any(a is not b for a, b in zip(cached, params)):
                self.__dict__["_param_list"] = params
                self.__dict__.pop("_wstruct_cache", None)
            return _DiTTrainFn.apply(self, x, t, labels, keep, *params)
        if not self.fused_shape:
            return self._generic_forward(x, t, labels)
        L, h = self._native()
        out = torch.empty_like(x)
        ws = self._workspace(L, n, n, 0)
        with torch.cuda.device(x.device):
            _lib.check(L.scldm_dit_forward(h, x.data_ptr(), t.data_ptr(), C.cast(labels, _lib.c_void_pp), out.data_ptr(), n,
                                           self._prec(), ws, _stream_ptr()), "scldm_dit_forward")
        del keep
        return out

    # ------------------------------------------------------------------ CFG plan shared by forward_with_cfg / sample_ode_cfg
    def _cfg_plan(self, condition, cfg_scale, B: int, dedup: bool):
        """Returns (ulabel_ptrs, n_urows, cell_row_ptr, n_pass, masks, scales, keepalive)."""
        names = self._class_names
        if condition is None or cfg_scale is None:
            return None, 0, None, 0, None, None, []
        half = {}
        for k, v in condition.items():
            if v.shape[0] != 2 * B:
                raise ValueError(f"Condition '{k}' length ({v.shape[0]}) must match batch size ({2 * B})")
            half[k] = v[B:].to(device=self.pos_embed.device, dtype=torch.long)
        if self.condition_strategy == "joint":
            for c in names:
                if c not in half:
                    raise KeyError(c)
            used = list(names)
            masks = [sum(1 << i for i in range(len(names)))]
            scales = [sum(cfg_scale.values()) / len(cfg_scale)]  # nnets.py:368
        else:
            used = [c for c in cfg_scale.keys()]
            for c in used:
                if c not in half:
                    raise KeyError(c)
                if c not in names:
                    raise KeyError(c)
            masks = [1 << names.index(c) for c in used]
            scales = [float(cfg_scale[c]) for c in used]
        keep = []
        cell_row_ptr = None
        n_u = B
        cols = {c: half[c].contiguous() for c in used}
        dense_total = 1
        for c in used:
            dense_total *= int(self.class_vocab_sizes[c]) + 1
        if dedup and used and dense_total < B and dense_total <= 4096:
            # Dense plan (round 6): one conditioning row per POSSIBLE label tuple (15 for the dentate vocabulary) instead of per tuple
            # present in this batch - no torch.unique (a sort, a scan and a dozen small launches per call).  With `deferred_label_check`
            # it is also sync-free (the prediction loop queues batch i + 1's solve while batch i's is still running): out-of-range labels
            # are clamped and counted on the device like the plain forward's and `check_labels()` raises for them; otherwise the count is
            # read here and raises at once, as the torch.unique path below does.
            sizes = [int(self.class_vocab_sizes[c]) + 1 for c in used]
            dkey = (tuple(used), str(self.pos_embed.device))
            dense = self.__dict__.setdefault("_dense_rows", {}).get(dkey)
            if dense is None:
                idx = torch.arange(dense_total, device=self.pos_embed.device, dtype=torch.long)
                strides, acc = [], 1
                for sz in reversed(sizes):
                    strides.append(acc)
                    acc *= sz
                strides = strides[::-1]
                dense = ({c: ((idx // st) % sz).contiguous() for c, st, sz in zip(used, strides, sizes)}, strides)
                self.__dict__["_dense_rows"][dkey] = dense
            ucols, strides = dense
            # (same label tensors as the last call - a host-driven solver evaluating forward_with_cfg again and again: the row map is
            # reused, nothing is launched and nothing is read back; the entry keeps the label tensors alive so that their addresses
            # cannot be handed to other tensors)
            key = ("dense",) + tuple((c, cols[c].data_ptr(), cols[c]._version, B) for c in used)
            hit = self._dedup_cache.get(key)
            if hit is None:
                inv = torch.zeros(B, dtype=torch.long, device=self.pos_embed.device)
                bad = torch.zeros((), dtype=torch.long, device=self.pos_embed.device)
                for c, st, sz in zip(used, strides, sizes):
                    col = cols[c]
                    bad = bad + ((col < 0) | (col >= sz)).sum()
                    inv = inv + col.clamp(0, sz - 1) * st
                if self.deferred_label_check:
                    acc_bad = self.__dict__.get("_label_bad")
                    self.__dict__["_label_bad"] = bad if acc_bad is None else acc_bad + bad
                elif int(bad):      # (the one host read of a call with new labels)
                    raise IndexError(f"{int(bad)} condition label(s) were outside their class vocabulary (index out of range in self)")
                hit = (inv.to(torch.int32).contiguous(), [cols[c] for c in used])
                self._dedup_cache.clear()
                self._dedup_cache[key] = hit
            inv = hit[0]
            cols, n_u, cell_row_ptr = ucols, dense_total, inv.data_ptr()
            keep.append(inv)
        elif dedup and used:
            key = tuple((c, cols[c].data_ptr(), cols[c]._version, B) for c in used)
            hit = self._dedup_cache.get(key)
            if hit is None:
                stacked = torch.stack([cols[c] for c in used], dim=1)
                uniq, inv = torch.unique(stacked, dim=0, return_inverse=True)
                lo, hi = uniq.min(dim=0).values.tolist(), uniq.max(dim=0).values.tolist()   # unique() synchronised already
                for i, c in enumerate(used):   # the reference's nn.Embedding raises on such labels (nnets.py:420,453)
                    if lo[i] < 0 or hi[i] > self.class_vocab_sizes[c]:
                        raise IndexError(f"condition '{c}' has labels outside [0, {self.class_vocab_sizes[c]}] (index out of range in self)")
                hit = ({c: uniq[:, i].contiguous() for i, c in enumerate(used)}, inv.to(torch.int32).contiguous(), [cols[c] for c in used])
                self._dedup_cache.clear()
                self._dedup_cache[key] = hit
            ucols, inv, _ = hit
            if ucols[used[0]].shape[0] < B:
                cols, n_u, cell_row_ptr = ucols, ucols[used[0]].shape[0], inv.data_ptr()
                keep.append(inv)
        ptrs = []
        for c in names:
            if c in cols:
                keep.append(cols[c])
                ptrs.append(cols[c].data_ptr())
            else:
                ptrs.append(None)
        n_pass = len(masks)
        m = (C.c_uint32 * max(n_pass, 1))(*masks)
        s = (C.c_float * max(n_pass, 1))(*scales)
        return _lib.ptr_array(ptrs), n_u, cell_row_ptr, n_pass, m, s, keep

    # ------------------------------------------------------------------ forward_with_cfg (nnets.py:336-378)
    def forward_with_cfg(self, x: torch.Tensor, t: torch.Tensor, condition: dict[str, torch.Tensor] | None = None,
                         cfg_scale: dict[str, float] | None = None) -> torch.Tensor:
        if self.training:   # the reference's forward_with_cfg works in either mode (it passes force_drop_ids=False, nnets.py:353,367,375)
            return self._composed_forward_with_cfg(x, t, condition, cfg_scale)
        if not self.fused_shape:
            xg, tg = _require_cuda_f32("x", x), _require_cuda_f32("t", t)
            if xg.shape[0] % 2 or xg.shape[1:] != (self.seq_len, self.n_embed_input) or tg.shape != (xg.shape[0],):
                raise ValueError(f"expected x (2B,{self.seq_len},{self.n_embed_input}) and t (2B,), got {tuple(xg.shape)}, {tuple(tg.shape)}")
            with torch.no_grad():
                return self._generic_forward_with_cfg(xg, tg, condition, cfg_scale)
        self._need_null_row("forward_with_cfg (the unconditional pass)")
        L, h = self._native()
        x = _require_cuda_f32("x", x)
        n = x.shape[0]
        B = n // 2
        if n != 2 * B or x.shape[1:] != (self.seq_len, self.n_embed_input) or tuple(t.shape) != (n,):
            raise ValueError(f"expected x (2B,{self.seq_len},{self.n_embed_input}) and t (2B,), got {tuple(x.shape)}, {tuple(t.shape)}")
        # An ODE solver broadcasts ONE scalar t over the batch (integrators.py:103-104): the unconditional pass then needs one
        # conditioning row and the conditional passes one per unique label tuple.  A stride-0 / one-element t proves it for
        # free (scldm_amd.transport passes such a view).  For a dense t (the reference's `ones(B) * t` under torchdiffeq) the
        # decision is taken ON DEVICE (t_stride 2): one small kernel compares the entries, both plans' conditioning kernels are
        # enqueued and the one that does not apply exits at once - no host synchronisation per evaluation.
        if t.stride(0) == 0 or n == 1:
            t_stride = 0
        else:
            t_stride = 2 if self.detect_uniform_t else 1
        tt = _require_cuda_f32("t", t[:1] if t_stride == 0 else t)
        ul, n_u, cell_row, n_pass, masks, scales, keep = self._cfg_plan(condition, cfg_scale, B, dedup=t_stride != 1)
        out = torch.empty_like(x)
        n_rows = (1 + n_pass * n_u) if t_stride == 0 else (2 * B + n_pass * B)
        ws = self._workspace(L, 2 * B + n_pass * B, n_rows, 0)
        with torch.cuda.device(x.device):
            _lib.check(L.scldm_dit_forward_cfg(h, x.data_ptr(), tt.data_ptr(), t_stride,
                                               C.cast(ul, _lib.c_void_pp) if ul is not None else None, n_u, cell_row, B, n_pass,
                                               masks, scales, out.data_ptr(), self._prec(), ws, _stream_ptr()),
                       "scldm_dit_forward_cfg")
        del keep
        return out

    def _composed_forward_with_cfg(self, x, t, condition, cfg_scale):
        """forward_with_cfg of a module in TRAINING mode: composed from `forward(..., force_drop_ids=False)` calls exactly as the
        reference composes it (nnets.py:336-378), so that its training-mode semantics carry over - the joint strategy draws its
        label-dropout mask whenever `self.training` (nnets.py:440-445, also inside a CFG evaluation), and the result is
        differentiable when gradients are enabled.  Eval mode takes the fused single-call path instead."""
        n = x.shape[0]
        B = n // 2
        if n != 2 * B or tuple(t.shape) != (n,):
            raise ValueError(f"expected x (2B,{self.seq_len},{self.n_embed_input}) and t (2B,), got {tuple(x.shape)}, {tuple(t.shape)}")
        self._need_null_row("forward_with_cfg (the unconditional pass)")
        uncond = {c: torch.full((n,), v, device=x.device, dtype=torch.long) for c, v in self.class_vocab_sizes.items()}
        u = self.forward(x, t, uncond, force_drop_ids=False)                      # :346-353
        u1, u2 = u[:B], u[B:]
        g = u2.clone()
        if condition is not None and cfg_scale is not None:
            x2, t2 = x[B:], t[B:]
            if self.condition_strategy == "joint":                               # :362-368
                full = {k: v[B:] for k, v in condition.items()}
                g = g + (sum(cfg_scale.values()) / len(cfg_scale)) * (self.forward(x2, t2, full, force_drop_ids=False) - u2)
            else:                                                                # :370-376
                for c, sc in cfg_scale.items():
                    g = g + float(sc) * (self.forward(x2, t2, {c: condition[c][B:]}, force_drop_ids=False) - u2)
        return torch.cat([u1, g], dim=0)

    def forward_with_cfg_joint(self, x: torch.Tensor, t: torch.Tensor, condition: dict[str, torch.Tensor] | None = None,
                               cfg_scale: dict[str, float] | None = None) -> torch.Tensor:
        """Mirror of nnets.py:299-334 (no caller in the reference; kept for API completeness): unconditional pass on every
        row, plus cfg_scale["cell_line"] times the conditional difference - composed from two `forward` calls as there."""
        uncond = {c: torch.full((len(x),), v, device=x.device, dtype=torch.long) for c, v in self.class_vocab_sizes.items()}
        with torch.no_grad():
            u = self.forward(x, t, uncond, force_drop_ids=False)
            g = u.clone()
            if condition is not None and cfg_scale is not None:
                g += cfg_scale["cell_line"] * (self.forward(x, t, condition, force_drop_ids=False) - u)
        return g

    # ------------------------------------------------------------------ fused sampler (transport.py:324-369 + models.py:801-812)
    @torch.no_grad()
    def sample_ode_cfg(self, z: torch.Tensor, condition: dict[str, torch.Tensor] | None, cfg_scale: dict[str, float] | None,
                       num_steps: int, sampling_method: str = "euler", atol: float = 1e-5, rtol: float = 1e-5) -> torch.Tensor:
        """Integrate dz/dt = forward_with_cfg(z, t) over linspace(0, 1, num_steps) entirely on device.

        `z` is the doubled state cat([z0, z0]) (2B,S,C); `condition` the doubled label dict; `num_steps` has the
        reference meaning (grid POINTS: num_steps-1 Euler evaluations).  Returns the final state (what the
        reference indexes with [-1], models.py:812).
        """
        if num_steps < 2:
            raise ValueError("num_steps must be >= 2 (grid points)")
        if sampling_method.lower() == "dopri5":   # adaptive solve: host-driven steps over the fused forward_with_cfg
            from .transport import Sampler, create_transport
            fn = Sampler(create_transport()).sample_ode(sampling_method="dopri5", num_steps=2, atol=atol, rtol=rtol)
            return fn(_require_cuda_f32("z", z), self.forward_with_cfg, condition=condition, cfg_scale=cfg_scale)[-1]
        if self.training or not self.fused_shape:
            # fixed-grid Euler / Heun over forward_with_cfg, one evaluation per call (same grid as the fused loop): shapes outside
            # the fused family, and modules left in training mode (the reference samples in whatever mode the caller left the
            # model in; forward_with_cfg then carries the training-mode label handling)
            fwd = self._generic_forward_with_cfg if not self.fused_shape else self.forward_with_cfg
            z = _require_cuda_f32("z", z).clone()
            method = sampling_method.lower()
            if method not in _lib.METHODS:
                raise KeyError(method)
            hstep = 1.0 / (num_steps - 1)
            for i in range(num_steps - 1):
                t0 = torch.full((z.shape[0],), i * hstep, device=z.device)
                k1 = fwd(z, t0, condition, cfg_scale)
                if method == "euler":
                    z = z + hstep * k1
                else:
                    t1 = torch.full((z.shape[0],), (i + 1) * hstep, device=z.device)
                    k2 = fwd(z + hstep * k1, t1, condition, cfg_scale)
                    z = z + (0.5 * hstep) * (k1 + k2)
            return z
        self._need_null_row("CFG sampling (the unconditional pass)")
        L, h = self._native()
        z = _require_cuda_f32("z", z).clone()
        n = z.shape[0]
        B = n // 2
        if n != 2 * B or z.shape[1:] != (self.seq_len, self.n_embed_input):
            raise ValueError(f"expected z (2B,{self.seq_len},{self.n_embed_input}), got {tuple(z.shape)}")
        ul, n_u, cell_row, n_pass, masks, scales, keep = self._cfg_plan(condition, cfg_scale, B, dedup=True)
        ws = self._workspace(L, 2 * B + n_pass * B, 1 + n_pass * n_u, 2 * B)
        with torch.cuda.device(z.device):
            _lib.check(L.scldm_sample_ode(h, z.data_ptr(), C.cast(ul, _lib.c_void_pp) if ul is not None else None, n_u, cell_row,
                                          B, n_pass, masks, scales, num_steps - 1, _lib.METHODS[sampling_method.lower()],
                                          self._prec(), ws, _stream_ptr()), "scldm_sample_ode")
        del keep
        return z

    # ------------------------------------------------------------------ bench hook
    def layers_per_launch(self) -> int:
        L, h = self._native()
        return int(L.scldm_dit_layers_per_launch(h))

    def block_timing(self, enable: bool | None = None):
        L, h = self._native()
        if enable is not None:
            L.scldm_dit_block_timing_enable(h, int(enable))
            return None
        n, ms = C.c_int(), C.c_double()
        _lib.check(L.scldm_dit_block_timing(h, C.byref(n), C.byref(ms)), "scldm_dit_block_timing")
        return n.value, ms.value
