"""TransformerVAE with the reference's API (src/scldm/vae.py:15-87); encode / decode run on the MI355X MCAB kernels."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
from torch._utils import _unflatten_dense_tensors

from . import _lib
from .layers import InputTransformerVAE, swiglu_hidden
from .nnets import Decoder, Encoder, _require_cuda_f32, _stream_ptr
from .stochastic_layers import NegativeBinomial, NegativeBinomialTransformerLayer


class _VAETrainFn(torch.autograd.Function):
    """TransformerVAE.forward with a HIP backward (scldm_vae_train_forward / _backward, include/scldm_hip.h): replaces torch
    autograd over the reference's module tree (vae.py:29-56) inside VAE.training_step (models.py:249-290).  Outputs mu, theta
    (B, G) and z (B, 16, n_lat) are ordinary differentiable tensors: the reference's own `-log_nb_positive(counts, mu, theta)`
    (models.py:243) - or scldm_amd.distributions.log_nb_positive, the fused form - sits on top of them unchanged."""

    @staticmethod
    def forward(ctx, module, counts_subset, genes_subset, genes, lib, *params):
        L, h = module._native()
        B, S = counts_subset.shape
        G = genes.shape[1]
        dev = counts_subset.device
        mu = torch.empty(B, G, device=dev, dtype=torch.float32)
        theta = torch.empty(B, G, device=dev, dtype=torch.float32)
        z = torch.empty(B, module.encoder.latent_dim, module.encoder.latent_embedding, device=dev, dtype=torch.float32)
        saved = torch.empty(L.scldm_vae_train_saved_bytes(h, B), dtype=torch.uint8, device=dev)
        ws = torch.empty(L.scldm_vae_train_workspace_bytes(h, B, S, G), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.scldm_vae_train_forward(h, counts_subset.data_ptr(), genes_subset.data_ptr(), B, S, genes.data_ptr(), lib.data_ptr(), G,
                                                 mu.data_ptr(), theta.data_ptr(), z.data_ptr(), saved.data_ptr(), ws.data_ptr(), _stream_ptr()),
                       "scldm_vae_train_forward")
        ctx.module, ctx.saved, ctx.ws = module, saved, ws
        ctx.inputs = (counts_subset, genes_subset, genes, lib)
        ctx.outs = (mu, theta, z)
        ctx.params = params
        ctx.param_versions = [p._version for p in params]
        return mu, theta, z

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dmu, dtheta, dz):
        module, params = ctx.module, ctx.params
        L, h = _lib.lib(), module._handle
        if [p._version for p in params] != ctx.param_versions:
            raise RuntimeError("TransformerVAE parameters were modified between forward and backward (the HIP backward reads them live)")
        counts_subset, genes_subset, genes, lib = ctx.inputs
        mu, theta, z = ctx.outs
        B, S = counts_subset.shape
        G = genes.shape[1]
        dev = mu.device
        prep = lambda t: None if t is None else t.contiguous().float()
        dmu, dtheta, dz = prep(dmu), prep(dtheta), prep(dz)
        # every gradient is a view of ONE zero-initialised buffer (parameters the kernels do not write, e.g. a frozen table, stay 0).
        # The offsets, the weight-pointer struct and (while the allocator hands back the same block) the gradient-pointer struct are
        # cached on the module: building two 175-field ctypes structs per step is host time a batch-32 step cannot hide.
        cache = module.__dict__.setdefault("_train_cache", {})
        pk = tuple(p.data_ptr() for p in params)
        if cache.get("pk") != pk:
            offs, total = {}, 0
            for p in params:       # dense: the kernels write gradients with 4-byte stores / atomics, nothing needs more alignment
                offs[id(p)] = total
                total += p.numel()
            cache.clear()
            cache.update(pk=pk, offs=offs, total=total, w=module._weights_struct(lambda t: t.data_ptr()))
        offs, total = cache["offs"], cache["total"]
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        base = flat.data_ptr()
        if cache.get("gbase") != base:
            cache["gbase"], cache["g"] = base, module._weights_struct(lambda t: base + 4 * offs[id(t)])
        w, keep_w = cache["w"]
        g, keep_g = cache["g"]
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.scldm_vae_train_backward(h, C.byref(w), C.byref(g), counts_subset.data_ptr(), genes_subset.data_ptr(), B, S,
                                                  genes.data_ptr(), lib.data_ptr(), G, mu.data_ptr(), theta.data_ptr(), z.data_ptr(),
                                                  ptr(dmu), ptr(dtheta), ptr(dz), ctx.saved.data_ptr(), ctx.ws.data_ptr(), _stream_ptr()),
                       "scldm_vae_train_backward")
        ctx.saved = ctx.ws = None
        views = _unflatten_dense_tensors(flat, params)      # one C++ call instead of 175 slices + views
        out = [v if ctx.needs_input_grad[5 + i] else None for i, v in enumerate(views)]
        return (None, None, None, None, None, *out)


class TransformerVAE(nn.Module):
    def __init__(self, encoder: Encoder, decoder: Decoder, decoder_head: NegativeBinomialTransformerLayer,
                 input_layer: InputTransformerVAE):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.decoder_head = decoder_head
        self.input_layer = input_layer
        self._handle = None
        # operand policy of the contractions of encode / decode / decode_sample - the per-gene MCAB / SwiGLU products and the Linears of
        # the 16-token trunks (LayerNorms, softmax, the trunks' 16 x 16 attention and the NB head are fp32 in every policy): "fp32" =
        # exact (parity path); "fp16" = TF32's mantissa, the arithmetic class the reference itself runs the VAE in
        # (set_float32_matmul_precision("high"), experiments/scripts/inference.py:26); "bf16" = 8 bits
        self.precision = "fp32"
        self._weights_key = None
        self._weights_fp = None
        self.check_weight_fingerprint = True
        self._ws = None
        self._keep = None

    # ------------------------------------------------------------------ native handle
    def _native(self):
        L = _lib.lib()
        emb = self.input_layer.gene_embedding.weight
        if emb.device.type != "cuda":
            raise RuntimeError("TransformerVAE parameters must live on a CUDA (ROCm) device; there is no CPU path")
        enc = self.encoder
        if self._handle is None:
            cfg = _lib.VaeConfig(n_genes=emb.shape[0] - 1, n_embed=emb.shape[1], n_inducing=enc.latent_dim,
                                 n_embed_latent=enc.latent_embedding, n_layer=enc.n_layer, n_head=enc.n_head,
                                 n_head_cross=enc.n_head_cross, hidden_dim=swiglu_hidden(enc.n_embed, enc.multiple_of),
                                 layernorm_eps=enc.layernorm_eps, positional_encoding=int(enc.pos_embed is not None),
                                 nb_temperature=float(self.decoder_head.t))
            h = C.c_void_p()
            with torch.cuda.device(emb.device):
                _lib.check(L.scldm_vae_create(C.byref(cfg), C.byref(h)), "scldm_vae_create")
            self._handle = h
        params = list(self.parameters())
        key = tuple(p.data_ptr() for p in params)
        ver = tuple(p._version for p in params)
        if key != self._weights_key:
            self._load_weights(L)             # new storages: the job / fingerprint tables are rebuilt, everything is re-packed
            self._weights_key, self._weights_fp = key, ver
        elif ver != self._weights_fp or self.check_weight_fingerprint:
            # same storages: an optimiser step (version counters moved) or possibly an in-place update through `.data` (EMA-style,
            # invisible to the counters).  Either way the C side compares a device-side fingerprint of every source tensor with that
            # of the packed copies and re-packs in stream order if it moved: five small launches, no host synchronisation (round 3
            # compared per-tensor norms on the host - one sync per encode / decode call - and re-issued ~150 launches per re-pack).
            with torch.cuda.device(emb.device):
                _lib.check(L.scldm_vae_refresh_weights(self._handle, _stream_ptr()), "scldm_vae_refresh_weights")
            self._weights_fp = ver
        return L, self._handle

    def invalidate_weights(self) -> None:
        """Force the next encode / decode to re-derive the packed weight copies from the parameters."""
        self._weights_key = None

    # copies / pickling: the native handle never travels (the copy builds its own at its first call)
    def __getstate__(self):
        state = self.__dict__.copy()
        state.update(_handle=None, _weights_key=None, _weights_fp=None, _ws=None, _keep=None)
        state.pop("_train_cache", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        for k in ("_handle", "_weights_key", "_weights_fp", "_ws", "_keep"):
            self.__dict__.setdefault(k, None)

    def _weights_struct(self, dp):
        """scldm_vae_weights filled with dp(parameter) -> device pointer (the same struct, with writable pointers, receives the
        gradients in scldm_vae_train_backward).  Returns (struct, keepalive)."""
        enc, dec = self.encoder, self.decoder
        n = enc.n_layer

        def block(b):
            return _lib.VaeBlock(ln1_w=dp(b.ln_1.weight), ln1_b=dp(b.ln_1.bias), attn_w=dp(b.attn.c_attn.weight),
                                 proj_w=dp(b.attn.c_proj.weight), ln2_w=dp(b.ln_2.weight), ln2_b=dp(b.ln_2.bias), w1=dp(b.mlp.w1.weight),
                                 w2=dp(b.mlp.w2.weight), cproj=dp(b.mlp.c_proj.weight))

        def cross(c):
            return _lib.VaeCross(ln1_w=dp(c.ln_1.weight), ln1_b=dp(c.ln_1.bias), ln1q_w=dp(c.ln_1q.weight), ln1q_b=dp(c.ln_1q.bias),
                                 attn_kv=dp(c.attn.c_attn.weight), attn_q=dp(c.attn.c_attn_q.weight), attn_proj=dp(c.attn.c_proj.weight),
                                 ln2_w=dp(c.ln_2.weight), ln2_b=dp(c.ln_2.bias), w1=dp(c.mlp.w1.weight), w2=dp(c.mlp.w2.weight),
                                 cproj=dp(c.mlp.c_proj.weight))

        eb = (_lib.VaeBlock * max(n, 1))(*[block(b) for b in enc.encoder_layers])
        db = (_lib.VaeBlock * max(n, 1))(*[block(b) for b in dec.decoder_layers])
        w = _lib.VaeWeights(gene_embedding=dp(self.input_layer.gene_embedding.weight), inducing_points=dp(enc.ca_layer.inducing_points),
                            enc_pos_embed=dp(enc.pos_embed) if enc.pos_embed is not None else None,
                            enc_latent_w=dp(enc.encoder_latent_input[0].weight), dec_latent_w=dp(dec.decoder_latent_input[1].weight),
                            theta=dp(self.decoder_head.theta.weight) if self.decoder_head.theta is not None else None,   # NULL = unshared theta
                            head_w=dp(self.decoder_head.params.weight),
                            head_b=dp(self.decoder_head.params.bias), enc_cross=cross(enc.ca_layer),
                            dec_cross=cross(dec.decoder_cross_attention), enc_blocks=eb, dec_blocks=db)
        return w, (eb, db)

    def _load_weights(self, L):
        for p in self.parameters():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("TransformerVAE parameters must be contiguous fp32")
        w, keep = self._weights_struct(lambda t: t.data_ptr())
        self._keep = keep
        with torch.cuda.device(self.input_layer.gene_embedding.weight.device):
            _lib.check(L.scldm_vae_load_weights(self._handle, C.byref(w), _stream_ptr()), "scldm_vae_load_weights")

    def _workspace(self, L, B: int, G: int) -> int:
        need = L.scldm_vae_workspace_bytes(self._handle, B, G)
        dev = self.input_layer.gene_embedding.weight.device
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws.data_ptr()

    KERNEL_KINDS = ("enc_pool", "enc_cell", "dec_cell", "dec_gene", "dec_finalize")

    def kernel_timing(self, enable: bool | None = None):
        """Measurement hook (bench.py): `kernel_timing(True / False)` switches HIP-event pairs around every MCAB kernel launch on /
        off; `kernel_timing()` drains them and returns {kernel: (launches, total ms)} (scldm_vae_kernel_timing)."""
        L, h = self._native()
        if enable is not None:
            L.scldm_vae_kernel_timing_enable(h, int(enable))
            return None
        out = {}
        for k, name in enumerate(self.KERNEL_KINDS):
            n, ms = C.c_int(), C.c_double()
            _lib.check(L.scldm_vae_kernel_timing(h, k, C.byref(n), C.byref(ms)), "scldm_vae_kernel_timing")
            out[name] = (n.value, ms.value)
        return out

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().scldm_vae_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API (vae.py:29-87)
    @torch.no_grad()
    def encode(self, counts: torch.Tensor, genes: torch.Tensor, counts_subset: torch.Tensor | None = None,
               genes_subset: torch.Tensor | None = None) -> torch.Tensor:
        c = counts_subset if counts_subset is not None else counts
        g = genes_subset if genes_subset is not None else genes
        L, h = self._native()
        c = _require_cuda_f32("counts", c)
        if not g.is_cuda:
            raise RuntimeError("genes must be a CUDA (ROCm) tensor")
        g = g.to(torch.long).contiguous()
        if c.dim() != 2 or g.shape != c.shape:
            raise ValueError(f"counts and genes must both be (B,S); got {tuple(c.shape)} and {tuple(g.shape)}")
        B, S = c.shape
        z = torch.empty(B, self.encoder.latent_dim, self.encoder.latent_embedding, device=c.device, dtype=torch.float32)
        ws = self._workspace(L, B, 1)
        with torch.cuda.device(c.device):
            _lib.check(L.scldm_vae_encode(h, c.data_ptr(), g.data_ptr(), B, S, z.data_ptr(), _lib.PRECISIONS[self.precision], ws,
                                          _stream_ptr()), "scldm_vae_encode")
        return z

    @torch.no_grad()
    def decode(self, z: torch.Tensor, genes: torch.Tensor, library_size: torch.Tensor,
               condition: dict[str, torch.Tensor] | None = None) -> torch.distributions.Distribution:
        L, h = self._native()
        z = _require_cuda_f32("z", z)
        if not genes.is_cuda:
            raise RuntimeError("genes must be a CUDA (ROCm) tensor")
        g = genes.to(torch.long).contiguous()
        lib = _require_cuda_f32("library_size", library_size).reshape(-1)
        B, G = g.shape
        if z.shape != (B, self.encoder.latent_dim, self.encoder.latent_embedding) or lib.shape[0] != B:
            raise ValueError(f"expected z (B,{self.encoder.latent_dim},{self.encoder.latent_embedding}) and library_size (B,1); got "
                             f"{tuple(z.shape)}, {tuple(library_size.shape)} for genes {tuple(g.shape)}")
        mu = torch.empty(B, G, device=z.device, dtype=torch.float32)
        theta = torch.empty(B, G, device=z.device, dtype=torch.float32)
        ws = self._workspace(L, B, G)
        with torch.cuda.device(z.device):
            _lib.check(L.scldm_vae_decode(h, z.data_ptr(), g.data_ptr(), lib.data_ptr(), B, G, mu.data_ptr(), theta.data_ptr(),
                                          _lib.PRECISIONS[self.precision], ws, _stream_ptr()), "scldm_vae_decode")
        return NegativeBinomial(mu=mu, theta=theta)

    @torch.no_grad()
    def decode_sample(self, z: torch.Tensor, genes: torch.Tensor, library_size: torch.Tensor, seed: int | None = None) -> torch.Tensor:
        """`decode(z, genes, library_size).sample()` in one call (models.py:819 after vae.py:71-87): the negative-binomial draw is
        fused into the decoder's normalisation pass, mu / theta never reach HBM.  Returns counts (B, G) fp32."""
        L, h = self._native()
        z = _require_cuda_f32("z", z)
        if not genes.is_cuda:
            raise RuntimeError("genes must be a CUDA (ROCm) tensor")
        g = genes.to(torch.long).contiguous()
        lib = _require_cuda_f32("library_size", library_size).reshape(-1)
        B, G = g.shape
        if z.shape != (B, self.encoder.latent_dim, self.encoder.latent_embedding) or lib.shape[0] != B:
            raise ValueError(f"expected z (B,{self.encoder.latent_dim},{self.encoder.latent_embedding}) and library_size (B,1); got "
                             f"{tuple(z.shape)}, {tuple(library_size.shape)} for genes {tuple(g.shape)}")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        counts = torch.empty(B, G, device=z.device, dtype=torch.float32)
        ws = self._workspace(L, B, G)
        with torch.cuda.device(z.device):
            _lib.check(L.scldm_vae_decode_sample(h, z.data_ptr(), g.data_ptr(), lib.data_ptr(), B, G, counts.data_ptr(), C.c_uint64(seed),
                                                 _lib.PRECISIONS[self.precision], ws, _stream_ptr()), "scldm_vae_decode_sample")
        return counts

    def forward(self, counts, genes, library_size, counts_subset=None, genes_subset=None):
        """(params, z) with params = {"mu", "theta"} (vae.py:29-56).  With gradients enabled and trainable parameters the outputs
        are differentiable: forward on the inference kernels + a hand-derived HIP backward (`_VAETrainFn`), fp32.  As in the
        reference, the encoder reads counts_subset / genes_subset (vae.py:37-40: no fallback to the full vectors in forward)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if counts_subset is None or genes_subset is None:
                raise ValueError("TransformerVAE.forward needs counts_subset / genes_subset (the reference passes them to the input layer, vae.py:37-40)")
            cs = _require_cuda_f32("counts_subset", counts_subset)
            if not genes_subset.is_cuda or not genes.is_cuda:
                raise RuntimeError("genes / genes_subset must be CUDA (ROCm) tensors")
            gs = genes_subset.to(torch.long).contiguous()
            g = genes.to(torch.long).contiguous()
            lib = _require_cuda_f32("library_size", library_size).reshape(-1)
            if cs.dim() != 2 or gs.shape != cs.shape or g.dim() != 2 or g.shape[0] != cs.shape[0] or lib.shape[0] != cs.shape[0]:
                raise ValueError(f"expected counts_subset / genes_subset (B,S), genes (B,G), library_size (B,1); got {tuple(cs.shape)}, "
                                 f"{tuple(gs.shape)}, {tuple(g.shape)}, {tuple(library_size.shape)}")
            if self.precision != "fp32":
                raise NotImplementedError("TransformerVAE training runs in fp32 (precision='fp32')")
            if self.decoder_head.theta is None:
                raise NotImplementedError("the HIP training backward is built for the shared-theta NB head (vae_base.yaml:62); "
                                          "the unshared-theta head (stochastic_layers.py:94-96) decodes only")
            params = tuple(self.parameters())
            mu, theta, z = _VAETrainFn.apply(self, cs, gs, g, lib, *params)
            return {"mu": mu, "theta": theta}, z
        with torch.no_grad():
            z = self.encode(counts, genes, counts_subset, genes_subset)
            nb = self.decode(z, genes, library_size)
            return {"mu": nb.mu, "theta": nb.theta}, z
