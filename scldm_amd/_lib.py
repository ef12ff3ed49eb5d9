"""ctypes binding of libscldm_hip.so (C ABI in include/scldm_hip.h).

There is NO CPU fallback: importing the product without the built HIP library raises.
Build it with `./build.sh` (hipcc --offload-arch=gfx950) or `python -c "import __graft_entry__ as g; g.build()"`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCLDM_LIB", os.path.join(_HERE, "libscldm_hip.so"))  # SCLDM_LIB: debug-build override

MAX_CLASSES = 8
PREC_FP32, PREC_BF16, PREC_BF16X3, PREC_FP16 = 0, 1, 2, 3
METHOD_EULER, METHOD_HEUN = 0, 1
PRECISIONS = {"fp32": PREC_FP32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "fp16": PREC_FP16}
METHODS = {"euler": METHOD_EULER, "heun": METHOD_HEUN}
OPT_CFG1_DIRECT = 1
OPT_TAIL_SPLIT = 2   # scldm_dit_set_option (include/scldm_hip.h)

c_float_p = C.POINTER(C.c_float)
c_void_pp = C.POINTER(C.c_void_p)


class DitConfig(C.Structure):
    _fields_ = [("n_embed", C.c_int), ("n_embed_input", C.c_int), ("n_layer", C.c_int), ("n_head", C.c_int),
                ("seq_len", C.c_int), ("hidden_dim", C.c_int), ("layernorm_eps", C.c_float), ("n_classes", C.c_int),
                ("class_vocab", C.c_int * MAX_CLASSES), ("has_null_row", C.c_int)]


class DitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos_embed", "t_w0", "t_b0", "t_w2", "t_b2", "in_w", "in_b", "fin_w", "fin_b",
                                          "fin_ada_w", "fin_ada_b")] + \
               [(n, c_void_pp) for n in ("class_emb", "attn_w", "attn_b", "proj_w", "proj_b", "w1", "w2", "cproj", "ada_w", "ada_b")]


class AdamwEntry(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong)]


class VaeConfig(C.Structure):
    _fields_ = [("n_genes", C.c_int), ("n_embed", C.c_int), ("n_inducing", C.c_int), ("n_embed_latent", C.c_int),
                ("n_layer", C.c_int), ("n_head", C.c_int), ("n_head_cross", C.c_int), ("hidden_dim", C.c_int),
                ("layernorm_eps", C.c_float), ("positional_encoding", C.c_int), ("nb_temperature", C.c_float)]


class VaeBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "attn_w", "proj_w", "ln2_w", "ln2_b", "w1", "w2", "cproj")]


class VaeCross(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "ln1q_w", "ln1q_b", "attn_kv", "attn_q", "attn_proj", "ln2_w", "ln2_b",
                                          "w1", "w2", "cproj")]


class VaeWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gene_embedding", "inducing_points", "enc_pos_embed", "enc_latent_w", "dec_latent_w",
                                          "theta", "head_w", "head_b")] + \
               [("enc_cross", VaeCross), ("dec_cross", VaeCross), ("enc_blocks", C.POINTER(VaeBlock)), ("dec_blocks", C.POINTER(VaeBlock))]


class AdamwLaunch(C.Structure):
    """scldm_adamw_launch (include/scldm_hip.h)."""
    _fields_ = [("table", C.c_void_p), ("count", C.c_int), ("n_blocks", C.c_int), ("step", C.c_void_p), ("found_inf", C.c_void_p),
                ("hyper", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("maximize", C.c_int), ("max_grad_norm", C.c_float), ("clip_ws", C.c_void_p)]


class TrainStepBuffers(C.Structure):
    """scldm_train_step_buffers (include/scldm_hip.h)."""
    _fields_ = [("row_elems", C.c_int)] + [(n, C.c_void_p) for n in ("t", "x0", "xt", "ut", "pred", "dpred", "labels", "loss_rows", "loss_mean",
                                                                      "ticket", "saved", "ws")]


class ScldmError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (./build.sh). "
                          "scldm_amd has no CPU fallback by design.")
    L = C.CDLL(LIB_PATH)
    L.scldm_last_error.restype = C.c_char_p
    L.scldm_version.restype = C.c_int
    L.scldm_dit_create.argtypes = [C.POINTER(DitConfig), C.POINTER(C.c_void_p)]
    L.scldm_dit_destroy.argtypes = [C.c_void_p]
    L.scldm_dit_destroy.restype = None
    L.scldm_dit_load_weights.argtypes = [C.c_void_p, C.POINTER(DitWeights), C.c_void_p]
    L.scldm_dit_refresh_weights.argtypes = [C.c_void_p, C.c_void_p]
    L.scldm_adamw_step.argtypes = [C.POINTER(AdamwEntry), C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 5 + [C.c_int, C.c_void_p]
    L.scldm_adamw_table_bytes.argtypes = [C.POINTER(AdamwEntry), C.c_int]
    L.scldm_adamw_table_bytes.restype = C.c_size_t
    L.scldm_adamw_table_build.argtypes = [C.POINTER(AdamwEntry), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    L.scldm_adamw_table_records_bytes.argtypes = [C.c_int]
    L.scldm_adamw_table_records_bytes.restype = C.c_size_t
    L.scldm_adamw_clip_workspace_bytes.argtypes = [C.c_int]
    L.scldm_adamw_clip_workspace_bytes.restype = C.c_size_t
    L.scldm_adamw_table_update.argtypes = [C.POINTER(AdamwEntry), C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]
    L.scldm_adamw_table_step.argtypes = [C.POINTER(AdamwLaunch), C.c_void_p]
    L.scldm_fm_prepare.argtypes = [C.c_void_p, c_void_pp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_fm_loss_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_dit_train_step.argtypes = [C.c_void_p, C.POINTER(DitWeights), C.POINTER(DitWeights), C.c_void_p, c_void_pp, C.POINTER(C.c_int), C.c_int,
                                       C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.POINTER(TrainStepBuffers), C.POINTER(AdamwLaunch), C.c_void_p]
    fpp = C.POINTER(C.c_void_p)
    L.scldm_rk_combine.argtypes = [C.c_void_p, C.c_void_p, fpp, c_float_p, C.c_int, C.c_longlong, C.c_void_p]
    L.scldm_rk_error.argtypes = [C.c_void_p, C.c_void_p, fpp, c_float_p, C.c_int, C.c_longlong, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.scldm_rk_dense.argtypes = [C.c_void_p, C.c_void_p, fpp, c_float_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_longlong,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_rk_poly.argtypes = [C.c_void_p] * 6 + [C.c_float] * 4 + [C.c_longlong, C.c_void_p]
    L.scldm_dit_train_fp16_state.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_void_p]
    L.scldm_dit_train_set_found_inf.argtypes = [C.c_void_p, C.c_void_p]
    L.scldm_dit_fp16_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p]
    L.scldm_dit_label_errors.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    L.scldm_dit_mod_width.argtypes = [C.c_void_p]
    L.scldm_dit_layers_per_launch.argtypes = [C.c_void_p]
    L.scldm_dit_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.scldm_dit_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.scldm_dit_workspace_bytes.restype = C.c_size_t
    L.scldm_dit_cond_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, c_void_pp, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_dit_forward_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_dit_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]
    L.scldm_dit_forward_cfg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_void_pp, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.POINTER(C.c_uint32), c_float_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_sample_ode.argtypes = [C.c_void_p, C.c_void_p, c_void_pp, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.POINTER(C.c_uint32), c_float_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_mfma_sustained_tflops.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.scldm_dit_block_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.scldm_dit_block_timing_enable.restype = None
    L.scldm_dit_block_timing.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.scldm_dit_set_debug_buffer.argtypes = [C.c_void_p, C.c_void_p]
    L.scldm_dit_train_saved_bytes.argtypes = [C.c_void_p, C.c_int]
    L.scldm_dit_train_saved_bytes.restype = C.c_size_t
    L.scldm_dit_train_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
    L.scldm_dit_train_workspace_bytes.restype = C.c_size_t
    L.scldm_dit_train_saved_bytes_for.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.scldm_dit_train_saved_bytes_for.restype = C.c_size_t
    L.scldm_dit_train_workspace_bytes_for.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.scldm_dit_train_workspace_bytes_for.restype = C.c_size_t
    L.scldm_dit_train_prepare.argtypes = [C.c_void_p, C.POINTER(DitWeights), C.c_int, C.c_int, C.c_void_p]
    L.scldm_dit_train_set_grad_events.argtypes = [C.c_void_p, c_void_pp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    # scldm_dit_grads has the field order of scldm_dit_weights, so DitWeights serves for both
    L.scldm_dit_train_forward.argtypes = [C.c_void_p, C.POINTER(DitWeights), C.c_void_p, C.c_void_p, c_void_pp, C.c_int, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_dit_train_backward.argtypes = [C.c_void_p, C.POINTER(DitWeights), C.POINTER(DitWeights), C.c_void_p, c_void_pp,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_fm_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.scldm_fm_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.scldm_fm_loss_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.scldm_vae_create.argtypes = [C.POINTER(VaeConfig), C.POINTER(C.c_void_p)]
    L.scldm_vae_destroy.argtypes = [C.c_void_p]
    L.scldm_vae_destroy.restype = None
    L.scldm_vae_load_weights.argtypes = [C.c_void_p, C.POINTER(VaeWeights), C.c_void_p]
    L.scldm_vae_kernel_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.scldm_vae_kernel_timing_enable.restype = None
    L.scldm_vae_kernel_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.scldm_vae_refresh_weights.argtypes = [C.c_void_p, C.c_void_p]
    L.scldm_vae_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.scldm_vae_workspace_bytes.restype = C.c_size_t
    L.scldm_vae_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_vae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_vae_decode_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                          C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_vae_train_saved_bytes.argtypes = [C.c_void_p, C.c_int]
    L.scldm_vae_train_saved_bytes.restype = C.c_size_t
    L.scldm_vae_train_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.scldm_vae_train_workspace_bytes.restype = C.c_size_t
    L.scldm_vae_train_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_vae_train_backward.argtypes = [C.c_void_p, C.POINTER(VaeWeights), C.POINTER(VaeWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_nb_loglik.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
    L.scldm_nb_loglik_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.scldm_nb_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
    L.scldm_tokenize_expressed.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_csr_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.scldm_csr_fill.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scldm_mmd_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.scldm_mmd_workspace_bytes.restype = C.c_size_t
    L.scldm_mmd_kernel_sum.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    L.scldm_sinkhorn_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.scldm_sinkhorn_workspace_bytes.restype = C.c_size_t
    L.scldm_wasserstein_sinkhorn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_longlong, C.c_float,
                                             C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    _lib = L
    return L


EXPORTS = ["scldm_last_error", "scldm_version", "scldm_dit_create", "scldm_dit_destroy", "scldm_dit_load_weights",
           "scldm_dit_refresh_weights", "scldm_dit_fp16_stats", "scldm_dit_train_fp16_state", "scldm_dit_train_set_found_inf", "scldm_dit_label_errors", "scldm_dit_mod_width", "scldm_dit_layers_per_launch", "scldm_dit_set_option", "scldm_dit_workspace_bytes", "scldm_dit_cond_rows", "scldm_dit_forward_rows",
           "scldm_dit_forward", "scldm_dit_forward_cfg", "scldm_sample_ode", "scldm_adamw_step", "scldm_adamw_table_bytes", "scldm_adamw_table_build", "scldm_adamw_table_records_bytes", "scldm_adamw_clip_workspace_bytes", "scldm_adamw_table_update", "scldm_adamw_table_step", "scldm_fm_prepare", "scldm_fm_loss_grad", "scldm_dit_train_step", "scldm_rk_combine", "scldm_rk_error", "scldm_rk_dense", "scldm_rk_poly", "scldm_mfma_sustained_tflops", "scldm_dit_block_timing_enable",
           "scldm_dit_block_timing", "scldm_dit_set_debug_buffer", "scldm_dit_train_saved_bytes", "scldm_dit_train_workspace_bytes", "scldm_dit_train_saved_bytes_for", "scldm_dit_train_workspace_bytes_for",
           "scldm_dit_train_prepare", "scldm_dit_train_set_grad_events", "scldm_dit_train_forward", "scldm_dit_train_backward", "scldm_fm_mix", "scldm_fm_loss", "scldm_fm_loss_bwd", "scldm_vae_create", "scldm_vae_destroy", "scldm_vae_load_weights", "scldm_vae_refresh_weights", "scldm_vae_kernel_timing_enable", "scldm_vae_kernel_timing",
           "scldm_vae_workspace_bytes", "scldm_vae_encode", "scldm_vae_decode", "scldm_vae_decode_sample", "scldm_vae_train_saved_bytes", "scldm_vae_train_workspace_bytes", "scldm_vae_train_forward", "scldm_vae_train_backward", "scldm_nb_loglik", "scldm_nb_loglik_bwd", "scldm_nb_sample", "scldm_tokenize_expressed", "scldm_csr_count", "scldm_csr_fill", "scldm_mmd_workspace_bytes",
           "scldm_mmd_kernel_sum", "scldm_sinkhorn_workspace_bytes", "scldm_wasserstein_sinkhorn"]


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ScldmError(f"{what} failed (code {rc}): {lib().scldm_last_error().decode()}")


def ptr_array(ptrs):
    """Host array of device pointers (None -> NULL)."""
    arr = (C.c_void_p * max(len(ptrs), 1))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
