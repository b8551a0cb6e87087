"""ctypes binding of libaudiocaption_hip.so (declared in include/audiocaption_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is
raised.  Tensors are passed as raw device pointers together with the caller's current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUDIOCAPTION_HIP_LIB") or os.path.join(_HERE, "libaudiocaption_hip.so")  # env: development builds

AC_MAX_LAYERS = 8
ABI_VERSION = 2   # include/audiocaption_hip.h AC_ABI_VERSION

c_float_p = ctypes.c_void_p
c_int_p = ctypes.c_void_p


class AcTrmLayer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b",
        "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w", "n2_b", "n3_w", "n3_b")]


class AcTrmWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "d_model", "nhead", "nlayers", "dim_ff", "vocab", "max_pos", "attn_emb_dim", "reserved")] + [
        (n, ctypes.c_void_p) for n in ("emb", "pe", "cls_w", "proj_w", "proj_b", "proj_ln_w", "proj_ln_b", "step_pk")] + [
        ("layer", AcTrmLayer * AC_MAX_LAYERS)]


_I, _L, _F, _P = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
_U64 = ctypes.c_ulonglong
_WP = ctypes.POINTER(AcTrmWeights)

# name -> (restype, argtypes); must list every symbol of include/audiocaption_hip.h
SIGNATURES = {
    "ac_abi_version": (_I, []),
    "ac_logmel": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _P]),
    "ac_conv3x3_bn_relu": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_conv3x3_bn_relu_winograd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_conv3x3_bn_relu_bf16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_conv3x3_bn_relu_bf16x3_gw": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_conv3x3_bn_relu_wino1d": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "ac_conv3x3_bn_relu_wino1d_drop": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _P, _P]),
    "ac_conv3x3_wino1d_splitk_floats": (_L, [_I, _I, _I, _I, _I]),
    "ac_conv3x3_bn_relu_wino1d_splitk": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _L, _P]),
    "ac_conv3x3_bn_relu_wino43": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "ac_conv3x3_bn_relu_wino43_drop": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _U64, _P, _P]),
    "ac_conv3x3_wino43_workgroups": (_L, [_I, _I, _I, _I]),
    "ac_conv3x3_block1_wino43": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _F, _U64, _P, _P]),
    "ac_conv3x3_block1_wino43_mfma": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _F, _U64, _P, _P]),
    "ac_conv3x3_block1_conv2_wino43": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_conv3x3_bn_relu_f16x2_gw": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ac_conv3x3_block1_f16x2": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "ac_linear_bf16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ac_conv3x3_first": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ac_conv3x3_first_f16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "ac_linear": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P]),
    "ac_gru_pack_whh": (_I, [_P, _P, _I, _P]),
    "ac_gru_layer": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_gru_split_workspace_bytes": (_L, [_I]),
    "ac_gru_layer_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_mean_with_lens": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ac_add_layernorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _L, _L, _P]),
    "ac_trm_memory": (_I, [_WP, _P, _I, _I, _P, _P, _P]),
    "ac_trm_step_pack_floats": (_L, [_WP]),
    "ac_trm_pack_step_weights": (_I, [_WP, _P, _P]),
    "ac_dec_wide_packed_floats": (_L, [_I, _I]),
    "ac_dec_wide_pack": (_I, [_P, _L, _I, _I, _P, _P]),
    "ac_dec_wide_gemm": (_I, [_I, _P, _L, _P, _L, _P, _P, _P, _L, _I, _P, _P, _F, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "ac_trm_workspace_floats": (_L, [_WP, _I, _I]),
    "ac_trm_greedy": (_I, [_WP, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "ac_conv3x3_skinny_workspace_floats": (_L, [_I, _I, _I, _I, _I]),
    "ac_conv3x3_bn_relu_skinny": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _F, _U64, _P, _P]),
    "ac_trm_cluster_pack_floats": (_L, [_WP]),
    "ac_trm_cluster_pack": (_I, [_WP, _P, _P]),
    "ac_trm_cluster_workspace_bytes": (_L, [_I]),
    "ac_trm_greedy_cluster": (_I, [_WP, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "ac_trm_forward_tokens": (_I, [_WP, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "ac_trm_beam_step": (_I, [_WP, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "ac_trm_beam_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_trm_beam_reorder": (_I, [_WP, _I, _I, _I, _P, _P, _P]),
    # training step (csrc/train.hip)
    "ac_gemm": (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _I, _F, _I, _F, _U64, _P, _L, _P, _I, _P]),
    "ac_gemm_bf16x3": (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _P, _I, _F, _I, _F, _U64, _P, _L, _P, _I, _P]),
    "ac_dropout": (_I, [_P, _P, _L, _F, _U64, _P, _L, _P]),
    "ac_mask_pos_scale": (_I, [_P, _P, _L, _F, _P]),
    "ac_build_prefix": (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _L, _I, _I, _P]),
    "ac_embed_fwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _F, _U64, _F, _U64, _P, _P]),
    "ac_embed_bwd": (_I, [_P, _P, _P, _L, _I, _F, _U64, _F, _U64, _P, _P]),
    "ac_dropadd_ln_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _F, _U64, _P, _F, _P]),
    "ac_dropadd_ln_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _L, _P, _P, _L, _I, _F, _U64, _P, _F, _P]),
    "ac_attn_seq_fwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                             _I, _I, _I, _F, _U64, _P, _P]),
    "ac_attn_seq_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _I,
                             _I, _I, _I, _I, _I, _F, _U64, _P, _P]),
    "ac_gather_rows": (_I, [_P, _P, _P, _L, _I, _P]),
    "ac_scatter_add_rows": (_I, [_P, _P, _P, _L, _I, _P]),
    "ac_specaug": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ac_sum_replicas": (_I, [_P, _P, _L, _I, _P]),
    "ac_rows_mean_w": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ac_transpose": (_I, [_P, _P, _I, _I, _I, _P]),
    "ac_colsum": (_I, [_P, _L, _P, _L, _I, _P]),
    "ac_argmax_rows": (_I, [_P, _L, _I, _I, _P, _L, _P]),
    "ac_label_smoothing_loss": (_I, [_P, _P, _L, _P, _I, _I, _I, _F, _F, _P, _P, _P, _F, _P, _P]),
    "ac_gru_layer_train": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_gru_layer_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_swa_update": (_I, [_P, _P, _L, _I, _P]),
    "ac_grad_sumsq": (_I, [_P, _L, _P, _P]),
    "ac_clip_coef": (_I, [_P, _F, _F, _P]),
    "ac_scale_by_coef": (_I, [_P, _L, _P, _P]),
    "ac_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _I, _P, _P]),
    "ac_adam_commit": (_I, [_P, _P, _P]),
    # EfficientNet-B2 encoder (csrc/effnet.hip)
    "ac_pointwise_conv": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _F, _P, _I, _P]),
    "ac_top_db_clamp": (_I, [_P, _L, _F, _P, _I, _P]),
    "ac_effnet_stem": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ac_effnet_depthwise": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ac_effnet_se_gate": (_I, [_P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_pw_gemm_packed_bytes": (_L, [_I, _I]),
    "ac_pw_gemm_pack": (_I, [_P, _P, _I, _I, _P]),
    "ac_pw_gemm_bf16x3": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _F, _P, _I, _P]),
    "ac_pw_gemm_bf16x3_ex": (_I, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _I, _F, _P, _I, _F, _U64, _P, _L, _P]),
    "ac_pw_gemm_pack_strided": (_I, [_P, _L, _L, _P, _I, _I, _P]),
    "ac_pw_gemm_pack_table": (_I, [_P, _I, _P]),
    "ac_effnet_se_gate_t": (_I, [_P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ac_effnet_expand_depthwise": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    # waveform ingest (csrc/ingest.hip)
    "ac_ingest_resample": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ac_mfma_bf16_probe": (_I, [_P, _I, _I, _P]),
    "ac_placement_probe": (_I, [_P, _I, _I, _I, _P]),
    "ac_stream_create_cu_mask": (_I, [_I, _I, _P]),
    "ac_stream_destroy": (_I, [_P]),
}

_lib = None

# Bumped whenever a HIP kernel rewrites parameters in place (fused Adam, parameter flattening): torch's tensor
# version counters do not see those writes, so the packed-weight caches of the inference path key on this as well.
PARAM_GENERATION = 0


# Per-tensor counters for the same purpose, keyed by id(parameter) with a weak reference so that a recycled id is
# never mistaken for the old tensor: the packed weights of a FROZEN sub-network (the Cnn14 under TrainEngine) must not
# be invalidated by every optimiser step on the OTHER parameters - captured graphs hold raw addresses of those packs.
_TENSOR_GENERATION = {}


def bump_param_generation(params=None):
    """``params``: the tensors a HIP kernel just rewrote in place (None: unknown - every cache keyed on the global
    counter is rebuilt; caches keyed per tensor are rebuilt only for tensors named here)."""
    global PARAM_GENERATION
    import weakref
    PARAM_GENERATION += 1
    if params is not None:
        for p_ in params:
            hit = _TENSOR_GENERATION.get(id(p_))
            n = hit[1] + 1 if (hit is not None and hit[0]() is p_) else 1
            _TENSOR_GENERATION[id(p_)] = (weakref.ref(p_), n)


def param_generation():
    return PARAM_GENERATION


def tensor_generation(t):
    hit = _TENSOR_GENERATION.get(id(t))
    return hit[1] if (hit is not None and hit[0]() is t) else 0


def param_generation_flat(engine):
    """Identity of the flat parameter storage a captured training graph was recorded against."""
    return engine.flat.flat.data_ptr()


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HipLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m audiocaption_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback on the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.ac_abi_version() != ABI_VERSION:
        raise HipLibraryError("libaudiocaption_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


AC_ERR_ARG, AC_ERR_LAUNCH = -1, -2
_ERR = {-1: "AC_ERR_ARG (arguments rejected)", -2: "AC_ERR_LAUNCH (HIP launch failed)"}


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("the HIP path needs tensors on a ROCm device (got a CPU tensor); "
                              "there is no CPU fallback")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32c(t):
    """Contiguous fp32 view/copy."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
