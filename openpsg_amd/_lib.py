"""ctypes binding of libpsg_hip.so (the C ABI declared in include/psg_hip.h).

The product path has NO CPU fallback: if the shared object is missing or a call fails, this
module raises `PsgHipError` - loudly - instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpsg_hip.so")

PSG_ABI_VERSION = 600            # include/psg_hip.h; checked against psg_version() of the loaded library
PSG_F32, PSG_BF16, PSG_F16 = 0, 1, 2
PSG_EMPTY_UNIFORM, PSG_EMPTY_UNMASKED = 0, 1
PSG_XATTN_MFMA, PSG_XATTN_SIMPLE, PSG_XATTN_MFMA_V1 = 0, 1, 2
PSG_TRACE_NONE, PSG_TRACE_SKINNY_GEMM, PSG_TRACE_CROSS_ATTN, PSG_TRACE_DECODE_LAYER = 0, 1, 2, 3


class PsgHipError(RuntimeError):
    pass


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

PSG_PRO_NONE, PSG_PRO_RMSNORM, PSG_PRO_DECODE_ATTN, PSG_PRO_SILU_MUL = 0, 1, 2, 3


class Prologue(C.Structure):
    """`psg_prologue` of include/psg_hip.h (row operation fused into a decode projection)."""
    _fields_ = [("kind", _i), ("in_splits", _i), ("input", _vp), ("resid", _vp), ("norm_w", _vp), ("eps", _f),
                ("heads", _i), ("ctx", _i), ("tok_pair", _vp), ("tok_pos", _vp), ("rope_cos", _vp), ("rope_sin", _vp),
                ("k_cache", _vp), ("v_cache", _vp), ("sync", _vp)]

# name -> argtypes (return type is int unless noted).  Mirrors include/psg_hip.h one to one;
# tests/test_abi_symbols.py checks the header and this table against the built library.
SIGNATURES = {
    "psg_version": [],
    "psg_create": [_i, C.POINTER(_vp)],
    "psg_destroy": [_vp],
    "psg_device_info": [_vp, C.POINTER(_i), C.c_char_p, _i],
    "psg_set_option": [_vp, C.c_char_p, _i],
    "psg_get_option": [_vp, C.c_char_p, C.POINTER(_i)],
    "psg_set_trace_buffer": [_vp, _i, _vp, _i64],
    "psg_patch_embed_workspace": [_vp, _i, _i, _i, _i, _i, C.POINTER(_i64)],
    "psg_patch_embed": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp],
    "psg_mask_grid": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "psg_object_bitmasks": [_vp, _vp, _i, _vp, _i, _vp, _i, _vp],
    "psg_qformer_embed": [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _vp, _i, _vp],
    "psg_add_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i64, _i, _vp, _i, _vp],
    "psg_add_layernorm_periodic": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _i64, _i, _vp, _i, _vp],
    "psg_add_layernorm_indexed": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _i64, _i, _vp, _i, _vp],
    "psg_add_layernorm_res32": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _i64, _i, _vp, _vp, _i, _vp],
    "psg_bias_gelu": [_vp, _vp, _vp, _i64, _i, _vp, _i, _vp],
    "psg_qformer_self_attn": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_qformer_self_attn_cls": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_qformer_cls_attn_input": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_qformer_self_attn_shared": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_qformer_cross_attn": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_exist_head": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "psg_topk": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "psg_gather_rows": [_vp, _vp, _i, _vp, _i64, _i, _i64, _vp, _i, _i64, _vp],
    "psg_gather_pair_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "psg_rmsnorm": [_vp, _vp, _vp, _i, _vp, _f, _i64, _i, _vp, _i, _i, _vp],
    "psg_rope_kvwrite": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "psg_train_object_bitmasks": [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "psg_bce_with_logits": [_vp, _vp, _vp, _i, _f, _vp, _vp],
    "psg_cross_entropy_rows": [_vp, _vp, _i64, _i, _vp, _vp, _i, _vp],
    "psg_llm_attn": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp],
    "psg_prefill_attn": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_prefill_attn_rope": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "psg_decode_attn": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "psg_silu_mul": [_vp, _vp, _i, _i64, _i, _vp, _i, _vp],
    "psg_skinny_gemm_plan": [_vp, _i, _i, _i, _i, C.POINTER(_i)],
    "psg_skinny_gemm": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "psg_batch_gemm_plan": [_vp, _i64, _i, _i, _i, _i, _i, C.POINTER(_i)],
    "psg_batch_gemm": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp],
    "psg_split_f16x2": [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp],
    "psg_rmsnorm_split2": [_vp, _vp, _vp, _i, _vp, _f, _i64, _i, _vp, _vp, _vp],
    "psg_split_gemm_w16_plan": [_vp, _i, _i, _i, _i, C.POINTER(_i)],
    "psg_split_gemm_w16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "psg_qformer_cross_attn_indexed": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "psg_skinny_gemm_w16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "psg_skinny_gemm_fused": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "psg_rmsnorm_split": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _i64, _i, _vp, _vp, _i, _vp],
    "psg_rope_kvwrite_scaled": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp],
    "psg_silu_mul_split": [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _i, _vp],
    "psg_decode_layer_workspace": [_vp, _i, _i, _i, C.POINTER(_i64), C.POINTER(_i64)],
    "psg_decode_layer_supported": [_vp, _i, _i, _i, _i, _i],
    "psg_decode_layers": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp],
    "psg_decode_layer": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f,
                         _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "psg_reduce_partials": [_vp, _vp, _i, _i64, _vp, _i, _vp],
    "psg_masked_mean_pool_workspace": [_vp, _i, _i, _i, _i, C.POINTER(_i64)],
    "psg_masked_mean_pool": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i64, _vp],
    "psg_masked_split_mean_pool": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i64, _vp],
    "psg_bilinear_scores": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "psg_dense_gemm": [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp],
    "psg_dense_gemm_ex": [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp],
    "psg_dense_gemm_tiled": [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "psg_interleave_gate_up": [_vp, _vp, _vp, _i, _i, _vp],
    "psg_dense_gemm_split": [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _i, _vp],
    "psg_train_layernorm_fwd": [_vp, _vp, _vp, _vp, _f, _i64, _i, _vp, _vp, _vp, _vp],
    "psg_train_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp],
    "psg_train_rmsnorm_fwd": [_vp, _vp, _vp, _f, _i64, _i, _vp, _vp, _vp],
    "psg_train_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp],
    "psg_split_f16x3": [_vp, _vp, _i64, _i, _i64, _i, _vp, _vp, _vp],
    "psg_scale_rows_cols": [_vp, _vp, _i64, _i, _vp, _vp, _vp],
    "psg_train_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _f, _vp, _vp, _vp],
    "psg_train_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _f, _vp, _vp, _vp, _vp],
    "psg_train_gelu_fwd": [_vp, _vp, _i64, _vp, _vp],
    "psg_train_gelu_bwd": [_vp, _vp, _vp, _i64, _vp, _vp],
    "psg_train_silu_mul_fwd": [_vp, _vp, _i64, _i, _vp, _vp],
    "psg_train_silu_mul_bwd": [_vp, _vp, _vp, _i64, _i, _vp, _vp],
    "psg_train_rope": [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp],
    "psg_train_ce_bwd": [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp],
    "psg_train_bce_bwd": [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp],
    "psg_greedy_step": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp],
}

_lib = None
_ctx = {}


def load():
    """dlopen the library and type every entry point.  Works without a GPU (no HIP call is made)."""
    global _lib
    if _lib is not None:
        return _lib
    # The library must share torch's HIP runtime (streams and device pointers come from torch):
    # torch bundles its own libamdhip64.so.7, and whichever copy is loaded first serves the whole
    # process, so make sure it is torch's.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PsgHipError(
            f"{LIB_PATH} is missing: build it with `python -m openpsg_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    lib.psg_last_error.restype = C.c_char_p
    lib.psg_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise PsgHipError(f"{LIB_PATH} does not export {name}; rebuild the library")
        fn.restype = _i
        fn.argtypes = argtypes
    built = lib.psg_version()
    if built != PSG_ABI_VERSION:
        raise PsgHipError(f"{LIB_PATH} was built with ABI version {built}, this binding is written against "
                          f"{PSG_ABI_VERSION} (include/psg_hip.h): rebuild with `python -m openpsg_amd.csrc.build --force`")
    _lib = lib
    return lib


def last_error() -> str:
    return load().psg_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise PsgHipError(f"{what} failed (status {rc}): {last_error()}")


def ctx(device_index: int):
    """One psg_ctx per device, created on first use."""
    if device_index not in _ctx:
        lib = load()
        h = _vp()
        check(lib.psg_create(int(device_index), C.byref(h)), "psg_create")
        _ctx[device_index] = h
    return _ctx[device_index]


def device_info(device_index: int):
    n = _i(0)
    buf = C.create_string_buffer(64)
    check(load().psg_device_info(ctx(device_index), C.byref(n), buf, 64), "psg_device_info")
    return n.value, buf.value.decode()


def set_option(device_index: int, name: str, value: int):
    check(load().psg_set_option(ctx(device_index), name.encode(), int(value)), f"psg_set_option({name})")


def get_option(device_index: int, name: str) -> int:
    v = _i(0)
    check(load().psg_get_option(ctx(device_index), name.encode(), C.byref(v)), f"psg_get_option({name})")
    return v.value


def set_trace_buffer(device_index: int, kind: int, tensor=None):
    """Per-wave cycle stamps of the next launches of `kind` go to `tensor` (int64, on the device); None = off."""
    if tensor is None or kind == PSG_TRACE_NONE:
        check(load().psg_set_trace_buffer(ctx(device_index), PSG_TRACE_NONE, None, 0), "psg_set_trace_buffer")
    else:
        check(load().psg_set_trace_buffer(ctx(device_index), int(kind), tensor.data_ptr(),
                                          tensor.numel() * tensor.element_size()), "psg_set_trace_buffer")
