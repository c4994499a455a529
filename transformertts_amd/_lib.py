"""ctypes binding of libttts_hip.so (the C ABI declared in include/ttts_hip.h).

There is deliberately no fallback: if the shared library has not been built (run
`python -m transformertts_amd.build` or `__graft_entry__.build()`), importing a kernel raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch  # noqa: F401  -- MUST precede the dlopen below: PyTorch-ROCm ships its own libamdhip64; if ours pulled the
#                system copy in first, the process would hold two HIP runtimes and every launch fails with
#                "no ROCm-capable device is detected"

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTTS_LIB", os.path.join(_HERE, "libttts_hip.so"))   # TTTS_LIB: development A/B builds

P, I, L, F, U, Z = c_void_p, c_int, c_int64, c_float, c_uint64, c_size_t

# name -> (restype, argtypes); mirrors include/ttts_hip.h one to one
SIGNATURES = {
    "ttts_last_error": (c_char_p, []),
    "ttts_abi_version": (I, []),
    "ttts_linear_fwd": (I, [P, P, P, P, P, L, I, I, I, F, U, P, I, I, P]),
    "ttts_linear_bwd_data": (I, [P, P, P, P, L, I, I, P, F, P]),
    "ttts_wgrad_workspace_bytes": (Z, [L, I, I, I]),
    "ttts_linear_bwd_weight": (I, [P, P, P, P, P, Z, L, I, I, I, I, I, P, P]),
    "ttts_split_bytes": (Z, [L, L]),
    "ttts_split_image_bytes": (Z, [L, L, I, I, I]),
    "ttts_gemm_tile_choice": (I, [L, I, I, I]),
    "ttts_weight_split": (I, [P, P, I, I, I, I, I, P]),
    "ttts_weight_split_units": (L, [L, L, I, I]),
    "ttts_weight_split_batched": (I, [P, I, L, P]),
    "ttts_linear_fwd_x6": (I, [P, P, P, P, P, L, I, I, I, F, U, P, I, I, P]),
    "ttts_linear_fwd_h3": (I, [P, P, P, P, P, L, I, I, I, F, U, P, I, I, P, P, P]),
    "ttts_conv1d_fwd_h3": (I, [P, P, P, P, I, I, I, I, I, P, P, P]),
    "ttts_conv1d_fwd_h3_bn_blocks": (I, [I, I, I, I, I]),
    "ttts_conv1d_fwd_h3_bn_chunk_rows": (I, [I, I, I, I, I]),
    "ttts_bn_train_stats_from_partials": (I, [P, I, P, P, P, P, P, I, F, F, P]),
    "ttts_bn_train_stats_from_partials_rows": (I, [P, I, P, I, P, P, P, P, P, I, F, F, P]),
    "ttts_bn_train_stats_twin": (I, [P, I, P, I, P, P, P, I, P, I, P, P, P, P, P, I, F, F, P]),
    "ttts_amax_partials": (I, [P, L, P, P]),
    "ttts_linear_bwd_data_h3": (I, [P, P, P, P, L, I, I, P, F, P, P, P]),
    "ttts_act_image": (I, [P, P, P, L, I, P]),
    "ttts_linear_fwd_h3i": (I, [P, P, P, P, P, P, L, I, I, I, F, U, P, P, P]),
    "ttts_linear_bwd_data_h3i": (I, [P, P, P, P, P, L, I, I, P, F, P, P]),
    "ttts_linear_fwd_h3d": (I, [P, P, P, P, P, L, I, I, I, F, U, P, P, P, P]),
    "ttts_linear_bwd_data_h3d": (I, [P, P, P, P, L, I, I, P, F, P, P, P]),
    "ttts_linear_fwd_h3d_img": (I, [P, P, P, P, P, L, I, I, P, P, I, P]),
    "ttts_head_image": (I, [P, L, P, L, P, L, I, P]),
    "ttts_conv1d_bwd_data_h3": (I, [P, P, P, I, I, I, I, I, P, P]),
    "ttts_linear_bwd_data_x6": (I, [P, P, P, P, L, I, I, P, F, P]),
    "ttts_conv1d_fwd_x6": (I, [P, P, P, P, I, I, I, I, I, P]),
    "ttts_conv1d_bwd_data_x6": (I, [P, P, P, I, I, I, I, I, P]),
    "ttts_linear_bwd_weight_x6": (I, [P, P, P, P, P, Z, L, I, I, I, I, I, P, P]),
    "ttts_conv1d_bwd_weight_x6": (I, [P, P, P, P, P, Z, I, I, I, I, I, I, P, P]),
    "ttts_linear_bwd_weight_h3": (I, [P, P, P, P, P, Z, L, I, I, I, I, I, P, P, P, P]),
    "ttts_conv1d_bwd_weight_h3": (I, [P, P, P, P, P, Z, I, I, I, I, I, I, P, P, P, P]),
    "ttts_linear_bwd_weight_h3_parts": (I, [P, P, P, P, I, P, Z, L, I, I, I, P, P, P, P]),
    "ttts_wgrad_group_ok": (I, [L, I, I, I]),
    "ttts_wgrad_group": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, P, P]),
    "ttts_conv1d_pack_bytes": (Z, [I, I, I]),
    "ttts_conv1d_pack_weight": (I, [P, P, P, I, I, I, P]),
    "ttts_conv1d_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "ttts_conv1d_bwd_data": (I, [P, P, P, I, I, I, I, I, P]),
    "ttts_conv1d_bwd_weight": (I, [P, P, P, P, P, Z, I, I, I, I, I, I, P, P]),
    "ttts_bn_workspace_bytes": (Z, [L, I]),
    "ttts_bn_train_stats": (I, [P, P, P, P, P, P, P, Z, L, I, F, F, P]),
    "ttts_bn_eval_stats": (I, [P, P, P, P, I, F, P]),
    "ttts_bn_apply_fwd": (I, [P, P, P, P, P, P, L, I, I, F, U, P, P, P]),
    "ttts_bn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, Z, L, I, I, F, U, P, I, P, I, P]),
    "ttts_layernorm_fwd": (I, [P, P, P, P, P, P, L, I, F, P, P, P, P]),
    "ttts_layernorm_bwd_workspace_bytes": (Z, [I]),
    "ttts_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, Z, L, I, I, P, P, P, P]),
    "ttts_layernorm_bwd_drop": (I, [P, P, P, P, P, P, P, P, P, Z, L, I, I, P, F, U, P, P, P, P, P, P]),
    "ttts_attention_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, U, P, P]),
    "ttts_attention_fwd_x6": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, U, P, P]),
    "ttts_attention_fwd_h3": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, U, P, P, P, P, P, P, P]),
    "ttts_attention_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, P]),
    "ttts_attention_bwd_x6": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, P]),
    "ttts_attention_bwd_h3": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, P, P, P, P, P, P, P, P]),
    "ttts_attention_fwd_img": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, U, P, P, P, P, L, L, L, P]),
    "ttts_attention_bwd_img": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, F, U, P, P, P, P,
                                   P, I, L, L, L, P]),
    "ttts_heads_pad": (I, [P, L, P, L, I, I, P]),
    "ttts_heads_unpad": (I, [P, P, L, L, I, I, P]),
    "ttts_embedding_fwd": (I, [P, P, P, L, I, I, P, P]),
    "ttts_embedding_bwd": (I, [P, P, P, L, I, I, I, P]),
    "ttts_posenc_fwd": (I, [P, P, P, P, I, I, I, F, U, P, P, P]),
    "ttts_posenc_bwd_workspace_bytes": (Z, []),
    "ttts_posenc_bwd": (I, [P, P, P, P, P, Z, I, I, I, F, U, P, I, P]),
    "ttts_relu_dropout_bwd": (I, [P, P, P, L, F, P, P]),
    "ttts_dropout_bwd": (I, [P, P, L, F, U, P, P, P]),
    "ttts_add": (I, [P, P, P, L, P]),
    "ttts_add3": (I, [P, P, P, P, L, P]),
    "ttts_collate_melspec": (I, [P, P, P, I, I, I, P]),
    "ttts_collate_phoneme": (I, [P, P, P, I, I, P]),
    "ttts_loss_workspace_bytes": (Z, []),
    "ttts_loss_fwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, F, P]),
    "ttts_loss_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P]),
    "ttts_zero": (I, [P, Z, P]),
    "ttts_reduce_queue_create": (P, []),
    "ttts_reduce_queue_destroy": (None, [P]),
    "ttts_reduce_queue_pending": (L, [P]),
    "ttts_reduce_queue_flush": (I, [P, P]),
    "ttts_reduce_queue_clear": (I, [P]),
    "ttts_sched_sampling_mix": (I, [P, P, P, P, P, I, I, I, F, I, U, P, P, P]),
    "ttts_grad_norm_workspace_bytes": (Z, []),
    "ttts_grad_norm": (I, [P, P, P, Z, L, P]),
    "ttts_adam_step": (I, [P, P, P, P, P, L, F, F, F, F, L, F, P, P]),
    "ttts_rowdot_fwd": (I, [P, P, P, P, L, I, P]),
    "ttts_rowdot_bwd_workspace_bytes": (Z, [I]),
    "ttts_rowdot_bwd": (I, [P, P, P, P, P, P, P, Z, L, I, I, P, P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raises (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -m transformertts_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild the extension")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().ttts_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
