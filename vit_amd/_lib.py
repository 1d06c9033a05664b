"""ctypes binding of libvit_amd.so (the C ABI of include/vit_amd.h).

There is NO fallback: if the shared library is missing or lacks a symbol, import of the
ops fails loudly (``VitLibraryError``). torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER, Structure

VIT_F32, VIT_BF16 = 0, 1
EPI_BIAS, EPI_GELU, EPI_QUICK_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_DQUICK_GELU = 0, 1, 2, 3, 4, 5

# VIT_AMD_LIB: another build of the same library (the host-sanitizer build of csrc/Makefile `asan`)
LIB_PATH = os.environ.get("VIT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvit_amd.so")


class VitLibraryError(RuntimeError):
    pass


class VitError(RuntimeError):
    """A C-ABI call returned a negative VIT_E_* code."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libvit_amd error {code}: {msg}")
        self.code = code


class vit_config(Structure):
    _fields_ = [("hidden_size", c_int32), ("num_hidden_layers", c_int32), ("num_attention_heads", c_int32),
                ("intermediate_size", c_int32), ("image_size", c_int32), ("patch_size", c_int32),
                ("num_channels", c_int32), ("hidden_act", c_int32), ("variant", c_int32), ("qkv_bias", c_int32),
                ("dtype", c_int32), ("layer_norm_eps", c_float)]


class vit_preproc_config(Structure):
    _fields_ = [("in_height", c_int32), ("in_width", c_int32), ("channels", c_int32), ("resize_height", c_int32),
                ("resize_width", c_int32), ("crop_height", c_int32), ("crop_width", c_int32), ("resample", c_int32),
                ("do_rescale", c_int32), ("do_normalize", c_int32), ("rescale_factor", c_double),
                ("image_mean", c_float * 4), ("image_std", c_float * 4)]


HEAD_LINEAR, HEAD_TANH = 0, 1

# name -> (restype, argtypes): every symbol include/vit_amd.h declares
SIGNATURES = {
    "vit_last_error": (c_char_p, []),
    "vit_abi_version": (c_int, []),
    "vit_create": (c_int, [POINTER(vit_config), POINTER(c_void_p)]),
    "vit_destroy": (c_int, [c_void_p]),
    "vit_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int64, c_void_p]),
    "vit_missing_weights": (c_int, [c_void_p, POINTER(c_char_p)]),
    "vit_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vit_forward_static": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vit_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_float,
                                  c_int, c_void_p]),
    "vit_gemm_bias_act": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "vit_attention_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vit_patch_embed_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p]),
    "vit_patch_kpad": (c_int, [c_int, c_int, c_int]),
    "vit_convert": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "vit_layernorm_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_int, c_void_p]),
    "vit_layernorm_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p]),
    "vit_gemm_ln_folded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p]),
    "vit_gemm_residual_row_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_void_p]),
    "vit_row_stats_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "vit_gemm_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_void_p]),
    "vit_attention_fwd_train": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vit_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "vit_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_float, c_void_p]),
    "vit_transpose_pad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vit_linear_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vit_train_stash_bytes": (c_size_t, [c_void_p, c_int]),
    "vit_train_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "vit_grad_bytes": (c_size_t, [c_void_p]),
    "vit_grad_slot": (c_int, [c_void_p, c_char_p, POINTER(c_size_t), POINTER(c_int64), POINTER(c_int64),
                              POINTER(c_int64)]),
    "vit_grad_layer_range": (c_int, [c_void_p, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "vit_train_init": (c_int, [c_void_p, c_void_p]),
    "vit_train_refresh_transposes": (c_int, [c_void_p, c_void_p]),
    "vit_forward_train": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                  c_size_t, c_void_p]),
    "vit_backward_stage": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "vit_sgd_step": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "vit_head_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_void_p]),
    "vit_pos_embed_interpolate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vit_clone_resized": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_void_p]),
    "vit_preproc_create": (c_int, [POINTER(vit_preproc_config), POINTER(c_void_p)]),
    "vit_preproc_destroy": (c_int, [c_void_p]),
    "vit_preproc_output_size": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "vit_preproc_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "vit_preproc_run": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitLibraryError(
            f"{LIB_PATH} not found: build it with `make -C vit_amd/csrc` (or __graft_entry__.build()). "
            "vit_amd has no CPU or eager fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise VitLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VitLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise VitError(rc, load().vit_last_error().decode(errors="replace"))
