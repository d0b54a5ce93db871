"""ctypes binding of libvitk.so (the C-ABI declared in include/vitk.h).

There is no fallback: importing this package without the built library, or calling an op on
a non-GPU tensor, raises.  Build with ``make -C vit-pytorch_amd/csrc`` (or
``__graft_entry__.build()``).
"""
import ctypes
import os
from contextlib import contextmanager

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITK_LIB") or os.path.join(_HERE, "libvitk.so")   # VITK_LIB: an instrumented build (tools/)

BF16, F32 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_PATCH = range(6)


class CastEntry(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_t", ctypes.c_void_p),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("ld_t", ctypes.c_int), ("tile0", ctypes.c_int)]


class EpilogueArgs(ctypes.Structure):
    _fields_ = [("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("aux", ctypes.c_void_p),
                ("pos", ctypes.c_void_p), ("ld_res", ctypes.c_long), ("ld_aux", ctypes.c_long),
                ("npatch", ctypes.c_int), ("dropout_p", ctypes.c_float), ("seed", ctypes.c_uint64)]


_vp, _i, _l, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64

# name -> argtypes; every function returns int (0 = ok)
PROTOTYPES = {
    "vitk_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp],
    "vitk_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitk_layernorm_bwd_static": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitk_layernorm_bwd_dz": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u64, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitk_gemm_nt": [_vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, ctypes.POINTER(EpilogueArgs), _i, _i, _vp],
    "vitk_gemm_tn_acc": [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _vp],
    "vitk_colsum_acc": [_vp, _vp, _i, _i, _l, _i, _vp],
    "vitk_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _f, _i, _vp],
    "vitk_attn_bwd": [_vp] * 11 + [_i, _i, _i, _i, _l, _l, _l, _f, _i, _vp],
    "vitk_attn_fwd_dropout": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _f, _f, _u64, _i, _vp],
    "vitk_attn_bwd_dropout": [_vp] * 11 + [_i, _i, _i, _i, _l, _l, _l, _f, _f, _u64, _i, _vp],
    "vitk_attn_probs": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _f, _i, _vp],
    "vitk_window_attn_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "vitk_debug_force_window_attn_valu": [_i],
    "vitk_gemm_tn_set_workspace": [_vp, ctypes.c_size_t],
    "vitk_mask_fold_fwd": [_vp, _vp, _vp, ctypes.c_long, _i, ctypes.c_float, _vp],
    "vitk_mask_fold_bwd": [_vp] * 5 + [ctypes.c_long, _i, ctypes.c_float, _vp],
    "vitk_rkr_fold_fwd": [_vp] * 7 + [_i, _i, _i, _vp],
    "vitk_rkr_fold_bwd": [_vp] * 12 + [_i, _i, _i, _vp],
    "vitk_scale_add": [_vp, _vp, _vp, _vp, _l, _l, _i, _vp],
    "vitk_scale_cast": [_vp, _vp, _vp, _l, _l, _i, _vp],
    "vitk_window_attn_bwd": [_vp] * 7 + [_i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "vitk_wstd_fwd": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "vitk_wstd_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "vitk_groupnorm_fwd": [_vp] * 7 + [_i, _i, _i, _i, _f, _i, _i, _vp],
    "vitk_groupnorm_bwd": [_vp] * 10 + [_i, _i, _i, _i, _i, _i, _i, _vp],
    "vitk_groupnorm_add_relu_fwd": [_vp] * 9 + [_i, _i, _i, _i, _f, _i, _vp],
    "vitk_groupnorm_add_relu_bwd": [_vp] * 13 + [_i, _i, _i, _i, _i, _i, _vp],
    "vitk_add_relu_fwd": [_vp, _vp, _vp, _vp, _l, _i, _vp],
    "vitk_add_relu_bwd": [_vp, _vp, _vp, _vp, _l, _i, _vp],
    "vitk_im2col_conv": [_vp, _vp] + [_i] * 8 + [_l] * 4 + [_i, _i, _vp],
    "vitk_col2im_conv": [_vp, _vp] + [_i] * 9 + [_i, _vp],
    "vitk_im2col": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "vitk_embed_cls": [_vp, _vp, _vp, _i, _i, _i, _f, _u64, _i, _vp],
    "vitk_embed_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _i, _i, _vp],
    "vitk_cast_weight": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "vitk_cast_weights_batched": [ctypes.POINTER(CastEntry), _i, _i, _vp],
    "vitk_dropout_bwd": [_vp, _vp, _l, _i, _f, _u64, _i, _i, _vp],
    "vitk_convert": [_vp, _vp, _l, _i, _i, _vp],
    "vitk_attn_set_oversubscription": [_i],
    "vitk_gemm_set_persistent": [_i],
    "vitk_sgd_norm_sq": [_vp, _i, _i, _vp, _vp],
    "vitk_sgd_step": [_vp, _i, _i, _vp, _f, _f, _f, _f, _f, _vp],
    "vitk_comm_unique_id": [_vp, ctypes.c_size_t],
    "vitk_comm_init": [_i, _i, _vp, ctypes.POINTER(ctypes.c_void_p)],
    "vitk_allreduce": [_vp, _vp, ctypes.c_size_t, _i, _i, _vp],
    "vitk_broadcast": [_vp, _vp, ctypes.c_size_t, _i, _i, _vp],
    "vitk_comm_destroy": [_vp],
}

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make -C vit-pytorch_amd/csrc` "
                      "(there is no fallback path)")
lib = ctypes.CDLL(LIB_PATH)
lib.vitk_version.restype = ctypes.c_int
lib.vitk_error_string.restype = ctypes.c_char_p
lib.vitk_error_string.argtypes = [ctypes.c_int]
lib.vitk_debug_force_attn_streaming.restype = None
lib.vitk_debug_force_attn_streaming.argtypes = [ctypes.c_int]
lib.vitk_debug_force_gemm_shape.restype = None
lib.vitk_debug_force_gemm_shape.argtypes = [ctypes.c_int]
lib.vitk_layernorm_bwd_workspace.restype = ctypes.c_size_t
lib.vitk_layernorm_bwd_workspace.argtypes = [ctypes.c_int, ctypes.c_int]
lib.vitk_groupnorm_workspace.restype = ctypes.c_size_t
lib.vitk_groupnorm_workspace.argtypes = [ctypes.c_int] * 4
for _name, _args in PROTOTYPES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int


class VitkError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise VitkError(f"{what}: {lib.vitk_error_string(rc).decode()} (code {rc})")


# ----------------------------------------------------------------------------- dtypes
# The residual stream (block inputs/outputs and their gradients) is always fp32; `compute dtype`
# is the type of every other activation and of the GEMM operands.
STREAM_DTYPE = torch.float32
_compute_dtype = torch.bfloat16


def set_compute_dtype(dtype):
    """Activation dtype of the kernels: torch.bfloat16 (default) or torch.float32 (parity checks)."""
    global _compute_dtype
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


@contextmanager
def compute_dtype(dtype):
    old = get_compute_dtype()
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        set_compute_dtype(old)


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise VitkError(f"unsupported dtype {dt}")


from . import ops  # noqa: E402,F401
