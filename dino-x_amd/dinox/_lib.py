"""ctypes binding of libdinox_hip.so (the C ABI declared in include/dinox.h).

The library is the product's only compute path: if it is missing or cannot be loaded this module
raises -- there is no eager/PyTorch/CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64; loading it FIRST makes this library's HIP dependency resolve to that same runtime.
# The other order puts two HIP runtimes in the process, and the second one to initialise finds no device
# ("no ROCm-capable device is detected"): measured on the MI355X box with `import dinox` ahead of `import torch`.
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DINOX_LIB") or os.path.join(_HERE, "libdinox_hip.so")     # DINOX_LIB: A/B another build of the same ABI

F32, BF16 = 0, 1
EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_RESIDUAL, EPI_ACCUM, EPI_AUXGRAD = 1, 2, 4, 8, 16, 32

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    """struct dinox_gemm_args (include/dinox.h)."""
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i64), ("N", i64), ("K", i64),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("batch", i64), ("strideA", i64), ("strideB", i64), ("strideC", i64),
        ("transA", i32), ("transB", i32), ("in_dtype", i32), ("out_dtype", i32), ("epilogue", i32),
        ("alpha", f32),
        ("bias", vp), ("residual", vp), ("ldr", i64), ("aux", vp), ("ldaux", i64), ("colsum", vp), ("ws", vp),
    ]


class BlockFwdArgs(C.Structure):
    """struct dinox_block_fwd_args (include/dinox.h)."""
    _fields_ = [
        ("V", i64), ("N", i64), ("D", i32), ("H", i32), ("heads", i32), ("train", i32), ("fuse_proj_ln", i32), ("fuse_fc2_ln", i32), ("eps", f32),
        ("x0", vp), ("xn1_in", vp), ("mean1_in", vp), ("rstd1_in", vp), ("xn1", vp), ("mean1", vp), ("rstd1", vp),
        ("qkv", vp), ("o", vp), ("lse", vp), ("x1", vp), ("xn2", vp), ("mean2", vp), ("rstd2", vp), ("act", vp), ("pre", vp), ("x2", vp),
        ("next_g", vp), ("next_b", vp), ("next_eps", f32), ("next_dtype", i32), ("yn", vp), ("meann", vp), ("rstdn", vp),
        ("n1w", vp), ("n1b", vp), ("n2w", vp), ("n2b", vp), ("wqkv", vp), ("wproj", vp), ("w1", vp), ("w2", vp),
        ("bqkv", vp), ("bproj", vp), ("b1", vp), ("b2", vp),
    ]


class BlockBwdArgs(C.Structure):
    """struct dinox_block_bwd_args (include/dinox.h)."""
    _fields_ = [
        ("V", i64), ("N", i64), ("D", i32), ("H", i32), ("heads", i32), ("reserved", i32),
        ("g", vp), ("g_lowp", vp), ("g_lowp_buf", vp),
        ("x0", vp), ("x1", vp), ("xn1", vp), ("xn2", vp), ("qkv", vp), ("o", vp), ("lse", vp), ("pre", vp), ("act", vp),
        ("mean1", vp), ("rstd1", vp), ("mean2", vp), ("rstd2", vp), ("n1w", vp), ("n2w", vp),
        ("wqkv_t", vp), ("wproj_t", vp), ("w1_t", vp), ("w2_t", vp),
        ("dwqkv", vp), ("dbqkv", vp), ("dwproj", vp), ("dbproj", vp), ("dw1", vp), ("db1", vp), ("dw2", vp), ("db2", vp),
        ("dn1w", vp), ("dn1b", vp), ("dn2w", vp), ("dn2b", vp),
        ("dpre", vp), ("dxn2", vp), ("d_o", vp), ("dqkv", vp), ("dxn1", vp), ("g1", vp), ("g1_lowp", vp), ("g0_lowp", vp),
        ("attn_ws", vp), ("ln_ws", vp), ("tn_ws", vp), ("tn_ws_bytes", i64),
    ]


# name -> (restype, argtypes); order and types mirror include/dinox.h exactly.
SIGNATURES = {
    "dinox_version": (i32, []),
    "dinox_last_error": (C.c_char_p, []),
    "dinox_device_ok": (i32, []),
    "dinox_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "dinox_gemm_kernel_name": (C.c_char_p, [C.POINTER(GemmArgs)]),
    "dinox_gemm_ws_bytes": (i64, [C.POINTER(GemmArgs)]),
    "dinox_colsum": (i32, [vp, vp, i64, i64, i64, i32, i32, vp]),
    "dinox_linear_residual_ln_ok": (i32, [i64, i32, i32]),
    "dinox_linear_residual_ln": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, i32, vp, vp, i64, i32, i32, vp]),
    "dinox_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]),
    "dinox_layernorm_bwd_ws_bytes": (i64, [i64, i32]),
    "dinox_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "dinox_linear_ln_bwd_ok": (i32, [i64, i32, i32]),
    "dinox_linear_ln_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "dinox_attention_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_attention_bwd_ws_bytes": (i64, [i32, i32, i32]),
    "dinox_attention_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_qkv_attention_ok": (i32, [i32, i32, i32, i32, i32]),
    "dinox_qkv_attention_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_patch_unfold": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_patch_unfold_ld": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dinox_tokens_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_tokens_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_scale_embed_fwd": (i32, [vp] * 12 + [i32, i32, i32, f32, vp]),
    "dinox_scale_embed_bwd_ws_bytes": (i64, [i32, i32, i32]),
    "dinox_scale_embed_bwd": (i32, [vp] * 17 + [i32, i32, i32, vp]),
    "dinox_dino_ce": (i32, [vp, vp, vp, f32, f32, f32, vp, vp, vp, i32, i32, vp]),
    "dinox_dino_ce_multi": (i32, [vp, vp, vp, f32, f32, f32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "dinox_colmean": (i32, [vp, vp, i32, i32, vp]),
    "dinox_center_ema": (i32, [vp, vp, f32, i32, vp]),
    "dinox_gram_normalize": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "dinox_sqsum": (i32, [vp, i64, f32, vp, vp, vp]),
    "dinox_gram_normalize_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "dinox_slice_views_lds_bytes": (i64, [i32, i32]),
    "dinox_slice_views": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "dinox_slice_views_patches": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dinox_softmax_rows": (i32, [vp, vp, i64, i32, i32, i64, vp]),
    "dinox_softmax_bwd_rows": (i32, [vp, vp, vp, f32, i64, i32, i32, i64, vp]),
    "dinox_koleo_normalize": (i32, [vp, vp, vp, vp, i64, i32, f32, vp]),
    "dinox_koleo_nn": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "dinox_koleo_loss": (i32, [vp, i32, f32, vp, vp]),
    "dinox_koleo_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, vp, vp]),
    "dinox_adamw_ema": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, f32, vp, vp, vp]),
    "dinox_adamw_ema_dev": (i32, [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, vp, vp, vp]),
    "dinox_sumsq": (i32, [vp, i64, vp, vp, vp]),
    "dinox_cast_bf16": (i32, [vp, vp, i64, vp]),
    "dinox_cast_transpose_bf16": (i32, [vp, vp, i32, i32, vp]),
    "dinox_cast_transpose_bf16_multi": (i32, [vp, vp, vp, i32, i64, vp]),
    "dinox_take_rows": (i32, [vp, vp, i64, i64, i32, i64, i32, vp]),
    "dinox_put_rows": (i32, [vp, vp, i64, i64, i32, i64, i32, i32, vp]),
    "dinox_axpy": (i32, [vp, vp, f32, i64, vp]),
    "dinox_lincomb3": (i32, [vp, vp, vp, f32, f32, vp, vp]),
    "dinox_zero": (i32, [vp, i64, vp]),
    "dinox_gelu_fwd": (i32, [vp, vp, i64, vp]),
    "dinox_gelu_bwd": (i32, [vp, vp, vp, i64, vp]),
    "dinox_block_forward": (i32, [C.POINTER(BlockFwdArgs), vp]),
    "dinox_block_backward": (i32, [C.POINTER(BlockBwdArgs), vp]),
    "dinox_gemm_timer_start": (i32, [i32]),
    "dinox_gemm_timer_stop": (i64, [C.c_char_p, i64]),
}


class DinoxLibraryError(ImportError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise DinoxLibraryError(
            f"{LIB_PATH} not found: the HIP kernel library is the only compute path of this package "
            "(no CPU/eager fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dino-x_amd/csrc`.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise DinoxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DinoxLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    return lib.dinox_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
