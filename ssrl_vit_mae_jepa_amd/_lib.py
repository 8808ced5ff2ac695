"""ctypes binding of libmae_hip.so (C ABI declared in include/mae_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is absent this module raises,
loudly, at import time.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C ssrl_vit_mae_jepa_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MAE_HIP_LIB") or _HERE / "lib" / "libmae_hip.so")  # MAE_HIP_LIB: an alternate build of the same library (kernel experiments)

MAE_F32, MAE_BF16, MAE_U8 = 0, 1, 2
PARAM_TRAINABLE, PARAM_FROZEN, PARAM_UNUSED, PARAM_MATRIX = 1, 2, 4, 8
LOSS_MSE, LOSS_SMOOTH_L1 = 0, 1
EPI_NONE, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_GELU_GRAD, EPI_MUL, EPI_GELU_ACT = 0, 1, 2, 3, 4, 5, 6
ABI_VERSION = 4


class MaeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads",
        "decoder_embed_dim", "decoder_depth", "decoder_num_heads", "mlp_ratio", "act_dtype", "pred_dim")] + [
        ("reserved", C.c_int32 * 4)]


class MaeHipError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"libmae_hip.so not found at {LIB_PATH}: the HIP extension is not built. "
            "Run `make -C ssrl_vit_mae_jepa_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for this path.")
    return C.CDLL(os.fspath(LIB_PATH), mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_pp = C.POINTER

# name -> (restype, argtypes); every symbol of include/mae_hip.h
SIGNATURES = {
    "mae_last_error": (C.c_char_p, []),
    "mae_abi_version": (C.c_int, []),
    "mae_engine_create": (C.c_int, [_pp(MaeConfig), _pp(_vp)]),
    "mae_engine_destroy": (None, [_vp]),
    "mae_engine_num_params": (_i64, [_vp]),
    "mae_engine_arena_elems": (_i64, [_vp]),
    "mae_engine_trainable_elems": (_i64, [_vp]),
    "mae_engine_param_info": (C.c_int, [_vp, _i64, _pp(C.c_char_p), _pp(_i64), _pp(_i64), _pp(_i32), _i64 * 4, _pp(_i32)]),
    "mae_engine_workspace_bytes": (_i64, [_vp, _i32, _i32]),
    "mae_engine_wcache_bytes": (_i64, [_vp]),
    "mae_engine_refresh_weights": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mae_mask_from_noise": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mae_engine_forward_encoder": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mae_engine_forward_decoder": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mae_patchify_gather": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mae_augment_crop_flip_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "mae_mse_loss": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp]),
    "mae_engine_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mae_engine_loss_and_grads": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mae_engine_encoder_grad_elems": (_i64, [_vp]),
    "mae_engine_backward_decoder": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "mae_engine_backward_encoder": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mae_engine_decoder_decode": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp]),
    "mae_engine_grad_ready_points": (_i32, [_vp, _pp(_i64), _i32]),
    "mae_engine_loss_and_grads_phased": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _i64, _vp, _vp, _vp, _vp, _pp(_vp), _i32, _vp]),
    "mae_engine_optimizer_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _vp, _vp, _vp]),
    "mae_engine_jepa_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32, _i32]),
    "mae_engine_jepa_loss_and_grads": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64,
                                                 _vp, _vp, _vp, _vp, _pp(_vp), _i32, _vp]),
    "mae_engine_grad_sumsq_range": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "mae_engine_clip_from_sumsq": (C.c_int, [_vp, _vp, _f32, _vp, _vp]),
    "mae_engine_adamw_range": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _i64, _vp, _i64, _i64, _vp]),
    "mae_engine_optimizer_step_ema": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _vp, _vp, _vp, _vp, _f32, _vp]),
    "mae_engine_timers_enable": (C.c_int, [_vp, _i32]),
    "mae_engine_timer_count": (_i32, [_vp]),
    "mae_engine_timer_name": (C.c_char_p, [_vp, _i32]),
    "mae_engine_timer_read": (C.c_int, [_vp, _i32, _pp(C.c_double), _pp(_i64), _pp(C.c_double), _pp(C.c_double)]),
    "mae_engine_timers_reset": (C.c_int, [_vp]),
    "mae_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mae_add_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mae_layernorm_bwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mae_linear_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mae_linear_wgrad_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "mae_linear_wgrad": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mae_linear_wgrad_pair_scratch_bytes": (C.c_int64, [_i64, _i32, _i32, _i32, _i32]),
    "mae_linear_wgrad_pair": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "mae_attention_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mae_attention_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as exc:  # pragma: no cover
        raise ImportError(f"libmae_hip.so does not export {_name}; rebuild the extension") from exc
    _fn.restype = _res
    _fn.argtypes = _args

if lib.mae_abi_version() != ABI_VERSION:  # pragma: no cover
    raise ImportError(f"libmae_hip.so ABI {lib.mae_abi_version()} != binding {ABI_VERSION}; rebuild the extension")


# Tensors whose addresses were handed to the library by ptr() stay referenced here until the call has returned (check()
# runs right after it): a temporary created inline in the call expression, e.g. ptr(x.float()), would otherwise be
# freed -- and its block possibly re-issued by the caching allocator for the next argument -- before the launch.
# Holding it until the launch is ENQUEUED is enough: the allocator reuses blocks in stream order.
_alive = threading.local()


def ptr(t, dtype=None) -> C.c_void_p:
    """Device address of a contiguous CUDA tensor (None -> NULL); the tensor is kept alive until the next check()."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("libmae_hip works on device tensors only (no CPU fallback); move the tensor to cuda")
    if not t.is_contiguous():
        raise RuntimeError("libmae_hip needs contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    keep = getattr(_alive, "tensors", None)
    if keep is None:
        keep = _alive.tensors = []
    keep.append(t)
    return C.c_void_p(t.data_ptr())


def check(rc: int) -> None:
    """Raise with the library's own message when a call returned non-zero."""
    keep = getattr(_alive, "tensors", None)
    if keep:
        keep.clear()
    if rc != 0:
        msg = lib.mae_last_error()
        raise MaeHipError(msg.decode("utf-8", "replace") if msg else f"libmae_hip error {rc}")
