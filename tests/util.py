"""Helpers shared by the GPU parity tests: call the C ABI on torch tensors."""
import ctypes as C

import torch

from ssrl_vit_mae_jepa_amd import _lib
from ssrl_vit_mae_jepa_amd._lib import check, lib
from ssrl_vit_mae_jepa_amd.mae import _ptr, _stream

F32, BF16 = _lib.MAE_F32, _lib.MAE_BF16
TDT = {F32: torch.float32, BF16: torch.bfloat16}


def stream(dev):
    return _stream(dev)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_err(a, b) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


# Device copies made inline in a call expression must outlive the launch: a temporary freed between two `_ptr(...)`
# arguments can be handed out again by the caching allocator for the next argument.
_KEEP = []


def dv(t: torch.Tensor, dtype=None) -> torch.Tensor:
    d = t.to("cuda") if dtype is None else t.to("cuda").to(dtype)
    _KEEP.append(d)
    return d


def release_device_copies() -> None:
    torch.cuda.synchronize()
    _KEEP.clear()
