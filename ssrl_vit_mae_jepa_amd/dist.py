"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm), gloo for CPU tests.

The MAE pretrain step shards by samples only (no cross-sample op on the path, loss is a mean), so the single exchange is
the gradient sum: every rank runs the native loss+grads with grad_scale = 1/world on its rows of the global batch, the flat
fp32 gradient arena is all-reduced ONCE, then every rank applies the same global-norm clip + AdamW (replicated optimizer).
Rank r owns rows [r*B/W, (r+1)*B/W) of the global image batch AND of the global noise tensor, so W-GPU masks equal the
1-GPU run.  Nothing like this exists in the reference (devices=1, scripts/training/pretrain_mae.py:118).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run sets them)."""
    rank, _local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of an n-row global batch that rank ``rank`` owns: contiguous, as even as possible (n need not divide by
    the world size: the shares then differ by one row and the step weights each rank's mean by its share)."""
    return n * rank // world, n * (rank + 1) // world


def shard_rows(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous row shard of a global-batch tensor (see ``shard_bounds``)."""
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi].contiguous()


def global_noise(global_batch: int, seq_len: int, seed: int, step: int, device: torch.device) -> torch.Tensor:
    """The mask noise of the whole global batch, identical on every rank (same generator seed)."""
    g = torch.Generator(device=device).manual_seed(seed + step)
    return torch.rand(global_batch, seq_len, generator=g, device=device)


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """One collective over the whole flat gradient arena (a single bucket: 89.5 MB fp32 for ViT-S/8)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat
