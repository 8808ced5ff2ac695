"""Input step before the hot path: batches of (B, 3, 96, 96) fp32 images in [-1, 1], resident on the GPU.

Mirrors what the reference's ``get_pretrain_dataloaders`` (src/data.py:45-106) hands the step, without torchvision
(not installed): the STL-10 unlabeled split is read straight from its binary file (uint8, column-major images), kept as
uint8 on the device (27.6 KB/image instead of 110.6 KB fp32) and normalised with ToTensor + Normalize(.5,.5) arithmetic
on the GPU; the 94 000 / 6 000 split uses ``random_split``'s permutation with generator seed 73 (src/data.py:74-80).
Because at data_fraction == 1.0 the reference overwrites the shared dataset's transform with the un-augmented one
(src/data.py:81), its training images are NOT augmented there; this loader reproduces exactly that case.
Without the dataset file (no network here) it serves seeded synthetic uniform[-1,1] images of the same shape.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Iterator, Optional, Tuple

import numpy as np
import torch

STL10_UNLABELED = Path("data") / "stl10_binary" / "unlabeled_X.bin"


def _load_stl10_unlabeled(path: Path, fraction: float) -> torch.Tensor:
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    n = raw.size // (3 * 96 * 96)
    n_use = int(n * fraction) if fraction < 1.0 else n
    imgs = np.asarray(raw[: n_use * 3 * 96 * 96]).reshape(n_use, 3, 96, 96)
    return torch.from_numpy(np.ascontiguousarray(imgs.transpose(0, 1, 3, 2)))  # file stores each plane column-major


def normalize_u8(x: torch.Tensor) -> torch.Tensor:
    """ToTensor (x/255) then Normalize(mean .5, std .5): (x/255 - .5)/.5, fp32."""
    return (x.to(torch.float32) / 255.0 - 0.5) / 0.5


def get_pretrain_batches(cfg: dict, device: torch.device, synthetic_images: Optional[int] = None,
                         seed: int = 73) -> Tuple[Callable[[int], Iterator[torch.Tensor]], Callable[[], Iterator[torch.Tensor]]]:
    pre = cfg["pretrain"]
    batch = int(pre.get("batch_size", 512))
    val_split = float(pre.get("val_split", 0.1))
    fraction = float(pre.get("data_fraction", 1.0))
    seed = int(cfg.get("seed", seed))
    if synthetic_images is None and STL10_UNLABELED.exists():
        data = _load_stl10_unlabeled(STL10_UNLABELED, fraction).to(device)  # uint8 on device
        fetch = lambda idx: normalize_u8(data[idx])  # noqa: E731
        n_total = data.shape[0]
    else:
        n_total = int(synthetic_images or 4 * batch)
        g = torch.Generator(device=device).manual_seed(seed)
        data = (torch.rand(n_total, 3, 96, 96, device=device, generator=g) * 2 - 1)
        fetch = lambda idx: data[idx]  # noqa: E731
    n_val = int(n_total * val_split)
    n_train = n_total - n_val
    perm = torch.randperm(n_total, generator=torch.Generator().manual_seed(seed))  # random_split's permutation
    train_idx, val_idx = perm[:n_train].to(device), perm[n_train:].to(device)
    print(f"Unlabeled pretrain split: {n_train} train, {n_val} val ({val_split * 100:.1f}% validation)")

    def train_batches(epoch: int) -> Iterator[torch.Tensor]:
        g = torch.Generator().manual_seed(seed + 1000 + epoch)  # DataLoader(shuffle=True): a fresh order per epoch
        order = train_idx[torch.randperm(n_train, generator=g).to(device)]
        for i in range(0, n_train, batch):  # no drop_last, like the reference
            yield fetch(order[i:i + batch]).contiguous()

    def val_batches() -> Iterator[torch.Tensor]:
        for i in range(0, n_val, batch):
            yield fetch(val_idx[i:i + batch]).contiguous()

    return train_batches, val_batches
