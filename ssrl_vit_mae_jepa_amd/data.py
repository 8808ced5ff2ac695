"""Input step before the hot path: batches of (B, 3, 96, 96) fp32 images in [-1, 1], resident on the GPU.

Mirrors what the reference's ``get_pretrain_dataloaders`` (src/data.py:45-106) hands the step, without torchvision
(not installed): the STL-10 unlabeled split is read straight from its binary file (uint8, column-major images), kept as
uint8 on the device (27.6 KB/image instead of 110.6 KB fp32) and normalised with ToTensor + Normalize(.5,.5) arithmetic
on the GPU; the 94 000 / 6 000 split uses ``random_split``'s permutation with generator seed 73 (src/data.py:74-80).
Augmentation follows what the reference actually does, quirk included (src/data.py:74-81): ``val_subset.dataset`` is the
object ``random_split`` was given, so with data_fraction == 1.0 the assignment overwrites the shared STL10 transform and
BOTH loaders serve un-augmented images, while with data_fraction < 1.0 it lands on the ``Subset`` wrapper, does nothing,
and BOTH loaders keep the training transform RandomResizedCrop(96, scale=(0.8, 1.0)) + RandomHorizontalFlip
(src/data.py:18-21).  That transform is applied here on the GPU to whole batches (crop box sampled like torchvision's
``get_params``, bilinear resampling through an affine grid, flip by sign of the x scale); it cannot be bit-identical to
PIL's fixed-point resize nor share its RNG stream, so it is the same distribution, not the same pixels.
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


def random_resized_crop_params(n: int, size: int, gen: torch.Generator, scale=(0.8, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params for n square images at once: up to 10 draws of (area, log-uniform aspect),
    the first whose box fits is kept, otherwise the whole image (ratio is inside the bounds for a square).  Returns
    integer (top, left, h, w) tensors on the generator's device."""
    dev = gen.device
    area = float(size * size)
    top = torch.zeros(n, dtype=torch.int64, device=dev); left = torch.zeros_like(top)
    h = torch.full_like(top, size); w = torch.full_like(top, size)
    done = torch.zeros(n, dtype=torch.bool, device=dev)
    log_lo, log_hi = float(np.log(ratio[0])), float(np.log(ratio[1]))
    for _ in range(10):
        target = area * (scale[0] + (scale[1] - scale[0]) * torch.rand(n, generator=gen, device=dev))
        ar = torch.exp(log_lo + (log_hi - log_lo) * torch.rand(n, generator=gen, device=dev))
        cw = torch.round(torch.sqrt(target * ar)).to(torch.int64)
        ch = torch.round(torch.sqrt(target / ar)).to(torch.int64)
        ok = (cw > 0) & (cw <= size) & (ch > 0) & (ch <= size) & ~done
        u, v = torch.rand(n, generator=gen, device=dev), torch.rand(n, generator=gen, device=dev)
        t = torch.floor(u * (size - ch + 1).clamp_min(1)).to(torch.int64)   # randint(0, size - h + 1)
        l = torch.floor(v * (size - cw + 1).clamp_min(1)).to(torch.int64)
        top = torch.where(ok, t, top); left = torch.where(ok, l, left)
        h = torch.where(ok, ch, h); w = torch.where(ok, cw, w)
        done |= ok
    return top, left, h, w


def augment_batch(x: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
    """RandomResizedCrop(size, scale=(0.8, 1.0)) + RandomHorizontalFlip() on a (B, C, S, S) float batch, on its device."""
    import torch.nn.functional as F
    B, _, S, _ = x.shape
    top, left, h, w = random_resized_crop_params(B, S, gen)
    flip = torch.rand(B, generator=gen, device=gen.device) < 0.5
    # affine grid in normalised coordinates (align_corners=False): output pixel centres map to the crop box
    sx = w.to(torch.float32) / S; sy = h.to(torch.float32) / S
    cx = (2.0 * left.to(torch.float32) + w.to(torch.float32)) / S - 1.0
    cy = (2.0 * top.to(torch.float32) + h.to(torch.float32)) / S - 1.0
    sx = torch.where(flip, -sx, sx)
    theta = torch.zeros(B, 2, 3, device=x.device)
    theta[:, 0, 0] = sx.to(x.device); theta[:, 0, 2] = cx.to(x.device)
    theta[:, 1, 1] = sy.to(x.device); theta[:, 1, 2] = cy.to(x.device)
    grid = F.affine_grid(theta, list(x.shape), align_corners=False)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=False)


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

    augment = fraction < 1.0  # the reference's quirk: see the module docstring
    aug_gen = torch.Generator(device=device).manual_seed(seed + 7)
    finish = (lambda x: augment_batch(x, aug_gen).contiguous()) if augment else (lambda x: x.contiguous())

    def train_batches(epoch: int) -> Iterator[torch.Tensor]:
        g = torch.Generator().manual_seed(seed + 1000 + epoch)  # DataLoader(shuffle=True): a fresh order per epoch
        order = train_idx[torch.randperm(n_train, generator=g).to(device)]
        for i in range(0, n_train, batch):  # no drop_last, like the reference
            yield finish(fetch(order[i:i + batch]))

    def val_batches() -> Iterator[torch.Tensor]:
        for i in range(0, n_val, batch):
            yield finish(fetch(val_idx[i:i + batch]))

    return train_batches, val_batches
