"""Input step before the hot path: batches of (B, 3, 96, 96) images, uint8 as the dataset stores them, on the GPU.

Mirrors what the reference's ``get_pretrain_dataloaders`` (src/data.py:45-106) hands the step, without torchvision
(not installed): the STL-10 unlabeled split is read straight from its binary file (uint8, column-major images) and kept
uint8 all the way INTO the engine (27.6 KB/image instead of 110.6 KB fp32): ToTensor + Normalize(.5,.5)
(src/data.py:22-23) is applied by the kernels that read pixels (csrc/k_pixels_u8.hip), bit-identical to the torch
expression ``normalize_u8``.  The dataset lives either on the device (``engine.data_on_device: true``, the default:
2.6 GB for STL-10 unlabeled) or in pinned host memory, streamed by ``PinnedBatchStream``: index gather on the host into
a pinned staging buffer, H2D copy on a side stream into one of two device buffers, so batch i+1 crosses PCIe while
batch i trains (55 MB per 2000-image batch: 0.9 ms at 63 GB/s against a 25 ms step).
The 94 000 / 6 000 split uses ``random_split``'s permutation with generator seed 73 (src/data.py:74-80).
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


def augment_batch(x: torch.Tensor, gen: Optional[torch.Generator] = None, params=None) -> torch.Tensor:
    """RandomResizedCrop(size, scale=(0.8, 1.0)) + RandomHorizontalFlip() on a (B, C, S, S) float batch, on its device
    (``params``: the output of ``draw_augment_params`` sliced to the batch's rows, instead of drawing them here)."""
    import torch.nn.functional as F
    B, _, S, _ = x.shape
    if params is None:
        top, left, h, w = random_resized_crop_params(B, S, gen)
        flip = torch.rand(B, generator=gen, device=gen.device) < 0.5
    else:
        top, left, h, w, flip = params
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


def draw_augment_params(n: int, size: int, gen: torch.Generator):
    """Crop boxes and flips of n images: (top, left, h, w, flip) integer tensors on the generator's device.  Drawn for the
    GLOBAL batch on every rank (five numbers per image) and sliced to the rank's rows, so that a data-parallel run augments
    every image exactly as the one-GPU run does."""
    top, left, h, w = random_resized_crop_params(n, size, gen)
    flip = torch.rand(n, generator=gen, device=gen.device) < 0.5
    return top, left, h, w, flip


def augment_batch_u8(x: torch.Tensor, gen: Optional[torch.Generator] = None, params=None) -> torch.Tensor:
    """The same augmentation on a uint8 device batch, by the HIP kernel behind ``mae_augment_crop_flip_u8`` (uint8 in, uint8
    out, as the reference's PIL transforms run before ToTensor): the crop boxes and flips are drawn here with the rule of
    ``random_resized_crop_params`` (or passed in as ``params``, already sliced to the batch's rows), the resampling runs in
    libmae_hip.so.  No torch fallback: a missing extension fails loudly."""
    from ._lib import check, lib, ptr
    from .mae import _stream
    if x.dtype != torch.uint8 or not x.is_cuda or x.dim() != 4 or x.shape[2] != x.shape[3]:
        raise ValueError(f"augment_batch_u8 needs a square (B, C, S, S) uint8 CUDA batch, got {tuple(x.shape)} {x.dtype} on {x.device}")
    B, C, S, _ = x.shape
    top, left, h, w, flip = draw_augment_params(B, S, gen) if params is None else params
    params = torch.stack([top, left, h, w, flip.to(torch.int64)], dim=1).to(device=x.device, dtype=torch.int32).contiguous()
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib.mae_augment_crop_flip_u8(ptr(x), ptr(params), B, C, S, ptr(out), _stream(x.device)))
    return out


class PinnedBatchStream:
    """uint8 dataset in pinned host memory -> device batches through two device buffers (double buffering).

    ``batches(order)`` yields ``data[order[i:i+batch]]`` as device tensors.  While the consumer works on buffer s, the next
    batch is gathered on the host into pinned staging buffer 1-s and copied on ``copy_stream``; the consumer's stream waits
    for the copy's event (never the host), and a buffer is rewritten only after the kernels that read it have finished
    (an event recorded on the consumer's stream when the next batch is requested)."""

    def __init__(self, data_u8: torch.Tensor, batch: int, device: torch.device):
        if data_u8.dtype != torch.uint8 or data_u8.is_cuda:
            raise ValueError("PinnedBatchStream takes a uint8 host tensor")
        self.device, self.batch = device, int(batch)
        self.data = data_u8.contiguous()
        shape = (self.batch,) + tuple(data_u8.shape[1:])
        self.stage = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        self.staged = [torch.cuda.Event() for _ in range(2)]   # the H2D copy out of a staging buffer has finished
        self._used = [False, False]

    def _submit(self, slot: int, idx: torch.Tensor) -> int:
        n = idx.numel()
        if self._used[slot]:
            self.staged[slot].synchronize()  # host: the previous copy out of this staging buffer is done
        torch.index_select(self.data, 0, idx, out=self.stage[slot][:n])
        with torch.cuda.stream(self.copy_stream):
            if self._used[slot]:
                self.copy_stream.wait_event(self.free[slot])  # device: the consumer finished with this buffer
            self.dev[slot][:n].copy_(self.stage[slot][:n], non_blocking=True)
            self.staged[slot].record(self.copy_stream)
            self.ready[slot].record(self.copy_stream)
        self._used[slot] = True
        return n

    def batches(self, order: torch.Tensor) -> Iterator[torch.Tensor]:
        order = order.to("cpu", torch.int64)
        return self.batches_of([order[i:i + self.batch] for i in range(0, order.numel(), self.batch)])

    def batches_of(self, chunks) -> Iterator[torch.Tensor]:
        """The same pipeline over explicit index chunks (each at most ``batch`` long; a chunk may be empty: a data-parallel rank's
        share of a short last batch)."""
        chunks = [c.to("cpu", torch.int64) for c in chunks]
        if not chunks:
            return
        n_next = self._submit(0, chunks[0])
        for i in range(len(chunks)):
            slot, n = i & 1, n_next
            if i + 1 < len(chunks):
                n_next = self._submit(1 - slot, chunks[i + 1])  # overlaps whatever the consumer enqueued for batch i-1
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.ready[slot])
            yield self.dev[slot][:n]
            self.free[slot].record(torch.cuda.current_stream(self.device))  # everything the consumer enqueued on it so far


class ShardedBatch:
    """What the training loops get per step under data parallelism: THIS rank's rows of a global batch (fetched / copied /
    augmented for these rows only), the global row count and the rank's row range inside it."""
    __slots__ = ("images", "global_rows", "lo", "hi")

    def __init__(self, images: torch.Tensor, global_rows: int, lo: int, hi: int):
        self.images, self.global_rows, self.lo, self.hi = images, int(global_rows), int(lo), int(hi)


def get_pretrain_batches(cfg: dict, device: torch.device, synthetic_images: Optional[int] = None, seed: int = 73, rank: int = 0,
                         world: int = 1) -> Tuple[Callable[[int], Iterator["ShardedBatch"]], Callable[[], Iterator["ShardedBatch"]]]:
    """(train_batches(epoch), val_batches()): iterators of ``ShardedBatch``.  The order of the global batches and their
    augmentation parameters do not depend on the world size; rank r fetches rows [gb r / W, gb (r + 1) / W) of each global
    batch only (the last batch of an epoch may be ragged: no batch is dropped, as in the reference, src/data.py:86-92).
    ``train_batches.steps_per_epoch`` = global batches per epoch."""
    pre = cfg["pretrain"]
    batch = int(pre.get("batch_size", 512))
    val_split = float(pre.get("val_split", 0.1))
    fraction = float(pre.get("data_fraction", 1.0))
    seed = int(cfg.get("seed", seed))
    on_device = bool(cfg.get("engine", {}).get("data_on_device", True))
    patch = int(cfg.get("model", {}).get("general", {}).get("patch_size", 8))
    img_size = int(cfg.get("model", {}).get("general", {}).get("image_size", 96))
    # the uint8 pixel kernels take patch sizes that are multiples of 4 (csrc/k_pixels_u8.hip); other geometries (the reference
    # class's default patch_size 6, ViT-L/14) are served normalised fp32 batches, which every engine kernel accepts
    keep_u8 = patch % 4 == 0 and img_size // max(1, patch) <= 64
    stream: Optional[PinnedBatchStream] = None
    if synthetic_images is None and STL10_UNLABELED.exists():
        host = _load_stl10_unlabeled(STL10_UNLABELED, fraction)
        n_total = host.shape[0]
        if on_device:
            data = host.to(device)  # uint8 on device; the engine normalises while it reads
            fetch = lambda idx: data[idx]  # noqa: E731
        else:
            stream = PinnedBatchStream(host.pin_memory(), (batch + world - 1) // world + 1, device)
            fetch = None
    else:
        n_total = int(synthetic_images or 4 * batch)
        g = torch.Generator(device=device).manual_seed(seed)
        data = (torch.rand(n_total, 3, 96, 96, device=device, generator=g) * 2 - 1)
        fetch = lambda idx: data[idx]  # noqa: E731
    n_val = int(n_total * val_split)
    n_train = n_total - n_val
    perm = torch.randperm(n_total, generator=torch.Generator().manual_seed(seed))  # random_split's permutation
    idx_dev = torch.device("cpu") if stream is not None else device
    train_idx, val_idx = perm[:n_train].to(idx_dev), perm[n_train:].to(idx_dev)
    if rank == 0:
        print(f"Unlabeled pretrain split: {n_train} train, {n_val} val ({val_split * 100:.1f}% validation)")

    augment = fraction < 1.0  # the reference's quirk: see the module docstring
    aug_gen = torch.Generator(device=device).manual_seed(seed + 7)

    def finish(x: torch.Tensor, lo: int, hi: int, gb: int) -> torch.Tensor:
        params = None
        if hi == lo:   # this rank holds no row of a short last batch (the generator still advances below, in step with the other ranks)
            if augment:
                draw_augment_params(gb, x.shape[-1], aug_gen)
            return (x if (x.dtype == torch.uint8 and keep_u8) else normalize_u8(x) if x.dtype == torch.uint8 else x).contiguous()
        if augment:  # parameters of the whole global batch (cheap), this rank's rows of them
            params = tuple(t[lo:hi] for t in draw_augment_params(gb, x.shape[-1], aug_gen))
        if x.dtype == torch.uint8 and x.is_cuda and keep_u8:
            return augment_batch_u8(x, params=params) if augment else x.contiguous()  # uint8 stays uint8: normalised inside the engine
        x = normalize_u8(x) if x.dtype == torch.uint8 else x
        return (augment_batch(x, params=params) if augment else x).contiguous()

    def shards(order: torch.Tensor):
        from .dist import shard_bounds
        for i in range(0, order.numel(), batch):  # no drop_last, like the reference
            idx = order[i:i + batch]
            lo, hi = shard_bounds(idx.numel(), rank, world)
            yield idx[lo:hi], idx.numel(), lo, hi

    def serve(order: torch.Tensor) -> Iterator[ShardedBatch]:
        if stream is not None:
            plan = list(shards(order))
            for (idx, gb, lo, hi), x in zip(plan, stream.batches_of([p[0] for p in plan])):
                yield ShardedBatch(finish(x, lo, hi, gb), gb, lo, hi)
        else:
            for idx, gb, lo, hi in shards(order):
                yield ShardedBatch(finish(fetch(idx), lo, hi, gb), gb, lo, hi)

    def train_batches(epoch: int) -> Iterator[ShardedBatch]:
        g = torch.Generator().manual_seed(seed + 1000 + epoch)  # DataLoader(shuffle=True): a fresh order per epoch
        return serve(train_idx[torch.randperm(n_train, generator=g).to(train_idx.device)])

    def val_batches() -> Iterator[ShardedBatch]:
        return serve(val_idx)

    train_batches.steps_per_epoch = (n_train + batch - 1) // batch
    train_batches.n_train = n_train
    return train_batches, val_batches
