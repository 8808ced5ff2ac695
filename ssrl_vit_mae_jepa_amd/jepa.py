"""I-JEPA on the MI355X engine (BASELINE.json configs[2] and [4]).

THE REFERENCE HAS NO I-JEPA CODE: "JEPA" appears in its README title and project name only (README.md:1,9,
pyproject.toml:2).  This module is therefore not a mirror of a reference file; it implements the specification of
DESIGN.md, section "I-JEPA" (I-JEPA paper: context encoder, EMA target encoder, narrow predictor, latent regression
loss, multi-block masks), with the reference's ViT pieces and configuration style, on the same engine:

  * ``IJEPA.net`` is a ``MaskedAutoencoder`` parameter table created with ``pred_dim = embed_dim``: its encoder is the
    context encoder, its "decoder" tensors are the predictor (``decoder.decoder_embed`` = predictor embed,
    ``decoder.mask_token``, ``decoder.decoder_pos_embed``, ``decoder.decoder_blocks.*``, ``decoder.decoder_norm``,
    ``decoder.decoder_pred`` : Dp -> D).  ``encoder.vit.cls_token`` / ``encoder.mask_token`` exist and are unused.
  * ``IJEPA.target_arena`` is the EMA target encoder: a second flat fp32 arena of the same layout.
  * one native call runs target forward, context forward, predictor, loss and backward
    (``mae_engine_jepa_loss_and_grads``); AdamW and the EMA update are one sweep (``mae_engine_optimizer_step_ema``).
  * ``sample_block_masks`` is integer-exact host logic (same ids as oracle/jepa_oracle.py::sample_masks).
There is no CPU fallback: the step needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from ._lib import check, lib
from .mae import MaskedAutoencoder, _ptr, _stream
from .training import MAEPretrainModule, lr_lambda


def _block_size(grid: int, scale, aspect, u_scale: float, u_aspect: float) -> Tuple[int, int]:
    s = scale[0] + u_scale * (scale[1] - scale[0])
    max_keep = int(grid * grid * s)
    a = aspect[0] + u_aspect * (aspect[1] - aspect[0])
    h = int(round(math.sqrt(max_keep * a)))
    w = int(round(math.sqrt(max_keep / a)))
    return max(1, min(h, grid - 1)), max(1, min(w, grid - 1))


def sample_block_masks(batch: int, grid: int, gen: torch.Generator, num_target_blocks: int = 4, target_scale=(0.15, 0.2),
                       target_aspect=(0.75, 1.5), context_scale=(0.85, 1.0)) -> Tuple[torch.Tensor, torch.Tensor]:
    """Multi-block masks of I-JEPA: one target block size and one context block size per batch, per image
    ``num_target_blocks`` target positions and one context position; context = context block minus the image's target
    blocks, truncated to the batch minimum.  Returns ``idx_context (B, k)`` and ``idx_target (B, nblk, m)``: int64 token
    ids in 1..N, ascending inside the context and row-major inside a block.  CPU tensors (host logic, as the public I-JEPA
    code samples masks in its collator); vectorised, integer-exact."""
    g, nb = grid, num_target_blocks
    u = torch.rand(4, generator=gen, dtype=torch.float64).tolist()
    th, tw = _block_size(g, target_scale, target_aspect, u[0], u[1])
    ch, cw = _block_size(g, context_scale, (1.0, 1.0), u[2], u[3])
    pos = torch.rand(batch, nb + 1, 2, generator=gen, dtype=torch.float64)
    t_top = (pos[:, :nb, 0] * (g - th + 1)).to(torch.int64)                     # (B, nb): int() truncation of a non-negative double
    t_left = (pos[:, :nb, 1] * (g - tw + 1)).to(torch.int64)
    rr, cc = torch.arange(th).view(1, 1, th, 1), torch.arange(tw).view(1, 1, 1, tw)
    tgt = (1 + (t_top.view(batch, nb, 1, 1) + rr) * g + (t_left.view(batch, nb, 1, 1) + cc)).reshape(batch, nb, th * tw)
    taken = torch.zeros(batch, g * g + 1, dtype=torch.bool)
    taken.scatter_(1, tgt.reshape(batch, -1), True)
    c_top = (pos[:, nb, 0] * (g - ch + 1)).to(torch.int64)
    c_left = (pos[:, nb, 1] * (g - cw + 1)).to(torch.int64)
    rr, cc = torch.arange(ch).view(1, ch, 1), torch.arange(cw).view(1, 1, cw)
    cblock = (1 + (c_top.view(batch, 1, 1) + rr) * g + (c_left.view(batch, 1, 1) + cc)).reshape(batch, ch * cw)
    inblock = torch.zeros(batch, g * g + 1, dtype=torch.bool)
    inblock.scatter_(1, cblock, True)
    ctx_map = inblock & ~taken
    counts = ctx_map.sum(1)
    for b in torch.nonzero(counts == 0).flatten().tolist():  # degenerate draw: one patch outside the targets (lowest id), or patch 1
        free = torch.nonzero(~taken[b, 1:]).flatten()
        ctx_map[b, int(free[0]) + 1 if free.numel() else 1] = True
    counts = ctx_map.sum(1)
    k = int(counts.min())
    order = torch.argsort((~ctx_map).to(torch.int8), dim=1, stable=True)        # the True columns first, ascending id
    return order[:, :k].contiguous(), tgt.contiguous()


class IJEPA(nn.Module):
    """Context encoder + predictor (trainable, ``net``) and EMA target encoder (``target_arena``)."""

    def __init__(self, general_cfg: Dict[str, Any], encoder_cfg: Dict[str, Any], predictor_cfg: Dict[str, Any]):
        super().__init__()
        D = int(encoder_cfg.get("embed_dim", 384))
        self.loss_kind = str(general_cfg.get("loss", "mse"))
        if self.loss_kind not in ("mse", "smooth_l1"):
            raise ValueError(f"loss must be 'mse' or 'smooth_l1', got {self.loss_kind!r}")
        self.num_target_blocks = int(general_cfg.get("num_target_blocks", 4))
        self.target_scale = tuple(general_cfg.get("target_scale", (0.15, 0.2)))
        self.target_aspect = tuple(general_cfg.get("target_aspect", (0.75, 1.5)))
        self.context_scale = tuple(general_cfg.get("context_scale", (0.85, 1.0)))
        general = {k: v for k, v in general_cfg.items() if k in ("image_size", "patch_size", "in_chans", "engine_precision")}
        general.setdefault("patch_size", 8)
        self.net = MaskedAutoencoder(dict(general, pred_dim=D), encoder_cfg, dict(
            decoder_embed_dim=int(predictor_cfg.get("pred_embed_dim", D // 2)), decoder_depth=int(predictor_cfg.get("pred_depth", 6)),
            decoder_num_heads=int(predictor_cfg.get("pred_num_heads", 6))))
        self.embed_dim = D
        self.grid = self.net.image_size // self.net.patch_size
        self.num_patches = self.grid * self.grid
        self.register_buffer("target_arena", self.net.flat_params.detach().clone())  # target starts as a copy of the context encoder
        self._target_wcache: Optional[torch.Tensor] = None
        self._target_version = -1
        self._workspace: Optional[torch.Tensor] = None

    # ---- what the data-parallel step and the optimizer need from "the model" ---------------------------------------------
    @property
    def engine(self):
        return self.net.engine

    @property
    def grad_buffer(self) -> torch.Tensor:
        return self.net.grad_buffer

    @property
    def flat_grads(self) -> torch.Tensor:
        return self.net.flat_grads

    @property
    def flat_params(self) -> torch.Tensor:
        return self.net.flat_params

    def grad_ready_points(self) -> List[int]:
        return self.net.grad_ready_points()

    def _require_cuda(self) -> torch.device:
        return self.net._require_cuda()

    def _weights(self):
        return self.net._weights()

    def mark_weights_fresh(self) -> None:
        self.net.mark_weights_fresh()

    def _scratch_f32(self) -> torch.Tensor:
        return self.net._scratch_f32()

    def named_flat_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.net.named_flat_views(flat)

    @torch.no_grad()
    def reset_target(self) -> None:
        """target encoder <- context encoder (after loading context weights from a checkpoint)."""
        self.target_arena.copy_(self.net.flat_params)

    def target_state_dict(self) -> Dict[str, torch.Tensor]:
        """The EMA encoder under the names a ViT consumer expects (``encoder.vit.*``)."""
        return {name: self.target_arena[off:off + numel].view(shape) for (name, off, numel, shape, _f) in self.net.engine.table
                if name.startswith("encoder.vit.")}

    def _target_weights(self) -> Optional[torch.Tensor]:
        dev = self._require_cuda()
        if self.target_arena.device != dev:
            raise RuntimeError("IJEPA: move the whole module with .to(device)")
        if self._target_wcache is None or self._target_wcache.device != dev:
            self._target_wcache = torch.zeros(self.net.engine.wcache_bytes, dtype=torch.uint8, device=dev)
            self._target_version = -1
        if self.net.engine.act == _lib.MAE_BF16 and self._target_version != self.target_arena._version:
            check(lib.mae_engine_refresh_weights(self.net.engine.handle, _ptr(self.target_arena), _ptr(self._target_wcache), _stream(dev)))
            self._target_version = self.target_arena._version
        return self._target_wcache

    def _ws(self, B: int, k: int, nblk: int, m: int) -> torch.Tensor:
        need = lib.mae_engine_jepa_workspace_bytes(self.net.engine.handle, B, k, nblk, m)
        if need < 0:
            raise ValueError(f"bad I-JEPA token counts (batch {B}, context {k}, {nblk} blocks of {m})")
        dev = self._require_cuda()
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            # the context length changes from step to step (batch minimum of the sampled masks): grow with headroom so that
            # a slightly longer context a few steps later does not reallocate again
            grown = self._workspace is not None
            self._workspace = None
            self._workspace = torch.empty(int(need * 1.25) if grown else need, dtype=torch.uint8, device=dev)
        return self._workspace

    def reserve_workspace(self, batch: int, max_context: Optional[int] = None, num_blocks: Optional[int] = None, max_block: Optional[int] = None) -> int:
        """Allocate the step workspace for the longest context expected (default: every patch), so that no step of a run
        pays for a reallocation when the sampled context grows.  Returns the bytes held."""
        k = int(max_context) if max_context is not None else self.num_patches
        nblk = int(num_blocks) if num_blocks is not None else self.num_target_blocks
        m = int(max_block) if max_block is not None else self.num_patches
        need = max(lib.mae_engine_jepa_workspace_bytes(self.net.engine.handle, batch, kk, nblk, m) for kk in {k, max(1, k // 2)})
        if need < 0:
            raise ValueError(f"bad I-JEPA token counts (batch {batch}, context {k}, {nblk} blocks of {m})")
        dev = self._require_cuda()
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        return int(self._workspace.numel())

    # ---- masks ---------------------------------------------------------------------------------------------------------------
    def sample_masks(self, batch: int, gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
        return sample_block_masks(batch, self.grid, gen, self.num_target_blocks, self.target_scale, self.target_aspect, self.context_scale)

    def _check_masks(self, images, idx_context, idx_target):
        dev = self._require_cuda()
        B = images.shape[0]
        if idx_context.dim() != 2 or idx_target.dim() != 3 or idx_context.shape[0] != B or idx_target.shape[0] != B:
            raise ValueError(f"idx_context must be (B, k) and idx_target (B, nblk, m); got {tuple(idx_context.shape)}, {tuple(idx_target.shape)}")
        out = []
        for idx in (idx_context, idx_target):
            if idx.is_cuda:  # already on the device: no host round trip in the step; ids are clamped into the patch range
                idx = idx.to(device=dev, dtype=torch.int64).clamp(1, self.num_patches).contiguous()
            else:            # host tensors (what sample_block_masks returns): strict check, then upload
                lo, hi = int(idx.min()), int(idx.max())
                if lo < 1 or hi > self.num_patches:
                    raise IndexError(f"I-JEPA token ids must be patch tokens 1..{self.num_patches} (got {lo}..{hi})")
                idx = idx.to(device=dev, dtype=torch.int64, non_blocking=True).contiguous()
            out.append(idx)
        return out[0], out[1]

    # ---- the step's native call ----------------------------------------------------------------------------------------------
    def loss_and_grads(self, images: torch.Tensor, idx_context: torch.Tensor, idx_target: torch.Tensor, grad_scale: float = 1.0,
                       ready_events=None, loss_out: Optional[torch.Tensor] = None, return_aux: bool = False):
        """target forward (EMA weights, no gradient) + context forward + predictor + latent loss + backward in one native
        call.  Gradients land in ``flat_grads``; returns the device scalar loss (and, with ``return_aux``, the target rows
        ``h`` and the predictions, both (B, nblk, m, D) fp32)."""
        net = self.net
        dev = self._require_cuda()
        images = net._check_images(images)
        idx_context, idx_target = self._check_masks(images, idx_context, idx_target)
        B, k = idx_context.shape
        nblk, m = idx_target.shape[1], idx_target.shape[2]
        ws = self._ws(B, k, nblk, m)
        net._gen_enc += 1; net._gen_dec += 1  # the net's own workspace is untouched, but its saved activations are not ours
        loss = torch.empty(1, dtype=torch.float32, device=dev) if loss_out is None else loss_out
        h = pred = None
        if return_aux:
            h = torch.empty(B, nblk, m, self.embed_dim, dtype=torch.float32, device=dev)
            pred = torch.empty_like(h)
        evs, n_ev = None, 0
        if ready_events is not None:
            n_ev = len(ready_events)
            evs = (C.c_void_p * n_ev)(*[C.c_void_p(ev.cuda_event) if ev is not None else C.c_void_p(0) for ev in ready_events])
        kind = _lib.LOSS_SMOOTH_L1 if self.loss_kind == "smooth_l1" else _lib.LOSS_MSE
        check(lib.mae_engine_jepa_loss_and_grads(
            net.engine.handle, _ptr(net.flat_params), _ptr(net._weights()), _ptr(self.target_arena), _ptr(self._target_weights()),
            _ptr(images), net._img_dt(images), _ptr(idx_context), _ptr(idx_target), B, k, nblk, m, kind, float(grad_scale),
            _ptr(ws), ws.numel(), _ptr(net.flat_grads), _ptr(loss, torch.float32), _ptr(h), _ptr(pred), evs, n_ev, _stream(dev)))
        return (loss, h, pred) if return_aux else loss

    @torch.no_grad()
    def target_features(self, images: torch.Tensor, idx_target: torch.Tensor) -> torch.Tensor:
        """layer_norm(target_encoder(images))[target blocks]: (B, nblk, m, D) fp32 (no gradient, nothing else computed)."""
        net = self.net
        dev = self._require_cuda()
        images = net._check_images(images)
        ctx = torch.ones(images.shape[0], 1, dtype=torch.int64, device=dev)
        ctx, idx_target = self._check_masks(images, ctx, idx_target)
        B, nblk, m = idx_target.shape
        ws = self._ws(B, 1, nblk, m)
        h = torch.empty(B, nblk, m, self.embed_dim, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib.mae_engine_jepa_loss_and_grads(
            net.engine.handle, _ptr(net.flat_params), _ptr(net._weights()), _ptr(self.target_arena), _ptr(self._target_weights()),
            _ptr(images), net._img_dt(images), _ptr(ctx), _ptr(idx_target), B, 1, nblk, m, _lib.LOSS_MSE, 1.0, _ptr(ws), ws.numel(),
            None, _ptr(loss), _ptr(h), None, None, 0, _stream(dev)))
        return h


class IJEPAPretrainModule(MAEPretrainModule):
    """The MAE pretrain module's optimiser, schedules, checkpoint layout and data-parallel gradient exchange, around the
    I-JEPA step.  Differences, all from the I-JEPA recipe: no gradient clipping, an EMA momentum schedule (linear
    ``ema_start`` -> ``ema_end`` over ``total_epochs * steps_per_epoch`` steps), masks sampled per step on the host."""

    def __init__(self, model_cfg: Dict[str, Any], training_cfg: Dict[str, Any]):
        nn.Module.__init__(self)
        self.hparams = {"model_cfg": model_cfg, "training_cfg": training_cfg}
        self.model = IJEPA(model_cfg["general"], model_cfg["encoder"], model_cfg["predictor"])
        self._init_training_state(training_cfg)
        self.gradient_clip_val = float("inf")
        self.ema_start = float(training_cfg.get("ema_start", 0.996))
        self.ema_end = float(training_cfg.get("ema_end", 1.0))
        self.steps_per_epoch = int(training_cfg.get("steps_per_epoch", 1000))
        self.mask_generator = torch.Generator().manual_seed(int(training_cfg.get("mask_seed", 73)))

    def on_train_epoch_start(self):
        self.log("ema_momentum", self.ema_momentum())

    def ema_momentum(self) -> float:
        """Linear ema_start -> ema_end over total_epochs * steps_per_epoch optimizer steps.  ``steps_per_epoch`` must be the
        number of steps the training loop really runs per epoch: scripts/training/pretrain_ijepa.py sets it from its loader
        (the config value, default 1000, is only a fallback: STL-10 unlabeled at batch 2000 has 47 steps per epoch, and a
        schedule stretched over 1000 would end the run at 0.9962 instead of 1.0)."""
        total = max(1, self.total_epochs * self.steps_per_epoch)
        return self.ema_start + (self.ema_end - self.ema_start) * min(self.global_step, total) / total

    def checkpoint_dict(self, epoch: int, weights_only: bool = False, extra=None):
        """The MAE module's checkpoint plus the state of the host mask sampler (a uint8 tensor: the safe loader reads it), so
        that a resumed run continues the mask sequence instead of replaying it from step 0."""
        ckpt = super().checkpoint_dict(epoch, weights_only=weights_only, extra=extra)
        ckpt["mask_generator_state"] = self.mask_generator.get_state().clone()
        ckpt["steps_per_epoch"] = int(self.steps_per_epoch)
        return ckpt

    def load_checkpoint_dict(self, ckpt) -> int:
        nxt = super().load_checkpoint_dict(ckpt)
        st = ckpt.get("mask_generator_state")
        if st is not None:
            self.mask_generator.set_state(st.to(torch.uint8).cpu())
        return nxt

    def forward(self, images: torch.Tensor):
        raise RuntimeError("IJEPAPretrainModule has no standalone forward: use fused_training_step(images) or model.loss_and_grads(...)")

    def optimizer_step(self, lr: Optional[float] = None, momentum: Optional[float] = None) -> torch.Tensor:
        """AdamW (unclipped) over the context encoder + predictor with the EMA update of the target encoder in the same sweep."""
        model = self.model
        dev = model._require_cuda()
        m, v, stats = self._opt_state()
        self._opt_steps += 1
        mom = self.ema_momentum() if momentum is None else momentum
        check(lib.mae_engine_optimizer_step_ema(
            model.engine.handle, _ptr(model.flat_params), _ptr(model.flat_grads), _ptr(m), _ptr(v), _ptr(model._weights()),
            float(self.current_lr() if lr is None else lr), 0.9, 0.999, 1e-8, float(self.weight_decay), float(self.gradient_clip_val),
            self._opt_steps, _ptr(stats), _ptr(model._scratch_f32()), _ptr(model.target_arena), _ptr(model._target_weights()), float(mom),
            _stream(dev)))
        model.mark_weights_fresh()
        model._target_version = model.target_arena._version  # the native sweep refreshed the target's operand copy itself
        return stats

    def fused_training_step(self, images: torch.Tensor, idx_context: Optional[torch.Tensor] = None, idx_target: Optional[torch.Tensor] = None,
                            lr: Optional[float] = None, momentum: Optional[float] = None, process_group=None,
                            global_rows: Optional[int] = None) -> torch.Tensor:
        """``global_rows``: rows of the global batch this step belongs to (see MAEPretrainModule.fused_training_step): the rank's
        mean loss and gradient are weighted by images.shape[0] / global_rows; a 0-row shard contributes zeros."""
        model = self.model
        rows = int(images.shape[0])
        weight = None if global_rows is None else rows / float(global_rows)
        if rows == 0:
            if global_rows is None:
                raise ValueError("an empty shard needs global_rows")
            compute = None
        else:
            if idx_context is None or idx_target is None:
                idx_context, idx_target = model.sample_masks(rows, self.mask_generator)
            compute = lambda scale, events, out: model.loss_and_grads(images, idx_context, idx_target, grad_scale=scale, ready_events=events, loss_out=out)  # noqa: E731
        loss = self._exchanged_loss_and_grads(compute, process_group, weight)
        self.optimizer_step(lr, momentum)
        self.global_step += 1
        self.log("train_loss", loss)
        return loss
