"""MAEPretrainModule: the reference's ``src/training/mae.py`` (lines 14-83) without Lightning, plus the fused
data-parallel step (native loss+grads -> one RCCL all-reduce -> native clip+AdamW).

Step semantics kept from the reference:
  * loss = MSELoss(preds, targets)                                          src/training/mae.py:40,48
  * AdamW(self.parameters(), lr = base*batch/256, weight_decay) one group   src/training/mae.py:59-65
  * per-EPOCH LambdaLR min((e+1)/warmup, 1) * 0.5(1+cos(pi e/total))        src/training/mae.py:67-76
  * per-epoch linear mask-ratio ramp written to model.mask_ratio            src/training/mae.py:78-83
  * clip_grad_norm_(1.0, L2) before the update                              scripts/training/pretrain_mae.py:124-125
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, Optional

import torch
from torch import nn

from ._lib import check, lib
from .mae import MaskedAutoencoder, _ptr, _stream


def lr_lambda(epoch: int, warmup_epochs: int, total_epochs: int) -> float:
    warmup = (epoch + 1) / max(1, warmup_epochs)
    cosine = 0.5 * (1 + math.cos(math.pi * epoch / total_epochs))
    return min(warmup, 1.0) * cosine


def mask_ratio_at(epoch: int, start: float, end: float, ramp_epochs: int) -> float:
    progress = min(epoch / max(1, ramp_epochs - 1), 1.0)
    return start + progress * (end - start)


class MAEPretrainModule(nn.Module):
    """Self-supervised pretraining for Masked Autoencoder (reference API, no Lightning dependency)."""

    def __init__(self, model_cfg: Dict[str, Any], training_cfg: Dict[str, Any]):
        super().__init__()
        self.hparams = {"model_cfg": model_cfg, "training_cfg": training_cfg}
        self.model = MaskedAutoencoder(
            general_cfg=model_cfg["general"], encoder_cfg=model_cfg["encoder"], decoder_cfg=model_cfg["decoder"])
        self.mask_start = training_cfg.get("mask_ratio_start", 0.5)
        self.mask_end = training_cfg.get("mask_ratio_end", 0.85)
        self.ramp_epochs = training_cfg.get("mask_ramp_epochs", 200)
        self.lr = float(training_cfg.get("base_learning_rate", 1.5e-4))
        self.weight_decay = float(training_cfg.get("weight_decay", 0.05))
        self.warmup_epochs = int(training_cfg.get("warmup_epochs", 20))
        self.total_epochs = int(training_cfg.get("total_epochs", 200))
        self.batch_size = int(training_cfg.get("batch_size", 512))
        self.criterion = torch.nn.MSELoss()
        self.current_epoch = 0
        self.global_step = 0
        self.logged: Dict[str, Any] = {}
        self.gradient_clip_val = 1.0
        # fused-step optimizer state (flat, trainable range)
        self._exp_avg: Optional[torch.Tensor] = None
        self._exp_avg_sq: Optional[torch.Tensor] = None
        self._stats: Optional[torch.Tensor] = None
        self._opt_steps = 0

    # ---- Lightning-shaped surface --------------------------------------------------------------
    def log(self, name: str, value, **_kw) -> None:
        self.logged[name] = value  # device tensors stay on device: no host sync in the step

    def forward(self, x: torch.Tensor):
        return self.model(x)

    def training_step(self, batch, batch_idx):
        imgs, _ = batch
        preds, targets = self(imgs)
        loss = self.criterion(preds, targets)
        self.log("train_loss", loss, prog_bar=True, on_epoch=True)
        return loss

    def validation_step(self, batch, batch_idx):
        imgs, _ = batch
        preds, targets = self(imgs)
        loss = self.criterion(preds, targets)
        self.log("val_loss", loss, prog_bar=True, on_epoch=True)
        return loss

    @property
    def effective_lr(self) -> float:
        return self.lr * self.batch_size / 256

    def configure_optimizers(self):
        from torch.optim import AdamW
        from torch.optim.lr_scheduler import LambdaLR
        optimizer = AdamW(self.parameters(), lr=self.effective_lr, weight_decay=self.weight_decay)
        scheduler = LambdaLR(optimizer, lambda e: lr_lambda(e, self.warmup_epochs, self.total_epochs))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "epoch", "name": "lr"}}

    def on_train_epoch_start(self):
        new_mask = mask_ratio_at(self.current_epoch, self.mask_start, self.mask_end, self.ramp_epochs)
        self.model.mask_ratio = new_mask
        self.log("mask_ratio", new_mask, prog_bar=True)

    # ---- fused native step ------------------------------------------------------------------------
    def current_lr(self) -> float:
        return self.effective_lr * lr_lambda(self.current_epoch, self.warmup_epochs, self.total_epochs)

    def _opt_state(self):
        dev = self.model.flat_params.device
        n = self.model.engine.trainable_elems
        if self._exp_avg is None or self._exp_avg.device != dev:
            self._exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
            self._exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
            self._stats = torch.zeros(8, dtype=torch.float32, device=dev)
        return self._exp_avg, self._exp_avg_sq, self._stats

    def optimizer_step(self, lr: Optional[float] = None) -> torch.Tensor:
        """clip_grad_norm_(gradient_clip_val) + AdamW over model.flat_grads, natively; returns [norm, clip_coef]."""
        model = self.model
        dev = model._require_cuda()
        m, v, stats = self._opt_state()
        self._opt_steps += 1
        check(lib.mae_engine_optimizer_step(
            model.engine.handle, _ptr(model.flat_params), _ptr(model.flat_grads), _ptr(m), _ptr(v), _ptr(model._weights()),
            float(self.current_lr() if lr is None else lr), 0.9, 0.999, 1e-8, float(self.weight_decay),
            float(self.gradient_clip_val), self._opt_steps, _ptr(stats), _ptr(model._scratch_f32()), _stream(dev)))
        model.mark_weights_fresh()  # the native step refreshed the operand copies itself
        return stats

    def fused_training_step(self, images: torch.Tensor, noise: Optional[torch.Tensor] = None, lr: Optional[float] = None,
                            process_group=None) -> torch.Tensor:
        """One whole pretrain step.  With torch.distributed initialised the local gradients (already divided by
        world size) are summed with ONE all-reduce over RCCL before the global-norm clip."""
        model = self.model
        dev = model._require_cuda()
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(process_group)
        if noise is None:
            noise = torch.rand(images.shape[0], model.sequence_length, device=dev)
        loss = model.loss_and_grads(images, noise, grad_scale=1.0 / world)
        if world > 1:
            torch.distributed.all_reduce(model.flat_grads, op=torch.distributed.ReduceOp.SUM, group=process_group)
        self.optimizer_step(lr)
        self.global_step += 1
        self.log("train_loss", loss)
        return loss

    def optimizer_state_dict(self) -> Dict[str, Any]:
        m, v, _ = self._opt_state()
        return {"step": self._opt_steps, "exp_avg": self.model.named_flat_views(m), "exp_avg_sq": self.model.named_flat_views(v)}
