"""MAEPretrainModule: the reference's ``src/training/mae.py`` (lines 14-83) without Lightning, plus the fused
data-parallel step: native loss+grads whose backward pass reports gradient-ready points -> bucketed all-reduce of the
flat gradient arena over RCCL, started on a side stream while the rest of the backward pass still runs -> native
global-norm clip + AdamW (which needs the WHOLE reduced gradient first: scripts/training/pretrain_mae.py:124-125).

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
import os
import warnings
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
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
        self._init_training_state(training_cfg)

    def _init_training_state(self, training_cfg: Dict[str, Any]) -> None:
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
        # data-parallel exchange: gradient buckets (arena ranges, in the order the backward pass finishes them)
        self.bucket_bytes = int(float(os.environ.get("MAE_DP_BUCKET_MB", "24")) * (1 << 20))
        self.overlap_exchange = os.environ.get("MAE_DP_OVERLAP", "1") != "0"
        self._buckets: Optional[List[Tuple[int, int, int]]] = None   # (ready point, begin, end) in floats of the grad buffer
        self._bucket_events: Optional[List[Optional[torch.cuda.Event]]] = None
        self._comm_stream: Optional[torch.cuda.Stream] = None

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
    def current_lr(self, epoch: Optional[int] = None) -> float:
        return self.effective_lr * lr_lambda(self.current_epoch if epoch is None else epoch, self.warmup_epochs, self.total_epochs)

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

    # ---- data-parallel gradient exchange -------------------------------------------------------------
    def gradient_buckets(self) -> List[Tuple[int, int, int]]:
        """Ranges of the gradient buffer to all-reduce, in the order the backward pass finishes them: consecutive
        gradient-ready points (decoder, encoder block depth-1, ..., block 0 + patch projection) are merged until a bucket
        holds at least ``bucket_bytes``.  The first bucket also covers the loss slot behind the gradients, so the
        global mean loss rides along without a collective of its own.  The buckets tile the whole buffer exactly."""
        if self._buckets is None:
            model = self.model
            n = model.engine.trainable_elems
            points = model.grad_ready_points()
            buckets, end = [], n + 1  # slot n carries the loss
            for j, off in enumerate(points):
                last = j == len(points) - 1
                if last or (end - off) * 4 >= self.bucket_bytes:
                    if off < end:
                        buckets.append((j, off, end))
                    end = off
            assert end == 0 and sum(b[2] - b[1] for b in buckets) == n + 1
            self._buckets = buckets
        return self._buckets

    def _exchange_state(self, dev):
        if self._bucket_events is None or self._comm_stream is None or self._comm_stream.device != dev:
            npts = len(self.model.grad_ready_points())
            evs: List[Optional[torch.cuda.Event]] = [None] * npts
            for j, _b, _e in self.gradient_buckets():
                evs[j] = torch.cuda.Event()
                evs[j].record(torch.cuda.current_stream(dev))  # torch creates the HIP event lazily: force it now
            self._bucket_events = evs
            self._comm_stream = torch.cuda.Stream(device=dev)
        return self._bucket_events, self._comm_stream

    def _exchanged_loss_and_grads(self, compute, process_group=None, weight: Optional[float] = None) -> torch.Tensor:
        """Run ``compute(grad_scale, ready_events, loss_out)`` (a native loss+grads call) and, with torch.distributed
        initialised, sum the gradient buckets over the ranks as the backward pass finishes them.  Returns the GLOBAL mean
        loss as a device scalar (no host sync); afterwards ``model.flat_grads`` holds the reduced gradients.
        ``weight`` = this rank's share of the global batch (local rows / global rows; default 1 / world, equal shards): the
        local mean loss and its gradient are scaled by it before the sum, so a ragged last batch (the reference never drops
        one, src/data.py:86-92) still yields the global mean.  ``compute`` is None on a rank that holds no row of the batch:
        it contributes zeros to the same sequence of collectives."""
        model = self.model
        dev = model._require_cuda()
        dist = torch.distributed
        world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1 and not (dist.is_available() and dist.is_initialized() and os.environ.get("MAE_DP_FORCE_EXCHANGE") == "1"):
            return compute(1.0, None, None)  # (MAE_DP_FORCE_EXCHANGE=1: rehearse the exchange on a one-rank RCCL group)
        w = 1.0 / world if weight is None else float(weight)
        n = model.engine.trainable_elems
        buf = model.grad_buffer
        loss_slot = buf[n:n + 1]
        if self.overlap_exchange:
            events, comm = self._exchange_state(dev)
            main = torch.cuda.current_stream(dev)
            comm.wait_stream(main)  # the previous step's optimizer has read the buffer before it is reduced into again
            if compute is None:
                buf[:n + 1].zero_()
                comm.wait_stream(main)
            else:
                compute(w, events, loss_slot)
            works = []
            with torch.cuda.stream(comm):
                for j, b, e in self.gradient_buckets():
                    if compute is not None:
                        comm.wait_event(events[j])
                    if compute is not None and b <= n < e:
                        loss_slot.mul_(w)  # the loss was written before this bucket's event (it opens the backward pass): weight it ahead of the sum
                    works.append(dist.all_reduce(buf[b:e], op=dist.ReduceOp.SUM, group=process_group, async_op=True))
                for wk in works:
                    wk.wait()
            main.wait_stream(comm)
        else:  # one blocking collective after the whole backward pass
            if compute is None:
                buf[:n + 1].zero_()
            else:
                compute(w, None, loss_slot)
                loss_slot.mul_(w)
            dist.all_reduce(buf[:n + 1], op=dist.ReduceOp.SUM, group=process_group)
        return loss_slot * 1.0  # sum over ranks of (share x local mean) = the global mean; a copy: the slot is rewritten next step

    # ---- sharded optimizer (MAE_DP_SHARDED_OPT=1): reduce-scatter -> shard sum of squares -> scalar all-reduce -> AdamW on the shard -> all-gather
    def _sharded_opt_world(self, process_group) -> int:
        dist = torch.distributed
        if os.environ.get("MAE_DP_SHARDED_OPT") != "1" or not (dist.is_available() and dist.is_initialized()):
            return 0
        world = dist.get_world_size(process_group)
        return world if world > 1 or os.environ.get("MAE_DP_FORCE_EXCHANGE") == "1" else 0

    def _sharded_step(self, compute, process_group, weight: Optional[float], lr: Optional[float], world: int) -> torch.Tensor:
        """The data-parallel step with the optimizer sharded over the ranks: every rank owns 1 / world of the
        trainable arena (64-float granules), receives only that slice of the summed gradient (reduce-scatter: half the bytes of
        an all-reduce), adds its slice's sum of squares to the others' (one scalar all-reduce: clip_grad_norm_ needs the norm of
        the WHOLE gradient), runs AdamW on the slice and all-gathers the updated parameters (the other half).  Same result as
        the replicated step up to the order in which the squares are summed (one ulp of the clip coefficient).  No overlap with
        the backward pass: the exchange is one collective over the padded arena."""
        model = self.model
        dev = model._require_cuda()
        dist = torch.distributed
        rank = dist.get_rank(process_group)
        n = model.engine.trainable_elems
        shard = ((n + world - 1) // world + 63) // 64 * 64          # ceil(n / world) rounded up to 64 floats
        npad = shard * world
        buf, arena = model.grad_buffer, model.flat_params
        if npad + 1 > buf.numel() or npad > arena.numel():
            raise RuntimeError(f"sharded optimizer: {world} ranks need {npad - n} floats of padding behind the {n} trainable elements")
        w = 1.0 / world if weight is None else float(weight)
        loss_slot = buf[n:n + 1]
        if compute is None:
            buf[:n + 1].zero_()
        else:
            compute(w, None, loss_slot)
            loss_slot.mul_(w)
        loss = loss_slot.clone()            # the slot lies inside the last rank's slice: the loss travels on its own
        dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=process_group)
        loss_slot.zero_()                   # not a gradient: keep it out of the reduce-scatter's last slice
        lo = rank * shard
        own = buf[lo:lo + shard]
        nccl = dist.get_backend(process_group) == "nccl"
        if nccl:
            dist.reduce_scatter_tensor(own, buf[:npad], op=dist.ReduceOp.SUM, group=process_group)   # in place: own is this rank's slice of the input
        else:                               # gloo (CPU rehearsal, one-GPU tests) has no reduce-scatter: the sum everywhere, then use the slice
            dist.all_reduce(buf[:npad], op=dist.ReduceOp.SUM, group=process_group)
        count = max(0, min(n, lo + shard) - lo)
        m, v, stats = self._opt_state()
        self._opt_steps += 1
        h, st = model.engine.handle, _stream(dev)
        sumsq = stats[4:5]
        check(lib.mae_engine_grad_sumsq_range(h, _ptr(model.flat_grads), lo if count else 0, count, _ptr(sumsq), _ptr(model._scratch_f32()), st))
        dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=process_group)
        check(lib.mae_engine_clip_from_sumsq(h, _ptr(sumsq), float(self.gradient_clip_val), _ptr(stats), st))
        check(lib.mae_engine_adamw_range(h, _ptr(arena), _ptr(model.flat_grads), _ptr(m), _ptr(v), _ptr(model._weights()),
                                         float(self.current_lr() if lr is None else lr), 0.9, 0.999, 1e-8, float(self.weight_decay),
                                         self._opt_steps, _ptr(stats), lo if count else 0, count, st))
        mine = arena[lo:lo + shard]          # the last rank's slice runs into the frozen tables behind the trainable range: identical on every rank
        if nccl:
            dist.all_gather_into_tensor(arena[:npad], mine, group=process_group)
        else:
            dist.all_gather([arena[r * shard:(r + 1) * shard] for r in range(world)], mine.clone(), group=process_group)
        if model.engine.act == _lib.MAE_BF16:   # bf16 copies of the other ranks' slices + every transposed copy
            check(lib.mae_engine_refresh_weights(h, _ptr(arena), _ptr(model._wcache), st))
        model.mark_weights_fresh()
        self._moments_sharded = (world, shard)   # exp_avg / exp_avg_sq are current on this rank's slice only until gather_optimizer_state()
        return loss

    def gather_optimizer_state(self, process_group=None) -> None:
        """After sharded steps every rank holds the AdamW moments of its own slice only: all-gather them so that
        ``optimizer_state_dict()`` / ``checkpoint_dict()`` (usually called on rank 0 alone) see the whole state.  A collective: EVERY
        rank calls it (the CLI does, once per epoch, before the checkpoint).  A no-op in the replicated mode."""
        if getattr(self, "_moments_sharded", None) is None:
            return
        dist = torch.distributed
        world, shard = self._moments_sharded
        rank = dist.get_rank(process_group)
        m, v, _ = self._opt_state()
        n = m.numel()
        for t in (m, v):
            full = torch.zeros(world * shard, dtype=t.dtype, device=t.device)
            lo = rank * shard
            cnt = max(0, min(n, lo + shard) - lo)
            full[lo:lo + cnt] = t[lo:lo + cnt]
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=process_group)   # slices are disjoint: the sum is the concatenation (works on gloo too)
            t.copy_(full[:n])
        self._moments_sharded = None

    def fused_training_step(self, images: torch.Tensor, noise: Optional[torch.Tensor] = None, lr: Optional[float] = None,
                            process_group=None, global_rows: Optional[int] = None) -> torch.Tensor:
        """One whole pretrain step.  With torch.distributed initialised every rank computes its rows of the global batch
        with the loss gradient scaled by its share of the batch (``images.shape[0] / global_rows``; 1 / world when
        ``global_rows`` is not given: equal shards), the gradient buckets are summed over RCCL as the backward pass finishes
        them (on a side stream, behind the engine's gradient-ready events), and every rank then applies the same
        global-norm clip + AdamW.  A rank whose shard of a ragged last batch is empty passes a 0-row ``images``.
        Returns the GLOBAL mean loss as a device scalar (no host sync)."""
        model = self.model
        dev = model._require_cuda()
        rows = int(images.shape[0])
        weight = None if global_rows is None else rows / float(global_rows)
        if rows == 0:
            if global_rows is None:
                raise ValueError("an empty shard needs global_rows")
            compute = None
        else:
            if noise is None:
                noise = torch.rand(rows, model.sequence_length, device=dev)
            compute = lambda scale, events, out: model.loss_and_grads(images, noise, grad_scale=scale, ready_events=events, loss_out=out)  # noqa: E731
        sharded = self._sharded_opt_world(process_group)
        if sharded:
            loss = self._sharded_step(compute, process_group, weight, lr, sharded)
        else:
            loss = self._exchanged_loss_and_grads(compute, process_group, weight)
            self.optimizer_step(lr)
        self.global_step += 1
        self.log("train_loss", loss)
        return loss

    # ---- optimizer state in the layout Lightning checkpoints carry (torch.optim.AdamW.state_dict()) ------------------
    def _named_param_index(self) -> List[Tuple[int, str, bool]]:
        """(index in self.parameters() order, state_dict name, has optimizer state) -- the reference hands every
        parameter to AdamW (src/training/mae.py:62); only those that receive a gradient get state."""
        trainable = set(self.model.named_flat_views(self.model.flat_grads))
        strip = lambda n: n[4:] if n.startswith("net.") else n  # noqa: E731  (IJEPA keeps its parameter table in .net)
        return [(i, strip(n), strip(n) in trainable) for i, (n, _p) in enumerate(self.model.named_parameters())]

    def optimizer_state_dict(self) -> Dict[str, Any]:
        if getattr(self, "_moments_sharded", None) is not None:
            raise RuntimeError("the AdamW moments are sharded over the ranks (MAE_DP_SHARDED_OPT=1): call gather_optimizer_state() on every rank first")
        m, v, _ = self._opt_state()
        mv, vv = self.model.named_flat_views(m), self.model.named_flat_views(v)
        state = {}
        if self._opt_steps > 0:
            for i, name, has in self._named_param_index():
                if has:
                    state[i] = {"step": torch.tensor(float(self._opt_steps)), "exp_avg": mv[name].detach().cpu().clone(),
                                "exp_avg_sq": vv[name].detach().cpu().clone()}
        group = {"lr": self.current_lr(self.current_epoch + 1), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "initial_lr": self.effective_lr,
                 "params": [i for i, _n, _h in self._named_param_index()]}
        return {"state": state, "param_groups": [group]}

    def lr_scheduler_state_dict(self) -> Dict[str, Any]:
        """LambdaLR.state_dict() as Lightning stores it (the lambda itself is not picklable and is saved as None).  Lightning steps
        the epoch-interval scheduler BEFORE ModelCheckpoint runs at the end of epoch e, so the file written there carries
        last_epoch = e + 1, _step_count = e + 2 and the learning rate of epoch e + 1 (also in the optimizer's param group)."""
        nxt = self.current_epoch + 1
        return {"base_lrs": [self.effective_lr], "last_epoch": nxt, "verbose": False,
                "_step_count": nxt + 1, "_get_lr_called_within_step": False, "_last_lr": [self.current_lr(nxt)],
                "lr_lambdas": [None]}

    def load_optimizer_state_dict(self, st: Dict[str, Any]) -> bool:
        """Accepts torch.optim.AdamW.state_dict() (what a reference Lightning ``last.ckpt`` holds, indexed in
        ``parameters()`` order) and this repository's round-1 layout ({step, exp_avg{name}, exp_avg_sq{name}}).
        Unknown layouts: weights-only resume with a warning.  Returns True when the state was restored."""
        m, v, _ = self._opt_state()
        mv, vv = self.model.named_flat_views(m), self.model.named_flat_views(v)
        try:
            if "state" in st and "param_groups" in st:
                by_index = {i: n for i, n, has in self._named_param_index() if has}
                if not st["state"]:
                    m.zero_(); v.zero_(); self._opt_steps = 0
                    return True
                missing = [n for i, n in by_index.items() if i not in st["state"]]
                if missing:
                    raise KeyError(f"no optimizer state for {missing[:3]}...")
                steps = set()
                for i, n in by_index.items():
                    e = st["state"][i]
                    if tuple(e["exp_avg"].shape) != tuple(mv[n].shape):
                        raise ValueError(f"optimizer state {i} has shape {tuple(e['exp_avg'].shape)}, parameter {n} {tuple(mv[n].shape)}")
                    mv[n].copy_(e["exp_avg"]); vv[n].copy_(e["exp_avg_sq"])
                    steps.add(int(float(e["step"])))
                if len(steps) != 1:
                    raise ValueError(f"per-parameter step counts differ: {sorted(steps)[:4]}")
                self._opt_steps = steps.pop()
                return True
            if "exp_avg" in st and "exp_avg_sq" in st:
                for k in mv:
                    mv[k].copy_(st["exp_avg"][k]); vv[k].copy_(st["exp_avg_sq"][k])
                self._opt_steps = int(st["step"])
                return True
            raise KeyError(f"unknown optimizer state layout with keys {sorted(st)[:6]}")
        except (KeyError, ValueError, TypeError, IndexError) as exc:
            warnings.warn(f"optimizer state not restored ({exc}); resuming from the weights only", RuntimeWarning)
            m.zero_(); v.zero_(); self._opt_steps = 0
            return False

    def checkpoint_dict(self, epoch: int, weights_only: bool = False, extra: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        """A Lightning-shaped checkpoint: ``state_dict`` keys carry the ``model.`` prefix the reference's loaders sniff
        (scripts/training/train_mae.py:105-109); ``optimizer_states`` / ``lr_schedulers`` / ``hyper_parameters`` /
        ``pytorch-lightning_version`` / ``loops`` are the keys ``Trainer.fit(ckpt_path=...)`` reads (uv.lock:1772-1773)."""
        ckpt: Dict[str, Any] = {
            "epoch": epoch, "global_step": self.global_step, "pytorch-lightning_version": "2.5.6",
            "state_dict": {f"model.{k}": t.detach().cpu().clone() for k, t in self.model.state_dict().items()},
            # Lightning's _Progress.load_state_dict reads both "total" and "current"
            "loops": {"fit_loop": {"epoch_progress": {"total": {"completed": epoch + 1, "processed": epoch + 1, "ready": epoch + 1, "started": epoch + 1},
                                                      "current": {"completed": epoch + 1, "processed": epoch + 1, "ready": epoch + 1, "started": epoch + 1}}}},
            "callbacks": {},
            "hyper_parameters": {"model_cfg": self.hparams["model_cfg"], "training_cfg": self.hparams["training_cfg"]},
        }
        if not weights_only:
            ckpt["optimizer_states"] = [self.optimizer_state_dict()]
            ckpt["lr_schedulers"] = [self.lr_scheduler_state_dict()]
        if extra:
            ckpt.update(extra)
        return ckpt

    def load_checkpoint_dict(self, ckpt: Dict[str, Any]) -> int:
        """Restore weights (+ optimizer state when present) from a checkpoint of this repository or of the reference's
        Lightning Trainer; returns the epoch to continue with."""
        sd = ckpt.get("state_dict", ckpt)
        sd = {k[len("model."):] if k.startswith("model.") else k: t for k, t in sd.items()}
        self.model.load_state_dict(sd, strict=True)
        if ckpt.get("optimizer_states"):
            self.load_optimizer_state_dict(ckpt["optimizer_states"][0])
        self.global_step = int(ckpt.get("global_step", 0))
        return int(ckpt.get("epoch", -1)) + 1

