"""Self-supervised MAE pretraining on the MI355X engine: the caller of the hot path.

Keeps the reference entry point's surface (scripts/training/pretrain_mae.py:21-36, 39-139): the same three flags, the
same YAML keys, the same output tree ``outputs/pretrain/<suffix>/{checkpoints/{best,last,epoch-XXX}.ckpt, logs/,
config.yaml, vit-mae.pt}``, seed 73, per-epoch LR / mask-ratio schedules, clip 1.0, validation each epoch, resume.
The Lightning Trainer is replaced by the plain loop below (Lightning is not installed here); data is either the STL-10
``unlabeled_X.bin`` file if present or synthetic 96x96x3 batches (there is no network for the download).

    python -m scripts.training.pretrain_mae --config configs/mae.yaml [--resume_from CKPT] [--output_dir_suffix NAME]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m scripts.training.pretrain_mae ...
"""
from __future__ import annotations

import argparse
import json
import os
import time
from pathlib import Path

import torch
import yaml

from ssrl_vit_mae_jepa_amd import dist as mdist
from ssrl_vit_mae_jepa_amd.data import get_pretrain_batches
from ssrl_vit_mae_jepa_amd.training import MAEPretrainModule

SEED = 73  # setup_reproducibility(seed=73), reference scripts/training/pretrain_mae.py:18


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Self-supervised MAE pretraining")
    parser.add_argument("--config", type=str, default="configs/mae.yaml")
    parser.add_argument("--resume_from", type=str, default=None, help="Path to checkpoint to resume from")
    parser.add_argument("--output_dir_suffix", type=str, default="mae_pretrain", help="Suffix for the output directory")
    # additions (not in the reference): bounded runs for smoke tests / benchmarks
    parser.add_argument("--max_epochs", type=int, default=None)
    parser.add_argument("--max_steps_per_epoch", type=int, default=None)
    parser.add_argument("--synthetic_images", type=int, default=None, help="use N synthetic images instead of STL-10")
    return parser.parse_args(argv)


def save_checkpoint(path: Path, module: MAEPretrainModule, epoch: int, weights_only: bool = False, extra=None) -> None:
    """Lightning-shaped checkpoint (MAEPretrainModule.checkpoint_dict), written atomically."""
    tmp = path.with_suffix(path.suffix + ".tmp")
    torch.save(module.checkpoint_dict(epoch, weights_only=weights_only, extra=extra), tmp)
    os.replace(tmp, path)


def load_checkpoint(path: str, module: MAEPretrainModule):
    """Returns (epoch to continue with, the checkpoint dict).  Files are opened with the safe loader only; a reference
    Lightning checkpoint whose hyper-parameters need unpickling is refused by it and reported as such."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return module.load_checkpoint_dict(ckpt), ckpt


def main(argv=None):
    args = parse_args(argv)
    with open(args.config, "r") as f:
        cfg = yaml.safe_load(f)
    pre_cfg, model_cfg, log_cfg = cfg["pretrain"], cfg["model"], cfg["logging"]
    eng_cfg = cfg.get("engine", {})
    model_cfg = dict(model_cfg, general=dict(model_cfg["general"], engine_precision=eng_cfg.get("precision", "bf16")))

    rank, local_rank, world = mdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("pretrain_mae: the MI355X engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mdist.init_from_env(device=dev)
    torch.manual_seed(SEED)

    output_dir = Path(log_cfg["output_dir_base"]) / "pretrain" / args.output_dir_suffix
    ckpt_dir = output_dir / "checkpoints"
    if rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        (output_dir / "logs").mkdir(exist_ok=True)
        with open(output_dir / "config.yaml", "w") as f_out:
            yaml.safe_dump(cfg, f_out)
        print(f"Saved config snapshot to: {output_dir / 'config.yaml'}")

    module = MAEPretrainModule(model_cfg=model_cfg, training_cfg=pre_cfg).to(dev)
    start_epoch, resumed = load_checkpoint(args.resume_from, module) if args.resume_from else (0, {})
    module = module.to(dev)
    L = module.model.sequence_length
    train_batches, val_batches = get_pretrain_batches(cfg, dev, synthetic_images=args.synthetic_images, seed=SEED, rank=rank, world=world)

    # resume bookkeeping: best validation loss so far (kept in last.ckpt, else read from the best.ckpt next to it)
    best_val, log_path = float("inf"), output_dir / "logs" / "metrics.jsonl"
    if args.resume_from:
        best_val = float(resumed.get("best_val_loss", best_val))
        best_path = Path(args.resume_from).with_name("best.ckpt")
        if best_val == float("inf") and best_path.exists():
            best_val = float(torch.load(best_path, map_location="cpu", weights_only=True).get("val_loss", best_val))
    total_epochs = int(pre_cfg["total_epochs"]) if args.max_epochs is None else min(int(pre_cfg["total_epochs"]), start_epoch + args.max_epochs)
    for epoch in range(start_epoch, total_epochs):
        module.current_epoch = epoch
        module.on_train_epoch_start()
        t0, seen, steps, loss_sum = time.perf_counter(), 0, 0, torch.zeros(1, device=dev)
        for step, sb in enumerate(train_batches(epoch)):
            if args.max_steps_per_epoch is not None and step >= args.max_steps_per_epoch:
                break
            # this rank's rows of the global batch (fetched for these rows only) and of the global noise (drawn whole: 145 floats per
            # image, so that W-GPU masks equal the one-GPU run); a ragged last batch is weighted by the rows each rank holds
            noise = mdist.global_noise(sb.global_rows, L, SEED, module.global_step, dev)[sb.lo:sb.hi].contiguous()
            loss = module.fused_training_step(sb.images, noise, global_rows=sb.global_rows)
            loss_sum += loss  # already the global mean (it rides through the gradient all-reduce)
            seen += sb.global_rows
            steps += 1
        # validation (masked-reconstruction MSE, no grad): every rank scores its rows of every batch, the sums are reduced
        val = torch.zeros(2, device=dev, dtype=torch.float64)  # [sum of per-image losses, images]
        with torch.no_grad():
            for sb in val_batches():
                imgs = sb.images
                if imgs.shape[0] == 0:
                    continue
                preds, targets = module.model(imgs)
                val[0] += torch.nn.functional.mse_loss(preds, targets).double() * imgs.shape[0]
                val[1] += imgs.shape[0]
        if world > 1:
            torch.distributed.all_reduce(val)
        train_loss = float(loss_sum.item()) / max(1, steps)
        val_loss = float(val[0].item()) / max(1.0, float(val[1].item()))
        dt = time.perf_counter() - t0
        module.gather_optimizer_state()   # a collective when the optimizer is sharded (MAE_DP_SHARDED_OPT=1): rank 0's checkpoint needs every slice
        if rank == 0:
            rec = dict(epoch=epoch, train_loss=train_loss, val_loss=val_loss, lr=module.current_lr(), mask_ratio=module.model.mask_ratio,
                       images_per_s=seen / dt)
            with open(log_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
            print(json.dumps(rec))
            if val_loss < best_val:
                best_val = val_loss
                save_checkpoint(ckpt_dir / "best.ckpt", module, epoch, extra={"val_loss": val_loss, "best_val_loss": best_val})
            save_checkpoint(ckpt_dir / "last.ckpt", module, epoch, extra={"val_loss": val_loss, "best_val_loss": best_val})
            if (epoch + 1) % 25 == 0:
                save_checkpoint(ckpt_dir / f"epoch-epoch={epoch:03d}.ckpt", module, epoch, weights_only=True)
    if rank == 0:
        model_path = output_dir / log_cfg["model_path"]
        torch.save({k: v.detach().cpu() for k, v in module.model.state_dict().items()}, model_path)
        print(f"Pretraining complete; model weights saved to: {model_path}")


if __name__ == "__main__":
    main()
