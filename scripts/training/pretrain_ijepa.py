"""Self-supervised I-JEPA pretraining on the MI355X engine.

THE REFERENCE HAS NO I-JEPA ENTRY POINT (nor any I-JEPA code: README.md:1,9 name it only).  This script gives the I-JEPA step
 the same harness as scripts/training/pretrain_mae.py: the same three flags, the same output tree
``outputs/pretrain/<suffix>/{checkpoints/{best,last}.ckpt, logs/metrics.jsonl, config.yaml, <model_path>}``, seed 73, per-epoch
learning-rate schedule, validation each epoch, resume, one process per GPU under torch.distributed.run.

    python -m scripts.training.pretrain_ijepa --config configs/ijepa_vits8.yaml [--resume_from CKPT] [--output_dir_suffix NAME]
"""
from __future__ import annotations

import json
import time
from pathlib import Path

import torch
import yaml

from scripts.training.pretrain_mae import SEED, load_checkpoint, parse_args, save_checkpoint
from ssrl_vit_mae_jepa_amd import dist as mdist
from ssrl_vit_mae_jepa_amd.data import get_pretrain_batches
from ssrl_vit_mae_jepa_amd.jepa import IJEPAPretrainModule


def main(argv=None):
    args = parse_args(argv)
    if args.output_dir_suffix == "mae_pretrain":
        args.output_dir_suffix = "ijepa_pretrain"
    with open(args.config, "r") as f:
        cfg = yaml.safe_load(f)
    pre_cfg, model_cfg, log_cfg = cfg["pretrain"], cfg["model"], cfg["logging"]
    eng_cfg = cfg.get("engine", {})
    model_cfg = dict(model_cfg, general=dict(model_cfg["general"], engine_precision=eng_cfg.get("precision", "bf16")))
    rank, local_rank, world = mdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("pretrain_ijepa: the MI355X engine has no CPU fallback")
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

    data_cfg = dict(cfg, pretrain=dict({"val_split": 0.06, "data_fraction": 1.0}, **pre_cfg))
    train_batches, val_batches = get_pretrain_batches(data_cfg, dev, synthetic_images=args.synthetic_images, seed=SEED, rank=rank, world=world)

    module = IJEPAPretrainModule(model_cfg=model_cfg, training_cfg=pre_cfg).to(dev)
    start_epoch, resumed = load_checkpoint(args.resume_from, module) if args.resume_from else (0, {})
    module = module.to(dev)
    model = module.model
    # the EMA momentum schedule runs over the optimizer steps this loop really takes (every global batch of an epoch, ragged last one
    # included, capped by --max_steps_per_epoch), not over the config's fallback count
    module.steps_per_epoch = train_batches.steps_per_epoch if args.max_steps_per_epoch is None else min(train_batches.steps_per_epoch, args.max_steps_per_epoch)
    best_val = float(resumed.get("best_val_loss", float("inf")))
    log_path = output_dir / "logs" / "metrics.jsonl"
    val_gen = torch.Generator().manual_seed(SEED + 1)
    total_epochs = int(pre_cfg["total_epochs"]) if args.max_epochs is None else min(int(pre_cfg["total_epochs"]), start_epoch + args.max_epochs)
    for epoch in range(start_epoch, total_epochs):
        module.current_epoch = epoch
        module.on_train_epoch_start()
        t0, seen, steps, loss_sum = time.perf_counter(), 0, 0, torch.zeros(1, device=dev)
        for step, sb in enumerate(train_batches(epoch)):
            if args.max_steps_per_epoch is not None and step >= args.max_steps_per_epoch:
                break
            ctx, tgt = model.sample_masks(sb.global_rows, module.mask_generator)  # masks of the GLOBAL batch (every rank draws the same), own rows kept
            loss = module.fused_training_step(sb.images, ctx[sb.lo:sb.hi].contiguous(), tgt[sb.lo:sb.hi].contiguous(), global_rows=sb.global_rows)
            loss_sum += loss
            seen += sb.global_rows
            steps += 1
        val = torch.zeros(2, device=dev, dtype=torch.float64)
        for sb in val_batches():
            imgs = sb.images
            if imgs.shape[0] == 0:
                continue
            ctx, tgt = model.sample_masks(imgs.shape[0], val_gen)
            val[0] += model.loss_and_grads(imgs, ctx, tgt).double()[0] * imgs.shape[0]  # the latent loss (its gradients are discarded)
            val[1] += imgs.shape[0]
        if world > 1:
            torch.distributed.all_reduce(val)
        train_loss = float(loss_sum.item()) / max(1, steps)
        val_loss = float(val[0].item()) / max(1.0, float(val[1].item()))
        if rank == 0:
            rec = dict(epoch=epoch, train_loss=train_loss, val_loss=val_loss, lr=module.current_lr(), ema_momentum=module.ema_momentum(),
                       images_per_s=seen / (time.perf_counter() - t0))
            with open(log_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
            print(json.dumps(rec))
            if val_loss < best_val:
                best_val = val_loss
                save_checkpoint(ckpt_dir / "best.ckpt", module, epoch, extra={"val_loss": val_loss, "best_val_loss": best_val})
            save_checkpoint(ckpt_dir / "last.ckpt", module, epoch, extra={"val_loss": val_loss, "best_val_loss": best_val})
    if rank == 0:
        model_path = output_dir / log_cfg["model_path"]
        sd = {k: v.detach().cpu() for k, v in model.net.state_dict().items()}                 # context encoder + predictor
        sd.update({f"target_encoder.{k[len('encoder.'):]}": v.detach().cpu().clone() for k, v in model.target_state_dict().items()})
        torch.save(sd, model_path)
        print(f"Pretraining complete; model weights saved to: {model_path}")


if __name__ == "__main__":
    main()
