#!/usr/bin/env python3
"""One data-parallel rank of the PRODUCT step (MAEPretrainModule.fused_training_step over libmae_hip.so), started as a
fresh child process by tests/test_gpu_dp.py (and usable by hand).  Every rank uses cuda:0 (a one-GPU box) and the gloo
backend; rank r takes rows [r*B/W, (r+1)*B/W) of the seeded global images and noise (B need not divide by W: the step weights the ranks by their rows), runs `--steps` whole steps and
saves its final parameters, losses and first-step masks.  With --world 1 it is the single-process full-batch step.

    python tests/dp_worker.py --rank 0 --world 2 --port 29511 --out /tmp/x --config micro --precision fp32
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, default=29511)
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="micro", choices=["micro", "vits8", "ijepa"])
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--global-batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ssrl_vit_mae_jepa_amd import IJEPAPretrainModule, MAEPretrainModule
    from ssrl_vit_mae_jepa_amd import dist as mdist

    if a.world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world), LOCAL_RANK="0")
        dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    jepa = a.config == "ijepa"
    if a.config in ("micro", "ijepa"):
        general = dict(image_size=32, patch_size=8 if not jepa else 4, in_chans=3, mask_ratio=0.75)
        encoder, decoder = dict(embed_dim=48, depth=3, num_heads=2), dict(decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)
    else:
        general = dict(image_size=96, patch_size=8, in_chans=3, mask_ratio=0.75)
        encoder, decoder = dict(embed_dim=384, depth=12, num_heads=6), dict(decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6)
    general["engine_precision"] = a.precision
    tcfg = dict(mask_ratio_start=0.75, mask_ratio_end=0.75, mask_ramp_epochs=5, total_epochs=800, warmup_epochs=20,
                batch_size=2000, base_learning_rate=1.5e-4, weight_decay=0.05)
    if jepa:  # the I-JEPA step shares the exchange (gradient-ready events, buckets, loss slot) with the MAE step
        module = IJEPAPretrainModule(dict(general=general, encoder=encoder, predictor=dict(pred_embed_dim=32, pred_depth=1, pred_num_heads=2)),
                                     dict(tcfg, steps_per_epoch=4, ema_start=0.9))
        net = module.model.net
    else:
        module = MAEPretrainModule(dict(general=general, encoder=encoder, decoder=decoder), tcfg)
        net = module.model
    net._init_weights(seed=73)  # identical parameters on every rank
    with torch.no_grad():  # non-zero biases / LayerNorm affine so every gradient term is exercised
        g = torch.Generator().manual_seed(7)
        for n, p in net.named_parameters():
            if p.requires_grad and p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    if jepa:
        module.model.reset_target()
    module = module.to(dev)
    module.on_train_epoch_start()
    model = module.model
    B = a.global_batch
    gi = torch.Generator().manual_seed(100)
    images = (torch.rand(B, 3, general["image_size"], general["image_size"], generator=gi) * 2 - 1).to(dev)
    losses, keep0 = [], None
    for step in range(a.steps):
        my_images = mdist.shard_rows(images, a.rank, a.world)
        if jepa:
            ctx, tgt = model.sample_masks(B, torch.Generator().manual_seed(step))  # masks of the global batch, own rows kept
            my_ctx, my_tgt = mdist.shard_rows(ctx, a.rank, a.world), mdist.shard_rows(tgt, a.rank, a.world)
            if step == 0:
                keep0 = my_ctx.clone()
            losses.append(module.fused_training_step(my_images, my_ctx, my_tgt, global_rows=B).clone())
        else:
            noise = mdist.global_noise(B, model.sequence_length, 73, step, dev)
            my_noise = mdist.shard_rows(noise, a.rank, a.world)
            if step == 0:
                keep0 = model.random_token_mask(my_images.shape[0], my_noise)[0].cpu() if my_images.shape[0] else torch.zeros(0, model.num_keep(), dtype=torch.int64)
            losses.append(module.fused_training_step(my_images, my_noise, global_rows=B).clone())
    torch.cuda.synchronize()
    if hasattr(module, "gather_optimizer_state") and a.world > 1:
        module.gather_optimizer_state()   # sharded optimizer: every rank's slice of the AdamW moments (a no-op otherwise)
    out = {"exp_avg": module._exp_avg.detach().cpu() if getattr(module, "_exp_avg", None) is not None else torch.zeros(1),
           "exp_avg_sq": module._exp_avg_sq.detach().cpu() if getattr(module, "_exp_avg_sq", None) is not None else torch.zeros(1),
           "params": model.flat_params.detach().cpu(), "losses": torch.cat(losses).cpu(), "keep0": keep0,
           "target": model.target_arena.detach().cpu() if jepa else torch.zeros(1),
           "stats": module._stats.cpu(), "buckets": module.gradient_buckets() if a.world > 1 else [],
           "overlap": module.overlap_exchange}
    torch.save(out, f"{a.out}/w{a.world}_r{a.rank}.pt")
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
