#!/usr/bin/env python3
"""Generates tests/golden/jepa_micro.npz from the CPU oracle (oracle/jepa_oracle.py).

PARITY UNPINNED: the reference holds no I-JEPA code (README.md:1,9 name it only), so these vectors come from our
restatement of DESIGN.md section 9.  They pin (a) the oracle and the mask sampler against drift and (b) the HIP fp32
engine against the oracle on committed inputs.  Run from the repo root:
    python tests/golden/make_golden_jepa.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import jepa_oracle as J  # noqa: E402

torch.set_float32_matmul_precision("highest")
CFG = J.JEPA_MICRO
PINNED_GRADS = ("encoder.vit.patch_embed.proj.weight", "encoder.vit.blocks.1.attn.proj.weight", "encoder.vit.norm.bias",
                "decoder.mask_token", "decoder.decoder_embed.weight", "decoder.decoder_pred.bias")


def weights(cfg, seed=73):
    """Context/predictor weights and a target encoder that has drifted from the context encoder."""
    params = J.init_params(cfg, seed)
    J.M.randomize_params(params)
    target = {k: v.clone() for k, v in params.items()}
    g = torch.Generator().manual_seed(seed + 5)
    for n in J.ema_names(cfg):
        target[n] = target[n] + 0.02 * torch.randn(target[n].shape, generator=g)
    return params, target


def case(B, seed, loss):
    cfg = J.JEPAConfig(**{**CFG.__dict__, "loss": loss})
    params, target = weights(cfg)
    images = J.M.synthetic_images(B, cfg.as_mae(), seed=seed)
    ctx, tgt = J.sample_masks(cfg, B, torch.Generator().manual_seed(seed + 1))
    l, grads, aux = J.loss_and_grads(params, target, cfg, images, ctx, tgt)
    out = {"images": images, "idx_context": ctx, "idx_target": tgt, "loss": l, "h": aux["h"], "pred": aux["pred"],
           "grad_norms": torch.stack([g.norm() for g in grads.values()])}
    for n in PINNED_GRADS:
        out["grad/" + n] = grads[n]
    # two whole steps (AdamW unclipped + EMA) at a fixed learning rate and momentum
    p2, t2, state = {k: v.clone() for k, v in params.items()}, {k: v.clone() for k, v in target.items()}, {}
    for step in (1, 2):
        J.train_step(p2, t2, cfg, state, images, ctx, tgt, 1e-3, step, 0.99)
    out["param_norms_after_2_steps"] = torch.stack([p2[n].norm() for n in J.trainable_names(cfg)])
    out["target_norms_after_2_steps"] = torch.stack([t2[n].norm() for n in J.ema_names(cfg)])
    return out


def main():
    blob = {}
    for tag, (B, seed, loss) in {"b3_mse": (3, 21, "mse"), "b4_sl1": (4, 22, "smooth_l1")}.items():
        for k, v in case(B, seed, loss).items():
            blob[f"{tag}/{k}"] = v.numpy()
    # the sampler on the headline geometry (12 x 12 grid): ids only
    ctx, tgt = J.sample_masks(J.JEPA_VIT_S8, 6, torch.Generator().manual_seed(3))
    blob["sampler_vits8/idx_context"], blob["sampler_vits8/idx_target"] = ctx.numpy(), tgt.numpy()
    np.savez_compressed(Path(__file__).with_name("jepa_micro.npz"), **blob)
    print("wrote", Path(__file__).with_name("jepa_micro.npz"), len(blob), "arrays")


if __name__ == "__main__":
    main()
