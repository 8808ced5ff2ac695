#!/usr/bin/env python3
"""Generates tests/golden/mae_micro.npz from the CPU oracle (oracle/mae_oracle.py).

PARITY UNPINNED w.r.t. lightly/timm: the reference's arithmetic cannot be imported here (packages absent, no network)
and the reference holds no fixture for this path, so these vectors come from our restatement.  They pin (a) the oracle
against drift and (b) the HIP fp32 engine against the oracle on committed inputs.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mae_oracle as O  # noqa: E402

torch.set_float32_matmul_precision("highest")
MICRO = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=2, num_heads=2,
                    decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)


def case(B, r, seed):
    cfg = MICRO
    params = O.init_params(cfg, 73)
    O.randomize_params(params)
    images = O.synthetic_images(B, cfg, seed=seed)
    noise = O.make_noise(B, cfg.sequence_length, torch.Generator().manual_seed(seed + 1))
    loss, grads, aux = O.loss_and_grads(params, cfg, images, noise, r)
    out = {"images": images, "noise": noise, "idx_keep": aux["idx_keep"], "idx_mask": aux["idx_mask"], "loss": loss,
           "x_pred": aux["x_pred"], "target": aux["target"], "x_encoded": aux["x_encoded"],
           "grad_norm": torch.linalg.vector_norm(torch.stack([g.norm() for g in grads.values()]))}
    out["grad_norms"] = torch.stack([g.norm() for g in grads.values()])
    for n in ("encoder.vit.cls_token", "decoder.mask_token", "encoder.vit.blocks.0.attn.qkv.bias", "decoder.decoder_pred.weight",
              "encoder.vit.patch_embed.proj.weight", "decoder.decoder_blocks.0.norm2.weight"):
        out["grad/" + n] = grads[n]
    # two optimizer steps at the YAML schedule's epoch-0 learning rate
    state, lr = {}, O.effective_lr(1.5e-4, 2000) * O.lr_lambda(0, 20, 800)
    p2 = {k: v.clone() for k, v in params.items()}
    for step in (1, 2):
        O.train_step(p2, cfg, state, images, noise, lr, step, r)
    out["param_norms_after_2_steps"] = torch.stack([p2[n].norm() for n in O.trainable_names(cfg)])
    out["cls_after_2_steps"] = p2["encoder.vit.cls_token"]
    return out


def main():
    blob = {}
    for tag, (B, r, seed) in {"b2_r75": (2, 0.75, 11), "b5_r50": (5, 0.5, 12)}.items():
        for k, v in case(B, r, seed).items():
            blob[f"{tag}/{k}"] = v.numpy()
    # tie case: equal noise values; pinned to the index-order (stable) permutation the HIP kernel produces
    noise = torch.rand(3, 17, generator=torch.Generator().manual_seed(5))
    noise[0, 5] = noise[0, 11]; noise[1, 2:7] = 0.25; noise[2, :] = 0.5
    ref = noise.clone(); ref[:, 0] = -1
    blob["ties/noise"] = noise.numpy()
    blob["ties/order_stable"] = torch.argsort(ref, dim=1, stable=True).numpy()
    np.savez_compressed(Path(__file__).with_name("mae_micro.npz"), **blob)
    print("wrote", Path(__file__).with_name("mae_micro.npz"), len(blob), "arrays")


if __name__ == "__main__":
    main()
