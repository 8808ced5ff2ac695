"""CPU oracle for the I-JEPA step.  TEST INFRASTRUCTURE ONLY (same import rule as mae_oracle.py).

NO REFERENCE CODE, PARITY UNPINNED BY CONSTRUCTION.  The reference repository names JEPA in its title only
(README.md:1,9; pyproject.toml:2; `grep -i jepa` over *.py / *.yaml finds nothing), and BASELINE.json's configs[2] / [4]
("I-JEPA (context+target encoders, EMA target, latent MSE)") therefore have no reference implementation, no fixture and
no golden vector.  This file is OUR restatement of the specification in DESIGN.md (section "I-JEPA"), which follows the
I-JEPA paper (Assran et al., CVPR 2023, sections 3 and 4 + appendix A.1), built on the reference's own ViT pieces as
restated in mae_oracle.py (timm blocks: src/models/mae.py:28-36; sin-cos position tables as lightly builds them):

  * context encoder   f: patch embed + position row of every CONTEXT token -> blocks -> norm           (no class token)
  * target encoder    f_bar: same network, EMA weights, sees every patch, no gradient; its output is passed through a
                      parameter-free LayerNorm over the feature dimension (eps 1e-5), then the target blocks are gathered
  * predictor         g: Linear(D, Dp) on the context output + predictor position rows; one shared mask token + position row
                      per target token; for each of the nblk target blocks the sequence [context | mask tokens] runs through
                      dp blocks + norm, the mask-token rows go through Linear(Dp, D)
  * loss              mean squared error over all (image, block, token, feature) elements (BASELINE.json: "latent MSE");
                      smooth-L1 (beta 1) as the alternative the public I-JEPA code uses
  * update            AdamW on context encoder + predictor; target <- m * target + (1 - m) * context encoder
  * masks             multi-block sampler: per batch ONE target block size (scale 0.15-0.2, aspect 0.75-1.5) and ONE context
                      block size (scale 0.85-1.0, aspect 1); per image 4 target positions and 1 context position; context =
                      context block minus the union of the image's target blocks, every image truncated to the batch minimum

Tensors and names: the engine reuses the MAE parameter table, so the predictor's tensors carry the decoder's names
(decoder.decoder_embed = predictor_embed, decoder.mask_token, decoder.decoder_pos_embed, decoder.decoder_blocks.*,
decoder.decoder_norm, decoder.decoder_pred (Dp -> D)); encoder.vit.cls_token and encoder.mask_token exist but are unused.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import mae_oracle as M


@dataclass(frozen=True)
class JEPAConfig:
    image_size: int = 96
    patch_size: int = 8
    in_chans: int = 3
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    pred_embed_dim: int = 192
    pred_depth: int = 6
    pred_num_heads: int = 6
    num_target_blocks: int = 4
    target_scale: Tuple[float, float] = (0.15, 0.2)
    target_aspect: Tuple[float, float] = (0.75, 1.5)
    context_scale: Tuple[float, float] = (0.85, 1.0)
    loss: str = "mse"

    def as_mae(self) -> M.MAEConfig:
        """The MAE-shaped view used for the shared ViT pieces (decoder_* = predictor_*)."""
        return M.MAEConfig(image_size=self.image_size, patch_size=self.patch_size, in_chans=self.in_chans, embed_dim=self.embed_dim,
                           depth=self.depth, num_heads=self.num_heads, decoder_embed_dim=self.pred_embed_dim,
                           decoder_depth=self.pred_depth, decoder_num_heads=self.pred_num_heads)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


JEPA_MICRO = JEPAConfig(image_size=32, patch_size=4, embed_dim=48, depth=2, num_heads=2, pred_embed_dim=32, pred_depth=1, pred_num_heads=2)
# BASELINE.json configs[2]: ViT-S/8 96 px I-JEPA (predictor 192 wide, 6 deep: half the encoder width as in the paper's ViT-L/H -> 384 table)
JEPA_VIT_S8 = JEPAConfig()
# BASELINE.json configs[4]: ViT-L/14 224 px I-JEPA (paper table: predictor 384 x 12)
JEPA_VIT_L14 = JEPAConfig(image_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16, pred_embed_dim=384, pred_depth=12, pred_num_heads=12)


def param_shapes(cfg: JEPAConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    shapes = M.param_shapes(cfg.as_mae())
    shapes["decoder.decoder_pred.weight"] = (cfg.embed_dim, cfg.pred_embed_dim)
    shapes["decoder.decoder_pred.bias"] = (cfg.embed_dim,)
    return shapes


def init_params(cfg: JEPAConfig, seed: int = 73) -> "OrderedDict[str, torch.Tensor]":
    p = M.init_params(cfg.as_mae(), seed)
    g = torch.Generator().manual_seed(seed + 1)
    D, Dp = cfg.embed_dim, cfg.pred_embed_dim
    bound = math.sqrt(6.0 / (D + Dp))
    p["decoder.decoder_pred.weight"] = (torch.rand(D, Dp, generator=g) * 2 - 1) * bound
    p["decoder.decoder_pred.bias"] = torch.zeros(D)
    return p


ENCODER_PREFIX = "encoder.vit."
UNUSED = ("encoder.mask_token", "encoder.vit.cls_token")  # the I-JEPA ViT has no class token


def trainable_names(cfg: JEPAConfig) -> List[str]:
    return [n for n in param_shapes(cfg) if n not in M.FROZEN and n not in UNUSED]


def ema_names(cfg: JEPAConfig) -> List[str]:
    """What the target encoder tracks: every trainable tensor of the context encoder."""
    return [n for n in trainable_names(cfg) if n.startswith(ENCODER_PREFIX)]


# ----------------------------------------------------------------------------
# multi-block mask sampler (integer-exact host logic; the product's sampler must return the same ids)
# ----------------------------------------------------------------------------
def block_size(grid: int, scale: Tuple[float, float], aspect: Tuple[float, float], u_scale: float, u_aspect: float) -> Tuple[int, int]:
    s = scale[0] + u_scale * (scale[1] - scale[0])
    max_keep = int(grid * grid * s)
    a = aspect[0] + u_aspect * (aspect[1] - aspect[0])
    h = int(round(math.sqrt(max_keep * a)))
    w = int(round(math.sqrt(max_keep / a)))
    return max(1, min(h, grid - 1)), max(1, min(w, grid - 1))


def sample_masks(cfg: JEPAConfig, batch: int, gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns idx_context (B, k) and idx_target (B, nblk, m), int64 token ids in 1..N.  Plain loops on purpose."""
    g, nb = cfg.grid, cfg.num_target_blocks
    u = torch.rand(4, generator=gen, dtype=torch.float64).tolist()
    th, tw = block_size(g, cfg.target_scale, cfg.target_aspect, u[0], u[1])
    ch, cw = block_size(g, cfg.context_scale, (1.0, 1.0), u[2], u[3])
    pos = torch.rand(batch, nb + 1, 2, generator=gen, dtype=torch.float64)
    targets, contexts = [], []
    for b in range(batch):
        taken = set()
        blocks = []
        for i in range(nb):
            top = int(pos[b, i, 0].item() * (g - th + 1))
            left = int(pos[b, i, 1].item() * (g - tw + 1))
            ids = [1 + (top + r) * g + (left + c) for r in range(th) for c in range(tw)]
            blocks.append(ids)
            taken.update(ids)
        top = int(pos[b, nb, 0].item() * (g - ch + 1))
        left = int(pos[b, nb, 1].item() * (g - cw + 1))
        ctx = [1 + (top + r) * g + (left + c) for r in range(ch) for c in range(cw)]
        ctx = [t for t in ctx if t not in taken]
        if not ctx:  # degenerate draw: keep one patch outside the targets (lowest id), or patch 1
            rest = [t for t in range(1, g * g + 1) if t not in taken]
            ctx = [rest[0] if rest else 1]
        targets.append(blocks)
        contexts.append(ctx)
    k = min(len(c) for c in contexts)
    return torch.tensor([c[:k] for c in contexts], dtype=torch.int64), torch.tensor(targets, dtype=torch.int64)


# ----------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------
def encode_tokens(p, cfg: JEPAConfig, images, idx, bf16=False):
    """ViT over the patch tokens `idx` (B, n) (ids 1..N): patch embed + position rows -> blocks -> norm.  No class token."""
    mc = cfg.as_mae()
    tok = M.patch_embed_all(images, p, mc, bf16)                       # (B, N, D)
    tok = tok + p["encoder.vit.pos_embed"][:, 1:, :]
    tok = torch.gather(tok, 1, (idx - 1).unsqueeze(-1).expand(-1, -1, tok.shape[-1]))
    for i in range(cfg.depth):
        tok = M._block(tok, p, f"encoder.vit.blocks.{i}", cfg.num_heads, bf16)
    return F.layer_norm(tok, (cfg.embed_dim,), p["encoder.vit.norm.weight"], p["encoder.vit.norm.bias"], M.LN_EPS)


def target_features(p_target, cfg: JEPAConfig, images, idx_target, bf16=False):
    """h (B, nblk, m, D): target encoder on every patch, parameter-free LayerNorm, gather of the target blocks."""
    B = images.shape[0]
    all_idx = torch.arange(1, cfg.num_patches + 1).repeat(B, 1)
    with torch.no_grad():
        h = encode_tokens(p_target, cfg, images, all_idx, bf16)
        h = F.layer_norm(h, (cfg.embed_dim,))
        nb, m = idx_target.shape[1], idx_target.shape[2]
        flat = (idx_target - 1).reshape(B, nb * m)
        return torch.gather(h, 1, flat.unsqueeze(-1).expand(-1, -1, cfg.embed_dim)).reshape(B, nb, m, cfg.embed_dim)


def predict(p, cfg: JEPAConfig, x_ctx, idx_context, idx_target, bf16=False):
    """(B, nblk, m, D): the predictor on [context | mask tokens of block i], for every block."""
    B, k, _ = x_ctx.shape
    nb, m = idx_target.shape[1], idx_target.shape[2]
    Dp = cfg.pred_embed_dim
    pos = p["decoder.decoder_pos_embed"][0]                                   # (L, Dp), row = token id
    x = M._r(M._linear(x_ctx, p["decoder.decoder_embed.weight"], p["decoder.decoder_embed.bias"], bf16), bf16)
    x = x + pos[idx_context]                                                  # (B, k, Dp)
    mask = p["decoder.mask_token"].reshape(1, 1, 1, Dp) + pos[idx_target]     # (B, nb, m, Dp)
    seq = torch.cat([x.unsqueeze(1).expand(-1, nb, -1, -1), mask], dim=2).reshape(B * nb, k + m, Dp)
    for i in range(cfg.pred_depth):
        seq = M._block(seq, p, f"decoder.decoder_blocks.{i}", cfg.pred_num_heads, bf16)
    seq = F.layer_norm(seq, (Dp,), p["decoder.decoder_norm.weight"], p["decoder.decoder_norm.bias"], M.LN_EPS)
    out = M._linear(seq[:, k:, :], p["decoder.decoder_pred.weight"], p["decoder.decoder_pred.bias"], bf16)
    return out.reshape(B, nb, m, cfg.embed_dim)


def latent_loss(pred, h, kind: str):
    return F.mse_loss(pred, h) if kind == "mse" else F.smooth_l1_loss(pred, h)


def loss_and_grads(p, p_target, cfg: JEPAConfig, images, idx_context, idx_target, bf16=False):
    names = trainable_names(cfg)
    leaves = {n: (p[n].detach().clone().requires_grad_(True) if n in names else p[n].detach()) for n in p}
    h = target_features(p_target, cfg, images, idx_target, bf16)
    x_ctx = encode_tokens(leaves, cfg, images, idx_context, bf16)
    pred = predict(leaves, cfg, x_ctx, idx_context, idx_target, bf16)
    loss = latent_loss(pred, h, cfg.loss)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names])
    return loss.detach(), OrderedDict(zip(names, grads)), dict(h=h, pred=pred.detach(), x_context=x_ctx.detach())


def ema_update(p_target, p, cfg: JEPAConfig, momentum: float) -> None:
    for n in ema_names(cfg):
        p_target[n].mul_(momentum).add_(p[n], alpha=1.0 - momentum)


def ema_momentum_at(step: int, total_steps: int, start: float = 0.996, end: float = 1.0) -> float:
    """Linear schedule of the paper (0.996 -> 1.0 over training)."""
    return start + (end - start) * min(step, total_steps) / max(1, total_steps)


def train_step(p, p_target, cfg: JEPAConfig, state, images, idx_context, idx_target, lr, step, momentum, weight_decay=0.05, bf16=False):
    """fwd + loss + bwd + AdamW (unclipped, as I-JEPA trains) + EMA of the target encoder."""
    loss, grads, aux = loss_and_grads(p, p_target, cfg, images, idx_context, idx_target, bf16)
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    M.adamw_step(p, grads, state, lr, step, weight_decay)
    ema_update(p_target, p, cfg, momentum)
    aux.update(grad_norm=total, grads=grads)
    return loss, aux


def flops_per_image_step(cfg: JEPAConfig, k: int, m: int) -> float:
    """2*MAC, GEMMs + attention matmuls.  Context encoder and predictor: forward + backward = 3 x forward; target encoder
    (every patch, no gradient): 1 x forward."""
    D, Dp, N, P, nb = cfg.embed_dim, cfg.pred_embed_dim, cfg.num_patches, cfg.patch_size ** 2 * cfg.in_chans, cfg.num_target_blocks
    enc = lambda t: 2 * t * P * D + cfg.depth * t * 24 * D * D + cfg.depth * 4 * t * t * D  # noqa: E731
    T = k + m
    pred = 2 * k * D * Dp + nb * (cfg.pred_depth * T * 24 * Dp * Dp + cfg.pred_depth * 4 * T * T * Dp + 2 * m * Dp * D)
    return 3.0 * (enc(k) + pred) + enc(N)
