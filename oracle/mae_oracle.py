"""CPU oracle for the MAE pretrain-step hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product path (``ssrl_vit_mae_jepa_amd``) never
does: it fails loudly when ``libmae_hip.so`` is missing.

PARITY UNPINNED.  The reference's arithmetic for this path lives in third-party
packages that are neither vendored in ``/root/reference`` nor installed here:
``lightly==1.5.22`` (uv.lock:866-867), ``timm==1.0.21`` (uv.lock:2163-2164),
``pytorch-lightning==2.5.6`` (uv.lock:1772-1773).  The reference has no test or
golden vector touching ``src/``.  This file restates the published algorithms of
those packages in plain fp32 torch, anchored on the reference's own call sites,
and is pinned only by the structure known-answers the reference records
(notebook.ipynb:987-994: 2.0 M params / 48.7 K non-trainable / 8.140 MB;
see ``param_census`` and tests/test_oracle_kat.py) plus a block-level cross-check
against the independent MAE in ``transformers`` (tests/test_oracle_crosscheck.py).

What follows what:
  * ``MAEConfig``            <- src/models/mae.py:15-52 (ctor dict defaults)
  * ``init_params``          <- lightly MaskedVisionTransformerTIMM/MAEDecoderTIMM
                                init as used at src/models/mae.py:38,45
  * ``mask_from_noise``      <- lightly utils.random_token_mask, called at
                                src/models/mae.py:79-83
  * ``forward_encoder``      <- src/models/mae.py:54-55 (lightly encode + timm blocks)
  * ``forward_decoder``      <- src/models/mae.py:57-75
  * ``forward``              <- src/models/mae.py:77-94
  * ``mse_loss``             <- src/training/mae.py:40,48
  * ``clip_grad_norm``       <- scripts/training/pretrain_mae.py:124-125
  * ``adamw_step``           <- src/training/mae.py:59-65 (torch.optim.AdamW defaults)
  * ``lr_lambda``/``mask_ratio_at`` <- src/training/mae.py:67-83
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # timm VisionTransformer / lightly MAEDecoderTIMM: partial(LayerNorm, eps=1e-6)


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class MAEConfig:
    image_size: int = 96
    patch_size: int = 6
    in_chans: int = 3
    mask_ratio: float = 0.75
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    decoder_embed_dim: int = 512
    decoder_depth: int = 4
    decoder_num_heads: int = 6
    mlp_ratio: int = 4

    @staticmethod
    def from_dicts(general: dict, encoder: dict, decoder: dict) -> "MAEConfig":
        # defaults exactly as src/models/mae.py:23-26, 32-34, 49-51
        return MAEConfig(
            mask_ratio=general.get("mask_ratio", 0.75),
            image_size=general.get("image_size", 96),
            patch_size=general.get("patch_size", 6),
            in_chans=general.get("in_chans", 3),
            embed_dim=encoder.get("embed_dim", 384),
            depth=encoder.get("depth", 12),
            num_heads=encoder.get("num_heads", 6),
            decoder_embed_dim=decoder.get("decoder_embed_dim", 512),
            decoder_depth=decoder.get("decoder_depth", 4),
            decoder_num_heads=decoder.get("decoder_num_heads", 6),
        )

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def sequence_length(self) -> int:  # src/models/mae.py:39-43
        return self.num_patches + 1

    @property
    def patch_dim(self) -> int:
        return self.patch_size * self.patch_size * self.in_chans

    def num_keep(self, mask_ratio: Optional[float] = None) -> int:
        r = self.mask_ratio if mask_ratio is None else mask_ratio
        # lightly random_token_mask: int(L*(1-r)), then max(1, .) because cls is never masked
        return max(1, int(self.sequence_length * (1 - r)))

    def validate(self) -> None:
        if self.image_size % self.patch_size:
            raise ValueError("image_size must be divisible by patch_size")
        if self.embed_dim % self.num_heads:  # timm Attention assert
            raise ValueError("dim should be divisible by num_heads")
        if self.decoder_embed_dim % self.decoder_num_heads:
            raise ValueError("decoder dim should be divisible by decoder_num_heads")


YAML_TINY = MAEConfig(image_size=96, patch_size=8, in_chans=3, embed_dim=144, depth=4, num_heads=6,
                      decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6)
VIT_S8_YAMLDEC = MAEConfig(image_size=96, patch_size=8, in_chans=3, embed_dim=384, depth=12, num_heads=6,
                           decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6)
# the code-default decoder width / depth (src/models/mae.py:49-50) with a valid head count
VIT_S8_DEC512 = MAEConfig(image_size=96, patch_size=8, in_chans=3, embed_dim=384, depth=12, num_heads=6,
                          decoder_embed_dim=512, decoder_depth=4, decoder_num_heads=8)
# BASELINE.json configs[3] / SURVEY 8d config 4: ViT-B/16 224 px with the MAE-paper decoder (512 x 8 x 16 heads);
# L = 197, k = 49, m = 148, P = 768.  The reference's ctor accepts it (src/models/mae.py:28-52) but never ran it.
VIT_B16_DEC512 = MAEConfig(image_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                           decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16)


# ----------------------------------------------------------------------------
# parameters (state_dict names of SURVEY.md 8b)
# ----------------------------------------------------------------------------
def _block_shapes(prefix: str, d: int, mlp_ratio: int) -> List[Tuple[str, Tuple[int, ...]]]:
    h = d * mlp_ratio
    return [
        (f"{prefix}.norm1.weight", (d,)), (f"{prefix}.norm1.bias", (d,)),
        (f"{prefix}.attn.qkv.weight", (3 * d, d)), (f"{prefix}.attn.qkv.bias", (3 * d,)),
        (f"{prefix}.attn.proj.weight", (d, d)), (f"{prefix}.attn.proj.bias", (d,)),
        (f"{prefix}.norm2.weight", (d,)), (f"{prefix}.norm2.bias", (d,)),
        (f"{prefix}.mlp.fc1.weight", (h, d)), (f"{prefix}.mlp.fc1.bias", (h,)),
        (f"{prefix}.mlp.fc2.weight", (d, h)), (f"{prefix}.mlp.fc2.bias", (d,)),
    ]


def param_shapes(cfg: MAEConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every tensor of ``MaskedAutoencoder.state_dict()`` in registration order."""
    D, Dd, L, p, C = cfg.embed_dim, cfg.decoder_embed_dim, cfg.sequence_length, cfg.patch_size, cfg.in_chans
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("encoder.mask_token", (1, 1, D)),
        ("encoder.vit.cls_token", (1, 1, D)),
        ("encoder.vit.pos_embed", (1, L, D)),
        ("encoder.vit.patch_embed.proj.weight", (D, C, p, p)),
        ("encoder.vit.patch_embed.proj.bias", (D,)),
    ]
    for i in range(cfg.depth):
        out += _block_shapes(f"encoder.vit.blocks.{i}", D, cfg.mlp_ratio)
    out += [("encoder.vit.norm.weight", (D,)), ("encoder.vit.norm.bias", (D,))]
    out += [
        ("decoder.mask_token", (1, 1, Dd)),
        ("decoder.decoder_pos_embed", (1, L, Dd)),
        ("decoder.decoder_embed.weight", (Dd, D)), ("decoder.decoder_embed.bias", (Dd,)),
    ]
    for i in range(cfg.decoder_depth):
        out += _block_shapes(f"decoder.decoder_blocks.{i}", Dd, cfg.mlp_ratio)
    out += [
        ("decoder.decoder_norm.weight", (Dd,)), ("decoder.decoder_norm.bias", (Dd,)),
        ("decoder.decoder_pred.weight", (cfg.patch_dim, Dd)), ("decoder.decoder_pred.bias", (cfg.patch_dim,)),
    ]
    return OrderedDict(out)


FROZEN = ("encoder.vit.pos_embed", "decoder.decoder_pos_embed")  # sin-cos, requires_grad=False
# encoder.mask_token is trainable but never reached by the MAE forward (encode() is called with
# idx_mask=None, src/models/mae.py:55), so its grad is None and AdamW/clip skip it.
NO_GRAD_ON_PATH = ("encoder.mask_token",)


def param_census(cfg: MAEConfig) -> Dict[str, int]:
    shapes = param_shapes(cfg)
    total = sum(math.prod(s) for s in shapes.values())
    frozen = sum(math.prod(shapes[n]) for n in FROZEN)
    return {"total": total, "frozen": frozen, "trainable": total - frozen, "bytes_fp32": 4 * total}


def module_inventory(cfg: MAEConfig) -> List[str]:
    """Names of the nn.Module objects the reference model tree holds (timm 1.0.21 + lightly 1.5.22),
    below the LightningModule root: used only for the 152-modules known answer (notebook.ipynb:987-994)."""
    def block(pfx):
        return [pfx, f"{pfx}.norm1", f"{pfx}.attn", f"{pfx}.attn.qkv", f"{pfx}.attn.q_norm", f"{pfx}.attn.k_norm",
                f"{pfx}.attn.attn_drop", f"{pfx}.attn.norm", f"{pfx}.attn.proj", f"{pfx}.attn.proj_drop",
                f"{pfx}.ls1", f"{pfx}.drop_path1", f"{pfx}.norm2", f"{pfx}.mlp", f"{pfx}.mlp.fc1", f"{pfx}.mlp.act",
                f"{pfx}.mlp.drop1", f"{pfx}.mlp.norm", f"{pfx}.mlp.fc2", f"{pfx}.mlp.drop2", f"{pfx}.ls2",
                f"{pfx}.drop_path2"]
    names = ["model", "model.encoder", "model.encoder.vit", "model.encoder.vit.patch_embed",
             "model.encoder.vit.patch_embed.proj", "model.encoder.vit.patch_embed.norm", "model.encoder.vit.pos_drop",
             "model.encoder.vit.patch_drop", "model.encoder.vit.norm_pre", "model.encoder.vit.blocks"]
    for i in range(cfg.depth):
        names += block(f"model.encoder.vit.blocks.{i}")
    names += ["model.encoder.vit.norm", "model.encoder.vit.fc_norm", "model.encoder.vit.head_drop",
              "model.encoder.vit.head", "model.decoder", "model.decoder.decoder_embed", "model.decoder.decoder_blocks"]
    for i in range(cfg.decoder_depth):
        names += block(f"model.decoder.decoder_blocks.{i}")
    names += ["model.decoder.decoder_norm", "model.decoder.decoder_pred", "criterion"]
    return names


def sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = True) -> torch.Tensor:
    """2-D sin-cos table as MAE-official / lightly build it (first half from the w coordinate)."""
    assert embed_dim % 4 == 0
    gh = torch.arange(grid_size, dtype=torch.float32)
    gw = torch.arange(grid_size, dtype=torch.float32)
    grid = torch.stack(torch.meshgrid(gw, gh, indexing="xy"), dim=0).reshape(2, -1)

    def one_d(dim: int, pos: torch.Tensor) -> torch.Tensor:
        omega = torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0)
        omega = 1.0 / (10000.0 ** omega)
        out = torch.einsum("m,d->md", pos.reshape(-1), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)

    emb = torch.cat([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], dim=1)
    if cls_token:
        emb = torch.cat([torch.zeros(1, embed_dim), emb], dim=0)
    return emb.unsqueeze(0)


def init_params(cfg: MAEConfig, seed: int = 73) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 parameters following the lightly/timm init recipe: Linear xavier-uniform / zero bias,
    LayerNorm 1/0, cls and mask tokens N(0, .02), patch projection xavier on its (D, C*p*p) view,
    frozen sin-cos position tables.  RNG order is ours (init-order parity with timm is unverifiable)."""
    g = torch.Generator().manual_seed(seed)
    params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith("pos_embed"):
            dim = shape[-1]
            params[name] = sincos_pos_embed(dim, cfg.grid, cls_token=True).contiguous()
        elif name.endswith("mask_token") or name.endswith("cls_token"):
            params[name] = torch.randn(shape, generator=g) * 0.02
        elif ".norm" in name or name.endswith("decoder_norm.weight") or name.endswith("decoder_norm.bias"):
            params[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith(".bias"):
            params[name] = torch.zeros(shape)
        else:  # Linear / conv-as-linear weight: xavier uniform on the 2-D view
            fan_out, fan_in = shape[0], math.prod(shape[1:])
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            params[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return params


def randomize_params(params: "OrderedDict[str, torch.Tensor]", seed: int = 7, scale: float = 0.05) -> None:
    """Perturb biases / LayerNorm affine / tokens in place so tests exercise every term (zero biases hide bugs)."""
    g = torch.Generator().manual_seed(seed)
    for n, t in params.items():
        if n in FROZEN:
            continue
        if n.endswith(".bias") or ".norm" in n or "decoder_norm" in n or n.endswith("_token"):
            t.add_(torch.randn(t.shape, generator=g) * scale)


# ----------------------------------------------------------------------------
# masking (lightly utils.random_token_mask; src/models/mae.py:79-83)
# ----------------------------------------------------------------------------
def make_noise(batch: int, seq_len: int, generator: torch.Generator) -> torch.Tensor:
    return torch.rand(batch, seq_len, generator=generator)


def mask_from_noise(noise: torch.Tensor, num_keep: int) -> Tuple[torch.Tensor, torch.Tensor]:
    noise = noise.clone()
    noise[:, 0] = -1  # the class token is never masked
    indices = torch.argsort(noise, dim=1)
    return indices[:, :num_keep], indices[:, num_keep:]


# ----------------------------------------------------------------------------
# building blocks (timm 1.0.21 VisionTransformer pieces)
# ----------------------------------------------------------------------------
def _r(x: torch.Tensor, bf16: bool) -> torch.Tensor:
    """bf16-emulation: round a GEMM operand to bf16 and back (fp32 accumulate stays)."""
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _linear(x, w, b, bf16):
    return F.linear(_r(x, bf16), _r(w, bf16), b)


def _attention(x, p, pfx, heads, bf16):
    B, T, C = x.shape
    hd = C // heads
    qkv = _linear(x, p[f"{pfx}.attn.qkv.weight"], p[f"{pfx}.attn.qkv.bias"], bf16)
    qkv = _r(qkv, bf16).reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    o = (_r(att, bf16) @ v).transpose(1, 2).reshape(B, T, C)
    # under bf16-mixed autocast the Linear output is bf16 before it is added to the fp32 residual stream
    return _r(_linear(_r(o, bf16), p[f"{pfx}.attn.proj.weight"], p[f"{pfx}.attn.proj.bias"], bf16), bf16)


def _block(x, p, pfx, heads, bf16):
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), p[f"{pfx}.norm1.weight"], p[f"{pfx}.norm1.bias"], LN_EPS)
    x = x + _attention(h, p, pfx, heads, bf16)
    h = F.layer_norm(x, (C,), p[f"{pfx}.norm2.weight"], p[f"{pfx}.norm2.bias"], LN_EPS)
    h = _linear(h, p[f"{pfx}.mlp.fc1.weight"], p[f"{pfx}.mlp.fc1.bias"], bf16)
    h = F.gelu(_r(h, bf16))  # nn.GELU() default = exact erf
    h = _r(_linear(h, p[f"{pfx}.mlp.fc2.weight"], p[f"{pfx}.mlp.fc2.bias"], bf16), bf16)
    return x + h


def patch_embed_all(images, p, cfg: MAEConfig, bf16=False):
    """timm PatchEmbed: Conv2d(C, D, k=s=patch) -> flatten(2).transpose(1,2): (B, N, D), patches row-major."""
    B, C, H, W = images.shape
    if H != cfg.image_size or W != cfg.image_size:
        raise ValueError(f"Input size ({H}x{W}) doesn't match model ({cfg.image_size})")
    x = F.conv2d(_r(images, bf16), _r(p["encoder.vit.patch_embed.proj.weight"], bf16),
                 p["encoder.vit.patch_embed.proj.bias"], stride=cfg.patch_size)
    return x.flatten(2).transpose(1, 2)


def forward_encoder(p, cfg: MAEConfig, images, idx_keep=None, bf16=False):
    """src/models/mae.py:54-55 -> lightly MaskedVisionTransformerTIMM.encode(images, idx_keep)."""
    B = images.shape[0]
    tok = patch_embed_all(images, p, cfg, bf16)
    tok = torch.cat([p["encoder.vit.cls_token"].expand(B, -1, -1), tok], dim=1)
    tok = tok + p["encoder.vit.pos_embed"]
    if idx_keep is not None:
        tok = torch.gather(tok, 1, idx_keep.unsqueeze(-1).expand(-1, -1, tok.shape[-1]))
    for i in range(cfg.depth):
        tok = _block(tok, p, f"encoder.vit.blocks.{i}", cfg.num_heads, bf16)
    D = cfg.embed_dim
    return F.layer_norm(tok, (D,), p["encoder.vit.norm.weight"], p["encoder.vit.norm.bias"], LN_EPS)


def forward_decoder(p, cfg: MAEConfig, x_encoded, idx_keep, idx_mask, bf16=False):
    """src/models/mae.py:57-75."""
    B = x_encoded.shape[0]
    Dd, L = cfg.decoder_embed_dim, cfg.sequence_length
    x_decode = _r(_linear(x_encoded, p["decoder.decoder_embed.weight"], p["decoder.decoder_embed.bias"], bf16), bf16)
    x_masked = p["decoder.mask_token"].repeat(B, L, 1)
    x_masked = torch.scatter(x_masked, 1, idx_keep.unsqueeze(-1).expand(-1, -1, Dd), x_decode.type_as(x_masked))
    x = x_masked + p["decoder.decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = _block(x, p, f"decoder.decoder_blocks.{i}", cfg.decoder_num_heads, bf16)
    x = F.layer_norm(x, (Dd,), p["decoder.decoder_norm.weight"], p["decoder.decoder_norm.bias"], LN_EPS)
    x_pred = torch.gather(x, 1, idx_mask.unsqueeze(-1).expand(-1, -1, Dd))
    return _linear(x_pred, p["decoder.decoder_pred.weight"], p["decoder.decoder_pred.bias"], bf16)


def patchify(images, patch_size):
    """lightly utils.patchify: (B,C,H,W) -> (B, N, p*p*C), per-patch order (py, px, c)."""
    N, C, H, W = images.shape
    if H != W or H % patch_size:
        raise ValueError("patchify needs a square image divisible by patch_size")
    g = H // patch_size
    x = images.reshape(N, C, g, patch_size, g, patch_size)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(N, g * g, patch_size * patch_size * C)


def unpatchify(patches, patch_size, channels=3):
    N, n, _ = patches.shape
    g = int(round(math.sqrt(n)))
    x = patches.reshape(N, g, g, patch_size, patch_size, channels)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(N, channels, g * patch_size, g * patch_size)


def build_target(images, idx_mask, cfg: MAEConfig):
    """src/models/mae.py:90-92."""
    patches = patchify(images, cfg.patch_size)
    idx = torch.clamp(idx_mask - 1, min=0)
    return torch.gather(patches, 1, idx.unsqueeze(-1).expand(-1, -1, patches.shape[-1]))


def forward(p, cfg: MAEConfig, images, noise, mask_ratio=None, bf16=False):
    """src/models/mae.py:77-94 with the noise draw made an explicit input."""
    idx_keep, idx_mask = mask_from_noise(noise, cfg.num_keep(mask_ratio))
    x_enc = forward_encoder(p, cfg, images, idx_keep, bf16)
    x_pred = forward_decoder(p, cfg, x_enc, idx_keep, idx_mask, bf16)
    target = build_target(images, idx_mask, cfg)
    return x_pred, target, idx_keep, idx_mask, x_enc


def mse_loss(pred, target):
    return F.mse_loss(pred, target)  # torch.nn.MSELoss() default: mean over all elements


# ----------------------------------------------------------------------------
# step semantics: clip + AdamW + schedules
# ----------------------------------------------------------------------------
def trainable_names(cfg: MAEConfig) -> List[str]:
    """Parameters that receive a gradient on the MAE path (what clip + AdamW actually touch)."""
    return [n for n in param_shapes(cfg) if n not in FROZEN and n not in NO_GRAD_ON_PATH]


def loss_and_grads(p, cfg, images, noise, mask_ratio=None, bf16=False):
    names = trainable_names(cfg)
    leaves = {n: (p[n].detach().clone().requires_grad_(True) if n in names else p[n].detach()) for n in p}
    x_pred, target, idx_keep, idx_mask, x_enc = forward(leaves, cfg, images, noise, mask_ratio, bf16)
    loss = mse_loss(x_pred, target)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names])
    return loss.detach(), OrderedDict(zip(names, grads)), dict(
        x_pred=x_pred.detach(), target=target.detach(), idx_keep=idx_keep, idx_mask=idx_mask, x_encoded=x_enc.detach())


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float = 1.0) -> Tuple[torch.Tensor, float]:
    """torch.nn.utils.clip_grad_norm_(L2): total = ||[||g_i||]||_2; coef = min(1, max_norm/(total+1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = float(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    for g in grads.values():
        g.mul_(coef)
    return total, coef


def adamw_step(p, grads, state, lr, step, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW single-group update (decoupled decay on every parameter that has a grad)."""
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for n, g in grads.items():
        m, v = state.setdefault(n, (torch.zeros_like(g), torch.zeros_like(g)))
        p[n].mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p[n].addcdiv_(m, denom, value=-lr / bc1)


def effective_lr(base_lr: float, batch_size: int) -> float:
    return base_lr * batch_size / 256  # src/training/mae.py:60


def lr_lambda(epoch: int, warmup_epochs: int, total_epochs: int) -> float:
    warmup = (epoch + 1) / max(1, warmup_epochs)  # src/training/mae.py:67-70
    cosine = 0.5 * (1 + math.cos(math.pi * epoch / total_epochs))
    return min(warmup, 1.0) * cosine


def mask_ratio_at(epoch: int, start: float, end: float, ramp_epochs: int) -> float:
    progress = min(epoch / max(1, ramp_epochs - 1), 1.0)  # src/training/mae.py:80-81
    return start + progress * (end - start)


def train_step(p, cfg, state, images, noise, lr, step, mask_ratio=None, weight_decay=0.05, max_norm=1.0, bf16=False):
    """zero_grad + fwd + MSE + bwd + clip(1.0) + AdamW: one Lightning automatic-optimisation step."""
    loss, grads, aux = loss_and_grads(p, cfg, images, noise, mask_ratio, bf16)
    total, coef = clip_grad_norm(grads, max_norm)
    adamw_step(p, grads, state, lr, step, weight_decay)
    aux.update(grad_norm=total, clip_coef=coef, grads=grads)
    return loss, aux


def synthetic_images(batch: int, cfg: MAEConfig, seed: int = 73) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)  # range of ToTensor+Normalize(.5,.5): src/data.py:22-23
    return torch.rand(batch, cfg.in_chans, cfg.image_size, cfg.image_size, generator=g) * 2 - 1


def flops_per_image_step(cfg: MAEConfig, mask_ratio: Optional[float] = None) -> float:
    """BASELINE.md section 3 formula; train step = 3 x forward."""
    L, k = cfg.sequence_length, cfg.num_keep(mask_ratio)
    m, P, D, Dd = L - k, cfg.patch_dim, cfg.embed_dim, cfg.decoder_embed_dim
    fwd = (2 * (k - 1) * P * D + cfg.depth * k * 24 * D * D + cfg.depth * 4 * k * k * D + 2 * k * D * Dd
           + cfg.decoder_depth * L * 24 * Dd * Dd + cfg.decoder_depth * 4 * L * L * Dd + 2 * m * Dd * P)
    return 3.0 * fwd
