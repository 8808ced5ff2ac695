"""Independent cross-check of the oracle's restated timm pieces (CPU): the pre-LN transformer block, exact-erf GELU,
qkv split order and patchify of the MAE implementation shipped in `transformers` (constructed from a local config, no
download), fed the same weights.  Not the reference -- a second, unrelated implementation of the same published model."""
import pytest
import torch

from oracle import mae_oracle as O

tr = pytest.importorskip("transformers")


@pytest.mark.parametrize("D,H,T", [(48, 2, 5), (144, 6, 36), (192, 6, 17)])
def test_block_matches_transformers_vitmae_layer(D, H, T):
    from transformers import ViTMAEConfig
    from transformers.models.vit_mae import modeling_vit_mae as M
    cfg = ViTMAEConfig(hidden_size=D, num_hidden_layers=1, num_attention_heads=H, intermediate_size=4 * D, image_size=32,
                       patch_size=8, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True,
                       attn_implementation="eager", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    layer = M.ViTMAELayer(cfg).eval()
    g = torch.Generator().manual_seed(D + T)
    p = {}
    for n, shape in O._block_shapes("blk", D, 4):
        p[n] = torch.randn(shape, generator=g) * (0.2 if n.endswith("weight") and len(shape) == 2 else 0.5)
    sd = layer.state_dict()
    qkv_w, qkv_b = p["blk.attn.qkv.weight"], p["blk.attn.qkv.bias"]
    # timm packs q, k, v as three consecutive row blocks of qkv.weight (reshape(B,N,3,H,hd))
    sd["attention.q_proj.weight" if "attention.q_proj.weight" in sd else "attention.attention.query.weight"] = qkv_w[:D]
    names = {k.split(".")[-2] for k in sd}
    def put(prefixes, w, b):
        for pre in prefixes:
            if pre + ".weight" in sd:
                sd[pre + ".weight"], sd[pre + ".bias"] = w, b
                return
        raise KeyError(prefixes)
    put(["attention.q_proj", "attention.attention.query"], qkv_w[:D], qkv_b[:D])
    put(["attention.k_proj", "attention.attention.key"], qkv_w[D:2 * D], qkv_b[D:2 * D])
    put(["attention.v_proj", "attention.attention.value"], qkv_w[2 * D:], qkv_b[2 * D:])
    put(["attention.o_proj", "attention.output.dense"], p["blk.attn.proj.weight"], p["blk.attn.proj.bias"])
    put(["layernorm_before"], p["blk.norm1.weight"], p["blk.norm1.bias"])
    put(["layernorm_after"], p["blk.norm2.weight"], p["blk.norm2.bias"])
    put(["mlp.fc1", "intermediate.dense"], p["blk.mlp.fc1.weight"], p["blk.mlp.fc1.bias"])
    put(["mlp.fc2", "output.dense"], p["blk.mlp.fc2.weight"], p["blk.mlp.fc2.bias"])
    layer.load_state_dict(sd)
    x = torch.randn(3, T, D, generator=g)
    with torch.no_grad():
        ref = layer(x)
        ref = ref[0] if isinstance(ref, tuple) else ref
        got = O._block(x, p, "blk", H, False)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5), float((got - ref).abs().max())


def test_patchify_matches_transformers():
    from transformers import ViTMAEConfig, ViTMAEForPreTraining
    cfg = ViTMAEConfig(hidden_size=48, num_hidden_layers=1, num_attention_heads=2, intermediate_size=96, image_size=32,
                       patch_size=8, num_channels=3, decoder_hidden_size=32, decoder_num_hidden_layers=1,
                       decoder_num_attention_heads=2, decoder_intermediate_size=64)
    model = ViTMAEForPreTraining(cfg)
    img = torch.randn(2, 3, 32, 32)
    assert torch.equal(model.patchify(img), O.patchify(img, 8))


# ---------------------------------------------------------------------------------------------------------------------
# Whole model: the wiring the oracle restates from /root/reference/src/models/mae.py:54-94 (class token prepended, position
# table added, visible tokens gathered; decoder embed, mask-token fill, scatter, position table, decode, gather of the masked
# tokens, predict; pixel targets; MSE) against transformers' ViTMAEForPreTraining fed the SAME weights, pixels and noise.
# Two convention differences, reconciled explicitly:
#   * lightly draws noise over all L = N + 1 tokens, forces the class token first and keeps int(L (1 - r)) tokens INCLUDING
#     it; transformers draws noise over the N patches and keeps int(N (1 - r')) patches PLUS the class token.  With the
#     oracle's noise columns 1.. as transformers' noise and r' chosen so that both keep the same count, the kept / masked
#     sets and their order (ascending noise) coincide.
#   * transformers predicts every token and averages per-patch means over the masked patches; the reference predicts the
#     masked tokens only and takes nn.MSELoss over them: equal when every row masks the same number of patches (it does).
# ---------------------------------------------------------------------------------------------------------------------
def _hf_model(cfg: O.MAEConfig, hf_ratio: float):
    from transformers import ViTMAEConfig, ViTMAEForPreTraining
    c = ViTMAEConfig(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.num_heads,
                     intermediate_size=4 * cfg.embed_dim, image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.in_chans,
                     layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, decoder_hidden_size=cfg.decoder_embed_dim,
                     decoder_num_hidden_layers=cfg.decoder_depth, decoder_num_attention_heads=cfg.decoder_num_heads,
                     decoder_intermediate_size=4 * cfg.decoder_embed_dim, mask_ratio=hf_ratio, norm_pix_loss=False,
                     attn_implementation="eager", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return ViTMAEForPreTraining(c).eval()


def _load_block(sd, pre, p, opre, D):
    def put(names, w, b):
        for n in names:
            if f"{pre}.{n}.weight" in sd:
                sd[f"{pre}.{n}.weight"], sd[f"{pre}.{n}.bias"] = w.clone(), b.clone()
                return
        raise KeyError((pre, names))
    qw, qb = p[f"{opre}.attn.qkv.weight"], p[f"{opre}.attn.qkv.bias"]
    put(["attention.q_proj", "attention.attention.query"], qw[:D], qb[:D])          # timm packs q | k | v as row blocks
    put(["attention.k_proj", "attention.attention.key"], qw[D:2 * D], qb[D:2 * D])
    put(["attention.v_proj", "attention.attention.value"], qw[2 * D:], qb[2 * D:])
    put(["attention.o_proj", "attention.output.dense"], p[f"{opre}.attn.proj.weight"], p[f"{opre}.attn.proj.bias"])
    put(["layernorm_before"], p[f"{opre}.norm1.weight"], p[f"{opre}.norm1.bias"])
    put(["layernorm_after"], p[f"{opre}.norm2.weight"], p[f"{opre}.norm2.bias"])
    put(["mlp.fc1", "intermediate.dense"], p[f"{opre}.mlp.fc1.weight"], p[f"{opre}.mlp.fc1.bias"])
    put(["mlp.fc2", "output.dense"], p[f"{opre}.mlp.fc2.weight"], p[f"{opre}.mlp.fc2.bias"])


def _load_oracle_weights(model, p, cfg: O.MAEConfig):
    sd = model.state_dict()
    sd["vit.embeddings.cls_token"] = p["encoder.vit.cls_token"].clone()
    sd["vit.embeddings.position_embeddings"] = p["encoder.vit.pos_embed"].clone()         # position tables travel as weights
    sd["vit.embeddings.patch_embeddings.projection.weight"] = p["encoder.vit.patch_embed.proj.weight"].clone()
    sd["vit.embeddings.patch_embeddings.projection.bias"] = p["encoder.vit.patch_embed.proj.bias"].clone()
    enc = "vit.layers" if "vit.layers.0.layernorm_before.weight" in sd else "vit.encoder.layer"   # module path differs between releases
    for i in range(cfg.depth):
        _load_block(sd, f"{enc}.{i}", p, f"encoder.vit.blocks.{i}", cfg.embed_dim)
    sd["vit.layernorm.weight"], sd["vit.layernorm.bias"] = p["encoder.vit.norm.weight"].clone(), p["encoder.vit.norm.bias"].clone()
    sd["decoder.mask_token"] = p["decoder.mask_token"].clone()
    sd["decoder.decoder_pos_embed"] = p["decoder.decoder_pos_embed"].clone()
    sd["decoder.decoder_embed.weight"], sd["decoder.decoder_embed.bias"] = p["decoder.decoder_embed.weight"].clone(), p["decoder.decoder_embed.bias"].clone()
    for i in range(cfg.decoder_depth):
        _load_block(sd, f"decoder.decoder_layers.{i}", p, f"decoder.decoder_blocks.{i}", cfg.decoder_embed_dim)
    sd["decoder.decoder_norm.weight"], sd["decoder.decoder_norm.bias"] = p["decoder.decoder_norm.weight"].clone(), p["decoder.decoder_norm.bias"].clone()
    sd["decoder.decoder_pred.weight"], sd["decoder.decoder_pred.bias"] = p["decoder.decoder_pred.weight"].clone(), p["decoder.decoder_pred.bias"].clone()
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


@pytest.mark.parametrize("cfg,ratio,hf_ratio,batch", [
    (O.MAEConfig(image_size=32, patch_size=8, embed_dim=48, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=2, decoder_num_heads=2), 0.75, 0.8, 3),
    (O.MAEConfig(image_size=48, patch_size=8, embed_dim=96, depth=3, num_heads=6, decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=4), 0.5, 0.52, 2),
    (O.MAEConfig(image_size=96, patch_size=8, embed_dim=144, depth=4, num_heads=6, decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6), 0.75, 0.755, 2),  # configs/mae.yaml
])
def test_whole_model_matches_transformers_vitmae(cfg, ratio, hf_ratio, batch):
    torch.set_float32_matmul_precision("highest")
    N, L = cfg.num_patches, cfg.sequence_length
    k = cfg.num_keep(ratio)
    assert int(N * (1 - hf_ratio)) == k - 1, "pick hf_ratio so that both implementations keep the same patches"
    p = O.init_params(cfg, 3)
    O.randomize_params(p, 5, 0.1)
    g = torch.Generator().manual_seed(11)
    images = torch.rand(batch, 3, cfg.image_size, cfg.image_size, generator=g) * 2 - 1
    noise = O.make_noise(batch, L, g)
    model = _hf_model(cfg, hf_ratio)
    _load_oracle_weights(model, p, cfg)

    loss_o, grads_o, aux = O.loss_and_grads(p, cfg, images, noise, mask_ratio=ratio)
    idx_keep, idx_mask = aux["idx_keep"], aux["idx_mask"]
    assert (idx_keep[:, 0] == 0).all()

    out = model(pixel_values=images, noise=noise[:, 1:].contiguous())
    # same kept set in the same order (ascending noise); transformers' mask marks exactly the oracle's masked patches
    hf_keep = torch.argsort(noise[:, 1:], dim=1)[:, :k - 1]
    assert torch.equal(hf_keep + 1, idx_keep[:, 1:])
    mask_o = torch.zeros(batch, N)
    mask_o.scatter_(1, idx_mask - 1, 1.0)
    assert torch.equal(out.mask, mask_o)
    # encoder output (class token + visible patches, final norm)
    latent = model.vit(pixel_values=images, noise=noise[:, 1:].contiguous()).last_hidden_state
    assert torch.allclose(latent, aux["x_encoded"], rtol=1e-5, atol=1e-5), float((latent - aux["x_encoded"]).abs().max())
    # decoder prediction on the masked set (transformers predicts every patch: pick the oracle's masked tokens, in its order)
    hf_pred = torch.gather(out.logits, 1, (idx_mask - 1).unsqueeze(-1).expand(-1, -1, cfg.patch_dim))
    assert torch.allclose(hf_pred, aux["x_pred"], rtol=1e-5, atol=1e-5), float((hf_pred - aux["x_pred"]).abs().max())
    # pixel targets and loss
    hf_target = torch.gather(model.patchify(images), 1, (idx_mask - 1).unsqueeze(-1).expand(-1, -1, cfg.patch_dim))
    assert torch.equal(hf_target, aux["target"])
    assert abs(float(out.loss.detach()) - float(loss_o)) <= 1e-5 * abs(float(loss_o)), (float(out.loss.detach()), float(loss_o))
    # a handful of gradients through the whole graph
    out.loss.backward()
    named = dict(model.named_parameters())
    D, Dd = cfg.embed_dim, cfg.decoder_embed_dim

    def qkv_grad(pre):
        names = [("attention.q_proj", "attention.k_proj", "attention.v_proj"), ("attention.attention.query", "attention.attention.key", "attention.attention.value")]
        for trio in names:
            if f"{pre}.{trio[0]}.weight" in named:
                return torch.cat([named[f"{pre}.{t}.weight"].grad for t in trio], 0)
        raise KeyError(pre)
    checks = [
        (named["vit.embeddings.patch_embeddings.projection.weight"].grad, grads_o["encoder.vit.patch_embed.proj.weight"]),
        (named["vit.embeddings.cls_token"].grad, grads_o["encoder.vit.cls_token"]),
        (qkv_grad("vit.layers.0" if "vit.layers.0.layernorm_before.weight" in named else "vit.encoder.layer.0"), grads_o["encoder.vit.blocks.0.attn.qkv.weight"]),
        (named["vit.layernorm.weight"].grad, grads_o["encoder.vit.norm.weight"]),
        (named["decoder.mask_token"].grad, grads_o["decoder.mask_token"]),
        (named["decoder.decoder_embed.weight"].grad, grads_o["decoder.decoder_embed.weight"]),
        (qkv_grad(f"decoder.decoder_layers.{cfg.decoder_depth - 1}"), grads_o[f"decoder.decoder_blocks.{cfg.decoder_depth - 1}.attn.qkv.weight"]),
        (named["decoder.decoder_pred.weight"].grad, grads_o["decoder.decoder_pred.weight"]),
    ]
    for got, want in checks:
        err = float((got - want).norm() / want.norm().clamp_min(1e-30))
        assert err < 1e-5, err
    assert named["vit.embeddings.position_embeddings"].grad is None and named["decoder.decoder_pos_embed"].grad is None   # frozen in both


@pytest.mark.parametrize("dim,grid", [(48, 4), (144, 12), (192, 12), (384, 12), (768, 14)])
def test_sincos_table_matches_transformers_builder(dim, grid):
    """transformers builds ViTMAE's frozen position tables as [sin_h | cos_h | sin_w | cos_w] and then swaps the two halves of
    the feature axis "to match the pretrained layout" (modeling_vit_mae.py, ViTMAEEmbeddings / ViTMAEDecoder.initialize_weights):
    that layout is MAE-official's [w | h], the one lightly writes into vit.pos_embed / decoder_pos_embed and the oracle
    restates (SURVEY 8c knew the [h | w] builder only).  The builder is called directly: constructing the model from a
    config leaves the tables at zero in this release (weights are initialised lazily)."""
    from transformers.models.vit_mae import modeling_vit_mae as M
    hf = M.build_2d_sinusoidal_position_embedding(height=grid, width=grid, embed_dim=dim, cls_token=True)
    half = dim // 2
    hf = torch.cat([hf[..., half:], hf[..., :half]], dim=-1)
    want = O.sincos_pos_embed(dim, grid, cls_token=True)[0]
    assert hf.shape == want.shape
    assert torch.allclose(hf, want, atol=2e-6), float((hf - want).abs().max())
