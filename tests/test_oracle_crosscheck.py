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
