"""Known-answer tests that pin the oracle (CPU, no GPU): the only numbers the reference records for this path
(notebook.ipynb:987-994 model summary; configs/mae.yaml schedule) plus structural properties of the restatement."""
import math

import pytest
import torch

from oracle import mae_oracle as O


def test_param_census_matches_reference_model_summary():
    c = O.param_census(O.YAML_TINY)
    assert c["total"] == 2_035_104            # "2.0 M" total params
    assert c["frozen"] == 48_720              # "48.7 K" non-trainable: both sin-cos position tables
    assert c["bytes_fp32"] == 8_140_416       # "8.140" MB
    assert len(O.module_inventory(O.YAML_TINY)) == 152  # "152" modules below the LightningModule


def test_vits8_census_and_flops_match_baseline_md():
    c = O.param_census(O.VIT_S8_YAMLDEC)
    assert c["total"] == 22_454_016 and c["frozen"] == 83_520 and c["trainable"] == 22_370_496
    assert abs(O.flops_per_image_step(O.VIT_S8_YAMLDEC) / 1e9 - 5.5803) < 1e-4
    assert abs(O.flops_per_image_step(O.YAML_TINY) / 1e9 - 1.1264) < 1e-4
    c = O.param_census(O.VIT_S8_DEC512)
    assert c["total"] == 34_405_824 and c["trainable"] == 34_275_904
    assert abs(O.flops_per_image_step(O.VIT_S8_DEC512) / 1e9 - 16.2442) < 1e-4
    c = O.param_census(O.VIT_B16_DEC512)  # BASELINE.json configs[3]; BASELINE.md section 3 last row
    assert c["trainable"] == 111_656_448
    assert (O.VIT_B16_DEC512.sequence_length, O.VIT_B16_DEC512.num_keep(), O.VIT_B16_DEC512.patch_dim) == (197, 49, 768)
    assert abs(O.flops_per_image_step(O.VIT_B16_DEC512) / 1e9 - 57.5245) < 1e-4


def test_state_dict_names_follow_survey_8b():
    names = list(O.param_shapes(O.YAML_TINY))
    assert names[:5] == ["encoder.mask_token", "encoder.vit.cls_token", "encoder.vit.pos_embed",
                         "encoder.vit.patch_embed.proj.weight", "encoder.vit.patch_embed.proj.bias"]
    assert "encoder.vit.blocks.3.mlp.fc2.bias" in names and "decoder.decoder_blocks.1.attn.qkv.weight" in names
    assert names[-2:] == ["decoder.decoder_pred.weight", "decoder.decoder_pred.bias"]
    s = O.param_shapes(O.YAML_TINY)
    assert s["encoder.vit.patch_embed.proj.weight"] == (144, 3, 8, 8) and s["decoder.decoder_pred.weight"] == (192, 192)
    assert s["encoder.vit.blocks.0.attn.qkv.weight"] == (432, 144) and s["encoder.vit.pos_embed"] == (1, 145, 144)


def test_num_keep_and_mask_properties():
    cfg = O.YAML_TINY
    assert cfg.sequence_length == 145 and cfg.num_keep(0.75) == 36 and cfg.num_keep(0.5) == 72 and cfg.num_keep(0.999) == 1
    noise = O.make_noise(64, 145, torch.Generator().manual_seed(1))
    keep, mask = O.mask_from_noise(noise, 36)
    assert keep.shape == (64, 36) and mask.shape == (64, 109) and keep.dtype == torch.int64
    assert bool((keep[:, 0] == 0).all())                       # class token always kept, always first
    assert torch.equal(torch.cat([keep, mask], 1).sort(1).values, torch.arange(145).repeat(64, 1))
    ref = noise.clone(); ref[:, 0] = -1
    assert bool((torch.gather(ref, 1, torch.cat([keep, mask], 1)).diff(dim=1) >= 0).all())  # ascending noise


def test_schedules_match_configs_mae_yaml():
    # lr_eff = 1.5e-4 * 2000/256 = 1.171875e-3; epoch-0 factor = 1/20 * 1
    assert abs(O.effective_lr(1.5e-4, 2000) - 1.171875e-3) < 1e-15
    assert abs(O.lr_lambda(0, 20, 800) - 0.05) < 1e-15
    assert abs(O.lr_lambda(19, 20, 800) - 0.5 * (1 + math.cos(math.pi * 19 / 800))) < 1e-15
    assert abs(O.lr_lambda(400, 20, 800) - 0.5) < 1e-12
    assert O.mask_ratio_at(0, 0.75, 0.75, 5) == 0.75
    assert abs(O.mask_ratio_at(2, 0.5, 0.85, 5) - (0.5 + 0.5 * 0.35)) < 1e-12 and O.mask_ratio_at(100, 0.5, 0.85, 5) == 0.85


def test_patchify_roundtrip_and_orders():
    img = torch.arange(2 * 3 * 16 * 16, dtype=torch.float32).reshape(2, 3, 16, 16)
    p = O.patchify(img, 8)
    assert p.shape == (2, 4, 192)
    assert torch.equal(O.unpatchify(p, 8), img)
    # element (py, px, c) of patch (ph, pw): c fastest
    assert p[1, 3, (2 * 8 + 5) * 3 + 1] == img[1, 1, 8 + 2, 8 + 5]
    with pytest.raises(ValueError):
        O.patchify(torch.zeros(1, 3, 16, 12), 4)


def test_clip_and_adamw_match_torch():
    torch.manual_seed(0)
    params = {"a": torch.randn(7, 5), "b": torch.randn(11)}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    opt = torch.optim.AdamW(ref.values(), lr=3e-3, weight_decay=0.05)
    state = {}
    for step in (1, 2, 3):
        grads = {k: torch.randn_like(v) * 3 for k, v in params.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        total_ref = torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)
        opt.step()
        total, _ = O.clip_grad_norm(grads, 1.0)
        O.adamw_step(params, grads, state, 3e-3, step, 0.05)
        assert abs(float(total) - float(total_ref)) < 1e-5
        for k in params:
            assert torch.allclose(params[k], ref[k].detach(), atol=1e-6, rtol=1e-5)


def test_oracle_backward_is_consistent_with_finite_differences():
    cfg = O.MAEConfig(image_size=16, patch_size=8, in_chans=1, embed_dim=16, depth=1, num_heads=2, decoder_embed_dim=16,
                      decoder_depth=1, decoder_num_heads=2)
    p = O.init_params(cfg, 3); O.randomize_params(p)
    p = {k: v.double() for k, v in p.items()}
    images = O.synthetic_images(2, cfg).double()
    noise = O.make_noise(2, cfg.sequence_length, torch.Generator().manual_seed(4))
    loss, grads, _ = O.loss_and_grads(p, cfg, images, noise, 0.5)
    name = "encoder.vit.blocks.0.mlp.fc1.bias"
    eps = 1e-6
    q = {k: v.clone() for k, v in p.items()}
    q[name][3] += eps
    lp = O.mse_loss(*O.forward(q, cfg, images, noise, 0.5)[:2])
    assert abs((lp - loss).item() / eps - grads[name][3].item()) < 1e-5
