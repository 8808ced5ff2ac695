"""Host-side mirror of the reference interface, exercised on CPU (no kernels run): construction, names, attributes,
schedules, config loading and error behaviour.  The compute entry points must refuse to run without a GPU."""
from pathlib import Path

import pytest
import torch
import yaml

from oracle import mae_oracle as O
from ssrl_vit_mae_jepa_amd import MAEPretrainModule, MaskedAutoencoder, lr_lambda, mask_ratio_at

ROOT = Path(__file__).resolve().parents[1]
CFG = yaml.safe_load(open(ROOT / "configs" / "mae.yaml"))


def yaml_model():
    m = CFG["model"]
    return MaskedAutoencoder(general_cfg=m["general"], encoder_cfg=m["encoder"], decoder_cfg=m["decoder"])


def test_yaml_model_matches_reference_summary_and_names():
    model = yaml_model()
    sd = model.state_dict()
    assert list(sd) == list(O.param_shapes(O.YAML_TINY))
    assert sum(p.numel() for p in model.parameters()) == 2_035_104
    assert sum(p.numel() for p in model.parameters() if not p.requires_grad) == 48_720
    assert model.mask_ratio == 0.75 and model.image_size == 96 and model.patch_size == 8 and model.in_chans == 3
    assert model.sequence_length == 145
    assert model.encoder.vit.embed_dim == 144 and hasattr(model.encoder, "encode") and hasattr(model.encoder.vit, "forward_features")
    assert not sd["encoder.vit.pos_embed"].requires_grad and not model.decoder.decoder_pos_embed.requires_grad
    assert torch.equal(sd["encoder.vit.pos_embed"], O.sincos_pos_embed(144, 12))
    assert torch.equal(sd["decoder.decoder_pos_embed"], O.sincos_pos_embed(192, 12))
    # init recipe: LayerNorm 1/0, zero biases, xavier-bounded matrices
    assert bool((sd["encoder.vit.blocks.0.norm1.weight"] == 1).all()) and bool((sd["decoder.decoder_pred.bias"] == 0).all())
    w = sd["encoder.vit.blocks.0.attn.qkv.weight"]
    assert float(w.abs().max()) <= (6.0 / (144 + 432)) ** 0.5 + 1e-6 and float(w.std()) > 0.01


def test_reference_code_defaults():
    # ctor defaults of src/models/mae.py:23-26,32-34,49-51; decoder default heads 6 does not divide 512 -> timm assert
    with pytest.raises(ValueError, match="divisible"):
        MaskedAutoencoder({}, {}, {})
    model = MaskedAutoencoder({}, {}, {"decoder_num_heads": 8})
    assert model.patch_size == 6 and model.sequence_length == 257 and model.encoder.vit.embed_dim == 384
    assert sum(p.numel() for p in model.parameters()) > 30_000_000


def test_flat_arena_survives_state_dict_and_views():
    model = yaml_model()
    params = O.init_params(O.YAML_TINY, 1)
    model.load_state_dict(params)
    for (name, off, numel, shape, flags), (p, *_r) in zip(model.engine.table, model._slots):
        assert p.data_ptr() == model.flat_params.data_ptr() + 4 * off  # still a view of the arena
        assert torch.equal(p.detach(), params[name])
    again = yaml_model()
    again.load_state_dict(model.state_dict())
    assert torch.equal(again.flat_params, model.flat_params)


def test_no_cpu_fallback():
    model = yaml_model()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(2, 3, 96, 96))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.forward_encoder(torch.zeros(2, 3, 96, 96))


def test_pretrain_module_surface_and_schedules():
    module = MAEPretrainModule(model_cfg=CFG["model"], training_cfg=CFG["pretrain"])
    assert isinstance(module.model, MaskedAutoencoder) and isinstance(module.criterion, torch.nn.MSELoss)
    assert abs(module.effective_lr - 1.171875e-3) < 1e-15
    opt = module.configure_optimizers()
    assert type(opt["optimizer"]).__name__ == "AdamW" and opt["lr_scheduler"]["interval"] == "epoch"
    group = opt["optimizer"].param_groups[0]
    assert len(opt["optimizer"].param_groups) == 1 and group["weight_decay"] == 0.05 and group["betas"] == (0.9, 0.999)
    assert sum(p.numel() for p in group["params"]) == 2_035_104  # self.parameters(): every tensor, one group
    assert abs(group["lr"] - 1.171875e-3 * 0.05) < 1e-12          # LambdaLR applied at construction (epoch 0)
    module.current_epoch = 0
    module.on_train_epoch_start()
    assert module.model.mask_ratio == 0.75 and module.logged["mask_ratio"] == 0.75
    ramp = MAEPretrainModule(model_cfg=CFG["model"], training_cfg={})  # code defaults .5 -> .85 over 200 epochs
    ramp.current_epoch = 100
    ramp.on_train_epoch_start()
    assert abs(ramp.model.mask_ratio - (0.5 + 100 / 199 * 0.35)) < 1e-12 and ramp.model.num_keep() == int(145 * (1 - ramp.model.mask_ratio))
    assert lr_lambda(0, 20, 800) == O.lr_lambda(0, 20, 800) and mask_ratio_at(3, .5, .85, 5) == O.mask_ratio_at(3, .5, .85, 5)


def test_drop_in_import_paths():
    from src.models.mae import MaskedAutoencoder as A
    from src.training.mae import MAEPretrainModule as B
    assert A is MaskedAutoencoder and B is MAEPretrainModule


def test_pretrain_cli_flags():
    from scripts.training.pretrain_mae import parse_args
    a = parse_args([])
    assert a.config == "configs/mae.yaml" and a.resume_from is None and a.output_dir_suffix == "mae_pretrain"


def test_random_resized_crop_matches_torchvision_semantics():
    """RandomResizedCrop(96, scale=(0.8, 1.0)) + flip of src/data.py:18-21, batched: crop boxes inside the image with
    area in [0.8, 1] of it and aspect in [3/4, 4/3] (up to integer rounding); an identity box reproduces the input."""
    import torch
    from ssrl_vit_mae_jepa_amd.data import augment_batch, random_resized_crop_params
    g = torch.Generator().manual_seed(0)
    top, left, h, w = random_resized_crop_params(4096, 96, g)
    assert ((top >= 0) & (left >= 0) & (top + h <= 96) & (left + w <= 96)).all()
    frac = (h * w).double() / (96 * 96)
    assert frac.min() > 0.78 and frac.max() <= 1.0 and 0.85 < frac.mean() < 0.90   # U[0.8, 1] thinned at the top: large non-square boxes are rejected
    ar = w.double() / h.double()
    assert ar.min() > 0.73 and ar.max() < 1.37
    x = torch.rand(8, 3, 96, 96, generator=g) * 2 - 1
    y = augment_batch(x, torch.Generator().manual_seed(1))
    assert y.shape == x.shape and y.min() >= -1 - 1e-6 and y.max() <= 1 + 1e-6 and not torch.equal(x, y)
    # the sampler mapping itself: a full-image box without flip is the identity (bilinear at pixel centres)
    import torch.nn.functional as F
    theta = torch.tensor([[[1.0, 0, 0], [0, 1.0, 0]]]).repeat(8, 1, 1)
    ident = F.grid_sample(x, F.affine_grid(theta, list(x.shape), align_corners=False), mode="bilinear", padding_mode="border", align_corners=False)
    assert torch.allclose(ident, x, atol=1e-5)


def test_checkpoints_are_consumable_by_the_reference_encoder_loader(tmp_path):
    """The reference's fine-tune script (scripts/training/train_mae.py:101-139) takes ``ckpt.get("state_dict", ckpt)``, finds
    the first of the prefixes 'model.encoder.' / 'encoder.' / 'module.encoder.', strips it and loads the rest into
    ``mae.encoder`` with strict=False.  Both files the pretrain CLI writes (Lightning-shaped .ckpt and the raw vit-mae.pt)
    must come through that procedure with every encoder tensor matched and nothing unexpected."""
    import torch
    from ssrl_vit_mae_jepa_amd import MAEPretrainModule, MaskedAutoencoder
    g = dict(image_size=32, patch_size=8, in_chans=3)
    e = dict(embed_dim=48, depth=2, num_heads=2)
    d = dict(decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)
    tcfg = dict(mask_ratio_start=0.75, mask_ratio_end=0.75, mask_ramp_epochs=5, total_epochs=8, warmup_epochs=2,
                batch_size=256, base_learning_rate=1.5e-4, weight_decay=0.05)
    torch.manual_seed(0)
    module = MAEPretrainModule(dict(general=g, encoder=e, decoder=d), tcfg)
    lightning_like = {"state_dict": {f"model.{k}": v.clone() for k, v in module.model.state_dict().items()}, "epoch": 0}
    raw = {k: v.clone() for k, v in module.model.state_dict().items()}
    torch.save(lightning_like, tmp_path / "last.ckpt"); torch.save(raw, tmp_path / "vit-mae.pt")
    for name, want_prefix in (("last.ckpt", "model.encoder."), ("vit-mae.pt", "encoder.")):
        ckpt = torch.load(tmp_path / name, map_location="cpu", weights_only=True)
        state_dict = ckpt.get("state_dict", ckpt)
        prefix = next((p for p in ("model.encoder.", "encoder.", "module.encoder.") if any(k.startswith(p) for k in state_dict)), None)
        assert prefix == want_prefix
        encoder_state = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        torch.manual_seed(1)
        fresh = MaskedAutoencoder(g, e, d)
        missing, unexpected = fresh.encoder.load_state_dict(encoder_state, strict=False)
        assert not missing and not unexpected and len(encoder_state) == 2 * 12 + 7  # 2 blocks x 12 tensors + mask/cls/pos/patch(2)/norm(2)
        for k, v in encoder_state.items():
            assert torch.equal(fresh.encoder.state_dict()[k], v)
        # timm names below `vit.` (what ViTClassifierTrainModule receives as pretrained_encoder)
        assert {"vit.cls_token", "vit.pos_embed", "vit.patch_embed.proj.weight", "vit.blocks.1.mlp.fc2.bias", "vit.norm.weight"} <= set(encoder_state)
