"""Input step before the path (src/data.py:15-106 of the reference): the STL-10 binary reader on a generated fixture and the
split / normalisation host logic (CPU); the uint8-in-engine pixel path and the pinned double-buffered H2D stream (-m gpu)."""
import numpy as np
import pytest
import torch

from ssrl_vit_mae_jepa_amd import data as D


def _write_stl10_bin(path, imgs):
    """torchvision STL10 layout: N images x 3 planes x 96 x 96, every plane stored COLUMN-major (the loader transposes
    (0, 1, 3, 2)); reference: STL10(DATA_DIR, split="unlabeled") at src/data.py:60-65."""
    np.ascontiguousarray(imgs.numpy().transpose(0, 1, 3, 2)).tofile(path)


def test_stl10_unlabeled_reader_on_a_generated_three_image_file(tmp_path):
    g = torch.Generator().manual_seed(5)
    imgs = torch.randint(0, 256, (3, 3, 96, 96), generator=g, dtype=torch.uint8)
    imgs[0, 0, 5, 90] = 201  # a pixel whose row/column swap would be noticed
    f = tmp_path / "unlabeled_X.bin"
    _write_stl10_bin(f, imgs)
    got = D._load_stl10_unlabeled(f, 1.0)
    assert got.dtype == torch.uint8 and got.shape == (3, 3, 96, 96) and got.is_contiguous()
    assert torch.equal(got, imgs) and int(got[0, 0, 5, 90]) == 201
    assert torch.equal(D._load_stl10_unlabeled(f, 0.67), imgs[:2])  # data_fraction: the first int(n * f) images (src/data.py:37-42)


def test_normalize_u8_is_totensor_then_normalize():
    u = torch.arange(256, dtype=torch.uint8)
    v = D.normalize_u8(u)
    assert v.dtype == torch.float32 and float(v[0]) == -1.0 and float(v[255]) == 1.0
    assert torch.equal(v, (u.float().div(255) - 0.5) / 0.5)  # ToTensor: /255; Normalize(.5, .5): (x - .5) / .5  (src/data.py:22-23)


@pytest.mark.gpu
@pytest.mark.parametrize("img,p,C,B", [(96, 8, 3, 5), (224, 16, 3, 2), (32, 4, 1, 3), (64, 16, 3, 4)])
def test_uint8_pixels_in_the_engine_equal_normalised_floats(dev, img, p, C, B):
    """MAE_U8 images: the three pixel readers normalise on the fly and produce the bits of the fp32-image path fed with
    normalize_u8(images) -- target bit-exact, visible-patch operand bit-exact, loss and gradients equal."""
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import MaskedAutoencoder
    cfg = O.MAEConfig(image_size=img, patch_size=p, in_chans=C, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)
    params = O.init_params(cfg, 73); O.randomize_params(params)
    model = MaskedAutoencoder(dict(image_size=img, patch_size=p, in_chans=C, mask_ratio=0.75, engine_precision="fp32"),
                              dict(embed_dim=64, depth=1, num_heads=2), dict(decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2))
    model.load_state_dict(params)
    model = model.to(dev)
    g = torch.Generator().manual_seed(img + p)
    u8 = torch.randint(0, 256, (B, C, img, img), generator=g, dtype=torch.uint8)
    u8[0, 0, 0, :4] = torch.tensor([0, 255, 1, 254], dtype=torch.uint8)
    f32 = D.normalize_u8(u8)
    noise = O.make_noise(B, cfg.sequence_length, torch.Generator().manual_seed(9))
    # oracle on the normalised floats
    loss_ref, grads_ref, aux = O.loss_and_grads(params, cfg, f32, noise)
    loss_u8, keep, mask = model.loss_and_grads(u8.to(dev), noise.to(dev), return_indices=True)
    g_u8 = model.flat_grads.clone()
    loss_f32 = model.loss_and_grads(f32.to(dev), noise.to(dev))
    assert torch.equal(model.patchify_gather(u8.to(dev), mask).cpu(), aux["target"])                    # bit-exact target
    assert torch.equal(model.patchify_gather(u8.to(dev), mask), model.patchify_gather(f32.to(dev), mask))
    assert abs(loss_u8.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    # same gradient bits as the fp32-image path (the scalar loss is summed over a different grid: equal to rounding)
    assert abs(loss_u8.item() - loss_f32.item()) <= 1e-6 * abs(loss_f32.item()) and torch.equal(g_u8, model.flat_grads)
    with torch.no_grad():
        assert torch.equal(model.forward_encoder(u8.to(dev), keep), model.forward_encoder(f32.to(dev), keep))
        feats = model.encoder.vit.forward_features(u8.to(dev))                                           # class token + every patch
        assert torch.equal(feats, model.encoder.vit.forward_features(f32.to(dev)))
        preds, targets = model(u8.to(dev), noise=noise.to(dev))
        assert torch.equal(targets.cpu(), aux["target"])
    # an index list that names one row of patches over and over is served completely (duplicates are legal for callers)
    dup = torch.full((B, 2 * (img // p) + 3), 2, dtype=torch.int64, device=dev)
    with torch.no_grad():
        assert torch.equal(model.forward_encoder(u8.to(dev), dup), model.forward_encoder(f32.to(dev), dup))


@pytest.mark.gpu
def test_uint8_full_size_batch_2000_matches_float_images(dev):
    from oracle import mae_oracle as O
    from ssrl_vit_mae_jepa_amd import MaskedAutoencoder
    cfg, B = O.VIT_S8_YAMLDEC, 2000
    torch.manual_seed(3)
    model = MaskedAutoencoder(dict(image_size=96, patch_size=8, in_chans=3, mask_ratio=0.75, engine_precision="bf16"),
                              dict(embed_dim=384, depth=12, num_heads=6), dict(decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6)).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    u8 = torch.randint(0, 256, (B, 3, 96, 96), generator=g, dtype=torch.uint8, device=dev)
    noise = torch.rand(B, cfg.sequence_length, generator=g, device=dev)
    a = model.loss_and_grads(u8, noise).clone()
    ga = model.flat_grads.clone()
    for _ in range(3):   # the band's tokens are summed in sorted order: the loss is the same float from run to run
        assert model.loss_and_grads(u8, noise).item() == a.item() and torch.equal(ga, model.flat_grads)
    b = model.loss_and_grads(D.normalize_u8(u8.cpu()).to(dev), noise)  # host normalisation = the reference's transform, bit for bit
    assert abs(a.item() - b.item()) <= 1e-5 * abs(b.item()) and torch.equal(ga, model.flat_grads)  # 84 M squared errors summed over two grids


@pytest.mark.gpu
def test_pinned_double_buffered_stream_serves_the_right_batches(dev):
    g = torch.Generator().manual_seed(2)
    data = torch.randint(0, 256, (37, 3, 16, 16), generator=g, dtype=torch.uint8)
    order = torch.randperm(37, generator=g)
    stream = D.PinnedBatchStream(data.pin_memory(), batch=8, device=dev)
    for epoch in range(2):  # buffers are reused across epochs
        got = []
        for x in stream.batches(order):
            assert x.is_cuda and x.dtype == torch.uint8
            y = x.float().sum(dim=(1, 2, 3))          # consumer work enqueued on the buffer before the next one is asked for
            got.append((x.clone(), y))
        torch.cuda.synchronize()
        assert [t[0].shape[0] for t in got] == [8, 8, 8, 8, 5]
        assert torch.equal(torch.cat([t[0] for t in got]).cpu(), data[order])
        assert torch.equal(torch.cat([t[1] for t in got]).cpu(), data[order].float().sum(dim=(1, 2, 3)))


# ----------------------------------------------------------------------------------------------------------------------
# augmentation in front of the path: RandomResizedCrop + RandomHorizontalFlip on uint8, the HIP resampler behind
# mae_augment_crop_flip_u8 against the torch restatement (affine_grid + grid_sample, align_corners=False, border)
# ----------------------------------------------------------------------------------------------------------------------
def _resample_reference(u8, params):
    """The crop boxes of `params` resampled with the torch ops augment_batch uses, on the raw 0..255 values, rounded to uint8."""
    import torch.nn.functional as F
    B, _, S, _ = u8.shape
    top, left, h, w, flip = (params[:, i].to(torch.float32) for i in range(5))
    sx = torch.where(flip > 0, -w / S, w / S); sy = h / S
    theta = torch.zeros(B, 2, 3)
    theta[:, 0, 0] = sx; theta[:, 0, 2] = (2.0 * left + w) / S - 1.0
    theta[:, 1, 1] = sy; theta[:, 1, 2] = (2.0 * top + h) / S - 1.0
    grid = F.affine_grid(theta, list(u8.shape), align_corners=False)
    out = F.grid_sample(u8.to(torch.float32), grid, mode="bilinear", padding_mode="border", align_corners=False)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,S", [(7, 3, 96), (3, 1, 32), (33, 3, 224)])
def test_hip_crop_resize_flip_matches_the_torch_restatement(dev, B, C, S):
    from ssrl_vit_mae_jepa_amd._lib import check, lib, ptr
    from ssrl_vit_mae_jepa_amd.mae import _stream
    g = torch.Generator().manual_seed(B + S)
    u8 = torch.randint(0, 256, (B, C, S, S), generator=g, dtype=torch.uint8)
    top, left, h, w = D.random_resized_crop_params(B, S, g)
    flip = (torch.rand(B, generator=g) < 0.5).to(torch.int64)
    params = torch.stack([top, left, h, w, flip], 1).to(torch.int32)
    params[0] = torch.tensor([0, 0, S, S, 0])      # identity box: the image itself
    params[1] = torch.tensor([0, 0, S, S, 1])      # identity box, mirrored
    ref = _resample_reference(u8, params)
    xd, pd = u8.to(dev), params.to(dev)
    out = torch.empty_like(xd)
    check(lib.mae_augment_crop_flip_u8(ptr(xd), ptr(pd), B, C, S, ptr(out), _stream(dev)))
    out = out.cpu()
    assert torch.equal(out[0], u8[0]) and torch.equal(out[1], u8[1].flip(-1))
    # bilinear weights are computed in a different order than affine_grid + grid_sample: the fp32 value can land on the other
    # side of a .5 rounding boundary for a few pixels; never more than one grey level
    diff = (out.to(torch.float32) - ref.round().clamp(0, 255)).abs()
    assert diff.max() <= 1 and (diff > 0).float().mean() < 1e-2   # random-noise images: every pixel sits on a steep gradient
    assert (out.to(torch.float32) - ref).abs().max() <= 0.5 + 1e-2


@pytest.mark.gpu
def test_augmented_uint8_batches_reach_the_engine_as_uint8(dev):
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randint(0, 256, (16, 3, 96, 96), generator=g, dtype=torch.uint8, device=dev)
    y = D.augment_batch_u8(x, torch.Generator(device=dev).manual_seed(9))
    y2 = D.augment_batch_u8(x, torch.Generator(device=dev).manual_seed(9))
    assert y.dtype == torch.uint8 and y.shape == x.shape and torch.equal(y, y2) and not torch.equal(y, x)
    # same boxes and flips as the float path draws from the same generator state
    z = D.augment_batch(D.normalize_u8(x.cpu()).to(dev), torch.Generator(device=dev).manual_seed(9))
    assert (D.normalize_u8(y.cpu()).to(dev) - z).abs().max() <= (0.5 + 1e-2) / 127.5
    with pytest.raises(ValueError):
        D.augment_batch_u8(x.float(), torch.Generator(device=dev).manual_seed(9))
