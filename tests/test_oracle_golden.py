"""The oracle against the committed vectors (tests/golden/mae_micro.npz, made by tests/golden/make_golden.py).
PARITY UNPINNED w.r.t. lightly/timm (see the generator's header): this guards the oracle against drift."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import mae_oracle as O

GOLD = np.load(Path(__file__).parent / "golden" / "mae_micro.npz")
MICRO = O.MAEConfig(image_size=32, patch_size=8, in_chans=3, embed_dim=48, depth=2, num_heads=2,
                    decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)


def T(key):
    return torch.from_numpy(GOLD[key])


@pytest.mark.parametrize("tag,r", [("b2_r75", 0.75), ("b5_r50", 0.5)])
def test_oracle_reproduces_golden(tag, r):
    params = O.init_params(MICRO, 73); O.randomize_params(params)
    images, noise = T(f"{tag}/images"), T(f"{tag}/noise")
    loss, grads, aux = O.loss_and_grads(params, MICRO, images, noise, r)
    assert torch.equal(aux["idx_keep"], T(f"{tag}/idx_keep")) and torch.equal(aux["idx_mask"], T(f"{tag}/idx_mask"))
    assert torch.equal(aux["target"], T(f"{tag}/target"))
    assert torch.allclose(loss, T(f"{tag}/loss"), rtol=1e-6)
    assert torch.allclose(aux["x_pred"], T(f"{tag}/x_pred"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(torch.stack([g.norm() for g in grads.values()]), T(f"{tag}/grad_norms"), rtol=1e-4)
    for k in GOLD.files:
        if k.startswith(f"{tag}/grad/"):
            assert torch.allclose(grads[k.split("/", 2)[2]], T(k), rtol=1e-4, atol=1e-7), k


def test_tie_case_contract():
    noise, stable = T("ties/noise"), T("ties/order_stable")
    ref = noise.clone(); ref[:, 0] = -1
    assert torch.equal(torch.argsort(ref, dim=1, stable=True), stable)
    keep, mask = O.mask_from_noise(noise, 4)  # lightly's (unstable) argsort: same keys position by position
    got = torch.cat([keep, mask], 1)
    assert torch.equal(torch.gather(ref, 1, got), torch.gather(ref, 1, stable))
