"""MI355X-native engine for the MAE pretrain step of giolucasd/ssrl-vit-mae-jepa.

``MaskedAutoencoder`` mirrors ``src/models/mae.py`` of the reference, ``MAEPretrainModule`` mirrors
``src/training/mae.py``; the arithmetic lives in ``lib/libmae_hip.so`` (HIP, gfx950), bound through ctypes.
Importing this package without the built library raises ImportError: there is no CPU fallback.
"""
from ._lib import MaeHipError, LIB_PATH  # noqa: F401  (import fails loudly when the extension is missing)
from .mae import MaskedAutoencoder, Engine  # noqa: F401
from .training import MAEPretrainModule, lr_lambda, mask_ratio_at  # noqa: F401
from .jepa import IJEPA, IJEPAPretrainModule, sample_block_masks  # noqa: F401

__all__ = ["MaskedAutoencoder", "MAEPretrainModule", "IJEPA", "IJEPAPretrainModule", "sample_block_masks", "Engine", "MaeHipError", "lr_lambda", "mask_ratio_at", "LIB_PATH"]
