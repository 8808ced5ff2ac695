"""Drop-in for the reference's ``src/models/mae.py``: ``from src.models.mae import MaskedAutoencoder`` now resolves to
the MI355X engine-backed module (same constructor dicts, attributes, methods and state_dict names)."""
from ssrl_vit_mae_jepa_amd.mae import MaskedAutoencoder  # noqa: F401

__all__ = ["MaskedAutoencoder"]
