"""Drop-in for the reference's ``src/training/mae.py`` (no Lightning needed): same class name, ctor and hooks."""
from ssrl_vit_mae_jepa_amd.training import MAEPretrainModule  # noqa: F401

__all__ = ["MAEPretrainModule"]
