"""MI355X-native engine for the v2 latent-diffusion hot path."""
from . import lib  # noqa: F401

__all__ = ["lib"]
