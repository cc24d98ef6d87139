"""`models.NARGP`, `models.GPDF`, `models.GPDFC` -- the reference's `src.models` namespace."""
from .presets import GPDF, GPDFC, NARGP  # noqa: F401

__all__ = ["NARGP", "GPDF", "GPDFC"]
