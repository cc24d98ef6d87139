"""Augmentation stencils (`src.augm_iterators` of the reference)."""
from .abstract_augm_iterator import AbstractAugmIterator  # noqa: F401
from .backward_augm_iterator import BackwardAugmentation  # noqa: F401
from .even_augm_iterator import EvenAugmentation  # noqa: F401

__all__ = ["AbstractAugmIterator", "BackwardAugmentation", "EvenAugmentation"]
