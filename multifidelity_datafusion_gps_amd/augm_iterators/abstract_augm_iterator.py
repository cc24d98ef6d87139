"""Augmentation stencils: integer offset vectors i such that the low-fidelity level is evaluated at
x + i*tau (interface of /root/reference/src/augm_iterators/abstract_augm_iterator.py:4-35)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class AbstractAugmIterator(metaclass=ABCMeta):
    """Re-iterable generator of stencil offsets (each a float vector of length `dim`)."""

    def __init__(self, n, dim=1):
        self.n = int(n)
        self.dim = int(dim)
        self._it = None

    @abstractmethod
    def offsets(self):
        """(count, dim) array of all offsets in iteration order"""

    def new_entries_count(self):
        return len(self.offsets())

    def reset(self):
        self._it = None

    def __iter__(self):
        return self

    def __next__(self):
        # the reference iterators reset themselves when exhausted, so the same object can be iterated again
        if self._it is None:
            self._it = iter(self.offsets())
        try:
            return np.array(next(self._it), dtype=float)
        except StopIteration:
            self.reset()
            raise
