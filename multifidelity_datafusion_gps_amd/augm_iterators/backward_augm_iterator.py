import numpy as np

from .abstract_augm_iterator import AbstractAugmIterator


class BackwardAugmentation(AbstractAugmIterator):
    """0, then -1*e_0 .. -1*e_{dim-1}, then -2*e_0 .. : n*dim + 1 offsets
    (sequence of /root/reference/src/augm_iterators/backward_augm_iterator.py:19-37; the only stencil the
    models instantiate, src/MFDataFusion.py:67)."""

    def offsets(self):
        out = [np.zeros(self.dim)]
        for step in range(1, self.n + 1):
            for j in range(self.dim):
                v = np.zeros(self.dim)
                v[j] = -step
                out.append(v)
        return np.array(out)

    def new_entries_count(self):
        return self.n * self.dim + 1
