import numpy as np

from .abstract_augm_iterator import AbstractAugmIterator


class EvenAugmentation(AbstractAugmIterator):
    """0, then for each step s = 1..n: -s*e_0 .. -s*e_{dim-1}, +s*e_0 .. +s*e_{dim-1}: 2*n*dim + 1 offsets
    (sequence of /root/reference/src/augm_iterators/even_augm_iterator.py:19-48)."""

    def offsets(self):
        out = [np.zeros(self.dim)]
        for step in range(1, self.n + 1):
            for sign in (-1.0, 1.0):
                for j in range(self.dim):
                    v = np.zeros(self.dim)
                    v[j] = sign * step
                    out.append(v)
        return np.array(out)

    def new_entries_count(self):
        return 2 * self.n * self.dim + 1
