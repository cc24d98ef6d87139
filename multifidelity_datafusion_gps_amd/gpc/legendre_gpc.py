"""Pseudo-spectral polynomial chaos for independent uniform inputs on a box, in plain numpy.

Stands where the reference uses chaospy (src/gpc/chaospy_wrapper.py:9-34: Gaussian quadrature of order q,
total-order expansion of order p, fit_quadrature, E / Var) -- chaospy is not available offline and its E/Var took
minutes at order 10 in 4-D (tests/test_mfgp_adapt_4d.py:72-78).  For uniform inputs the orthonormal basis is the
shifted Legendre family, so the projection is one (basis x nodes) matrix-vector product and
mean = c_0, variance = sum_{alpha != 0} c_alpha^2 exactly.  The model is evaluated ONCE on the whole tensor grid
((q+1)^d rows): one predictive panel on the GPU instead of chaospy's point loop.
"""
import itertools

import numpy as np

from .base import AbstractGPC


class LegendreGPC(AbstractGPC):

    def __init__(self, function: callable, lower, upper, polynomial_order=8, quadrature_order=8):
        self.lower = np.asarray(lower, dtype=np.float64).reshape(-1)
        self.upper = np.asarray(upper, dtype=np.float64).reshape(-1)
        self.dim = self.lower.size
        self.polynomial_order, self.quadrature_order = int(polynomial_order), int(quadrature_order)
        self._build()
        self.coefficients = None
        super().__init__(function)

    def _build(self):
        d, p, q = self.dim, self.polynomial_order, self.quadrature_order
        t, w = np.polynomial.legendre.leggauss(q + 1)            # nodes / weights on [-1, 1], q + 1 points per axis
        w = w / 2.0                                              # uniform density
        grids = np.meshgrid(*([t] * d), indexing="ij")
        T = np.stack([g.reshape(-1) for g in grids], axis=1)     # ((q+1)^d, d) in [-1, 1]
        W = np.ones(len(T))
        for k, g in enumerate(np.meshgrid(*([w] * d), indexing="ij")):
            W = W * g.reshape(-1)
        self.quad_points = (self.lower + (T + 1.0) * 0.5 * (self.upper - self.lower)).T   # (d, nq) like chaospy
        self.quad_weights = W
        # orthonormal Legendre values per axis: sqrt(2n+1) P_n(t)
        V = np.stack([np.sqrt(2 * n + 1.0) * np.polynomial.legendre.Legendre.basis(n)(t) for n in range(p + 1)])
        idx1d = [np.searchsorted(t, T[:, k]) for k in range(d)]  # node index of every grid point per axis
        self.multi_indices = [a for a in itertools.product(range(p + 1), repeat=d) if sum(a) <= p]
        Phi = np.ones((len(self.multi_indices), len(T)))
        for j, a in enumerate(self.multi_indices):
            for k in range(d):
                if a[k]:
                    Phi[j] *= V[a[k]][idx1d[k]]
        self._PhiW = Phi * W[None, :]

    def calculate_coefficients(self):
        evaluations = np.asarray(self.function(self.quad_points.T), dtype=np.float64).reshape(-1)
        self.coefficients = self._PhiW @ evaluations
        return self.coefficients

    def get_mean(self):
        return float(self.coefficients[0])

    def get_var(self):
        return float(np.sum(self.coefficients[1:] ** 2))

    def update_order(self, new_order):
        self.polynomial_order, self.quadrature_order = int(new_order), int(new_order)
        self._build()
