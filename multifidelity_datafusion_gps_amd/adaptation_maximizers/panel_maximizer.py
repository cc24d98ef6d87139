import numpy as np

from .abstract_maximizer import AbstractMaximizer


class PanelMaximizer(AbstractMaximizer):
    """Variance maximiser over a fixed candidate panel: ONE predictive panel per acquisition instead of a DIRECT run.

    The form BASELINE.json's configuration 5 is stated in ("predictive-variance panels sharded across 8 x MI355X";
    SURVEY.md 8(d): "candidate panel N* = 65536 Sobol/uniform points per acquisition"): `n_candidates` points of a scrambled
    Sobol sequence (or seeded uniform draws) in the box go through `model_predict` in a single call -- on the HIP path one
    K(X*, X) panel + one variance product per chunk of the panel, its rows sharded over the ranks when the model carries a
    communicator (`MultifidelityDataFusion.predict`) -- and the candidate with the largest predictive variance
    is returned.  Same interface and return convention as the reference's maximisers
    (/root/reference/src/adaptation_maximizers/abstract_maximizer.py:5-28: `(x_opt, -max variance)`).

    The panel is the same for every acquisition of a run (drawn once per dimension; `resample=True` draws a fresh one per
    call from the same generator): an acquired point has variance ~ noise afterwards, so it is not chosen twice.  Ties go
    to the first candidate in panel order, on every rank alike (the gathered variances are identical bit for bit)."""

    def __init__(self, n_candidates=65536, sampler="sobol", seed=0, resample=False):
        super().__init__()
        if n_candidates < 1:
            raise ValueError("n_candidates must be positive")
        if sampler not in ("sobol", "uniform"):
            raise ValueError("sampler must be 'sobol' or 'uniform'")
        self.n_candidates, self.sampler, self.seed, self.resample = int(n_candidates), sampler, seed, bool(resample)
        self._unit = {}     # dimension -> (n_candidates, d) points of the unit cube
        self._gen = {}
        self.last_info = None

    def _unit_panel(self, d):
        if d in self._unit and not self.resample:
            return self._unit[d]
        if d not in self._gen:
            if self.sampler == "sobol":
                from scipy.stats import qmc
                self._gen[d] = qmc.Sobol(d=d, scramble=True, seed=self.seed)
            else:
                self._gen[d] = np.random.default_rng(self.seed)
        g = self._gen[d]
        if self.sampler == "sobol":
            import warnings
            m = int(np.log2(self.n_candidates))
            if (1 << m) == self.n_candidates:
                pts = g.random_base2(m)                     # a power of two keeps the sequence's balance properties
            else:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    pts = g.random(self.n_candidates)
        else:
            pts = g.uniform(size=(self.n_candidates, d))
        self._unit[d] = np.ascontiguousarray(pts, dtype=float)
        return self._unit[d]

    def candidates(self, lower_bound, upper_bound):
        lo, hi = np.asarray(lower_bound, dtype=float).reshape(-1), np.asarray(upper_bound, dtype=float).reshape(-1)
        return lo + (hi - lo) * self._unit_panel(lo.size)

    def maximize(self, model_predict: callable, lower_bound: np.ndarray, upper_bound: np.ndarray):
        C = self.candidates(lower_bound, upper_bound)
        _, var = model_predict(C)                       # ONE panel (sharded over the ranks by the model's predict)
        var = np.asarray(var).reshape(-1)
        k = int(np.argmax(var))
        self.last_info = {"evaluations": int(C.shape[0]), "panels": 1, "argmax": k}
        return C[k].copy(), -float(var[k])
