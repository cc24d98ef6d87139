import numpy as np

from .abstract_maximizer import AbstractMaximizer
from .direct import direct_minimize, gablonsky_direct


class ScipyDirectMaximizer(AbstractMaximizer):
    """Variance maximiser with scipydirect's defaults (original DIRECT, eps=1e-4, maxf=20000, maxT=6000):
    the default maximiser of the models (/root/reference/src/MFDataFusion.py:59; wrapper
    src/adaptation_maximizers/scipydirect_wrapper.py:16-31).  `faithful=True` routes through Gablonsky's code
    (scipy.optimize.direct) one point per call instead of the batched DIRECT."""

    def __init__(self, eps=1e-4, maxf=20000, maxT=6000, algmethod=0, verbose=False, faithful=False):
        super().__init__()
        self.eps, self.maxf, self.maxT, self.algmethod, self.verbose = eps, maxf, maxT, algmethod, verbose
        self.faithful = faithful

    def maximize(self, model_predict: callable, lower_bound: np.ndarray, upper_bound: np.ndarray):
        def acquisition(Xb):
            _, var = model_predict(np.atleast_2d(Xb))
            return -np.asarray(var).reshape(-1)

        solver = gablonsky_direct if self.faithful else direct_minimize
        x, fun, self.last_info = solver(acquisition, lower_bound, upper_bound, eps=self.eps, maxf=self.maxf,
                                        maxT=self.maxT, algmethod=self.algmethod)
        if self.verbose:
            print("Selected point", x, fun)
        return x, fun
