import numpy as np

from .abstract_maximizer import AbstractMaximizer
from .direct import direct_minimize


class DIRECT1Maximizer(AbstractMaximizer):
    """Variance maximiser on the locally-biased DIRECT-L with 50 iterations: the controls of
    /root/reference/src/adaptation_maximizers/DIRECT1_maximizer.py:15-16 (maxT=50, algmethod=1)."""

    def __init__(self):
        super().__init__()
        self.maxT = 50
        self.algmethod = 1

    def maximize(self, model_predict: callable, lower_bound: np.ndarray, upper_bound: np.ndarray):
        def acquisition(Xb):
            _, var = model_predict(np.atleast_2d(Xb))
            return -np.asarray(var).reshape(-1)

        xopt, fopt, self.last_info = direct_minimize(acquisition, lower_bound, upper_bound, maxT=self.maxT,
                                                     algmethod=self.algmethod)
        return xopt, fopt
