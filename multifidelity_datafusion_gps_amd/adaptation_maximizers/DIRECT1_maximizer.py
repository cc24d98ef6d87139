import numpy as np

from .abstract_maximizer import AbstractMaximizer
from .direct import direct_minimize, gablonsky_direct


class DIRECT1Maximizer(AbstractMaximizer):
    """Variance maximiser on the locally-biased DIRECT-L with 50 iterations: the controls of
    /root/reference/src/adaptation_maximizers/DIRECT1_maximizer.py:15-16 (maxT=50, algmethod=1).

    faithful=False (default): the batched DIRECT of direct.py -- one predictive panel per iteration.
    faithful=True: Gablonsky's own code through scipy.optimize.direct (the implementation the `DIRECT` package
    wraps), called one point at a time exactly like the reference's callback; slower, same trajectory as the
    Fortran wrapper."""

    def __init__(self, faithful: bool = False):
        super().__init__()
        self.maxT = 50
        self.algmethod = 1
        self.faithful = faithful

    def maximize(self, model_predict: callable, lower_bound: np.ndarray, upper_bound: np.ndarray):
        def acquisition(Xb):
            _, var = model_predict(np.atleast_2d(Xb))
            return -np.asarray(var).reshape(-1)

        if self.faithful:
            xopt, fopt, self.last_info = gablonsky_direct(acquisition, lower_bound, upper_bound, maxT=self.maxT,
                                                          algmethod=self.algmethod)
        else:
            xopt, fopt, self.last_info = direct_minimize(acquisition, lower_bound, upper_bound, maxT=self.maxT,
                                                         algmethod=self.algmethod)
        return xopt, fopt
