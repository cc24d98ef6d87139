"""Round-based driver that alternates adaptation of a multi-fidelity GP with a refresh of the polynomial-chaos
moments of its posterior mean.

Same public surface as the reference's driver (/root/reference/src/gpc/mfgp_gpc.py:3-27 -- constructor arguments,
`adapt()`, and the history attributes its experiment scripts read: `mean_history`, `var_history`, `cost_history`,
`mse_history`, `adapt_per_steps`, `calculate_mse`), organised here around one snapshot record per round: every
history is a view of the same list of records, so the histories cannot get out of step with each other.
"""
from collections import namedtuple

import numpy as np

_Round = namedtuple("_Round", "mean var cost mse")


class MFGP_GPC(object):

    adapt_per_steps = 5      # high-fidelity acquisitions per round (the reference's constant)

    def __init__(self, mfgp_obj, gpc_obj, num_adapts: int, init_cost: float, X_test: np.ndarray = None,
                 Y_test: np.ndarray = None, verbose: bool = False):
        self.mfgp_obj = mfgp_obj
        self.gpc_obj = gpc_obj
        self.num_adapts = int(num_adapts)
        self.verbose = verbose
        self.X_test, self.Y_test = X_test, Y_test
        self._rounds = []
        self.gpc_obj.calculate_coefficients()          # moments of the model as handed over = round 0
        self._snapshot(cost=init_cost)

    # ---- bookkeeping ---------------------------------------------------------------------------------
    @property
    def calculate_mse(self):
        return self.X_test is not None and self.Y_test is not None

    def _snapshot(self, cost):
        mse = self.mfgp_obj.get_mse(self.X_test, self.Y_test) if self.calculate_mse else None
        self._rounds.append(_Round(self.gpc_obj.get_mean(), self.gpc_obj.get_var(), cost, mse))

    def _column(self, field):
        return [getattr(r, field) for r in self._rounds]

    mean_history = property(lambda self: self._column("mean"))
    var_history = property(lambda self: self._column("var"))
    cost_history = property(lambda self: self._column("cost"))

    @property
    def mse_history(self):
        if not self.calculate_mse:
            raise AttributeError("mse_history needs X_test and Y_test")
        return self._column("mse")

    # ---- the rounds ------------------------------------------------------------------------------------
    def _posterior_mean(self, x):
        return self.mfgp_obj.predict(x)[0]

    def adapt(self, **adapt_kwargs):
        """`num_adapts` rounds: acquire `adapt_per_steps` high-fidelity points, re-project the posterior mean on the
        polynomial basis (ONE predictive panel over the quadrature grid), record moments / cost / test error.  The
        cost of a round is the number of acquisitions the model actually made (early stopping shortens it)."""
        for k in range(self.num_adapts):
            if self.verbose:
                print("adaptation round %d of %d" % (k + 1, self.num_adapts))
            self.mfgp_obj.adapt(self.adapt_per_steps, **adapt_kwargs)
            self.gpc_obj.update_function(self._posterior_mean)
            self._snapshot(cost=self._rounds[-1].cost + self.mfgp_obj.adapt_steps)
