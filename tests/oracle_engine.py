"""TEST DOUBLE (tests/ only): an object with the _lib.Engine interface whose arithmetic is the CPU oracle.
It exists so that the host logic around the engine (L-BFGS-B driving, restart sharding, predictive-row
sharding, the model surface) can be exercised on CPU with gloo; the product never sees it."""
import numpy as np

from oracle import gp_oracle as orc


class OracleEngine:
    def __init__(self, device=None):
        self.n = self.d = self.n_parts = 0
        self.n_evals = 0

    def close(self):
        pass

    def set_data(self, X, Y):
        self.X, self.Y = np.asarray(X, float), np.asarray(Y, float).reshape(-1)
        self.n, self.d = self.X.shape

    def set_kernel(self, parts):
        self.parts = [tuple(int(v) for v in p) for p in parts]
        self.n_parts = len(self.parts)
        self.n_params = orc.layout(self.parts)[1]

    def eval(self, theta, noise, jitter=1e-8, want_grad=True):
        self.n_evals += 1
        self.theta, self.noise = np.asarray(theta, float), float(noise)
        # IEEE specials are part of the arithmetic at the edge of the parameter domain (r / l overflows to inf, exp(-inf) = 0), as in
        # GPy; what comes out of them is asserted in tests/test_host_logic.py::test_evaluations_at_the_edge_of_the_parameter_domain
        with np.errstate(all="ignore"):
            self.st = self._inference(jitter)
        return (self.st["nlml"], self.st["grad"]) if want_grad else self.st["nlml"]

    def _inference(self, jitter):
        return orc.inference(self.parts, self.theta, self.noise, self.X, self.Y, want_grad=True, const_jitter=jitter)

    def nlml_grad(self):
        return self.st["grad"]

    def predict(self, Xs, want_var=True, include_noise=True):
        mu, var = orc.predict_stable(self.parts, self.theta, self.noise, self.X, self.st, np.asarray(Xs, float),
                                     include_noise=include_noise)
        return mu, (var if want_var else None)

    def timings(self):
        return {}


class BatchOracleEngine(OracleEngine):
    """the double with `eval_batch` (B evaluations, one after the other): what makes the host layer take its lock-stepped paths on
    the CPU (tests/test_lockstep_host.py).  Kept apart from OracleEngine: the recorded call sequences of the reference-L3 fixtures
    are those of the sequential order."""
    MAX_BATCH = 16

    def eval_batch(self, thetas, noises, jitters=1e-8, want_grad=True):
        thetas = np.atleast_2d(np.asarray(thetas, float))
        B = len(thetas)
        noises = np.broadcast_to(np.asarray(noises, float), (B,))
        jitters = np.broadcast_to(np.asarray(jitters, float), (B,))
        nlml, grads, status = np.zeros(B), np.zeros((B, thetas.shape[1] + 1)), np.zeros(B, dtype=np.int32)
        for b in range(B):
            self.n_evals += 1
            try:
                with np.errstate(all="ignore"):
                    st = orc.inference(self.parts, thetas[b], float(noises[b]), self.X, self.Y, want_grad=True, const_jitter=float(jitters[b]))
                nlml[b], grads[b] = st["nlml"], st["grad"]
            except np.linalg.LinAlgError:
                status[b] = 1
        return nlml, (grads if want_grad else None), status
