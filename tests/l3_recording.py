"""Recording engine double for the L3 replay fixtures (tests / fixture generator only).

`RecordingEngine` is the tests-only oracle double (tests/oracle_engine.py) that also logs every call the host layer
makes across the engine boundary -- set_data (with the design matrix it was given), set_kernel, eval (theta, noise,
jitter), predict (rows, flags).  Run once under the REFERENCE's own L3 (tests/golden/make_reference_l3.py: the files of
/root/reference/src executed over multifidelity_datafusion_gps_amd.compat) and once under this package's L3
(tests/test_reference_l3.py), the two logs must coincide: same augmented design matrices, same sequence of
hyper-parameter assignments, same predictions."""
import hashlib

import numpy as np

from tests.oracle_engine import OracleEngine


class Recorder:
    def __init__(self):
        self.calls = []          # (op, payload) in call order, over all engines of the run
        self.quiet = False       # True: predict calls are only counted and hashed (the DIRECT callback issues thousands)
        self.n_predict = 0
        self.predict_hash = hashlib.sha256()

    def engine_factory(self):
        rec = self
        ids = {"n": 0}

        class RecordingEngine(OracleEngine):
            def __init__(self, device=None):
                super().__init__(device)
                self.eid = ids["n"]
                ids["n"] += 1

            def set_data(self, X, Y):
                super().set_data(X, Y)
                rec.calls.append(("set_data", dict(e=self.eid, X=np.array(X, dtype=np.float64), Y=np.array(Y, dtype=np.float64).reshape(-1))))

            def set_kernel(self, parts):
                super().set_kernel(parts)
                rec.calls.append(("set_kernel", dict(e=self.eid, parts=[tuple(int(v) for v in p) for p in parts])))

            def eval(self, theta, noise, jitter=1e-8, want_grad=True):
                out = super().eval(theta, noise, jitter, want_grad)
                rec.calls.append(("eval", dict(e=self.eid, theta=np.array(theta, dtype=np.float64), noise=float(noise),
                                               jitter=float(jitter), nlml=float(self.st["nlml"]))))
                return out

            def predict(self, Xs, want_var=True, include_noise=True):
                mu, var = super().predict(Xs, want_var=True, include_noise=include_noise)
                Xs = np.array(Xs, dtype=np.float64)
                if rec.quiet:
                    rec.n_predict += 1
                    rec.predict_hash.update(Xs.tobytes())
                else:
                    rec.calls.append(("predict", dict(e=self.eid, Xs=Xs, include_noise=bool(include_noise), mean=mu.copy(), var=var.copy())))
                return mu, (var if want_var else None)

        return RecordingEngine

    # ---- (de)serialisation: a flat dict of arrays for np.savez; the (many) evaluations share one table ---------------------
    def to_arrays(self, prefix=""):
        out = {prefix + "ops": np.array([op for op, _ in self.calls])}
        evals = [p for op, p in self.calls if op == "eval"]
        width = max([len(p["theta"]) for p in evals] + [0])
        table = np.full((len(evals), 4 + width), np.nan)
        for i, p in enumerate(evals):
            table[i, :4] = (p["e"], p["noise"], p["jitter"], p["nlml"])
            table[i, 4:4 + len(p["theta"])] = p["theta"]
        out[prefix + "evals"] = table
        for i, (op, p) in enumerate(self.calls):
            if op == "eval":
                continue
            for k, v in p.items():
                out["%sc%04d_%s" % (prefix, i, k)] = np.asarray(v)
        return out


def calls_from_arrays(z, prefix=""):
    ops = [str(o) for o in z[prefix + "ops"]]
    table = z[prefix + "evals"]
    calls, n_eval = [], 0
    for i, op in enumerate(ops):
        if op == "eval":
            row = table[n_eval]
            n_eval += 1
            theta = row[4:]
            calls.append((op, dict(e=int(row[0]), noise=row[1], jitter=row[2], nlml=row[3], theta=theta[~np.isnan(theta)])))
            continue
        p = {}
        stem = "%sc%04d_" % (prefix, i)
        for k in z.files:
            if k.startswith(stem):
                p[k[len(stem):]] = z[k]
        calls.append((op, p))
    return calls


def assert_same_calls(got, want, what=""):
    """bit-for-bit: the two host layers drove the engine identically"""
    assert [op for op, _ in got] == [op for op, _ in want], "%s: different call sequence" % what
    for i, ((op, g), (_, w)) in enumerate(zip(got, want)):
        for k in w:
            gv, wv = np.asarray(g[k]), np.asarray(w[k])
            assert gv.shape == wv.shape and np.array_equal(gv, wv), "%s: call %d (%s): %s differs" % (what, i, op, k)
