import os, sys
os.environ.setdefault("MFGP_HW_QUEUES", "2")
sys.path.insert(0, "/root/repo")
import numpy as np
import multifidelity_datafusion_gps_amd as mf
from multifidelity_datafusion_gps_amd import _lib
from tests import cases
col = lambda f: (lambda x: f(x)[:, None])
seen = []
orig_b, orig_e = _lib.Engine.eval_batch, _lib.Engine.eval
def eb(self, thetas, noises, jitters=1e-8, want_grad=True):
    out = orig_b(self, thetas, noises, jitters, want_grad)
    for t, f in zip(np.atleast_2d(thetas), out[0]): seen.append((np.min(t), np.max(t), f))
    return out
def ev(self, theta, noise, jitter=1e-8, want_grad=True):
    out = orig_e(self, theta, noise, jitter, want_grad)
    seen.append((np.min(theta), np.max(theta), out[0] if want_grad else out))
    return out
_lib.Engine.eval_batch, _lib.Engine.eval = eb, ev
n_hf = 2048
M = type("M", (mf.NARGP,), dict(lf_max_iters=50, first_run_max_iters=50, restart_max_iters=50, eval_cap=50, restart_lockstep=False, restart_concurrency=1))
rng = np.random.default_rng(1)
X_lf = rng.uniform(size=(2 * n_hf, 4))
m = M(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), seed=3)
X = rng.uniform(size=(n_hf, 4))
m.fit(X)
a = np.array(seen)
print("evaluations", len(a), "smallest parameter visited %.3g" % a[:, 0].min(), "largest %.3g" % a[:, 1].max())
for lo, hi, f in a[a[:, 0] < 1e-100]: print("  visited min theta %.3g max %.3g -> nlml %.6g" % (lo, hi, f))
