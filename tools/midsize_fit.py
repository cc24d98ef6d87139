"""A mid-size fit as a user of the reference would run it (the ARD recipe: 1 + 6 L-BFGS-B runs, here with a budget of 50
evaluations per run), sequential restarts against concurrent ones.  usage: midsize_fit.py N_hf [N_hf ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import multifidelity_datafusion_gps_amd as mf  # noqa: E402
from tests import cases  # noqa: E402


def col(f):
    return lambda x: f(x)[:, None]


for n_hf in [int(a) for a in sys.argv[1:]] or [1024]:
    line = "N_hf=%d (N_lf=%d):" % (n_hf, 2 * n_hf)
    ref = None
    for conc in (1, 2, 3):
        class M(mf.NARGP):
            lf_max_iters = first_run_max_iters = restart_max_iters = 50
            eval_cap = 50
            restart_concurrency = conc
        rng = np.random.default_rng(1)
        X_lf = rng.uniform(size=(2 * n_hf, 4))
        m = M(4, col(cases.hf_4d), None, lf_X=X_lf, lf_Y=col(cases.lf_4d)(X_lf), seed=3)
        X = rng.uniform(size=(n_hf, 4))
        m.fit(X)                      # warm: plans, allocations
        t0 = time.perf_counter()
        m.fit(X)
        dt = time.perf_counter() - t0
        th = np.array([p.value for p in m.hf_model.parameters()])
        if ref is None:
            ref = th
        line += "  conc %d: %.0f ms (%d evals)%s" % (conc, dt * 1e3, m.hf_model.n_evals, "" if np.array_equal(th, ref) else " DIFFERENT theta")
        m.close()
    print(line, flush=True)
