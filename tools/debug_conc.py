import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multifidelity_datafusion_gps_amd as mf
from tests import cases
def hf2(x): return cases.hf_2d(x)[:, None]
def lf2(x): return cases.lf_2d(x)[:, None]
rng = np.random.default_rng(21)
X_hf = rng.uniform(size=(150, 2))
for conc in (1, 3):
    model = mf.NARGP(2, hf2, lf2, seed=5)
    model.first_run_max_iters, model.restart_max_iters, model.restart_concurrency = 40, 40, conc
    model.fit(X_hf)
    print("conc", conc, [(round(r.f_opt, 6), r.n_evals, r.status) for r in model.hf_model.optimization_runs])
    print("   params", [p.value for p in model.hf_model.parameters()])
    model.close()
