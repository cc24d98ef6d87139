import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from multifidelity_datafusion_gps_amd import NARGP, sharding
from multifidelity_datafusion_gps_amd._lib import Engine
class A: pass
args = A(); args.evals = 20; args.restarts = 6; args.seed = 1234; args.concurrency = int(sys.argv[1]) if len(sys.argv) > 1 else 1
data = bench.make_data(8192, 8192, 8192)
engines = {"lf": Engine(0), "hf": Engine(0), "hf#1": Engine(0), "hf#2": Engine(0)}
X_lf, Y_lf, X_hf, X_st = data
class B(NARGP):
    lf_max_iters = 20; first_run_max_iters = 20; restart_max_iters = 20; num_restarts = 6; restart_concurrency = args.concurrency
for rep in range(2):
    t0 = time.perf_counter()
    m = B(4, f_exact=bench.hf_4d, f_low=None, lf_X=X_lf, lf_Y=Y_lf, seed=1234, engines=engines)
    t1 = time.perf_counter()
    Y = m.f_exact(X_hf); t2 = time.perf_counter()
    Xa = m._augment_data(X_hf); t3 = time.perf_counter()
    m.fit(X_hf); t4 = time.perf_counter()
    mean, var = m.predict(X_st); t5 = time.perf_counter()
    print("ctor+LF fit %.0f ms (LF evals %d) | f_exact %.1f | augment %.1f | HF fit %.0f ms (evals %d) | predict %.0f ms" % (
        (t1 - t0) * 1e3, m.lf_model.n_evals, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, m.hf_model.n_evals, (t5 - t4) * 1e3))
