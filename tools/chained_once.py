"""A handful of level-chained predict calls (NARGP: LF means of the test rows on the device, then the HF predict) at N_lf = 16384,
N_hf = 8128 -- the call the reference's adaptation loop issues per acquisition evaluation (src/abstractMFGP.py:317-359 through
src/MFDataFusion.py:106-156) -- for kernel traces, and its host-clock latency against the plain predict of the same rows.

    python tools/chained_once.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(3)
Xl = rng.uniform(size=(16384, 4)); Xh = rng.uniform(size=(8128, 4))
lf = Engine(0); hf = Engine(0)
lf.set_data(Xl, cases.lf_4d(Xl)); lf.set_kernel(cases.single(cases.RBF, 4)); lf.factorize(np.array([1.0, 0.4]), 1e-4)
offs = np.zeros((1, 4))
aug = lf.augment(Xh, offs)
hf.set_data(aug, cases.hf_4d(Xh)); hf.set_kernel(cases.composite(4, 1)); hf.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01)
for ns in (1, 4, 16, 64):
    Xs = rng.uniform(size=(ns, 4))
    A = lf.augment(Xs, offs)
    for _ in range(10):
        hf.predict_chained(lf, Xs, offs); hf.predict(A); lf.predict(Xs, want_var=False)
    t0 = time.perf_counter()
    for _ in range(reps): hf.predict_chained(lf, Xs, offs)
    t1 = time.perf_counter()
    for _ in range(reps): hf.predict(A)
    t2 = time.perf_counter()
    for _ in range(reps): lf.predict(Xs, want_var=False)
    t3 = time.perf_counter()
    print("N* = %2d: chained %.4f ms, HF predict alone %.4f ms, LF mean-only predict alone %.4f ms" %
          (ns, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3, (t3 - t2) / reps * 1e3), flush=True)
lf.close(); hf.close()
