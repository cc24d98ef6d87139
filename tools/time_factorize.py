"""gradient-free factorisation (mfgp_factorize: K build + Cholesky + inverse, no K^-1) of a single-RBF 3-D GP: ms per size"""
import sys, os
import os as _os; _os.environ.setdefault("MFGP_TIMING", "1")   # start / end stamps of a call at every size (off by default below Np = 4096)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases

e = Engine(0)
for N in [int(a) for a in sys.argv[1:]] or [4096]:
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 3)); Y = cases.hf_3d(X)
    e.set_data(X, Y); e.set_kernel(cases.single(cases.RBF, 3))
    th, nz = np.array([1.1, 0.4]), 0.01 * Y.var()
    for _ in range(3):
        e.factorize(th, nz)
    acc = 0.0
    for _ in range(5):
        e.factorize(th, nz); acc += e.timings()["total_ms"] / 5
    print("N=%d factorize %.3f ms" % (N, acc))
