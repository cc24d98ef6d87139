"""A handful of predict calls (N* = 1, 2, 4, 8, 16, 32, 48, 64) and rank-1 appends at N = 8128 (Np = 8192) -- the workload of the counter
passes over the adaptation loop's kernels (tools/gpu_session.sh adapt): few dispatches, one kernel name per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(os.environ.get("PREDV_REPS", "5"))
n0 = N - 64
rng = np.random.default_rng(N)
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
e = Engine(0)
e.set_data(Xa[:n0], Y[:n0]); e.set_kernel(cases.composite(4, 1))
e.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01)
for ns in (1, 2, 4, 8, 16, 32, 48, 64):
    for _ in range(reps):
        e.predict(Xa[:ns] + 0.01)
for i in range(reps):
    assert e.append_row(Xa[n0 + i], Y[n0 + i])
e.close()
print("done")
