import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
e = Engine(0)
N = 512
rng = np.random.default_rng(N)
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
e.set_data(Xa, Y); e.set_kernel(cases.composite(4, 1))
e.factorize(np.array([1.2, 1.1, 0.9, 0.6, 0.4, 0.8]), 0.01)
Xs = Xa[:1] + 0.01
for _ in range(5): e.predict(Xs)
t0 = time.perf_counter()
for _ in range(200): e.predict(Xs)
print("predict N=512 N*=1: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
