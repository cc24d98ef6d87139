import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
from tests import cases
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(N)
X = rng.uniform(size=(N, 4)); Y = cases.hf_4d(X)
Xa = np.hstack([X, cases.lf_4d(X)[:, None]])
parts = cases.composite(4, 1)
thetas = [np.abs(rng.normal(1.0, 0.3, 6)) + 0.1 for _ in range(12)]
noise = 0.01 * Y.var()
engs = [Engine(0) for _ in range(3)]
for e in engs:
    e.set_data(Xa, Y); e.set_kernel(parts)
seq = [engs[0].eval(t, noise) for t in thetas]
out = [[None] * len(thetas) for _ in engs]
def work(k):
    for i, t in enumerate(thetas):
        out[k][i] = engs[k].eval(t, noise)
ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
bad = 0
for k in range(3):
    for i in range(len(thetas)):
        d = abs(out[k][i][0] - seq[i][0]) + np.abs(out[k][i][1] - seq[i][1]).max()
        if d != 0.0:
            bad += 1
            print("engine", k, "theta", i, "nlml", out[k][i][0], seq[i][0], "dgrad", np.abs(out[k][i][1] - seq[i][1]).max())
print("mismatches:", bad)
