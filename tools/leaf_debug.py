import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
e = Engine(0)
rng = np.random.default_rng(0)
B = rng.normal(size=(128, 128)); A = B @ B.T + 128 * np.eye(128)
L, X, ld, rc = e.dbg_leaf(A)
Lr = np.linalg.cholesky(A)
D = np.abs(L - Lr)
bad = np.argwhere(D > 1e-9)
print("info", rc, "n bad", len(bad), "first bad entries", bad[:12].tolist())
print(np.array2string(D[:18, :18] > 1e-9, max_line_width=200).replace("False", ".").replace("True", "X"))
Y = np.tril(X[:16, :16]); Yr = np.linalg.inv(Lr[:16, :16])
DY = np.abs(Y - Yr)
print("Y00 max err", DY.max())
print(np.array2string(DY > 1e-9, max_line_width=200).replace("False", ".").replace("True", "X"))
