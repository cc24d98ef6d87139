"""Phase stamps (s_memtime) of the 128x128 leaf kernel: MFGP_LEAF_STAMPS=1 python tools/leaf_stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multifidelity_datafusion_gps_amd._lib import Engine
e = Engine(0)
rng = np.random.default_rng(0)
B = rng.normal(size=(128, 128)); A = B @ B.T + 128 * np.eye(128)
for _ in range(3):
    L, X, ld, rc = e.dbg_leaf(A)
print("max |L L^T - A| = %.2e, max |X L - I| = %.2e" % (np.abs(L @ L.T - A).max(), np.abs(np.tril(X) @ L - np.eye(128)).max()))
