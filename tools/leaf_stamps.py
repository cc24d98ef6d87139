"""Phase-by-phase cycle stamps of ONE leaf (mfgp_leaf_cholinv_f64 alone on the GPU) -- LAB BUILD ONLY: the library must have been built
with MFGP_BUILD_DEFINES=-DMFGP_LEAF_STAMPS (tools/gpu_session.sh leaf_stamps does that and restores the shipped build afterwards).
Prints, per panel, how long wave 0 (the pivot wave) and the slowest / fastest worker wave took for each phase and how long each waited at
the two workgroup barriers: who is the critical path of the panel loop.  Unit: thousands of shader cycles (s_memtime; 2.4 GHz alone)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from multifidelity_datafusion_gps_amd import _lib  # noqa: E402

e = _lib.Engine(0)
lib = _lib.load_library()
if not hasattr(lib, "mfgp_lab_leaf_stamps"):
    sys.exit("this library was built without -DMFGP_LEAF_STAMPS")
rng = np.random.default_rng(0)
M = rng.standard_normal((128, 128))
A = M @ M.T + 128 * np.eye(128)
for rep in range(3):
    e.dbg_leaf(A)
buf = (ctypes.c_ulonglong * (8 * 9 * 8))()
assert lib.mfgp_lab_leaf_stamps(buf) == 0
st = np.array(buf, dtype=np.float64).reshape(8, 9, 8)
t0 = st[:, 0, 0].min()
tick = 1.0       # s_memtime counts shader cycles on this part (DESIGN.md "matrix pipe": 2.39 GHz in bare loops); printed as k cycles
us = lambda v: (v - t0) * 1e-3
print("# one leaf alone; thousands of shader cycles (s_memtime) since the first wave's start")
print("load + first micro-Cholesky: wave 0 done at %.2f, last loader at %.2f, barrier passed %.2f" % (us(st[0, 0, 1]), us(st[1:, 0, 1].max()), us(st[:, 0, 2].max())))
for jb in range(7):
    p = jb + 1
    w0, wk = st[0, p], st[1:, p]
    print("panel %d: phase A  wave0 %.2f (solve + diag update), workers %.2f..%.2f | barrier | phase B  wave0 micro-Cholesky %.2f, workers' updates %.2f..%.2f, + panel output %.2f..%.2f | panel %.2f"
          % (jb, (w0[1] - w0[0]) * tick * 1e-3, (wk[:, 1] - wk[:, 0]).min() * tick * 1e-3, (wk[:, 1] - wk[:, 0]).max() * tick * 1e-3,
             (w0[3] - w0[2]) * tick * 1e-3, (wk[:, 3] - wk[:, 2]).min() * tick * 1e-3, (wk[:, 3] - wk[:, 2]).max() * tick * 1e-3,
             (wk[:, 4] - wk[:, 2]).min() * tick * 1e-3, (wk[:, 4] - wk[:, 2]).max() * tick * 1e-3, (st[:, p, 5].max() - st[:, p, 0].min()) * tick * 1e-3))
print("end of the leaf (last output written): %.2f" % us(st[:, 8, 6].max()))
