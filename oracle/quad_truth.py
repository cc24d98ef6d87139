"""ctypes loader of oracle/quad_truth.c -- the exact-GP quantities in IEEE binary128 (113-bit) arithmetic on fp64 inputs.

TEST INFRASTRUCTURE ONLY (tests/ call it as a checker; the product never imports anything under oracle/).  It answers a question
the fp64 oracle cannot: when the HIP path and oracle/gp_oracle.py differ by 3e-10, which of them is 3e-10 from the true value?
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libquad_truth.so")
SRC = os.path.join(HERE, "quad_truth.c")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.run(["make", "-C", HERE, "-B", "libquad_truth.so"], check=True, capture_output=True)
    return LIB


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        lib.quad_gp.restype = ctypes.c_int
        lib.quad_gp.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, ip, dp, ctypes.c_double, ctypes.c_double,
                                ctypes.c_int, dp] + [dp] * 8
        lib.quad_set_threads.argtypes = [ctypes.c_int]
        # the cores this process may use, at most 16 (the GPU box's CPU share; its affinity mask shows the whole machine)
        lib.quad_set_threads(int(os.environ.get("QUAD_THREADS", min(16, len(os.sched_getaffinity(0))))))
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def evaluate(parts, theta, noise, X, y, Xs=None, jitter=1e-8, want_grad=True, want_K=False, want_Kinv=False):
    """-> dict(nlml, logdet, alpha, grad [P + 1: kernel parameters, then the noise variance], mean, var [latent], K, Kinv), every entry the
    correctly rounded fp64 image of the quad-precision value (up to the ~1e-30 * cond(Ky) the quad evaluation itself carries)"""
    lib = _load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    n, d = X.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    pa = np.ascontiguousarray(np.array([[int(v) for v in p] for p in parts], dtype=np.int32).reshape(-1))
    P = sum(1 + ((c1 - c0) if (t & 0x100) else 1) for t, c0, c1, _ in parts)
    assert theta.size == P, (theta.size, P)
    ns = 0 if Xs is None else int(np.asarray(Xs).shape[0])
    Xs_c = None if Xs is None else np.ascontiguousarray(Xs, dtype=np.float64)
    out = dict(nlml=ctypes.c_double(), logdet=ctypes.c_double(), alpha=np.empty(n), grad=np.empty(P + 1) if want_grad else None,
               mean=np.empty(ns) if ns else None, var=np.empty(ns) if ns else None, K=np.empty((n, n)) if want_K else None,
               Kinv=np.empty((n, n)) if want_Kinv else None)
    rc = lib.quad_gp(n, d, _p(X), _p(y), len(parts), pa.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _p(theta), float(noise), float(jitter),
                     ns, _p(Xs_c), _p(out["K"]), ctypes.byref(out["nlml"]), ctypes.byref(out["logdet"]), _p(out["alpha"]), _p(out["grad"]),
                     _p(out["mean"]), _p(out["var"]), _p(out["Kinv"]))
    if rc != 0:
        raise RuntimeError("quad_gp failed: %d" % rc)
    out["nlml"], out["logdet"] = out["nlml"].value, out["logdet"].value
    return out
