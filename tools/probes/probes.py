"""ctypes wrapper of tools/probes/libmfgp_probes.so (hardware probes; test / tool code, not part of the product)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmfgp_probes.so")
_dp = ctypes.POINTER(ctypes.c_double)


def _lib():
    lib = ctypes.CDLL(LIB)
    for name in ("mfgp_probe_basic", "mfgp_probe_bw"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int32
        fn.argtypes = [ctypes.c_int32, _dp]
    return lib


def basic(device=0):
    """-> (bare fp64 MFMA TFLOP/s, 1 GiB device-copy GB/s)"""
    out = np.zeros(2)
    rc = _lib().mfgp_probe_basic(int(device), out.ctypes.data_as(_dp))
    if rc:
        raise RuntimeError("mfgp_probe_basic failed (%d)" % rc)
    return float(out[0]), float(out[1])


def bandwidth(device=0):
    """-> (write-only GB/s, read-only GB/s) of a 1 GiB grid-stride stream, 16 B per lane"""
    out = np.zeros(2)
    rc = _lib().mfgp_probe_bw(int(device), out.ctypes.data_as(_dp))
    if rc:
        raise RuntimeError("mfgp_probe_bw failed (%d)" % rc)
    return float(out[0]), float(out[1])
