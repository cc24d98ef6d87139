"""ctypes wrapper of tools/probes/libmfgp_probes.so (hardware probes; test / tool code, not part of the product)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmfgp_probes.so")
_dp = ctypes.POINTER(ctypes.c_double)


def _lib():
    lib = ctypes.CDLL(LIB)
    for name in ("mfgp_probe_basic", "mfgp_probe_detail", "mfgp_probe_fp64_shapes"):
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


def detail(device=0):
    out = np.zeros(24)
    rc = _lib().mfgp_probe_detail(int(device), out.ctypes.data_as(_dp))
    if rc:
        raise RuntimeError("mfgp_probe_detail failed (%d)" % rc)
    names = ["1w/SIMD x8acc", "2w/SIMD x8acc", "4w/SIMD x8acc", "1w/SIMD x1acc"]
    d = {n: dict(tflops=out[3 * i], cycles_per_mfma=out[3 * i + 1], clock_ghz=out[3 * i + 2])
         for i, n in enumerate(names)}
    d["valu_fma_f64_tflops"] = {"2w/SIMD": out[12], "4w/SIMD": out[13]}
    d["valu_plus_mfma_tflops"] = {"2w/SIMD": out[14], "4w/SIMD": out[15]}
    d["valu_fma_f64_three_vgpr_operands_tflops"] = {"2w/SIMD": out[16], "4w/SIMD": out[17]}
    d["mfma_i8_tops"] = out[18]        # v_mfma_i32_16x16x64_i8, bare loop
    d["mfma_bf16_tflops"] = out[19]    # v_mfma_f32_16x16x32_bf16, bare loop
    d["hbm_write_only_gbs"] = out[20]
    d["hbm_read_only_gbs"] = out[21]
    return d


def fp64_shapes(device=0):
    """bare fp64 MFMA loops by instruction shape, wave count and operand data (round 3): per configuration TFLOP/s,
    shader cycles per MFMA per wave (s_memtime, median over waves), in-kernel clock (s_memtime / s_memrealtime), launch ms"""
    out = np.zeros(32)
    rc = _lib().mfgp_probe_fp64_shapes(int(device), out.ctypes.data_as(_dp))
    if rc:
        raise RuntimeError("mfgp_probe_fp64_shapes failed (%d)" % rc)
    names = ["16x16x4 random 1w/SIMD", "16x16x4 random 2w/SIMD", "16x16x4 random 4w/SIMD", "16x16x4 ZERO 4w/SIMD",
             "4x4x4_4b random 1w/SIMD", "4x4x4_4b random 2w/SIMD", "4x4x4_4b random 4w/SIMD", "4x4x4_4b ZERO 4w/SIMD"]
    return {n: dict(tflops=round(out[4 * i], 2), cycles_per_mfma_per_wave=round(out[4 * i + 1], 2),
                    clock_ghz=round(out[4 * i + 2], 3), launch_ms=round(out[4 * i + 3], 2)) for i, n in enumerate(names)}


if __name__ == "__main__":
    import json
    import sys
    if "shapes" in sys.argv[1:]:
        print(json.dumps({"fp64_shapes": fp64_shapes()}, indent=1))
    else:
        print(json.dumps({"basic": basic(), "detail": detail(), "fp64_shapes": fp64_shapes()}, indent=1))
