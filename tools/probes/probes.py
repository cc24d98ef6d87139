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


def sustained(shape=1, launches=120, device=0):
    """`launches` back-to-back ~25 ms launches of the bare fp64 MFMA loop (shape 1 = 4x4x4_4b, 0 = 16x16x4): per launch
    (ms, TFLOP/s, in-kernel clock GHz) -- does the burst rate hold over seconds?"""
    lib = ctypes.CDLL(LIB)
    lib.mfgp_probe_sustained.restype = ctypes.c_int32
    lib.mfgp_probe_sustained.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _dp]
    out = np.zeros(3 * launches)
    rc = lib.mfgp_probe_sustained(int(device), int(shape), int(launches), out.ctypes.data_as(_dp))
    if rc:
        raise RuntimeError("mfgp_probe_sustained failed (%d)" % rc)
    return out.reshape(launches, 3)


def _smi_sampler(stop, rows, period=0.25):
    """power / clock / temperature as rocm-smi reports them, sampled beside a sustained run (best effort: the tool may be
    missing or refuse an ordinary user)"""
    import subprocess
    import time
    t0 = time.time()
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                               text=True, timeout=5)
            rows.append((round(time.time() - t0, 2), r.stdout.strip()[:1500] if r.returncode == 0 else "rc %d %s" % (r.returncode, r.stderr[:200])))
        except Exception as ex:  # noqa: BLE001
            rows.append((round(time.time() - t0, 2), repr(ex)[:200]))
            return
        time.sleep(period)


if __name__ == "__main__":
    import json
    import sys
    if "sustained" in sys.argv[1:]:
        import threading
        res = {}
        for shape, name in ((1, "4x4x4_4b"), (0, "16x16x4")):
            stop, rows = threading.Event(), []
            th = threading.Thread(target=_smi_sampler, args=(stop, rows))
            th.start()
            a = sustained(shape, 160 if shape == 1 else 60)
            stop.set()
            th.join()
            t = np.cumsum(a[:, 0]) / 1e3
            res[name] = {"launches": len(a), "seconds": round(float(t[-1]), 2),
                         "tflops_first_5": [round(float(x), 2) for x in a[:5, 1]],
                         "tflops_by_half_second": [round(float(a[(t > lo) & (t <= lo + 0.5), 1].mean()), 2)
                                                   for lo in np.arange(0, t[-1] - 0.25, 0.5)],
                         "clock_ghz_by_half_second": [round(float(a[(t > lo) & (t <= lo + 0.5), 2].mean()), 3)
                                                      for lo in np.arange(0, t[-1] - 0.25, 0.5)],
                         "tflops_last_second": round(float(a[t > t[-1] - 1.0, 1].mean()), 2),
                         "rocm_smi": rows[:3] + rows[-3:]}
        print(json.dumps({"sustained": res}, indent=1))
    elif "shapes" in sys.argv[1:]:
        print(json.dumps({"fp64_shapes": fp64_shapes()}, indent=1))
    else:
        print(json.dumps({"basic": basic(), "detail": detail(), "fp64_shapes": fp64_shapes()}, indent=1))
