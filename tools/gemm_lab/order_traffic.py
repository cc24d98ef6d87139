"""Traffic experiment on the product's 128-tile kernel body (lab variant o128_w22_k16s2_m0 = gemm_nt_dma<128,128,2,2,16,2>): one
launch of 64 x 64 tiles at n = 8192, the tasks enumerated row-major / with every task on the same operand panels (no operand
traffic) / dealt to the XCDs in bi x bj super-blocks.  usage: order_traffic.py time   (all configurations, short and >= 1 s runs)
                                                                order_traffic.py one <K> <order> <bi> <bj>   (a few launches: under rocprofv3 --pmc)"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libgemm_lab.so"))
lib.lab_variant_name.restype = ctypes.c_char_p
names = [lib.lab_variant_name(i).decode() for i in range(lib.lab_num_variants())]
V = names.index(os.environ.get("LAB_VARIANT", "o128_w22_k16s2_m0"))
lib.lab_run_order.restype = ctypes.c_int
lib.lab_run_order.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_double)]
CONFIGS = [("row-major", 0, 0, 0), ("same panels", 1, 0, 0), ("xcd 1x8", 2, 1, 8), ("xcd 2x4", 2, 2, 4), ("xcd 4x8", 2, 4, 8),
           ("xcd 8x4", 2, 8, 4), ("xcd 8x8", 2, 8, 8), ("xcd 4x4", 2, 4, 4), ("xcd 16x16", 2, 16, 16)]


def run(K, order, bi, bj, reps):
    ms = ctypes.c_double()
    rc = lib.lab_run_order(V, 8192, K, order, bi, bj, reps, ctypes.byref(ms))
    assert rc == 0, rc
    return ms.value


if sys.argv[1] == "one":
    K, order, bi, bj = map(int, sys.argv[2:6])
    print("%.3f ms" % run(K, order, bi, bj, 3))
else:
    for K in (512, 2048):
        fl = 2.0 * 8192 * 8192 * K
        for name, order, bi, bj in CONFIGS:
            short = run(K, order, bi, bj, 5)
            reps = max(5, int(1500.0 / short))
            steady = run(K, order, bi, bj, reps)
            print("K=%-5d %-12s burst %7.3f ms %5.1f TFLOP/s | %4d launches back to back %7.3f ms %5.1f TFLOP/s" % (
                K, name, short, fl / short / 1e9, reps, steady, fl / steady / 1e9), flush=True)
