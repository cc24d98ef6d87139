"""Where the dispatcher places the workgroups of a grid that does not fill the chip (GPU box).  usage: placement.py"""
import ctypes, os, collections
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfgp_probes.so"))
lib.mfgp_probe_placement.restype = ctypes.c_int32
lib.mfgp_probe_placement.argtypes = [ctypes.c_int32] * 4 + [ctypes.POINTER(ctypes.c_uint32)]
for G, lds in ((500, 65536), (480, 65536), (256, 65536), (300, 65536), (250, 131072 + 1280), (512, 65536)):
    out = np.zeros(2 * G, dtype=np.uint32)
    rc = lib.mfgp_probe_placement(0, G, lds, 300, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    assert rc == 0, rc
    hw, xcc = out[0::2], out[1::2] & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    hist = collections.Counter(per_cu.values())
    per_xcc = collections.Counter(xcc.tolist())
    print("G = %3d x %6d B LDS: %3d distinct CUs hold workgroups; workgroups per CU -> number of CUs: %s; per XCC: %s; "
          "first 16 (xcc, se, sh, cu): %s" % (G, lds, len(per_cu), dict(sorted(hist.items())), dict(sorted(per_xcc.items())),
                                             list(zip(xcc[:16].tolist(), se[:16].tolist(), sh[:16].tolist(), cu[:16].tolist()))))
