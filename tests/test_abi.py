"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mfgp.h declares,
and the product fails LOUDLY (no CPU fallback) when there is no HIP device.  (-m "not gpu")"""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "mfgp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfgp_[A-Za-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from multifidelity_datafusion_gps_amd import _lib
    assert _header_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_builds_loads_and_exports_every_symbol():
    from multifidelity_datafusion_gps_amd import _lib, build
    path = build.build()  # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    for sym in _header_symbols():
        assert hasattr(lib, sym), sym
    _lib.load_library()  # prototypes resolve


def test_struct_layouts_match_header():
    from multifidelity_datafusion_gps_amd import _lib
    assert ctypes.sizeof(_lib.KernPart) == 16
    assert ctypes.sizeof(_lib.Timings) == 13 * 8      # + timed (round 3)
    assert ctypes.sizeof(_lib.Counters) == 19 * 8     # + timed_evals, timed_predict_var_flops


def test_no_cpu_fallback_engine_fails_loudly_without_gpu():
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    from multifidelity_datafusion_gps_amd import _lib
    with pytest.raises(_lib.EngineUnavailable) as ei:
        _lib.Engine(0)
    assert "no HIP device" in str(ei.value) or "failed" in str(ei.value)
    import multifidelity_datafusion_gps_amd as mf
    with pytest.raises(_lib.EngineUnavailable):
        mf.NARGP(2, lambda x: x[:, :1], lambda x: x[:, :1]).fit(np.zeros((3, 2)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multifidelity_datafusion_gps_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", txt, flags=re.M), \
                    "%s imports test infrastructure" % os.path.join(dirpath, f)
                assert "gp_oracle" not in txt, "%s references the oracle" % os.path.join(dirpath, f)
