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


def test_row_block_ownership_balances_the_lower_triangle():
    """mfgp_row_block_owner (pure host function of the C-ABI, no device): serpentine block-cyclic 0 1 .. G-1 G-1 .. 1 0 ..; the packed
    LOWER parts of every rank's blocks -- what mfgp_allgather_rows moves -- are equal to within a few blocks, so the padded
    all-gather carries ~ half the bytes of full rows (at the bench size on 8 ranks: 0.508)"""
    from multifidelity_datafusion_gps_amd._lib import Engine
    own = Engine.row_block_owner
    assert [own(b, 3) for b in range(8)] == [0, 1, 2, 2, 1, 0, 0, 1] and own(5, 1) == 0 and own(-1, 4) == -1
    for nblk, size in ((64, 8), (64, 3), (32, 2), (128, 6)):
        fill = [0] * size
        for b in range(nblk):
            fill[own(b, size)] += 128 * 128 * (b + 1)
        moved = size * max(fill)                                  # doubles in the padded all-gather
        full = (128 * nblk) ** 2                                  # doubles of full rows
        assert 0.5 <= moved / full <= 0.5 + 1.5 * size / nblk, (nblk, size, moved / full)


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


def test_committed_counters_are_only_reported_for_the_library_build_they_were_taken_with(tmp_path, monkeypatch):
    """profiles/r0N_pmc.json / r0N_mfma_counters.json carry the hash of the sources they were taken with;
    bench.py reports their numbers only when that hash is the one embedded in the library it loaded (mfgp_build_id), and
    null + the reason otherwise -- a kernel change can no longer keep stale counters in the bench line."""
    import json
    import bench
    from multifidelity_datafusion_gps_amd import _lib, build
    lib_id = _lib.build_id()
    assert lib_id == build.source_hash() and len(lib_id) == 16          # the in-tree library was built from the in-tree sources
    prof = tmp_path / "profiles"
    prof.mkdir()
    pmc = {"n": 8192, "csrc_hash": lib_id, "sweep": {"traffic_bytes": 123}}
    mf = {"csrc_hash": lib_id, "pmcA_eval": {"mfgp_gemm_nt_f64_t128": {"mfma_busy": 0.8}}}
    (prof / "pmc.json").write_text(json.dumps(pmc))
    (prof / "mfma.json").write_text(json.dumps(mf))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_FILE", "profiles/pmc.json")
    monkeypatch.setattr(bench, "MFMA_FILE", "profiles/mfma.json")
    assert bench.pmc_traffic("sweep", 8192, lib_id) == (123, None)
    assert bench.mfma_busy(lib_id)[0] == {"mfgp_gemm_nt_f64_t128": 0.8}
    assert bench.pmc_traffic("sweep", 4096, lib_id)[0] is None           # another size
    # edit the hash: nulls, with the reason
    pmc["csrc_hash"] = mf["csrc_hash"] = "0123456789abcdef"
    (prof / "pmc.json").write_text(json.dumps(pmc))
    (prof / "mfma.json").write_text(json.dumps(mf))
    t, why = bench.pmc_traffic("sweep", 8192, lib_id)
    assert t is None and "0123456789abcdef" in why and lib_id in why
    b, why = bench.mfma_busy(lib_id)
    assert b is None and "0123456789abcdef" in why
    # no stamp at all (the round-3 files): nulls as well
    del pmc["csrc_hash"]
    (prof / "pmc.json").write_text(json.dumps(pmc))
    assert bench.pmc_traffic("sweep", 8192, lib_id) == (None, "profiles/pmc.json carries no csrc_hash (taken before the sources were stamped)")


def test_the_planners_rule_for_distributing_the_cholesky_is_arithmetic_on_a_measurement():
    """VERDICT r5 #4 / ADVICE r5: whether a shared evaluation's Cholesky is distributed over the rank group is decided from the
    MEASURED cost of one collective of that group (mfgp_comm_calibrate) -- never without one.  The rule itself is host arithmetic
    inside the library (plan.cpp dist_cholesky_pays), exported for exactly this check: saving = (1 - 1/G) N^3/3 flops at 60 TFLOP/s,
    cost = (2 nblk - 1) collectives, taken at saving > 1.25 x cost."""
    import ctypes
    from multifidelity_datafusion_gps_amd import _lib
    lib = _lib.load_library()
    sv, ct = ctypes.c_double(), ctypes.c_double()

    def pays(nblk, size, us):
        return lib.mfgp_dist_cholesky_pays(nblk, size, us, ctypes.byref(sv), ctypes.byref(ct)), sv.value, ct.value

    assert pays(128, 8, 0.0)[0] == 0                       # no measurement: replicated, whatever the size
    assert pays(256, 8, -1.0)[0] == 0
    # collectives on the chain, one exchange per block column inside a macro panel: nblk - 1 all-gathers + one broadcast per macro
    # panel (5 block columns per macro from 80 block columns, 4 from 56): 153 at N = 16384, 79 at N = 8192, 307 at N = 32768
    yes, saving, cost = pays(128, 8, 20.0)                 # N = 16384 on 8 ranks, 20 us per collective
    assert yes == 1 and abs(saving - 0.875 * 16384.0 ** 3 / 3 / 60e12 * 1e3) < 1e-9 and abs(cost - 153 * 20e-3) < 1e-12
    assert pays(128, 8, 150.0)[0] == 0                     # ... at 150 us the 153 collectives cost more than they save
    _, _, cost64 = pays(64, 8, 30.0)
    assert abs(cost64 - 79 * 30e-3) < 1e-12
    assert pays(64, 8, 30.0)[0] == 0 and pays(64, 8, 20.0)[0] == 1      # N = 8192: break-even between 20 and 30 us
    assert pays(64, 1, 1.0)[0] == 0                        # a group of one has nothing to distribute
    assert pays(256, 2, 100.0)[0] == 1 and pays(64, 2, 100.0)[0] == 0   # two ranks over a slow transport: only the large size
