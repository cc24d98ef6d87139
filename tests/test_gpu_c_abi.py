"""The C-ABI from a plain C program (gcc, no Python / torch in that process): include/mfgp.h is the drop-in boundary
for ANY host language.  The client's numbers are checked against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests import tolerances as tol
from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multifidelity_datafusion_gps_amd")


def test_plain_c_client_matches_oracle(tmp_path):
    exe = tmp_path / "abi_client"
    cmd = ["gcc", "-O1", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "abi_client.c"),
           "-o", str(exe), "-L", PKG, "-lmfgp_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    c = cases.make_case("nargp_4d_n64")
    X, Y, Xs, parts, theta, noise = c["X"], c["Y"], c["Xs"], c["parts"], c["theta"], c["noise"]
    case = tmp_path / "case.txt"
    with open(case, "w") as f:
        f.write("%d %d %d %d\n" % (X.shape[0], X.shape[1], len(parts), Xs.shape[0]))
        for p in parts:
            f.write("%d %d %d %d\n" % tuple(p))
        f.write(" ".join(repr(float(t)) for t in theta) + "\n" + repr(float(noise)) + "\n")
        for arr in (X, Y, Xs):
            f.write(" ".join(repr(float(v)) for v in np.asarray(arr).reshape(-1)) + "\n")
    out = subprocess.run([str(exe), str(case)], check=True, capture_output=True, text=True, timeout=120).stdout
    lines = out.strip().splitlines()
    assert "gfx950" in lines[0]
    nlml = float([l for l in lines if l.startswith("nlml")][0].split()[1])
    grad = np.array([float(l.split()[1]) for l in lines if l.startswith("grad")])
    pred = np.array([[float(v) for v in l.split()[1:]] for l in lines if l.startswith("pred")])
    st = orc.inference(parts, theta, noise, X, Y)
    tol.check_nlml(nlml, st["nlml"], label="c_abi_client")
    tol.check_grad(grad, st["grad"], label="c_abi_client")
    mu, var = orc.predict_stable(parts, theta, noise, X, st, Xs)
    ys = max(1.0, np.abs(Y).max())
    tol.check_pred(pred[:, 0], mu, ys, label="c_abi_client", what="mean")
    tol.check_pred(pred[:, 1], var, ys, label="c_abi_client", what="var")
    # mfgp_eval_batch from plain C: set 0 bitwise the single evaluation, the other sets against the oracle at their points
    assert [l for l in lines if l.startswith("batch_set0_bitwise")][0].split()[1] == "1"
    fb = [float(v) for v in [l for l in lines if l.startswith("batch_nlml")][0].split()[1:]]
    for b in (1, 2):
        st_b = orc.inference(parts, np.array(theta) * (1.0 + 0.1 * b), noise * (1.0 + 0.5 * b), X, Y, want_grad=False)
        assert fb[b] == pytest.approx(st_b["nlml"], rel=1e-10)
    from multifidelity_datafusion_gps_amd import build
    assert [l for l in lines if l.startswith("build_id")][0].split()[1] == build.source_hash()
    rc_line = [l for l in lines if l.startswith("null_predict_rc")][0]
    assert int(rc_line.split()[1]) < 0 and "NULL" in rc_line
