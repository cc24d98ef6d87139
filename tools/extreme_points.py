"""What an evaluation at the edge of the parameter domain returns -- the HIP engine beside the CPU double (tests/oracle_engine.py) through
the same host layer (engine.GPRegression._objective_grads): optimizer-space points whose softplus image is ~1e-304 or ~700."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multifidelity_datafusion_gps_amd import engine as gp
from tests.oracle_engine import OracleEngine

rng = np.random.default_rng(0)
X = rng.uniform(size=(60, 2)); Y = (np.sin(5 * X[:, :1]) + X[:, 1:])
POINTS = {"sane": [0.5, 0.2, -3.0], "tiny lengthscale": [0.5, -700.0, -3.0], "tiny variance": [-700.0, 0.2, -3.0], "tiny noise": [0.5, 0.2, -700.0],
          "huge lengthscale": [0.5, 700.0, -3.0], "huge variance": [700.0, 0.2, -3.0], "all tiny": [-700.0, -700.0, -700.0],
          "huge variance, tiny noise": [700.0, 3.0, -700.0], "nan": [np.nan, 0.2, -3.0], "inf": [0.5, np.inf, -3.0]}
use_gpu = "--cpu" not in sys.argv
for name, x in POINTS.items():
    row = []
    for label, mk in (("double", lambda: OracleEngine()),) + ((("hip", lambda: None),) if use_gpu else ()):
        m = gp.GPRegression(X, Y, kernel=gp.RBF(2), engine=mk())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f, g = m._objective_grads(np.array(x))
        row.append("%s: f = %-12.6g %s g = %s" % (label, f, "FAILED" if f == np.finfo(float).max else "      ", np.array2string(g, precision=4)))
    print("%-26s %s" % (name, "  |  ".join(row)), flush=True)
