"""Exact-GP path timing on the MI355X: covariance assembly, blocked Cholesky, solves (HIP events inside the library)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd                      # noqa: E402
from gpboost_amd import shim            # noqa: E402

gpboost_amd.set_device(0)
for n in [int(a) for a in sys.argv[1:]] or [2000, 8192, 16384]:
    rng = np.random.default_rng(1)
    coords = rng.uniform(size=(n, 2)); y = rng.standard_normal(n)
    ex = shim.ExactState(coords); ex.set_y(y)
    ex.nll_terms(1, 10.0, np.sqrt(3.0) / 0.1)
    vals = []
    for _ in range(3):
        t, _, ms3 = ex.nll_terms(1, 10.0, np.sqrt(3.0) / 0.1)
        vals.append((t[0], t[1], ms3[0], ms3[1], ms3[2]))
    ms = np.median(np.array([v[2:] for v in vals]), axis=0)
    assert all(v[0] == vals[0][0] and v[1] == vals[0][1] for v in vals), "not reproducible run to run"
    print("n=%d: assembly %.3f ms, cholesky %.3f ms (%.1f TFLOP/s), solves %.3f ms; yPy=%.10g logdet=%.10g" %
          (n, ms[0], ms[1], n ** 3 / 3.0 / (ms[1] * 1e-3) / 1e12, ms[2], vals[0][0], vals[0][1]))
    ex.close()
