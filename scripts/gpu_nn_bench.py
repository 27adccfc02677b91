"""Model creation (ordering + ordered neighbour search + upload) on the MI355X at BASELINE's sizes; prints seconds per creation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd          # noqa: E402

gpboost_amd.set_device(0)
for n, d, m in ((100000, 2, 30), (1000000, 2, 30), (1000000, 3, 40)):
    rng = np.random.default_rng(1)
    coords = rng.uniform(size=(n, d))
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
        ts.append(time.perf_counter() - t0)
        del mdl
    print("n=%d d=%d m=%d: model creation %.3f s (best of 3: %s)" % (n, d, m, min(ts), ", ".join("%.3f" % t for t in ts)), flush=True)
