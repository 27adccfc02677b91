"""PCIe-inclusive rate of the headline metric: GPB_EvalNegLogLikelihood through the reference-shaped C API, i.e. with the host
response vector (8 MB at n = 1e6) permuted and uploaded on every call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
n, m = 1000000, 30
rng = np.random.default_rng(1)
coords = rng.uniform(size=(n, 2)); y = rng.standard_normal(n)
mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
cp = np.array([0.1, 1.0, 0.1])
for k in range(5):
    mdl.neg_log_likelihood(cp, y)
ts = []
for k in range(20):
    t0 = time.perf_counter(); mdl.neg_log_likelihood(cp * (1 + 0.001 * k), y); ts.append(time.perf_counter() - t0)
print("GPB_EvalNegLogLikelihood incl. host permutation + H2D of y: median %.3f ms -> %.1f evals/s" % (1e3 * np.median(ts), 1 / np.median(ts)))
