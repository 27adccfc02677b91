"""Timing of the Vecchia-Laplace path (BASELINE config 4) on the MI355X, with the reference on the host beside it.
    python scripts/gpu_laplace.py [n] [m] [--ref] [--pivchol | --fitc] [--lik <likelihood>] [--fit]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
from tests import cases

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 30
if "--surface" in sys.argv:      # labels around a smooth latent surface (as in the parity tests)
    coords, y = cases.synthetic_binary(n, 2, seed=1)
else:                            # coords U[0,1]^2, y ~ Bernoulli(0.5), default_rng(1)
    rng = np.random.default_rng(1)
    coords = rng.uniform(size=(n, 2))
    y = (rng.uniform(size=n) < 0.5).astype(np.float64)
LIK = sys.argv[sys.argv.index("--lik") + 1] if "--lik" in sys.argv else "bernoulli_logit"
if LIK != "bernoulli_logit":     # round 5: the likelihoods with auxiliary parameters at config 4's size -- a smooth latent surface, the response drawn from the likelihood
    rng = np.random.default_rng(1)
    coords = rng.uniform(size=(n, 2))
    lat = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, 1]) + 0.2
    y = {"poisson": lambda: rng.poisson(np.exp(lat)).astype(np.float64),
         "gamma": lambda: rng.gamma(2.0, np.exp(lat) / 2.0),
         "negative_binomial": lambda: rng.negative_binomial(3.0, 3.0 / (3.0 + np.exp(lat))).astype(np.float64),
         "lognormal": lambda: np.exp(lat - 0.1 + np.sqrt(0.2) * rng.normal(size=n)),
         "t": lambda: lat + 0.35 * rng.standard_t(4, size=n)}[LIK]()
t0 = time.perf_counter()
mdl = gpboost_amd.GPModel(likelihood=LIK, gp_coords=coords, cov_function="exponential", gp_approx="vecchia",
                          num_neighbors=m, vecchia_ordering="random", seed=1)
print("setup %.3f s" % (time.perf_counter() - t0), flush=True)
if "--fitc" in sys.argv:         # cg_preconditioner_type = "fitc" (200 inducing points by kmeans++)
    mdl.set_optim_params({"cg_preconditioner_type": "fitc"})
    print("preconditioner:", mdl.get_cg_preconditioner_type(), flush=True)
if "--pivchol" in sys.argv:      # cg_preconditioner_type = "pivoted_cholesky" (rank 50) instead of the default "vadu"
    mdl.set_optim_params({"cg_preconditioner_type": "pivoted_cholesky"})
    print("preconditioner:", mdl.get_cg_preconditioner_type(), flush=True)
cp = np.array([1.0, 0.1])
res = {}
for k in range(3):
    t0 = time.perf_counter()
    v = mdl.neg_log_likelihood(cp * (1 + 0.01 * k), y)
    dt = time.perf_counter() - t0
    info = mdl.laplace_info()
    print("eval %d: negll %.10f  %.3f s  %s" % (k, v, dt, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in info.items()}), flush=True)
    res["gpu_s_%d" % k] = dt; res["negll_%d" % k] = v
if "--fit" in sys.argv:          # lbfgs fit (covariance and auxiliary parameters) at this size
    t0 = time.perf_counter()
    mdl.fit(y)
    dt = time.perf_counter() - t0
    aux = mdl.get_aux_pars() if mdl.get_num_aux_pars() > 0 else None
    print("fit %s n=%d: %d iterations, cov pars %s, aux pars %s, negll %.6f, %.2f s" % (LIK, n, mdl.get_num_optim_iter(), np.asarray(mdl.get_cov_pars()).ravel(), aux,
                                                                                  mdl.get_current_neg_log_likelihood(), dt), flush=True)
if "--ref" in sys.argv:
    from oracle import refdrv
    if refdrv.available():
        rm = refdrv.RefCAPIModel(coords, "exponential", 0.5, m, "random", 1, threads=-1, likelihood="bernoulli_logit")
        for k in range(2):
            t0 = time.perf_counter(); rv = rm.neg_log_likelihood(cp * (1 + 0.01 * k), y); dt = time.perf_counter() - t0
            print("reference eval %d: negll %.10f  %.3f s  rel diff %.3e  cores %d" % (k, rv, dt, abs(rv - res["negll_%d" % k]) / abs(rv), os.cpu_count()), flush=True)
