"""VIF x non-Gaussian likelihood at config 4's size on the MI355X: one evaluation and a short lbfgs fit (scripts/gpu_run.sh py:...)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lik = sys.argv[2] if len(sys.argv) > 2 else "bernoulli_logit"
rng = np.random.default_rng(1)
c = rng.uniform(size=(n, 2))
lat = 0.9 * np.sin(5 * c[:, 0]) * np.cos(3 * c[:, 1])
y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-1.5 * lat))).astype(np.float64) if lik == "bernoulli_logit" else rng.poisson(np.exp(0.5 * lat)).astype(np.float64)
t0 = time.perf_counter()
m = gpboost_amd.GPModel(likelihood=lik, gp_coords=c, cov_function="exponential", gp_approx="full_scale_vecchia", num_neighbors=30, num_ind_points=200, vecchia_ordering="random", seed=1)
print("setup %.2f s" % (time.perf_counter() - t0), flush=True)
for k in range(3):
    t0 = time.perf_counter()
    v = m.neg_log_likelihood(np.array([1.0 + 0.01 * k, 0.1]), y)
    i = m.laplace_info()
    print("eval %d: negll %.6f in %.3f s (factor %.1f ms, mode %.1f ms / %d Newton / %d CG, logdet %.1f ms / %d Lanczos)" %
          (k, v, time.perf_counter() - t0, i["ms_factor"], i["ms_mode"], i["newton_it"], i["cg_it"], i["ms_logdet"], i["lanczos_it"]), flush=True)
t0 = time.perf_counter()
m.fit(y, params={"optimizer_cov": "lbfgs", "init_cov_pars": np.array([1.0, 0.1]), "maxit": 5})
print("lbfgs fit: %d iterations in %.2f s -> %s, negll %.6f" % (m.get_num_optim_iter(), time.perf_counter() - t0, m.get_cov_pars(), m.get_current_neg_log_likelihood()), flush=True)
