"""Full-scale Vecchia (VIF) on the MI355X: model creation, likelihood evaluation and the analytic gradient at n = 1e5, m = 30, 200 inducing points
(tests/cases.py: vif_u2d_n100000_exp_m30_k200_random), checked against the reference's values (tests/golden/vif_ref.npz, vif_grad_ref.npz).
    python scripts/gpu_vif_bench.py [case]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
from tests import cases

name = sys.argv[1] if len(sys.argv) > 1 else "vif_u2d_n100000_exp_m30_k200_random"
n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
coords, y = cases.vif_data(name)
g = np.load(os.path.join(ROOT, "tests", "golden", "vif_grad_ref.npz"))
t0 = time.perf_counter()
mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="full_scale_vecchia", num_neighbors=m, num_ind_points=k,
                          vecchia_ordering=ordering, seed=seed)
t_setup = time.perf_counter() - t0
cp = np.array([0.1, 1.0, 0.1])
v = mdl.neg_log_likelihood(cp, y)
ref = float(g[name + "_negll_0"])
out = {"case": name, "setup_s": round(t_setup, 3), "negll": v, "negll_rel_err": abs(v - ref) / abs(ref)}
ts = []
for r in range(5):
    t0 = time.perf_counter(); mdl.neg_log_likelihood(cp * (1 + 0.001 * r)); ts.append(time.perf_counter() - t0)
out["eval_ms"] = round(1e3 * float(np.median(ts)), 3)
nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
gref = g[name + "_grad_0"]
out["grad"] = grad.tolist(); out["grad_ref"] = gref.tolist(); out["grad_rel_err"] = float(np.abs(grad - gref).max() / np.abs(gref).max())
ts = []
for r in range(5):
    t0 = time.perf_counter(); mdl.neg_log_likelihood_and_gradient(cp * (1 + 0.001 * r), y); ts.append(time.perf_counter() - t0)
out["eval_with_grad_ms"] = round(1e3 * float(np.median(ts)), 3)
out["grad_cost_in_evaluations"] = round(out["eval_with_grad_ms"] / out["eval_ms"], 2)
t0 = time.perf_counter()
mdl.fit(y, params={"optimizer_cov": "lbfgs", "init_cov_pars": cp})
out["lbfgs_fit_s"] = round(time.perf_counter() - t0, 3); out["fit_iterations"] = int(mdl.get_num_optim_iter()); out["fit_cov_pars"] = mdl.get_cov_pars().tolist()
print(json.dumps(out))
