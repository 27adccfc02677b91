"""Evidence run (GPU box, via scripts/run_reference_package_on_gpu.sh): the reference's unmodified Python package -- a scratch copy under
oracle/_ref/refpkg, git-ignored, never committed -- drives lib_gpboost_amd.so on the MI355X and reproduces the R suite's goldens
(R-package/tests/testthat/test_GPModel_gaussian_process.R:1144-1148, 1316-1334; test_GPModel_non_Gaussian_data.R) with ITS OWN
GPModel class: creation, likelihood evaluation, fit (GPB_OptimCovPar through the package's fit()), summary getters, prediction."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpboost_amd.libpath import find_lib_path   # noqa: E402
sys.modules.setdefault("optuna", types.ModuleType("optuna"))
fake = types.ModuleType("gpboost.libpath")
fake.find_lib_path = lambda: [find_lib_path()]
sys.modules["gpboost.libpath"] = fake
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref", "refpkg"))
import numpy as np            # noqa: E402
import gpboost as gpb         # noqa: E402   (the reference's package)
from oracle import orc        # noqa: E402   (only for the R fixture's inputs)

print("package:", gpb.__file__, "| library:", find_lib_path(), flush=True)
coords, y = orc.r_fixture()
m = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none", likelihood="gaussian")
nll = m.neg_log_likelihood(cov_pars=np.array([0.1, 1.6, 0.2]), y=y)
print("neg_log_likelihood = %.7f (R golden 124.2252524)" % nll, flush=True)
assert abs(nll - 124.2252524) < 1e-6
params = {"optimizer_cov": "gradient_descent", "lr_cov": 0.1, "use_nesterov_acc": True, "acc_rate_cov": 0.5, "delta_rel_conv": 1e-6,
          "maxit": 1000, "convergence_criterion": "relative_change_in_parameters",
          "init_cov_pars": np.array([np.var(y, ddof=1) / 2, np.var(y, ddof=1) / 2, 0.0])}
from scipy.spatial.distance import pdist   # noqa: E402
params["init_cov_pars"][2] = pdist(coords).mean() / 3
m = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none", likelihood="gaussian")
m.fit(y=y, params=params)
cp = m.get_cov_pars(format_pandas=False)
print("fit: cov pars", np.asarray(cp).ravel(), "iterations", m._get_num_optim_iter(), "nll %.7f" % m.get_current_neg_log_likelihood(), flush=True)
assert m._get_num_optim_iter() == 378
assert np.abs(np.asarray(cp).ravel()[:3] - np.array([0.03297349, 1.07691542, 0.11378505])).sum() < 1e-6
assert abs(m.get_current_neg_log_likelihood() - 122.7680889) < 1e-6
coord_test = np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]])
m.set_prediction_data(vecchia_pred_type="order_obs_first_cond_obs_only", num_neighbors_pred=30)
pred = m.predict(y=y, gp_coords_pred=coord_test, predict_cov_mat=True, predict_response=True)
print("predict: mu", pred["mu"], "cov diag", np.diag(pred["cov"]), flush=True)
assert np.abs(pred["mu"] - np.array([0.06968068, 0.06967750, 0.44208925])).sum() < 1e-6
assert np.abs(np.diag(pred["cov"]) - np.array([0.6214955, 0.6215069, 0.4199531])).sum() < 1e-6
print(m.summary() if hasattr(m, "summary") else "", flush=True)
# linear regression term: the package's fit(y, X) -> GPB_OptimLinRegrCoefCovPar, get_coef -> GPB_GetCoef, predict(X_pred) (tests/cases.py:COEF_CASES,
# tests/golden/optim_coef_ref.npz = the reference LIBRARY on the same inputs)
from tests import cases       # noqa: E402
gc = np.load(os.path.join(ROOT, "tests", "golden", "optim_coef_ref.npz"))
cx, yx, X, mc, init, cfg, Xp = cases.coef_case("r_m30_none_wls_default")
mx = gpb.GPModel(gp_coords=cx, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none", likelihood="gaussian")
mx.fit(y=yx, X=X)
coef = np.asarray(mx.get_coef(std_err=True, format_pandas=False))
print("fit with X: cov pars", np.asarray(mx.get_cov_pars(format_pandas=False)).ravel(), "coef", coef.ravel(), "iterations", mx._get_num_optim_iter(), flush=True)
assert mx._get_num_optim_iter() == int(gc["r_m30_none_wls_default_num_it"])
assert np.allclose(coef[0].ravel() if coef.ndim > 1 else coef[:2], gc["r_m30_none_wls_default_coef"], rtol=1e-6)
assert np.allclose(coef[1].ravel() if coef.ndim > 1 else coef[2:], gc["r_m30_none_wls_default_coef_sd"], rtol=1e-6)
px = mx.predict(gp_coords_pred=cases.COEF_PRED_COORDS, X_pred=Xp, predict_var=True, predict_response=True)
print("predict with X_pred: mu", px["mu"], "var", px["var"], flush=True)
assert np.allclose(px["mu"], gc["r_m30_none_wls_default_pred_mu"], rtol=1e-6) and np.allclose(px["var"], gc["r_m30_none_wls_default_pred_var"], rtol=1e-6)
print(mx.summary() if hasattr(mx, "summary") else "", flush=True)
# non-Gaussian: the package's GPModel with likelihood = "bernoulli_logit" (Vecchia-Laplace, iterative methods) on seeded data
rng = np.random.default_rng(21)
c2 = rng.uniform(size=(2000, 2))
lat = 1.5 * np.sin(5 * c2[:, 0]) * np.cos(3 * c2[:, -1]) + 0.3
y2 = (rng.uniform(size=2000) < 1.0 / (1.0 + np.exp(-lat))).astype(np.float64)
mb = gpb.GPModel(gp_coords=c2, cov_function="exponential", gp_approx="vecchia", num_neighbors=20, vecchia_ordering="random", likelihood="bernoulli_logit", seed=1)
g = np.load(os.path.join(ROOT, "tests", "golden", "laplace_ref.npz"))
v = mb.neg_log_likelihood(cov_pars=np.array([1.0, 0.1]), y=y2)
print("bernoulli_logit neg_log_likelihood = %.9f (reference library %.9f)" % (v, float(g["lap_u2d_n2000_exp_m20_negll_0"])), flush=True)
assert abs(v - float(g["lap_u2d_n2000_exp_m20_negll_0"])) <= 1e-8 * abs(v)
# likelihoods with an auxiliary parameter (round 5): the package's GPModel(likelihood = "gamma") -- evaluation at a given shape (aux_pars -> GPB_SetOptimConfig(init_aux_pars)),
# the fit with the shape estimated (the package's default estimate_aux_pars = True) and get_aux_pars (GPB_GetAuxPars) against the reference LIBRARY's values
# (tests/golden/laplace_aux_ref.npz)
ga = np.load(os.path.join(ROOT, "tests", "golden", "laplace_aux_ref.npz"))
for name in ("gamma_n1500", "negbin_n1500"):
    ac = cases.LAPLACE_AUX_CASES[name]
    c = cases.LAPLACE_CASES[ac["model"]]
    ca, ya = cases.make_aux_data(ac)
    kw = dict(gp_coords=ca, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"], vecchia_ordering=c["ordering"],
              likelihood=ac["lik"], seed=c["seed"])
    mg = gpb.GPModel(**kw)
    v = mg.neg_log_likelihood(cov_pars=np.asarray(c["cov_pars"][0], dtype=np.float64), y=ya, aux_pars=np.array([ac["aux"]]))
    print("%s neg_log_likelihood(aux_pars = %g) = %.9f (reference library %.9f)" % (ac["lik"], ac["aux"], v, float(ga[name + "_negll_0"])), flush=True)
    assert abs(v - float(ga[name + "_negll_0"])) <= 1e-8 * abs(v)
    mg = gpb.GPModel(**kw)
    mg.fit(y=ya)
    cpg = np.asarray(mg.get_cov_pars(format_pandas=False)).ravel()
    aux = np.asarray(mg.get_aux_pars(format_pandas=False)).ravel()
    print("%s fit: cov pars %s, shape %s, %d iterations, nll %.7f (reference library: %s, %s, %d, %.7f)" %
          (ac["lik"], cpg, aux, mg._get_num_optim_iter(), mg.get_current_neg_log_likelihood(), ga[name + "_fit_cov_pars"], ga[name + "_fit_aux"],
           int(ga[name + "_fit_num_it"]), float(ga[name + "_fit_negll"])), flush=True)
    assert mg._get_num_optim_iter() == int(ga[name + "_fit_num_it"])
    assert np.allclose(cpg[:2], ga[name + "_fit_cov_pars"], rtol=1e-4) and np.allclose(aux[:1], ga[name + "_fit_aux"], rtol=1e-4)
    assert abs(mg.get_current_neg_log_likelihood() - float(ga[name + "_fit_negll"])) <= 1e-7 * abs(float(ga[name + "_fit_negll"]))
# sample weights and proportions (round 5): the package's GPModel(likelihood = "poisson" / "gamma", weights = w) and GPModel(likelihood = "binomial", weights = trials)
# -- evaluation at given parameters and the tight-threshold fit against the reference LIBRARY's values (tests/golden/laplace_weights_ref.npz)
gw = np.load(os.path.join(ROOT, "tests", "golden", "laplace_weights_ref.npz"))
for name in ("w_poisson_n2000", "w_gamma_n1500", "binomial_logit_n1500"):
    wc = cases.LAPLACE_WEIGHT_CASES[name]
    c = cases.LAPLACE_CASES[wc["model"]]
    cw, yw, ww = cases.make_weight_data(wc)
    lik_pkg = "binomial" if wc["lik"] == "binomial_logit" else wc["lik"]          # (the package's alias, ParseLikelihoodAlias)
    kw = dict(gp_coords=cw, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"], vecchia_ordering=c["ordering"],
              likelihood=lik_pkg, seed=c["seed"], weights=ww)
    mw = gpb.GPModel(**kw)
    mw.set_optim_params(params=dict(cases.LAPLACE_TIGHT))
    akw = dict(aux_pars=np.array([wc["aux"]])) if "aux" in wc else {}
    v = mw.neg_log_likelihood(cov_pars=np.asarray(c["cov_pars"][0], dtype=np.float64), y=yw, **akw)
    print("%s with weights: neg_log_likelihood = %.9f (reference library %.9f)" % (lik_pkg, v, float(gw[name + "_negll_direct"])), flush=True)
    assert abs(v - float(gw[name + "_negll_direct"])) <= 1e-8 * abs(v)
    mw = gpb.GPModel(**kw)
    mw.fit(y=yw, params=dict(cases.LAPLACE_TIGHT))
    cpw = np.asarray(mw.get_cov_pars(format_pandas=False)).ravel()
    print("%s with weights, fit: cov pars %s, %d iterations, nll %.7f (reference library: %s, %d, %.7f)" %
          (lik_pkg, cpw[:2], mw._get_num_optim_iter(), mw.get_current_neg_log_likelihood(), gw[name + "_fit_tight_cov_pars"], int(gw[name + "_fit_tight_num_it"]),
           float(gw[name + "_fit_tight_negll"])), flush=True)
    assert mw._get_num_optim_iter() == int(gw[name + "_fit_tight_num_it"])
    assert np.allclose(cpw[:2], gw[name + "_fit_tight_cov_pars"], rtol=1e-6)
# round 5, the widening of this round through the package: tests/route_a_driver.py's scenario round5_widening (a Student-t fit with two auxiliary parameters, the refusal to
# change the preconditioner after a fit, a t evaluation with the fitc preconditioner, a beta fit with pivoted_cholesky) on the MI355X against the reference LIBRARY's results
# (tests/golden/route_a_round5_widening_ref.json = the same driver on oracle/_ref/lib_gpboost_ref.so)
import json, subprocess       # noqa: E402
r5 = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "route_a_driver_gpu.py"), find_lib_path(), "round5_widening", ROOT], capture_output=True, text=True, cwd=ROOT)
lines5 = [l for l in r5.stdout.splitlines() if l.startswith("RESULT ")]
assert lines5, r5.stdout[-2000:] + r5.stderr[-3000:]
ours5 = json.loads(lines5[-1][7:]); ref5 = json.load(open(os.path.join(ROOT, "tests", "golden", "route_a_round5_widening_ref.json")))
assert sorted(ours5) == sorted(ref5)
for k5 in ref5:
    x5, y5 = np.asarray(ours5[k5], dtype=float), np.asarray(ref5[k5], dtype=float)
    if k5 == "num_it" or k5 == "beta_num_it": assert np.array_equal(x5, y5), (k5, x5, y5)
    elif k5.startswith("stoch_"): assert np.allclose(x5, y5, rtol=0.2), k5
    elif k5.startswith("flat_"): assert np.allclose(x5, y5, rtol=1e-5, atol=1e-8), (k5, x5, y5)
    else: assert np.allclose(x5, y5, rtol=1e-6, atol=1e-8), (k5, x5, y5)
print("round5_widening through the package: t fit cov pars %s, (scale, df) %s, %d iterations; beta fit (pivoted_cholesky) cov pars %s, precision %s -- equal to the reference library's" %
      (ours5["t_cov_pars"], ours5["t_aux"], ours5["num_it"], ours5["flat_beta_cov_pars"], ours5["flat_beta_aux"]), flush=True)
# ... and the lognormal likelihood (fourth slice): scenario round5_lognormal against tests/golden/route_a_round5_lognormal_ref.json (the same driver on oracle/_ref/lib_gpboost_ref.so)
r6 = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "route_a_driver_gpu.py"), find_lib_path(), "round5_lognormal", ROOT], capture_output=True, text=True, cwd=ROOT)
lines6 = [l for l in r6.stdout.splitlines() if l.startswith("RESULT ")]
assert lines6, r6.stdout[-2000:] + r6.stderr[-3000:]
ours6 = json.loads(lines6[-1][7:]); ref6 = json.load(open(os.path.join(ROOT, "tests", "golden", "route_a_round5_lognormal_ref.json")))
assert sorted(ours6) == sorted(ref6)
for k6 in ref6:
    x6, y6 = np.asarray(ours6[k6], dtype=float), np.asarray(ref6[k6], dtype=float)
    if k6 == "num_it": assert np.array_equal(x6, y6), (k6, x6, y6)
    elif k6.startswith("stoch_"): assert np.allclose(x6, y6, rtol=0.2), k6
    elif k6.startswith("stochm_"): assert np.allclose(x6, y6, rtol=5e-3), k6
    else: assert np.allclose(x6, y6, rtol=1e-6, atol=1e-8), (k6, x6, y6)
print("round5_lognormal through the package: fit cov pars %s, log-variance %s, %d iterations, negll %.8f; evaluation with pivoted_cholesky %.8f -- equal to the reference library's" %
      (ours6["ln_cov_pars"], ours6["ln_aux"], ours6["num_it"], ours6["ln_nll"], ours6["ln_nll_eval_pivchol"]), flush=True)
# round 6: gp_approx = "full_scale_vecchia" with non-Gaussian likelihoods through the package (scenario round6_vif_non_gaussian: a logit evaluation and lbfgs fit, a gamma fit
# with its shape) against tests/golden/route_a_round6_vif_non_gaussian_ref.json (the same driver on oracle/_ref/lib_gpboost_ref.so)
r7 = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "route_a_driver_gpu.py"), find_lib_path(), "round6_vif_non_gaussian", ROOT], capture_output=True, text=True, cwd=ROOT)
lines7 = [l for l in r7.stdout.splitlines() if l.startswith("RESULT ")]
assert lines7, r7.stdout[-2000:] + r7.stderr[-3000:]
ours7 = json.loads(lines7[-1][7:]); ref7 = json.load(open(os.path.join(ROOT, "tests", "golden", "route_a_round6_vif_non_gaussian_ref.json")))
assert sorted(ours7) == sorted(ref7)
for k7 in ref7:
    x7, y7 = np.asarray(ours7[k7], dtype=float), np.asarray(ref7[k7], dtype=float)
    if k7 in ("num_it", "vg_num_it"): assert np.array_equal(x7, y7), (k7, x7, y7)
    elif k7.startswith("stoch_"): assert np.allclose(x7, y7, rtol=0.45), (k7, x7, y7)      # (the reference SIMULATES these variances, nsim_var_pred samples: up to 26 % off the exact values seen)
    elif k7.startswith("stochm_"): assert np.allclose(x7, y7, rtol=1e-2), (k7, x7, y7)
    elif k7.startswith("flat_"): assert np.allclose(x7, y7, rtol=5e-5, atol=1e-8), (k7, x7, y7)
    else: assert np.allclose(x7, y7, rtol=1e-6, atol=1e-8), (k7, x7, y7)
print("round6_vif_non_gaussian through the package: logit evaluation %.8f, fit cov pars %s in %d iterations (negll %.8f); gamma fit cov pars %s, shape %s, %d iterations -- equal to the reference library's" %
      (ours7["vl_nll_eval"], ours7["vl_cov_pars"], ours7["num_it"], ours7["vl_nll"], ours7["flat_vg_cov_pars"], ours7["flat_vg_aux"], ours7["vg_num_it"][0]), flush=True)
print("REFERENCE PACKAGE ON MI355X: OK", flush=True)
