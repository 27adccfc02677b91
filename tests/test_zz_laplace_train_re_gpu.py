"""GPU: GPB_PredictREModelTrainingDataRandomEffects for the non-Gaussian Vecchia models -- the mode of the latent process at the training locations and
diag((Sigma^-1 + W)^-1) (re_model_template.h:4683-4725; Likelihood::CalcVarLaplaceApproxVecchia) -- against the unmodified reference's exact ("cholesky")
values, tests/golden/laplace_train_re_ref.npz (oracle/make_golden.py laplace_train_re).  The oracle side of the same comparison:
tests/test_laplace_predvar.py.

This file sorts last on purpose: its device half (gpb_hip_vecchia_laplace_mode_var = the block solves of the predictive variances on unit vectors,
which ARE validated on the MI355X, profiles/r03_zzz_predvar_quick.log) was added after the GPU budget of round 3 was spent; its first device run was the driver's at the end of round 3 (all passed), since round 4 these are hard tests."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "laplace_train_re_ref.npz")
pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit", "poisson"])
def test_training_data_random_effects_of_non_gaussian_models(lib_built, lik):
    import gpboost_amd as gpb
    g = np.load(GOLD)
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    mu = mdl.predict_training_data_random_effects(y=y, cov_pars=cp)
    np.testing.assert_allclose(mu, g[lik + "_mu"], rtol=1e-5, atol=1e-6)
    mv = mdl.predict_training_data_random_effects(y=y, cov_pars=cp, predict_var=True)
    np.testing.assert_allclose(mv[:, 0], g[lik + "_mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv[:, 1], g[lik + "_var"], rtol=1e-5)


def test_training_data_random_effects_with_repeated_locations(lib_built):
    import gpboost_amd as gpb
    g = np.load(GOLD)
    cf, sh, m, ordering, seed = cases.LAPLACE_DUP_CASES["dup_mat15_m20_random"]
    coords, y, fe, _ = cases.laplace_dup_data("bernoulli_logit")
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering=ordering, seed=seed)
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    mv = mdl.predict_training_data_random_effects(y=y, cov_pars=np.asarray(cases.LAPLACE_DUP_COV_PARS[0], dtype=np.float64), predict_var=True)
    np.testing.assert_allclose(mv[:, 0], g["dup_bernoulli_logit_mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv[:, 1], g["dup_bernoulli_logit_var"], rtol=1e-5)


def test_r_suite_prediction_goldens_of_the_logit_model(orc, lib_built):
    """R-package/tests/testthat/test_GPModel_non_Gaussian_data.R:2510-2537 on the device: the exact GP as a Vecchia model on all predecessors
    (num_neighbors = n - 1 = 99, prediction on all 100 observed points: the 128-lane kernels), latent mean / variances / response mean at the R test's
    fitted parameters.  The oracle side: tests/test_laplace_predvar.py::test_oracle_reproduces_the_r_suite_prediction_goldens."""
    import gpboost_amd as gpb
    coords, y = orc.r_fixture_logit()
    n = len(y)
    ct = np.array([[0.1, 0.9], [0.11, 0.91], [0.7, 0.55]])
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=n - 1,
                      vecchia_ordering="none")
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    cp = np.array([1.4300136, 0.1891952])
    pr = mdl.predict(y=y, gp_coords_pred=ct, cov_pars=cp, predict_var=True, predict_response=False, num_neighbors_pred=n)
    assert np.abs(pr["mu"] - [-0.7792960, -0.7876208, 0.5476390]).sum() < 1e-6
    assert np.abs(pr["var"] - [1.024266883, 1.022897212, 0.7395745025]).sum() < 2e-6
    pr = mdl.predict(y=y, gp_coords_pred=ct, cov_pars=cp, predict_var=True, predict_response=True, num_neighbors_pred=n)
    assert np.abs(pr["mu"] - [0.3442815, 0.3426873, 0.6159933]).sum() < 1e-6
    assert np.abs(pr["var"] - pr["mu"] * (1 - pr["mu"])).sum() < 1e-12


def test_r_suite_prediction_goldens_of_the_probit_model(orc, lib_built):
    """test_GPModel_non_Gaussian_data.R:1391-1432 on the device (Vecchia on all predecessors = the exact GP): latent mean / variances at cov_pars
    (1, 0.2) without a linear predictor, and with the fitted linear predictor handed over as fixed effects (fixed_effects / fixed_effects_pred of
    GPB_PredictREModel) the latent and the response predictions.  Oracle side: tests/test_laplace_predvar.py."""
    import ctypes as C
    import gpboost_amd as gpb
    from gpboost_amd.basic import _lib, _safe_call
    from tests.test_laplace_predvar import R_PROBIT_CT, r_probit_design
    coords, y = orc.r_fixture_probit()
    n = len(y)
    mdl = gpb.GPModel(likelihood="bernoulli_probit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=n - 1,
                      vecchia_ordering="none")
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    cp = np.array([1.0, 0.2])
    pr = mdl.predict(y=y, gp_coords_pred=R_PROBIT_CT, cov_pars=cp, predict_var=True, predict_response=False, num_neighbors_pred=n)
    assert np.abs(pr["mu"] - [0.01874013, 0.01200800, 0.20498871]).sum() < 1e-5
    assert np.abs(pr["var"] - [0.6105248, 0.6093745, 0.4235374]).sum() < 1e-6
    # with the linear predictor: the C API's fixed_effects / fixed_effects_pred arguments (GPModel.predict of this package does not expose them)
    X, Xt, beta = r_probit_design()
    fe = np.ascontiguousarray(X @ beta); fep = np.ascontiguousarray(Xt @ beta)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    cpc = np.asfortranarray(R_PROBIT_CT)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    for resp, emu, evar in ((False, [0.3389905, 0.1512445, -0.1039307], [0.6193228722, 0.6159348965, 0.4291674143]),
                            (True, [0.6050312, 0.5473537, 0.4653610], [0.2389684, 0.2477576, 0.2488001])):
        out = np.empty(6)
        _safe_call(_lib().GPB_PredictREModel(mdl.handle, P(yv), C.c_int(3), P(out), C.c_bool(False), C.c_bool(True), C.c_bool(resp), C.c_bool(False),
                                             C.c_bool(False), C.c_int(0), C.c_int(0), C.c_void_p(), C.c_void_p(), C.c_void_p(), P(cpc), C.c_void_p(),
                                             P(cp), C.c_void_p(), C.c_bool(False), P(fe), P(fep)))
        assert np.abs(out[:3] - emu).sum() < 1e-6
        assert np.abs(out[3:] - evar).sum() < 1e-6


@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit", "poisson"])
def test_standard_errors_of_non_gaussian_models_after_a_fit(lib_built, lik):
    """GPB_GetCovPar(calc_std_dev = true) after GPModel.fit of a non-Gaussian model: the numerical Jacobian of the DEVICE gradient (host half tested
    on the CPU with the oracle as evaluator, tests/test_laplace_predvar.py) against the reference's standard errors after its own fit
    (tests/golden/laplace_stderr_ref.npz); tolerance as there (the quantity is a second difference of an iteratively computed gradient)."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_stderr_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.fit(y)
    out = mdl.get_cov_pars(std_err=True)
    np.testing.assert_allclose(out[:2], g[lik + "_cov_pars"], rtol=1e-4)
    np.testing.assert_allclose(out[2:], g[lik + "_std"], rtol=0.05)


@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit", "poisson"])
def test_fit_and_predict_with_covariates_for_non_gaussian_models(lib_built, lik):
    """GPModel.fit(y, X) of a non-Gaussian model on the device (coefficients in the lbfgs vector; host optimiser tested on the CPU with the oracle as
    evaluator, tests/test_laplace_coef.py) against the reference's own fit with its default tolerances (tests/golden/laplace_coef_ref.npz): iterations
    +- 1, likelihood 1e-5, estimates within the noise of gradients from CG solves stopped at |r| < 1e-2; then the latent prediction with X_pred."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_coef_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y, X = cases.laplace_coef_data(lik, 2)
    key = "%s_p2" % lik
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.fit(y, X=X, params={"init_coef_aux_pars_from_iid_model": False})          # as the fixture's fit (intercept from the data, zeros otherwise)
    assert abs(mdl.get_num_optim_iter() - int(g[key + "_num_it"])) <= 1
    nll = mdl.get_current_neg_log_likelihood()
    assert abs(nll - float(g[key + "_negll"])) <= 1e-5 * abs(nll)
    np.testing.assert_allclose(mdl.get_cov_pars(), g[key + "_cov_pars"], rtol=0.06)
    np.testing.assert_allclose(mdl.get_coef(), g[key + "_coef"], rtol=0.02, atol=2e-3)
    pr = mdl.predict(y=y, gp_coords_pred=g[key + "_pred_coords"], X_pred=g[key + "_pred_X"], predict_var=False, predict_response=False)
    np.testing.assert_allclose(pr["mu"], g[key + "_pred_latent_mu"], rtol=0.02, atol=5e-3)
    pr = mdl.predict(y=y, gp_coords_pred=g[key + "_pred_coords"], X_pred=g[key + "_pred_X"], predict_var=True, predict_response=True)
    assert np.all(np.isfinite(pr["mu"])) and np.all(pr["var"] > 0)
    with pytest.raises(gpb.GPBoostError, match="covariate_data_pred"):
        mdl.predict(y=y, gp_coords_pred=g[key + "_pred_coords"], predict_var=False, predict_response=False)


def test_standard_errors_of_the_coefficients_of_a_non_gaussian_model(lib_built):
    """GPB_GetCoef(calc_std_dev = true) after fit(y, X) of a probit model on the device (CalcStdDevCoefNonGaussian: numerical Jacobian of X' grad_F;
    host half tested on the CPU, tests/test_laplace_coef.py) against the reference's values (25 %: "(very) approximate" in the reference's own words)."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_coef_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y, X = cases.laplace_coef_data("bernoulli_probit", 3)
    mdl = gpb.GPModel(likelihood="bernoulli_probit", gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.fit(y, X=X, params={"init_coef_aux_pars_from_iid_model": False})
    out = mdl.get_coef(std_err=True)
    np.testing.assert_allclose(out[:3], g["bernoulli_probit_p3_coef"], rtol=0.02, atol=2e-3)
    np.testing.assert_allclose(out[3:], g["bernoulli_probit_p3_coef_sd"], rtol=0.25)


def test_fit_with_covariates_from_the_iid_model_coefficients(lib_built):
    """The packages' default (init_coef_aux_pars_from_iid_model = true): GPModel.fit(y, X) of a logit model starts from the GLM coefficients and ends at
    the reference's estimates (tests/golden/laplace_coef_ref.npz, iid_* entries; those were fitted with tight tolerances, this fit with the defaults:
    estimates within the gradient noise, likelihood 1e-5)."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_coef_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y, X = cases.laplace_coef_data("bernoulli_logit", 3)
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.fit(y, X=X)
    assert abs(mdl.get_num_optim_iter() - int(g["iid_bernoulli_logit_num_it"])) <= 2
    np.testing.assert_allclose(mdl.get_cov_pars(), g["iid_bernoulli_logit_cov_pars"], rtol=0.06)
    np.testing.assert_allclose(mdl.get_coef(), g["iid_bernoulli_logit_coef"], rtol=0.03, atol=3e-3)


@pytest.mark.parametrize("lik", ["bernoulli_logit", "poisson"])
def test_cond_all_prediction_of_non_gaussian_models(lib_built, lik):
    """'latent_order_obs_first_cond_all' through GPB_PredictREModel: device factor rows of the latent process for the appended points, host forward
    substitution with Bp and the rows of Bp^-1 Bpo, device quadratic forms (gpb_hip_vecchia_laplace_quad_forms) -- against the reference's exact values
    (tests/golden/laplace_predvar_ref.npz, cond_all_* entries): covariance matrix, variances, response predictions."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_predvar_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    cpc = g["coords_pred_cond_all"]
    pc = mdl.predict(y=y, gp_coords_pred=cpc, cov_pars=cp, predict_cov_mat=True, predict_response=False, vecchia_pred_type="latent_order_obs_first_cond_all",
                     num_neighbors_pred=40)
    np.testing.assert_allclose(pc["mu"], g["cond_all_%s_latent_mu" % lik], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pc["cov"], g["cond_all_%s_latent_cov" % lik], rtol=1e-5, atol=1e-8)
    pv = mdl.predict(y=y, gp_coords_pred=cpc, cov_pars=cp, predict_var=True, predict_response=False)
    np.testing.assert_allclose(pv["var"], np.diag(g["cond_all_%s_latent_cov" % lik]), rtol=1e-5)
    pr = mdl.predict(y=y, gp_coords_pred=cpc, cov_pars=cp, predict_var=True, predict_response=True)
    np.testing.assert_allclose(pr["mu"], g["cond_all_%s_resp_mu" % lik], rtol=1e-5)
    np.testing.assert_allclose(pr["var"], g["cond_all_%s_resp_var" % lik], rtol=1e-5)


def test_r_suite_joint_covariance_golden_of_the_logit_model(orc, lib_built):
    """test_GPModel_non_Gaussian_data.R:2527-2531 on the device path: the joint latent predictive covariance incl. its off-diagonal entries, with
    'latent_order_obs_first_cond_all' on all predecessors (oracle side: tests/test_laplace_predvar.py)."""
    import gpboost_amd as gpb
    coords, y = orc.r_fixture_logit()
    n = len(y)
    ct = np.array([[0.1, 0.9], [0.11, 0.91], [0.7, 0.55]])
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=n - 1,
                      vecchia_ordering="none")
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    pc = mdl.predict(y=y, gp_coords_pred=ct, cov_pars=np.array([1.4300136, 0.1891952]), predict_cov_mat=True, predict_response=False,
                     vecchia_pred_type="latent_order_obs_first_cond_all", num_neighbors_pred=n + 2)
    assert np.abs(pc["mu"] - [-0.7792960, -0.7876208, 0.5476390]).sum() < 1e-6
    exp_cov = [1.024266883e+00, 9.215203622e-01, 5.561463409e-05, 9.215203622e-01, 1.022897212e+00, 2.028646043e-05, 5.561463409e-05, 2.028646043e-05,
               7.395745025e-01]
    assert np.abs(pc["cov"].ravel() - exp_cov).sum() < 5e-6


@pytest.mark.parametrize("tag,est,n_cov", [("fix_range", [1, 0], 0), ("fix_var", [0, 1], 0), ("fix_var_p2", [0, 1], 2)])
def test_fits_with_covariance_parameters_held_fixed(lib_built, tag, est, n_cov):
    """estimate_cov_par_index for non-Gaussian models: parameters marked 0 have no gradient entry (likelihoods.h:6622), lbfgs leaves them at their initial
    values; standard errors: NaN for them, the Hessian of the others alone is inverted (re_model_template.h:11052-11079).  Against the reference's fits
    (tests/golden/laplace_coef_ref.npz, est_* entries; both sides with tight solver tolerances)."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_coef_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y, X = cases.laplace_coef_data("bernoulli_logit", max(n_cov, 1))
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    params = {"init_cov_pars": [0.5, 0.2], "cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13, "estimate_cov_par_index": est,
              "init_coef_aux_pars_from_iid_model": False}
    mdl.fit(y, X=X if n_cov else None, params=params)
    key = "est_" + tag
    assert mdl.get_num_optim_iter() == int(g[key + "_num_it"])
    np.testing.assert_allclose(mdl.get_cov_pars(), g[key + "_cov_pars"], rtol=1e-4)
    assert abs(mdl.get_current_neg_log_likelihood() - float(g[key + "_negll"])) <= 1e-7 * abs(float(g[key + "_negll"]))
    if n_cov:
        np.testing.assert_allclose(mdl.get_coef(), g[key + "_coef"], rtol=1e-4)
    se = mdl.get_cov_pars(std_err=True)[2:]
    ref = g[key + "_std"]
    assert np.array_equal(np.isnan(se), np.isnan(ref))
    np.testing.assert_allclose(se[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=0.05)


@pytest.mark.parametrize("lik,iid", [("poisson", False), ("poisson", True), ("bernoulli_logit", False), ("bernoulli_logit", True)])
def test_fit_with_covariates_and_sample_weights(lib_built, lik, iid):
    """Round 6: sample weights TOGETHER with covariates for a non-Gaussian model -- the weighted start of the intercept (FindInitialIntercept, likelihoods.h:1455-1560), the
    weighted constants of the step cap (:2618-2660), the iid model created with the weights (re_model.cpp:401-409), weights in every per-datum term on the device -- against the
    unmodified reference at cases.LAPLACE_TIGHT (tests/golden/laplace_coef_weights_ref.npz): the reference's iteration count, likelihood 1e-7, estimates 1e-3 (the reference's own
    fits move by 2e-5 / 5e-9 between two runs: its parallel sums are not reproducible).  The logit fits are stopped after 8 iterations (their optimum is degenerate on these data)."""
    import gpboost_amd as gpb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace_coef_weights_ref.npz"))
    c = cases.LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    coords, y, X = cases.laplace_coef_data(lik, 3)
    w = cases.laplace_coef_weights(c["n"])
    key = "%s_%s" % (lik, "iid" if iid else "noiid")
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"], weights=w)
    params = dict(cases.LAPLACE_TIGHT, init_coef_aux_pars_from_iid_model=iid)
    if lik == "bernoulli_logit":
        params["maxit"] = 8
    mdl.fit(y, X=X, params=params)
    assert mdl.get_num_optim_iter() == int(g[key + "_num_it"]), (mdl.get_num_optim_iter(), int(g[key + "_num_it"]))
    nll = mdl.get_current_neg_log_likelihood()
    # (the logit fits are cut off half way down a steep valley towards their degenerate optimum -- variance 20 -> 5e2 --: eight steps amplify last-digit differences of the
    #  gradients, seen 6e-7 on the value between the reference and the C restatement of the same algorithm; the converged Poisson fits are held tight)
    ntol, etol = (1e-5, 2e-2) if lik == "bernoulli_logit" else (1e-7, 1e-3)
    assert abs(nll - float(g[key + "_negll"])) <= ntol * abs(nll), (nll, float(g[key + "_negll"]))
    np.testing.assert_allclose(mdl.get_cov_pars(), g[key + "_cov_pars"], rtol=etol)
    np.testing.assert_allclose(mdl.get_coef(), g[key + "_coef"], rtol=etol, atol=1e-5)
