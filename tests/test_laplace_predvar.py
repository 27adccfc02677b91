"""Predictive variances and response-scale predictions of the non-Gaussian (Vecchia-Laplace) models at new locations
(PredictLaplaceApproxVecchia, include/GPBoost/likelihoods.h:8563-8824; PredictResponse, :9626-9672; 'latent_order_obs_first_cond_obs_only').

The latent variance is Dp + b_p' (Sigma^-1 + W)^-1 b_p.  The reference computes it exactly with matrix_inversion_method = "cholesky" (:8783-8821)
and estimates it with nsim_var_pred random vectors with "iterative" (:8637-8745; platform-dependent generators, ~10 % scatter: stored in the
fixture for information only).  Fixture: tests/golden/laplace_predvar_ref.npz (oracle/make_golden.py laplace_predvar, the unmodified reference).

CPU: the oracle (orc.vecchia_laplace_predict, orc.predict_response) against the reference; the host half of the product (Gauss-Hermite rule,
PredictResponse) against the oracle.   GPU: GPB_PredictREModel on the device path against the same fixture."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "laplace_predvar_ref.npz")
LIKS = ("bernoulli_logit", "bernoulli_probit", "poisson")
CASE = "lap_u2d_n1500_mat15_m30"
DUP = "dup_mat15_m20_random"


def _range_const(ct):
    return {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct]


@pytest.mark.parametrize("lik", LIKS)
def test_oracle_reproduces_the_reference(orc, lik):
    g = np.load(GOLD)
    c = cases.LAPLACE_CASES[CASE]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    mu, var = orc.vecchia_laplace_predict(co, nn, ct, cp[0], _range_const(ct) / cp[1], y[perm], g["coords_pred"], 2 * c["m"], likelihood=lik,
                                          cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    np.testing.assert_allclose(mu, g[lik + "_cholesky_latent_mu"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(var, g[lik + "_cholesky_latent_var"], rtol=1e-7)
    rm, rv = orc.predict_response(lik, mu, var, True)
    np.testing.assert_allclose(rm, g[lik + "_cholesky_resp_mu"], rtol=1e-7)
    np.testing.assert_allclose(rv, g[lik + "_cholesky_resp_var"], rtol=1e-7)
    # the reference's own stochastic estimate scatters around the exact value
    assert np.abs(g[lik + "_iterative_latent_var"] / g[lik + "_cholesky_latent_var"] - 1).max() < 0.25


@pytest.mark.parametrize("lik", ("bernoulli_logit", "poisson"))
def test_oracle_reproduces_the_reference_with_repeated_locations(orc, lik):
    g = np.load(GOLD)
    cf, sh, m, ordering, seed = cases.LAPLACE_DUP_CASES[DUP]
    coords, y, fe, cpd = cases.laplace_dup_data(lik)
    cpd2 = np.vstack([cpd, cpd[:5]])
    n = coords.shape[0]
    perm = orc.shuffle(n, seed)
    cs, ys = coords[perm], y[perm]
    uniq, uidx = orc.unique_locations(cs)
    cu = cs[uniq]
    nn = orc.neighbors(cu, m)
    ct = orc.cov_type_id(cf, sh)
    cp = cases.LAPLACE_DUP_COV_PARS[0]
    mu, var = orc.vecchia_laplace_predict(cu, nn, ct, cp[0], _range_const(ct) / cp[1], ys, cpd2, 2 * m, likelihood=lik, unique_idx=uidx,
                                          cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    np.testing.assert_allclose(mu, g["dup_%s_latent_mu" % lik], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(var, g["dup_%s_latent_var" % lik], rtol=1e-7)
    rm, rv = orc.predict_response(lik, mu, var, True)
    np.testing.assert_allclose(rm, g["dup_%s_resp_mu" % lik], rtol=1e-7)
    np.testing.assert_allclose(rv, g["dup_%s_resp_var" % lik], rtol=1e-7)


def test_oracle_reproduces_the_r_suite_prediction_goldens(orc):
    """R-package/tests/testthat/test_GPModel_non_Gaussian_data.R:2510-2537 (exact GP, Bernoulli logit, Laplace with the Cholesky factor): at the fitted
    parameters (1.4300136, 0.1891952) the latent predictive mean (-0.7792960, -0.7876208, 0.5476390), the diagonal of the latent predictive covariance
    (1.024266883, 1.022897212, 0.7395745025) and the response mean (0.3442815, 0.3426873, 0.6159933) at three new locations, two of them 0.014 apart.
    A Vecchia approximation that conditions on ALL predecessors / all observed points is the exact GP, so the oracle's prediction (mode by its
    iterative finder, variances Dp + Bpo (Sigma^-1 + W)^-1 Bpo') must reproduce them; the off-diagonal covariances of the golden belong to the exact
    GP's JOINT prediction and are not part of 'latent_order_obs_first_cond_obs_only'.  Tolerances: the R test's (1e-6 / 1e-3 summed; the printed
    parameters carry 7 digits, seen 8e-7 on the variances)."""
    coords, y = orc.r_fixture_logit()
    n = len(y)
    nn = orc.neighbors(coords, n - 1)
    ct = np.array([[0.1, 0.9], [0.11, 0.91], [0.7, 0.55]])
    mu, var = orc.vecchia_laplace_predict(coords, nn, 0, 1.4300136, 1.0 / 0.1891952, y, ct, n, likelihood="bernoulli_logit", cg_delta_conv=1e-8,
                                          delta_conv_mode=1e-13)
    assert np.abs(mu - [-0.7792960, -0.7876208, 0.5476390]).sum() < 1e-6
    assert np.abs(var - [1.024266883, 1.022897212, 0.7395745025]).sum() < 2e-6
    rm, rv = orc.predict_response("bernoulli_logit", mu, var, True)
    assert np.abs(rm - [0.3442815, 0.3426873, 0.6159933]).sum() < 1e-6
    assert np.abs(rv - rm * (1 - rm)).sum() < 1e-15


R_PROBIT_CT = np.array([[0.1, 0.9], [0.11, 0.91], [0.7, 0.55]])


def r_probit_design():
    """X / X_test / fitted coefficients of test_GPModel_non_Gaussian_data.R:84, 1393, 1398 (n = 100)."""
    n = 100
    i = np.arange(1, n + 1)
    X = np.c_[np.ones(n), np.sin((i - n / 2) ** 2 * 2 * np.pi / n)]
    return X, np.c_[np.ones(3), [-0.5, 0.2, 1.0]], np.array([0.3983333, -0.2653886])


def test_oracle_reproduces_the_r_suite_probit_prediction_goldens(orc):
    """test_GPModel_non_Gaussian_data.R:1391-1432 (exact GP, Bernoulli probit) at cov_pars (1, 0.2), as a Vecchia model on all predecessors:
    without a linear predictor the latent mean (0.01874013, 0.01200800, 0.20498871) / variances (0.6105248, 0.6093745, 0.4235374); with the fitted
    linear predictor X beta as fixed effects the latent mean (0.3389905, 0.1512445, -0.1039307), the diagonal of the covariance (0.6193228722,
    0.6159348965, 0.4291674143), the response mean (0.6050312, 0.5473537, 0.4653610) and variance (0.2389684, 0.2477576, 0.2488001).  (The golden
    means come from the reference's Newton iteration with its default stopping rule: they are defined to ~1e-6.)"""
    coords, y = orc.r_fixture_probit()
    n = len(y)
    nn = orc.neighbors(coords, n - 1)
    kw = dict(likelihood="bernoulli_probit", cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    mu, var = orc.vecchia_laplace_predict(coords, nn, 0, 1.0, 1.0 / 0.2, y, R_PROBIT_CT, n, **kw)
    assert np.abs(mu - [0.01874013, 0.01200800, 0.20498871]).sum() < 1e-5
    assert np.abs(var - [0.6105248, 0.6093745, 0.4235374]).sum() < 1e-6
    X, Xt, beta = r_probit_design()
    mu, var = orc.vecchia_laplace_predict(coords, nn, 0, 1.0, 1.0 / 0.2, y, R_PROBIT_CT, n, fixed_effects=X @ beta, **kw)
    mu = mu + Xt @ beta
    assert np.abs(mu - [0.3389905, 0.1512445, -0.1039307]).sum() < 1e-6
    assert np.abs(var - [0.6193228722, 0.6159348965, 0.4291674143]).sum() < 1e-6
    rm, rv = orc.predict_response("bernoulli_probit", mu, var, True)
    assert np.abs(rm - [0.6050312, 0.5473537, 0.4653610]).sum() < 1e-6
    assert np.abs(rv - [0.2389684, 0.2477576, 0.2488001]).sum() < 1e-6


GOLD_TR = os.path.join(os.path.dirname(__file__), "golden", "laplace_train_re_ref.npz")


@pytest.mark.parametrize("lik", LIKS)
def test_oracle_training_data_random_effects_match_the_reference(orc, lik):
    """Mode and diag((Sigma^-1 + W)^-1) at the training locations (GPB_PredictREModelTrainingDataRandomEffects, re_model_template.h:4683-4725;
    tests/golden/laplace_train_re_ref.npz from the reference with matrix_inversion_method = "cholesky")."""
    g = np.load(GOLD_TR)
    c = cases.LAPLACE_CASES[CASE]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    mode, v = orc.vecchia_laplace_train_re(co, nn, ct, cp[0], _range_const(ct) / cp[1], y[perm], likelihood=lik, cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    mu = np.empty_like(mode); mu[perm] = mode
    var = np.empty_like(v); var[perm] = v
    np.testing.assert_allclose(mu, g[lik + "_mu"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(var, g[lik + "_var"], rtol=1e-7)


def test_oracle_training_data_random_effects_with_repeated_locations(orc):
    g = np.load(GOLD_TR)
    cf, sh, m, ordering, seed = cases.LAPLACE_DUP_CASES[DUP]
    coords, y, fe, _ = cases.laplace_dup_data("bernoulli_logit")
    n = coords.shape[0]
    perm = orc.shuffle(n, seed)
    cs, ys = coords[perm], y[perm]
    uniq, uidx = orc.unique_locations(cs)
    cu = cs[uniq]
    ct = orc.cov_type_id(cf, sh)
    cp = cases.LAPLACE_DUP_COV_PARS[0]
    mode, v = orc.vecchia_laplace_train_re(cu, orc.neighbors(cu, m), ct, cp[0], _range_const(ct) / cp[1], ys, likelihood="bernoulli_logit",
                                           unique_idx=uidx, cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    mu = np.empty(n); mu[perm] = mode[uidx]
    var = np.empty(n); var[perm] = v[uidx]
    np.testing.assert_allclose(mu, g["dup_bernoulli_logit_mu"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(var, g["dup_bernoulli_logit_var"], rtol=1e-7)


def test_host_gauss_hermite_rule(orc, lib_built):
    """GPB_HIP_GaussHermiteHost (computed: Sturm bisection + Newton on the orthonormal recurrence) against numpy's rule and against the first /
    middle entries of the table the reference ships (GH_nodes_ / adaptive_GH_weights_, likelihoods.h:17486, :17546)."""
    lib = C.CDLL(lib_built)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for order in (1, 2, 5, 30, 64):
        x = np.empty(order); w = np.empty(order)
        assert lib.GPB_HIP_GaussHermiteHost(order, P(x), P(w)) == 0
        xo, wo = orc.gauss_hermite_adaptive(order)
        np.testing.assert_allclose(x, xo, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(w, wo, rtol=1e-12)
    x = np.empty(30); w = np.empty(30)
    lib.GPB_HIP_GaussHermiteHost(30, P(x), P(w))
    assert abs(x[0] + 6.863345293529891581061) < 1e-13 and abs(w[0] - 0.83424747101276179534) < 1e-13
    assert np.all(x[:15] == -x[:14:-1]) and np.all(w[:15] == w[:14:-1])
    assert lib.GPB_HIP_GaussHermiteHost(0, P(x), P(w)) == -1


@pytest.mark.parametrize("lik", LIKS)
def test_host_predict_response_matches_the_oracle(orc, lib_built, lik):
    lib = C.CDLL(lib_built)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(5)
    m = 2.5 * rng.normal(size=400)
    v = np.concatenate([rng.uniform(1e-6, 0.1, size=100), rng.uniform(0.1, 6.0, size=300)])
    for predict_var in (False, True):
        mm, vv = m.copy(), v.copy()
        assert lib.GPB_HIP_PredictResponseHost(lik.encode(), m.size, P(mm), P(vv), C.c_bool(predict_var), C.c_double(1e-8)) == 0
        om, ov = orc.predict_response(lik, m, v, predict_var)
        np.testing.assert_allclose(mm, om, rtol=1e-12)
        if predict_var:
            np.testing.assert_allclose(vv, ov, rtol=1e-12)
        else:
            assert np.array_equal(vv, v)
    lib.LGBM_GetLastError.restype = C.c_char_p
    assert lib.GPB_HIP_PredictResponseHost(b"tweedie", 1, P(m.copy()), P(v.copy()), C.c_bool(False), C.c_double(1e-8)) == -1       # ("t": on the path since round 5)
    assert b"'tweedie'" in lib.LGBM_GetLastError()


@pytest.mark.gpu
@pytest.mark.parametrize("lik", LIKS)
def test_device_path_reproduces_the_reference(lib_built, lik):
    """Latent variances by block CG on the device (gpb_hip_vecchia_laplace_predict) and the response predictions from them, against the reference's
    exact ("cholesky") values.  Tolerance: the device finds the mode with the iterative solver at cg_delta_conv = 1e-8 /
    delta_conv_mode_finding = 1e-13 (as tests/test_laplace_gpu.py's prediction test), the variance solves stop at a residual norm of 1e-8."""
    import gpboost_amd as gpb
    g = np.load(GOLD)
    c = cases.LAPLACE_CASES[CASE]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    pr = mdl.predict(y=y, gp_coords_pred=g["coords_pred"], cov_pars=cp, predict_var=True, predict_response=False)
    np.testing.assert_allclose(pr["mu"], g[lik + "_cholesky_latent_mu"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pr["var"], g[lik + "_cholesky_latent_var"], rtol=1e-5)
    pr = mdl.predict(y=y, gp_coords_pred=g["coords_pred"], cov_pars=cp, predict_var=True, predict_response=True)
    np.testing.assert_allclose(pr["mu"], g[lik + "_cholesky_resp_mu"], rtol=1e-5)
    np.testing.assert_allclose(pr["var"], g[lik + "_cholesky_resp_var"], rtol=1e-5)
    pr = mdl.predict(y=y, gp_coords_pred=g["coords_pred"], cov_pars=cp, predict_var=False, predict_response=True)      # the default call of GPModel.predict
    np.testing.assert_allclose(pr["mu"], g[lik + "_cholesky_resp_mu"], rtol=1e-5)
    # covariance matrix of the latent process: its diagonal are the variances; symmetric; positive definite
    pc = mdl.predict(y=y, gp_coords_pred=g["coords_pred"][:20], cov_pars=cp, predict_cov_mat=True, predict_response=False)
    np.testing.assert_allclose(np.diag(pc["cov"]), g[lik + "_cholesky_latent_var"][:20], rtol=1e-5)
    assert np.array_equal(pc["cov"], pc["cov"].T) and np.linalg.eigvalsh(pc["cov"]).min() > 0
    with pytest.raises(gpb.GPBoostError, match="not supported when predicting the response"):
        mdl.predict(y=y, gp_coords_pred=g["coords_pred"], cov_pars=cp, predict_cov_mat=True, predict_response=True)


@pytest.mark.gpu
@pytest.mark.parametrize("lik", ("bernoulli_logit", "poisson"))
def test_device_path_with_repeated_locations(orc, lib_built, lik):
    """Repeated training locations (the GP on the unique ones) and repeated prediction locations (one random effect each,
    re_model_template.h:3976-3988): variances against the reference, the covariance matrix against the oracle's dense computation."""
    import gpboost_amd as gpb
    g = np.load(GOLD)
    cf, sh, m, ordering, seed = cases.LAPLACE_DUP_CASES[DUP]
    coords, y, fe, cpd = cases.laplace_dup_data(lik)
    cpd2 = np.vstack([cpd, cpd[:5]])
    cp = np.asarray(cases.LAPLACE_DUP_COV_PARS[0], dtype=np.float64)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering=ordering, seed=seed)
    mdl.set_optim_params({"cg_delta_conv": 1e-8, "delta_conv_mode_finding": 1e-13})
    pr = mdl.predict(y=y, gp_coords_pred=cpd2, cov_pars=cp, predict_var=True, predict_response=False)
    np.testing.assert_allclose(pr["mu"], g["dup_%s_latent_mu" % lik], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pr["var"], g["dup_%s_latent_var" % lik], rtol=1e-5)
    pr = mdl.predict(y=y, gp_coords_pred=cpd2, cov_pars=cp, predict_var=True, predict_response=True)
    np.testing.assert_allclose(pr["mu"], g["dup_%s_resp_mu" % lik], rtol=1e-5)
    np.testing.assert_allclose(pr["var"], g["dup_%s_resp_var" % lik], rtol=1e-5)
    # covariance matrix: oracle (dense) on the unique prediction locations, expanded with the incidence of the repeats
    n = coords.shape[0]
    perm = orc.shuffle(n, seed)
    cs, ys = coords[perm], y[perm]
    uniq, uidx = orc.unique_locations(cs)
    cu = cs[uniq]
    ct = orc.cov_type_id(cf, sh)
    _, _, cov_u = orc.vecchia_laplace_predict(cu, orc.neighbors(cu, m), ct, cp[0], _range_const(ct) / cp[1], ys, cpd, 2 * m, likelihood=lik,
                                              unique_idx=uidx, want_cov=True, cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    idx = np.concatenate([np.arange(len(cpd)), np.arange(5)])
    pc = mdl.predict(y=y, gp_coords_pred=cpd2, cov_pars=cp, predict_cov_mat=True, predict_response=False)
    np.testing.assert_allclose(pc["cov"], cov_u[np.ix_(idx, idx)], rtol=1e-5, atol=1e-8)


# ---- standard errors of the covariance parameters of non-Gaussian models ---------------------------------------------------------------------
GOLD_SE = os.path.join(os.path.dirname(__file__), "golden", "laplace_stderr_ref.npz")


@pytest.mark.parametrize("lik", LIKS)
def test_host_standard_errors_of_non_gaussian_models(orc, lib_built, lik):
    """GPB_HIP_LaplaceStdErrorsWithCallback = CalcStdDevCovParAuxParsNonGaussian (re_model_template.h:11029-11117; the host half of
    GPB_GetCovPar(calc_std_dev = true) for non-Gaussian models) with the ORACLE as the evaluator, at the reference's fitted parameters:
    (i) against a restatement of the same sequence in numpy (same evaluations, same warm starts): 1e-8;
    (ii) against the unmodified reference's own standard errors after its own fit (tests/golden/laplace_stderr_ref.npz): 5 % -- the quantity is a
    second difference of a gradient that carries the noise of an iterative mode finder (1e-8 relative change of its objective) and of CG solves
    stopped at |r| < 1e-2, over a step of 1e-4: the reference's own value moves by ~1 % with the mode it is warm-started from."""
    from tests.optim_harness import LAPLACE_FN, OracleLaplaceEvaluator
    g = np.load(GOLD_SE)
    c = cases.LAPLACE_CASES[CASE]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    rc_ = _range_const(ct)
    cp = g[lik + "_cov_pars"]
    th = np.array([cp[0], rc_ / cp[1]])
    ev = OracleLaplaceEvaluator(orc, co, nn, ct, y[perm], lik)
    o3 = (C.c_double * 3)()
    assert ev._fn(None, 0, th[0], th[1], o3) == 0                      # the state a fit leaves behind: the mode at theta
    mode0 = ev.mode.copy()
    lib = C.CDLL(lib_built)
    lib.GPB_HIP_LaplaceStdErrorsWithCallback.argtypes = [C.c_void_p, C.c_double, LAPLACE_FN, C.c_void_p, C.c_void_p]
    se = np.empty(2)
    assert lib.GPB_HIP_LaplaceStdErrorsWithCallback(th.ctypes.data, C.c_double(rc_), ev.cb, None, se.ctypes.data) == 0
    assert [op for op, *_ in ev.calls] == [0, 1, 1, 1, 1, 0]
    # (i) the same sequence in numpy
    mode = mode0
    lp = np.log(th); delta = np.maximum(np.abs(lp) * 1e-4, 1e-4)
    H = np.zeros((2, 2))
    for i in range(2):
        gr = []
        for sgn in (1.0, -1.0):
            t = th.copy(); t[i] *= np.exp(sgn * delta[i])
            _, g2, mode = orc.vecchia_laplace_grad(co, nn, ct, t[0], t[1], y[perm], likelihood=lik, mode_init=mode, want_mode=True)
            gr.append(np.asarray(g2))
        H[i] = (gr[0] - gr[1]) / (2 * delta[i])
    H = 0.5 * (H + H.T)
    se_np = np.array([cp[0], cp[1]]) * np.sqrt(np.diag(np.linalg.inv(H)))
    np.testing.assert_allclose(se, se_np, rtol=1e-8)
    # (ii) the reference
    np.testing.assert_allclose(se, g[lik + "_std"], rtol=0.05)


def test_host_standard_errors_report_an_indefinite_hessian(lib_built):
    """A gradient that does not grow with the parameter has no positive definite Jacobian: NaN for both entries (the reference warns and returns NaN)."""
    from tests.optim_harness import LAPLACE_FN

    def fn(ctx, op, var, a, out3):
        out3[0] = 0.0; out3[1] = -np.log(var); out3[2] = np.log(a)
        return 0
    cb = LAPLACE_FN(fn)
    lib = C.CDLL(lib_built)
    lib.GPB_HIP_LaplaceStdErrorsWithCallback.argtypes = [C.c_void_p, C.c_double, LAPLACE_FN, C.c_void_p, C.c_void_p]
    th = np.array([1.3, 4.0]); se = np.zeros(2)
    assert lib.GPB_HIP_LaplaceStdErrorsWithCallback(th.ctypes.data, C.c_double(1.0), cb, None, se.ctypes.data) == 0
    assert np.all(np.isnan(se))


# ---- 'latent_order_obs_first_cond_all': prediction points that condition on each other ---------------------------------------------------------
@pytest.mark.parametrize("lik", ("bernoulli_logit", "poisson"))
def test_oracle_cond_all_prediction_matches_the_reference(orc, lik):
    """PredictLaplaceApproxVecchia with CondObsOnly = false (likelihoods.h:8603-8606, 8790-8821): mean = -Bp^-1 Bpo mode, covariance
    Bp^-1 Dp Bp^-T + (Bp^-1 Bpo) (Sigma^-1 + W)^-1 (Bp^-1 Bpo)' -- against the unmodified reference ("cholesky"), 30 prediction points of which 15
    sit within 0.05 of each other (tests/golden/laplace_predvar_ref.npz, cond_all_* entries), full covariance matrix and response predictions."""
    g = np.load(GOLD)
    c = cases.LAPLACE_CASES[CASE]
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    mu, var, cov = orc.vecchia_laplace_predict(co, nn, ct, cp[0], _range_const(ct) / cp[1], y[perm], g["coords_pred_cond_all"], 40, likelihood=lik,
                                               cond_obs_only=False, want_cov=True, cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    np.testing.assert_allclose(mu, g["cond_all_%s_latent_mu" % lik], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(cov, g["cond_all_%s_latent_cov" % lik], rtol=1e-6, atol=1e-9)
    rm, rv = orc.predict_response(lik, mu, var, True)
    np.testing.assert_allclose(rm, g["cond_all_%s_resp_mu" % lik], rtol=1e-6)
    np.testing.assert_allclose(rv, g["cond_all_%s_resp_var" % lik], rtol=1e-6)


def test_oracle_reproduces_the_r_suite_joint_covariance_golden(orc):
    """test_GPModel_non_Gaussian_data.R:2527-2531: the exact GP's JOINT latent predictive covariance of the logit model, off-diagonal entries included
    (0.9215203622 between the two points 0.014 apart) = 'latent_order_obs_first_cond_all' on all predecessors (num_neighbors_pred = n + 2, as the
    R suite's own Vecchia tests use it, :1491-1497)."""
    coords, y = orc.r_fixture_logit()
    n = len(y)
    nn = orc.neighbors(coords, n - 1)
    ct = np.array([[0.1, 0.9], [0.11, 0.91], [0.7, 0.55]])
    mu, var, cov = orc.vecchia_laplace_predict(coords, nn, 0, 1.4300136, 1.0 / 0.1891952, y, ct, n + 2, likelihood="bernoulli_logit", cond_obs_only=False,
                                               want_cov=True, cg_delta_conv=1e-8, delta_conv_mode=1e-13)
    assert np.abs(mu - [-0.7792960, -0.7876208, 0.5476390]).sum() < 1e-6
    exp_cov = [1.024266883e+00, 9.215203622e-01, 5.561463409e-05, 9.215203622e-01, 1.022897212e+00, 2.028646043e-05, 5.561463409e-05, 2.028646043e-05,
               7.395745025e-01]
    assert np.abs(cov.ravel() - exp_cov).sum() < 5e-6
