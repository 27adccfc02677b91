"""Gaussian Vecchia prediction types 'order_pred_first', 'latent_order_obs_first_cond_obs_only' and 'latent_order_obs_first_cond_all'
(CalcPredVecchiaPredictedFirstOrder / CalcPredVecchiaLatentObservedFirstOrder, src/GPBoost/Vecchia_utils.cpp:2203-2666).

CPU: the oracle's dense restatements against the unmodified reference's outputs (tests/golden/predtypes_ref.npz, oracle/make_golden.py
predtypes) and the R suite's goldens (R-package/tests/testthat/test_GPModel_gaussian_process.R:1497-1552).
GPU: the library's path (neighbour search + factor of ALL points of the joint ordering on the device, the conditional precision assembled on
the host and inverted by the dense MFMA Cholesky on the device) through GPB_SetPredictionData / GPB_PredictREModel against the same fixtures."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "predtypes_ref.npz")


def _tol(pt):
    """'order_pred_first': factor rows with the nugget on every diagonal -- 1e-8 like the likelihood.  'latent_*': the factor rows are those of the
    LATENT process (no nugget, diagonal x (1 + 1e-10), Vecchia_utils.cpp:2589): the neighbour systems of close points have condition numbers up
    to ~1e10, so their solutions -- in the reference as much as here -- are only defined to ~1e-6 relative; two correct evaluation orders differ by
    that much (measured: the oracle's dense restatement against the reference 8e-7)."""
    return dict(rtol=1e-8, atol=1e-10) if pt == "order_pred_first" else dict(rtol=1e-5, atol=1e-7)

R_GOLDENS = {   # test_GPModel_gaussian_process.R:1497-1552: cov_pars (0.02, 1.2, 0.9), 30 neighbours, predict_response = TRUE
    "order_pred_first": ([0.08498682, 0.08502034, 0.49572748],
                         [1.189037e-01, 9.888624e-02, -1.080005e-05, 9.888624e-02, 1.189065e-01, -1.079431e-05, -1.080005e-05, -1.079431e-05, 8.101757e-02]),
    "latent_order_obs_first_cond_obs_only": ([0.08616985, 0.08616384, 0.48721314],
                                             [1.189100e-01, 7.324225e-03, -5.851427e-07, 7.324225e-03, 1.189129e-01, -5.850749e-07, -5.851427e-07,
                                              -5.850750e-07, 8.107749e-02]),
    "latent_order_obs_first_cond_all": ([0.08616985, 0.08616377, 0.48721314],
                                        [1.189100e-01, 9.889258e-02, -5.851418e-07, 9.889258e-02, 1.189129e-01, -5.850764e-07, -5.851418e-07,
                                         -5.850764e-07, 8.107749e-02]),
}


def _oracle_predict(orc, name, pt, predict_response):
    n, d, cf, sh, m, ordering, seed, npred, mpred, cp = cases.PREDTYPE_CASES[name]
    coords, y, cpred = cases.predtype_data(name)
    perm = orc.shuffle(n, seed) if ordering == "random" else np.arange(n)
    ct = orc.cov_type_id(cf, sh)
    ptr = orc.transform_cov_pars(ct, np.asarray(cp, dtype=np.float64))
    if pt == "order_pred_first":
        return orc.predict_pred_first(coords[perm], y[perm], cpred, ct, ptr, mpred, predict_response)
    return orc.predict_latent(coords[perm], y[perm], cpred, ct, ptr, mpred, pt.endswith("cond_obs_only"), predict_response)


@pytest.mark.parametrize("name", list(cases.PREDTYPE_CASES))
def test_oracle_reproduces_the_reference(orc, name):
    g = np.load(GOLD)
    for pt in cases.PRED_TYPES:
        mu, cov = _oracle_predict(orc, name, pt, True)
        np.testing.assert_allclose(mu, g["%s_%s_mu" % (name, pt)], **_tol(pt))
        np.testing.assert_allclose(cov, g["%s_%s_cov" % (name, pt)], **_tol(pt))
        _, covl = _oracle_predict(orc, name, pt, False)
        np.testing.assert_allclose(np.diag(covl), g["%s_%s_latent_var" % (name, pt)], **_tol(pt))
        if name.startswith("pt_r100"):
            assert np.abs(mu - R_GOLDENS[pt][0]).sum() < 1e-6
            assert np.abs(cov.ravel() - R_GOLDENS[pt][1]).sum() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.PREDTYPE_CASES))
def test_device_path_reproduces_the_reference(name, lib_built):
    import gpboost_amd
    g = np.load(GOLD)
    n, d, cf, sh, m, ordering, seed, npred, mpred, cp = cases.PREDTYPE_CASES[name]
    coords, y, cpred = cases.predtype_data(name)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m, vecchia_ordering=ordering, seed=seed)
    for pt in cases.PRED_TYPES:
        mdl.set_prediction_data(vecchia_pred_type=pt, num_neighbors_pred=mpred)
        pr = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=np.asarray(cp), predict_cov_mat=True, predict_response=True)
        np.testing.assert_allclose(pr["mu"], g["%s_%s_mu" % (name, pt)], **_tol(pt))
        np.testing.assert_allclose(pr["cov"], g["%s_%s_cov" % (name, pt)], **_tol(pt))
        pv = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=np.asarray(cp), predict_var=True, predict_response=False)
        np.testing.assert_allclose(pv["mu"], g["%s_%s_mu" % (name, pt)], **_tol(pt))
        np.testing.assert_allclose(pv["var"], g["%s_%s_latent_var" % (name, pt)], **_tol(pt))
        pm = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=np.asarray(cp), predict_var=False)       # mean only: the solve without the inverse
        np.testing.assert_allclose(pm["mu"], g["%s_%s_mu" % (name, pt)], **_tol(pt))
        if name.startswith("pt_r100"):
            assert np.abs(pr["mu"] - R_GOLDENS[pt][0]).sum() < 1e-6
            assert np.abs(pr["cov"].ravel() - R_GOLDENS[pt][1]).sum() < 1e-6


@pytest.mark.gpu
def test_limits_fail_loudly(lib_built):
    import gpboost_amd
    coords, y = cases.synthetic(300, 2, seed=5)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    mdl.set_prediction_data(vecchia_pred_type="latent_order_obs_first_cond_all", num_neighbors_pred=10)
    with pytest.raises(gpboost_amd.GPBoostError, match="Duplicates found among training and test coordinates"):
        mdl.predict(y=y, gp_coords_pred=coords[:5], cov_pars=np.array([0.1, 1.0, 0.2]), predict_var=True)


TRAIN_RE = os.path.join(os.path.dirname(__file__), "golden", "train_re_ref.npz")


@pytest.mark.parametrize("name", list(cases.PREDTYPE_CASES))
def test_training_data_random_effects_oracle_reproduces_the_reference(orc, name):
    """PredictTrainingDataRandomEffects with calc_var (re_model_template.h:4496-4514): mean = y - y_aux, var = sigma2 (1 - diag(B' D^-1 B))."""
    g = np.load(TRAIN_RE)
    n, d, cf, sh, m, ordering, seed, npred, mpred, cp = cases.PREDTYPE_CASES[name]
    coords, y, _ = cases.predtype_data(name)
    perm = orc.shuffle(n, seed) if ordering == "random" else np.arange(n)
    ct = orc.cov_type_id(cf, sh)
    pt = orc.transform_cov_pars(ct, np.asarray(cp, dtype=np.float64))
    nn = orc.neighbors(coords[perm], min(m, n - 1))
    A, D, bad = orc.vecchia_factor(coords[perm], nn, ct, pt[1], pt[2], gauss=True)
    mu, var = orc.train_random_effects(A, D, nn, y[perm], pt[0])
    out_mu = np.empty(n); out_var = np.empty(n); out_mu[perm] = mu; out_var[perm] = var
    np.testing.assert_allclose(out_mu, g[name + "_mu"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out_var, g[name + "_var"], rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.PREDTYPE_CASES))
def test_training_data_random_effects_on_device(name, lib_built):
    import gpboost_amd
    g = np.load(TRAIN_RE)
    n, d, cf, sh, m, ordering, seed, npred, mpred, cp = cases.PREDTYPE_CASES[name]
    coords, y, _ = cases.predtype_data(name)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m, vecchia_ordering=ordering, seed=seed)
    out = mdl.predict_training_data_random_effects(y=y, cov_pars=np.asarray(cp), predict_var=True)
    np.testing.assert_allclose(out[:, 0], g[name + "_mu"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out[:, 1], g[name + "_var"], rtol=1e-8, atol=1e-12)
    mu = mdl.predict_training_data_random_effects(y=y, cov_pars=np.asarray(cp))
    np.testing.assert_allclose(mu, g[name + "_mu"], rtol=1e-8, atol=1e-10)


def _perm_case():
    n, d = 1500, 2
    coords, _ = cases.synthetic(n, d, seed=5)
    y = np.sin(4 * coords[:, 0]) + 0.3 * np.random.default_rng(6).standard_normal(n)
    return coords, y, np.random.default_rng(7).uniform(size=(40, d)), np.array([0.1, 1.0, 0.1])


def test_reference_orders_pred_first_variances_by_its_cholesky_permutation(orc):
    """A divergence that is the reference's, recorded with its evidence: for 'order_pred_first' the reference computes the predictive
    covariance as (L^-1)' L^-1 with L = CholFact.CholFactMatrix() (Vecchia_utils.cpp:2424-2441) -- the factor of the conditional
    precision AFTER the sparse Cholesky's fill-reducing permutation -- and never undoes the permutation: it returns P cov P' (and the
    permuted variances) next to a mean in the caller's order.  With prediction points close together (the R suite's three points, the
    cases above) the permutation is the identity; on 40 scattered points it is not.  The oracle (dense, no permutation) reproduces the
    reference's mean to 1e-15 and its covariance exactly up to ONE permutation of rows and columns."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pred_first_perm_ref.npz"))
    coords, y, cpred, cp = _perm_case()
    mu, cov = orc.predict_pred_first(coords, y, cpred, 0, orc.transform_cov_pars(0, cp), 15, True)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-9, atol=1e-12)
    var_ref = np.diag(g["cov"])
    assert np.abs(var_ref - np.diag(cov)).max() > 1e-3                                   # not in the caller's order ...
    np.testing.assert_allclose(np.sort(var_ref), np.sort(np.diag(cov)), rtol=1e-9)       # ... but the same numbers
    P = np.array([int(np.argmin(np.abs(np.diag(cov) - v))) for v in var_ref])
    assert sorted(P.tolist()) == list(range(40))
    np.testing.assert_allclose(g["cov"], cov[np.ix_(P, P)], rtol=1e-8, atol=1e-12)        # one permutation explains the whole matrix


@pytest.mark.gpu
def test_device_pred_first_on_scattered_points_is_in_the_callers_order(orc, lib_built):
    import gpboost_amd
    coords, y, cpred, cp = _perm_case()
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=15, vecchia_ordering="none")
    mdl.set_prediction_data(vecchia_pred_type="order_pred_first", num_neighbors_pred=15)
    pr = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, predict_cov_mat=True, predict_response=True)
    mu, cov = orc.predict_pred_first(coords, y, cpred, 0, orc.transform_cov_pars(0, cp), 15, True)
    np.testing.assert_allclose(pr["mu"], mu, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(pr["cov"], cov, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_prediction_without_y_conditions_on_the_stored_response_not_on_a_leftover_residual(orc, lib_built):
    """An evaluation with fixed_effects leaves y - F on the device; a later prediction without y_data and without an offset conditions on y_vec_
    (SetYCalcCovCalcYAuxForPred, re_model_template.h:11141-11166), so the stored response goes up again (gpb_c_api.cpp: prediction_response)."""
    import gpboost_amd
    coords, y, cpred, cp = _perm_case()
    kw = dict(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=15, vecchia_ordering="none")
    fresh = gpboost_amd.GPModel(**kw)
    want = fresh.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, predict_var=True)
    mdl = gpboost_amd.GPModel(**kw)
    mdl.neg_log_likelihood(cp, y)                                                    # y_vec_ = y
    mdl.neg_log_likelihood(cp, y, fixed_effects=np.linspace(-1., 1., len(y)))        # the device now holds y - F
    got = mdl.predict(gp_coords_pred=cpred, cov_pars=cp, predict_var=True)
    np.testing.assert_allclose(got["mu"], want["mu"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got["var"], want["var"], rtol=1e-12, atol=1e-14)
