"""Full-scale Vecchia ("VIF": gp_approx = "full_scale_vecchia" / "vif") approximation, Gaussian likelihood -- SURVEY.md section 8 row f4.

Pins: tests/golden/vif_ref.npz = the UNMODIFIED reference's GPB_EvalNegLogLikelihood with gp_approx = "full_scale_vecchia" on
tests/cases.py:VIF_CASES (oracle/make_golden.py vif; include/GPBoost/re_model_template.h:8151-8200, 9646-9745, 9785-9806, 2950-2966;
src/GPBoost/Vecchia_utils.cpp:1463-1500; src/GPBoost/GP_utils.cpp:208-308 for the kmeans++ inducing points).
CPU: the oracle's restatement (oracle/orc.py: vif_setup / vif_terms) against those values.  GPU: the device path (host kmeans++ from the
model's generator, vif_kernels.hip: cross-covariances, whitening, residual-process factor; Woodbury matrix through the Gram kernel)
against the same values at north_star's 1e-8 -- including n = 1e5 with 200 inducing points -- and against the oracle's inducing points."""
import os

import numpy as np
import pytest

from tests import cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vif_ref.npz")
SMALL = [k for k, v in cases.VIF_CASES.items() if v[0] <= 3000]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_reproduces_the_reference(orc, name):
    g = np.load(GOLDEN)
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    setup = orc.vif_setup(coords, m, k, ordering, seed)
    for j, cp in enumerate(cps):
        v = orc.vif_nll(coords, y, np.asarray(cp), cf, sh, m, k, ordering, seed, setup=setup)
        ref = float(g["%s_negll_%d" % (name, j)])
        assert abs(v - ref) <= 1e-10 * abs(ref), (name, j, v, ref)


GRAD_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vif_grad_ref.npz")
GRAD_PARS = [(0.1, 1.0, 0.1), (0.3, 0.6, 0.25)]           # oracle/make_golden.py: VIF_GRAD_PARS


def _ref_grad_to_terms(gref, pt, n):
    """the reference's gradient entries wrt log(sigma2, ratio, a) -> what they pin of (quad, g_var, g_range), g_p = g1_p / sigma2 + g2_p"""
    return n - 2.0 * gref[0], gref[1], gref[2]           # quad / sigma2, gradient entries of the two covariance parameters


@pytest.mark.parametrize("name", SMALL)
def test_oracle_gradient_reproduces_the_reference(orc, name):
    """orc.vif_grad_terms (restatement of CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i + the derivative branches of the residual-process factor)
    against the unmodified reference's CalcGradPars and its B_grad / D_grad (tests/golden/vif_grad_ref.npz, oracle/make_golden.py vif_grad)."""
    g = np.load(GRAD_GOLDEN)
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    ct = orc.cov_type_id(cf, sh)
    perm, co, nn, ip = orc.vif_setup(coords, m, k, ordering, seed)
    rows = g[name + "_rows"]
    for j, cp in enumerate(GRAD_PARS):
        pt = orc.transform_cov_pars(ct, np.asarray(cp))
        np.testing.assert_allclose(pt, g["%s_pars_trans_%d" % (name, j)], rtol=1e-14)
        quad, logdet, gg, dA, dD, A, D = orc.vif_grad_terms(co, nn, ip, ct, pt[1], pt[2], y[perm])
        nll = quad / 2 / pt[0] + logdet / 2 + n / 2 * (np.log(pt[0]) + np.log(2 * np.pi))
        ref = float(g["%s_negll_%d" % (name, j)])
        assert abs(nll - ref) <= 1e-10 * abs(ref)
        grad = np.array([-quad / pt[0] / 2 + n / 2, gg[0, 0] / pt[0] + gg[0, 1], gg[1, 0] / pt[0] + gg[1, 1]])
        gref = g["%s_grad_%d" % (name, j)]
        np.testing.assert_allclose(grad, gref, rtol=1e-9, atol=1e-9 * np.abs(gref).max())
        for p in range(2):
            rA, rD = g["%s_dA%d_%d" % (name, p, j)], g["%s_dD%d_%d" % (name, p, j)]
            np.testing.assert_allclose(dA[p][rows], rA, rtol=0, atol=1e-9 * np.abs(rA).max())
            np.testing.assert_allclose(dD[p][rows], rD, rtol=0, atol=1e-10 * np.abs(rD).max())


@pytest.mark.parametrize("name", ["vif_u2d_n1500_exp_m15_k40_random", "vif_u3d_n2000_mat25_m20_k64_random"])
def test_host_half_of_the_gradient_against_the_reference(orc, lib_built, name):
    """The product's host half (gpb_c_api.cpp: vif_terms_core -- Sigma_m, Woodbury matrix, the k x k inverses and traces of the analytic gradient)
    with a numpy restatement of the two device passes behind it (tests/vif_harness.py): the reference's gradient to 1e-9."""
    import ctypes
    from gpboost_amd.libpath import find_lib_path
    from tests import vif_harness as vh
    g = np.load(GRAD_GOLDEN)
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    ct = orc.cov_type_id(cf, sh)
    perm, co, nn, ip = orc.vif_setup(coords, m, k, ordering, seed)
    lib = ctypes.CDLL(find_lib_path())
    for j, cp in enumerate(GRAD_PARS):
        pt = orc.transform_cov_pars(ct, np.asarray(cp))
        t7 = vh.host_terms(lib, vh.NumpyDevice(co, nn, ip, ct, pt[1], pt[2], y[perm]))
        nll = t7[0] / 2 / pt[0] + t7[1] / 2 + n / 2 * (np.log(pt[0]) + np.log(2 * np.pi))
        ref = float(g["%s_negll_%d" % (name, j)])
        assert abs(nll - ref) <= 1e-10 * abs(ref)
        grad = np.array([-t7[0] / pt[0] / 2 + n / 2, t7[3] / pt[0] + t7[4], t7[5] / pt[0] + t7[6]])
        gref = g["%s_grad_%d" % (name, j)]
        np.testing.assert_allclose(grad, gref, rtol=1e-9, atol=1e-9 * np.abs(gref).max())


@pytest.fixture(scope="module")
def gpb(lib_built):
    import gpboost_amd
    assert gpboost_amd.device_count() > 0, "no GPU visible: the -m gpu tests must run on the MI355X box"
    return gpboost_amd


def _model(gpb, name, approx="full_scale_vecchia"):
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx=approx, num_neighbors=m, num_ind_points=k,
                      vecchia_ordering=ordering, seed=seed)
    return mdl, coords, y, cps


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.VIF_CASES))
def test_device_likelihood_against_the_reference(gpb, name):
    g = np.load(GOLDEN)
    mdl, coords, y, cps = _model(gpb, name)
    for j, cp in enumerate(cps):
        v = mdl.neg_log_likelihood(np.asarray(cp), y)
        ref = float(g["%s_negll_%d" % (name, j)])
        assert abs(v - ref) <= 1e-8 * abs(ref), (name, j, v, ref)          # north_star: fp64 log-likelihood within 1e-8 relative
    # repeated evaluations are bit-identical (fixed schedules and reduction orders), y = NULL uses the resident response
    assert mdl.neg_log_likelihood(np.asarray(cps[0])) == mdl.neg_log_likelihood(np.asarray(cps[0]), y)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.VIF_CASES))
def test_device_gradient_against_the_reference(gpb, name):
    """The analytic gradient on the device (vif_kernels.hip: derivative mode of the residual-process factor + four n x k x k products; host half
    gpb_c_api.cpp vif_terms_core) against the UNMODIFIED reference's CalcGradPars (CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i,
    re_model_template.h:2205-2330) at two parameter sets of every case, incl. n = 1e5 with 200 inducing points: north_star's 1e-8."""
    g = np.load(GRAD_GOLDEN)
    mdl, coords, y, cps = _model(gpb, name)
    for j, cp in enumerate(GRAD_PARS):
        nll, grad = mdl.neg_log_likelihood_and_gradient(np.asarray(cp), y)
        ref = float(g["%s_negll_%d" % (name, j)])
        assert abs(nll - ref) <= 1e-8 * abs(ref), (name, j, nll, ref)
        gref = g["%s_grad_%d" % (name, j)]
        np.testing.assert_allclose(grad, gref, rtol=1e-8, atol=1e-8 * np.abs(gref).max(), err_msg="%s %d" % (name, j))
    # bit-reproducible
    n1, g1 = mdl.neg_log_likelihood_and_gradient(np.asarray(GRAD_PARS[0]), y)
    n2, g2 = mdl.neg_log_likelihood_and_gradient(np.asarray(GRAD_PARS[0]), y)
    assert n1 == n2 and np.array_equal(g1, g2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_device_derivative_factors_against_the_reference(gpb, name):
    """dA_i, dD_i of the residual-process factor (vif_resid_grad_kernel) against the reference's B_grad / D_grad rows (200 sampled rows per case)."""
    g = np.load(GRAD_GOLDEN)
    mdl, coords, y, cps = _model(gpb, name)
    rows = g[name + "_rows"]
    for j, cp in enumerate(GRAD_PARS):
        mdl.neg_log_likelihood(np.asarray(cp), y)
        for p in range(2):
            dA, dD = mdl.vif_grad_factor(np.asarray(cp), p)
            rA, rD = g["%s_dA%d_%d" % (name, p, j)], g["%s_dD%d_%d" % (name, p, j)]
            np.testing.assert_allclose(dA[rows][:, :rA.shape[1]], rA, rtol=0, atol=1e-8 * np.abs(rA).max(), err_msg="%s dA%d %d" % (name, p, j))
            np.testing.assert_allclose(dD[rows], rD, rtol=0, atol=1e-9 * np.abs(rD).max(), err_msg="%s dD%d %d" % (name, p, j))


@pytest.mark.gpu
def test_alias_ordering_structure_and_rejections(gpb, orc):
    name = "vif_u2d_n1500_exp_m15_k40_random"
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    mdl, coords, y, _ = _model(gpb, name, approx="vif")                    # alias of "full_scale_vecchia" (re_model_template.h:207-209)
    perm, nn = mdl.vecchia_structure()
    perm_o, co, nn_o, ip = orc.vif_setup(coords, m, k, ordering, seed)
    assert np.array_equal(perm, perm_o) and np.array_equal(nn, nn_o)
    g = np.load(GOLDEN)
    assert abs(mdl.neg_log_likelihood(np.asarray(cps[0]), y) - float(g[name + "_negll_0"])) <= 1e-8 * abs(float(g[name + "_negll_0"]))
    with pytest.raises(gpb.GPBoostError, match="full-scale Vecchia"):
        mdl.predict(y, coords[:5], np.asarray(cps[0]), vecchia_pred_type="order_pred_first")       # (fatal in the reference too for this approximation, re_model_template.h:4072-4075)
    with pytest.raises(gpb.GPBoostError, match="gp_approx"):
        gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="full_scale_vecchia_correlation_based", num_neighbors=m, num_ind_points=k)
    with pytest.raises(gpb.GPBoostError, match="num_ind_points"):
        gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vif", num_neighbors=m, num_ind_points=400)


@pytest.mark.gpu
def test_nelder_mead_fit_of_a_vif_model(gpb):
    """A fit with likelihood evaluations only (optimizer_cov = "nelder_mead": the reference's simplex search as it ships it): the optimum is
    a likelihood the reference's own evaluation confirms -- checked through the oracle at the fitted parameters."""
    from oracle import orc as _orc
    name = "vif_u2d_n1500_exp_m15_k40_none"
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    mdl, coords, y, _ = _model(gpb, name)
    start = mdl.neg_log_likelihood(np.asarray(cps[0]), y)
    mdl.fit(y, params={"optimizer_cov": "nelder_mead", "init_cov_pars": np.asarray(cps[0]), "maxit": 200})
    cp = mdl.get_cov_pars()
    end = mdl.get_current_neg_log_likelihood()
    assert end < start and np.all(cp > 0)
    chk = _orc.vif_nll(coords, y, cp, cf, sh, m, k, ordering, seed)
    assert abs(end - chk) <= 1e-8 * abs(chk)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vif_u2d_n1500_exp_m15_k40_none", "vif_u2d_n3000_mat15_m30_k100_random", "vif_u3d_n2000_mat25_m20_k64_random"])
def test_lbfgs_fit_follows_the_reference(gpb, name):
    """The reference's default optimiser (lbfgs) on a VIF model, with the analytic gradient on the device (round 4; round 3 took fourth-order
    differences of the likelihood).  The fits of the unmodified reference (tests/golden/vif_fit_ref.npz, oracle/make_golden.py vif_fit) are
    reproduced: same number of iterations, estimates 1e-6, likelihood 1e-8."""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "vif_fit_ref.npz"))
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    mdl, coords, y, _ = _model(gpb, name)
    mdl.fit(y, params={"optimizer_cov": "lbfgs", "init_cov_pars": np.asarray(cps[0])})
    assert mdl.get_num_optim_iter() == int(g[name + "_num_it"])
    np.testing.assert_allclose(mdl.get_cov_pars(), g[name + "_cov_pars"], rtol=1e-6)
    ref = float(g[name + "_negll"])
    assert abs(mdl.get_current_neg_log_likelihood() - ref) <= 1e-8 * abs(ref)
    # the gradient against second-order differences of the device likelihood (NOT the parity test -- that is test_device_gradient_against_the_reference;
    # the reference differentiates the un-jittered Sigma_m, so its gradient is off the likelihood's exact derivative by ~1e-6 relative)
    cp = np.asarray(cps[0])
    nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    from oracle import orc as _orc
    ct = _orc.cov_type_id(cf, sh)
    pt = _orc.transform_cov_pars(ct, cp)
    cc = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct]

    def f(logp):
        s2, ratio, a = np.exp(logp)
        return mdl.neg_log_likelihood(np.array([s2, ratio * s2, cc / a]), y)
    lp = np.log(pt); fd = np.empty(3)
    for j in range(3):
        e = np.zeros(3); e[j] = 1e-4
        fd[j] = (f(lp + e) - f(lp - e)) / 2e-4
    np.testing.assert_allclose(grad, fd, rtol=1e-5, atol=1e-5 * np.abs(fd).max())


PRED_CASES = ["vif_u2d_n1500_exp_m15_k40_none", "vif_u2d_n3000_mat15_m30_k100_random", "vif_u3d_n2000_mat25_m20_k64_random"]


@pytest.mark.parametrize("name", PRED_CASES)
def test_prediction_oracle_reproduces_the_reference(orc, name):
    """VIF prediction 'order_obs_first_cond_obs_only': the conditional law under the model (orc.vif_predict_obs_only) against the unmodified
    reference (tests/golden/vif_pred_ref.npz, oracle/make_golden.py vif_pred)."""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "vif_pred_ref.npz"))
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    cpred = np.random.default_rng(51).uniform(size=(25, d))
    perm, co, nn, ip = orc.vif_setup(coords, m, k, ordering, seed)
    ct = orc.cov_type_id(cf, sh)
    pt = orc.transform_cov_pars(ct, np.asarray(cps[0]))
    for tag, mp in (("m", m), ("2m", 2 * m)):
        mu, var = orc.vif_predict_obs_only(co, nn, ip, ct, pt, y[perm], cpred, mp, True)
        np.testing.assert_allclose(mu, g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(var, g["%s_%s_var" % (name, tag)], rtol=1e-8)
        _, lvar = orc.vif_predict_obs_only(co, nn, ip, ct, pt, y[perm], cpred, mp, False)
        np.testing.assert_allclose(lvar, g["%s_%s_latent_var" % (name, tag)], rtol=1e-7)


def _condall_points(d):
    rng = np.random.default_rng(52)          # (oracle/make_golden.py: vif_pred_points)
    return np.vstack([rng.uniform(size=(25, d)), 0.37 + 0.02 * rng.uniform(size=(15, d))])


@pytest.mark.parametrize("name", PRED_CASES)
def test_prediction_cond_all_oracle_reproduces_the_reference(orc, name):
    """Round 5 -- VIF prediction 'order_obs_first_cond_all' (the reference's only other prediction type for full-scale Vecchia models): neighbours among
    the observed AND the preceding prediction points, mean / variances / covariance matrix by forward substitution with Bp and the low-rank part
    T W^-1 T' (orc.vif_predict_cond_all) against the unmodified reference (tests/golden/vif_pred_condall_ref.npz)."""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "vif_pred_condall_ref.npz"))
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    coords, y = cases.vif_data(name)
    cpred = _condall_points(d)
    perm, co, nn, ip = orc.vif_setup(coords, m, k, ordering, seed)
    ct = orc.cov_type_id(cf, sh)
    pt = orc.transform_cov_pars(ct, np.asarray(cps[0]))
    for tag, mp in (("m", m), ("2m", 2 * m)):
        mu, var, cov = orc.vif_predict_cond_all(co, nn, ip, ct, pt, y[perm], cpred, mp, True, want_cov=True)
        np.testing.assert_allclose(mu, g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(var, g["%s_%s_var" % (name, tag)], rtol=1e-8)
        _, lvar = orc.vif_predict_cond_all(co, nn, ip, ct, pt, y[perm], cpred, mp, False)
        np.testing.assert_allclose(lvar, g["%s_%s_latent_var" % (name, tag)], rtol=1e-7)
        if tag == "m":
            np.testing.assert_allclose(cov, g["%s_%s_cov" % (name, tag)], rtol=1e-7, atol=1e-11)
    # the type differs from 'cond_obs_only' on these points (the cluster's points condition on each other)
    mu_o, var_o = orc.vif_predict_obs_only(co, nn, ip, ct, pt, y[perm], cpred, m, True)
    assert np.abs(var_o - g["%s_m_var" % name]).max() > 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRED_CASES)
def test_device_prediction_cond_all_against_the_reference(gpb, name):
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "vif_pred_condall_ref.npz"))
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    mdl, coords, y, _ = _model(gpb, name)
    cpred = _condall_points(d)
    cp = np.asarray(cps[0])
    for tag, mp in (("m", m), ("2m", 2 * m)):
        kw = dict(y=y, gp_coords_pred=cpred, cov_pars=cp, num_neighbors_pred=mp, vecchia_pred_type="order_obs_first_cond_all")
        pr = mdl.predict(predict_var=True, predict_response=True, **kw)
        np.testing.assert_allclose(pr["mu"], g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(pr["var"], g["%s_%s_var" % (name, tag)], rtol=1e-8)
        pl = mdl.predict(predict_var=True, predict_response=False, **kw)
        np.testing.assert_allclose(pl["var"], g["%s_%s_latent_var" % (name, tag)], rtol=1e-7)
        pm = mdl.predict(**kw)
        np.testing.assert_allclose(pm["mu"], g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        if tag == "m":
            pc = mdl.predict(predict_cov_mat=True, predict_response=True, **kw)
            np.testing.assert_allclose(pc["cov"], g["%s_%s_cov" % (name, tag)], rtol=1e-7, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRED_CASES)
def test_device_prediction_against_the_reference(gpb, name):
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "vif_pred_ref.npz"))
    n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
    mdl, coords, y, _ = _model(gpb, name)
    cpred = np.random.default_rng(51).uniform(size=(25, d))
    cp = np.asarray(cps[0])
    for tag, mp in (("m", m), ("2m", 2 * m)):
        pr = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, predict_var=True, predict_response=True, num_neighbors_pred=mp)
        np.testing.assert_allclose(pr["mu"], g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(pr["var"], g["%s_%s_var" % (name, tag)], rtol=1e-8)
        pl = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, predict_var=True, predict_response=False, num_neighbors_pred=mp)
        np.testing.assert_allclose(pl["var"], g["%s_%s_latent_var" % (name, tag)], rtol=1e-7)
        pm = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, num_neighbors_pred=mp)
        np.testing.assert_allclose(pm["mu"], g["%s_%s_mu" % (name, tag)], rtol=1e-8, atol=1e-10)
        if tag == "m":      # covariance matrix: the residuals of two prediction points are independent given the observed ones, the inducing part couples them
            pc = mdl.predict(y=y, gp_coords_pred=cpred, cov_pars=cp, predict_cov_mat=True, predict_response=True, num_neighbors_pred=mp)
            np.testing.assert_allclose(pc["cov"], g["%s_%s_cov" % (name, tag)], rtol=1e-7, atol=1e-11)
    # after a fit: the estimated parameters and the resident response
    mdl.fit(y, params={"optimizer_cov": "lbfgs", "init_cov_pars": cp})
    pr = mdl.predict(gp_coords_pred=cpred, predict_var=True)
    assert np.all(np.isfinite(pr["mu"])) and np.all(pr["var"] > 0)
