"""GPU (MI355X): the Student-t likelihood (and, fourth slice of round 5, the lognormal likelihood: cases.LAPLACE_T_CASES entries with lik = "lognormal" -- one auxiliary parameter, the
variance of log y, the same constant-information structure) on the Vecchia-Laplace path (round 5, third slice) -- location = the latent value, auxiliary parameters
(scale, df) both estimated, the reference's default approximation "fisher_laplace" (likelihoods.h:384-423: the information is the constant Fisher information
(df + 1) / (df + 3) / scale^2) -- through the C ABI against the UNMODIFIED reference (tests/golden/laplace_t_ref.npz, oracle/make_golden.py laplace_t):
  * value at the default thresholds; value + gradient wrt (log sigma1^2, log a, log scale, log df) at cases.LAPLACE_TIGHT from the reference's own CalcGradPars (1e-8),
    without and with fixed effects; the parts of the auxiliary gradient against the oracle;
  * the model surface: evaluation, lbfgs fits with both auxiliary parameters in the vector (the reference's iteration counts), GPB_GetAuxPars (two values, "scale_SEP_df"),
    latent and response predictions.
(File name: sorts last -- added in round 5.)"""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RC = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}
TIGHT_ORC = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])


@pytest.fixture(scope="module")
def gpb(lib_built):
    import gpboost_amd
    assert gpboost_amd.device_count() > 0, "no GPU visible: the -m gpu tests must run on the MI355X box"
    return gpboost_amd


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_T_CASES))
def test_value_and_gradient_match_the_reference(gpb, orc, name):
    from gpboost_amd import shim
    tc = cases.LAPLACE_T_CASES[name]
    lik = tc.get("lik", "t")          # (lognormal, link 7: ONE auxiliary parameter -- the variance of log y -- and the same constant-information structure, likelihoods.h:505-513)
    naux = len(tc["aux"])
    c = cases.LAPLACE_CASES[tc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_t_ref.npz"))
    coords, y = cases.make_t_data(tc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    st = shim.VecchiaState(co, c["m"])
    st.set_neighbors(nn)
    st.laplace_set_likelihood(lik)
    st.laplace_set_response_real(y[perm])
    st.laplace_set_aux(tc["aux"])
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    negll, _ = st.laplace_logit(ct, cp[0], a)
    ref0 = float(g[name + "_negll_0"])
    assert abs(negll - ref0) <= 1e-8 * abs(ref0), (negll, ref0)
    for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
        st.laplace_set_fixed_effects(fe)
        nll_t, grad_t = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == (2 + naux,)
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=3e-8 * np.abs(ref).max())      # (the scale's trace term multiplies the block CG's 1e-8 stopping error by dW / d log scale = -2 W: seen 1.6e-8 of the gradient's scale on the d = 3 case with fixed effects)
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-8 * abs(ref_v), (nll_t, ref_v)
    # other auxiliary parameters: against the oracle (itself pinned to the reference above); the boosting gradient is -d log p / d loc alone (no determinant / implicit part)
    st.laplace_set_fixed_effects(None)
    for aux2 in (((0.8, 2.2), (0.25, 15.0)) if lik == "t" else ((0.6,), (0.05,))):
        st.laplace_set_aux(aux2)
        nll2, grad2 = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=lik, aux=aux2, **TIGHT_ORC)
        assert abs(nll2 - on) <= 1e-8 * abs(on), (aux2, nll2, on)
        np.testing.assert_allclose(grad2, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
        ga = st.laplace_grad_aux()
        assert ga.shape == (4 * naux,) and ga[3] == 0.0 and ga[-1] == 0.0              # no implicit part
    with pytest.raises(gpb.GPBoostError, match="not > 0"):
        st.laplace_set_aux((0.5, -1.0) if lik == "t" else (-0.5,))
    with pytest.raises(gpb.GPBoostError, match="parameters"):
        st.laplace_set_aux(0.5 if lik == "t" else (0.5, 2.0))
    if lik == "lognormal":
        with pytest.raises(gpb.GPBoostError, match="> 0"):
            st.laplace_set_response_real(-y[perm])
    st.close()


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_T_CASES))
def test_model_api_evaluation_fit_and_prediction_follow_the_reference(gpb, name):
    tc = cases.LAPLACE_T_CASES[name]
    c = cases.LAPLACE_CASES[tc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_t_ref.npz"))
    coords, y = cases.make_t_data(tc)
    lik = tc.get("lik", "t")
    kw = dict(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
              num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(**kw)
    assert mdl.get_num_aux_pars() == len(tc["aux"])
    v = mdl.neg_log_likelihood(cp, y, aux_pars=list(tc["aux"]))
    ref0 = float(g[name + "_negll_0"])
    assert abs(v - ref0) <= 1e-8 * abs(ref0), (v, ref0)
    np.testing.assert_allclose(mdl.get_aux_pars(), tc["aux"], rtol=0, atol=0)
    cases.check_predictions_against_reference(gpb, kw, g, name, y, cp, aux=list(tc["aux"]))
    for key, cfg, rtol, ntol in (("_fit", {}, 1e-3, 1e-7), ("_fit_tight", dict(cases.LAPLACE_TIGHT), 1e-6, 1e-8)):
        m2 = gpb.GPModel(**kw)
        m2.fit(y, params=dict(cfg))
        assert m2.get_num_optim_iter() == int(g[name + key + "_num_it"]), (key, m2.get_num_optim_iter(), int(g[name + key + "_num_it"]))
        np.testing.assert_allclose(m2.get_cov_pars(), g[name + key + "_cov_pars"], rtol=rtol)
        np.testing.assert_allclose(m2.get_aux_pars(), g[name + key + "_aux"], rtol=rtol)
        nll = m2.get_current_neg_log_likelihood()
        assert abs(nll - float(g[name + key + "_negll"])) <= ntol * abs(nll)


@pytest.mark.parametrize("case", ["t_n1500", "lognormal_n1500"])
@pytest.mark.parametrize("pcn,rank", [("pivoted_cholesky", 50), ("fitc", 80)])
def test_auxiliary_gradient_with_the_low_rank_preconditioners(gpb, orc, pcn, rank, case):
    """t x pivoted_cholesky / fitc: the pivoted_cholesky / fitc branches of CalcLogDetStochDerivAuxParVecchia (likelihoods.h:16800-16837) -- W^-1 P^-1 Z recomputed from the
    probes, the deterministic traces by pc_aux_sums -- against the oracle (pinned to the reference's CalcGradPars for these combinations at 8e-9, DESIGN.md 4.6;
    lognormal x pivoted_cholesky: value 1e-15, gradient 2.5e-12 of its scale in a one-off run of the unmodified reference, same section)."""
    from gpboost_amd import shim
    tc = cases.LAPLACE_T_CASES[case]
    lik = tc.get("lik", "t")
    c = cases.LAPLACE_CASES[tc["model"]]
    coords, y = cases.make_t_data(tc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    st = shim.VecchiaState(co, c["m"])
    st.set_neighbors(nn)
    st.laplace_set_likelihood(lik)
    st.laplace_set_response_real(y[perm])
    st.laplace_set_aux(tc["aux"])
    st.laplace_set_preconditioner(pcn, rank)
    ip = orc.vif_setup(coords, c["m"], rank, c["ordering"], c["seed"])[3] if pcn == "fitc" else None
    if ip is not None:
        st.laplace_set_inducing_points(ip)
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    nll, grad = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    ctx = orc.fitc_preconditioner(co, ip, ct, cp[0], a) if pcn == "fitc" else orc.pivoted_cholesky_preconditioner(co, ct, cp[0], a, rank=rank)
    with ctx:
        on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=lik, aux=tc["aux"], **TIGHT_ORC)
    assert abs(nll - on) <= 1e-8 * abs(on), (nll, on)
    assert grad.shape == (2 + len(tc["aux"]),)
    np.testing.assert_allclose(grad, og, rtol=1e-8, atol=3e-8 * np.abs(og).max())
    st.close()


def test_model_api_t_with_fixed_df_follows_the_reference(gpb):
    """Round 6: likelihood "t_fix_df" -- Student-t with the degrees of freedom HELD at likelihood_additional_param and only the scale estimated (estimate_df_t_ = false:
    likelihoods.h:384-407, :10466-10471; the df stay in the optimiser's vector with a zero gradient, :16179-16183, and SetAuxPars takes over the first
    num_aux_pars_estim_ = 1 values only, :2780-2789) -- and the reading of likelihood_additional_param by GPB_CreateREModel, against the unmodified
    reference (tests/golden/laplace_t_fixdf_ref.npz, oracle/make_golden.py laplace_t_fixdf): evaluation 1e-8, fits with its iteration counts."""
    g = np.load(os.path.join(GOLD, "laplace_t_fixdf_ref.npz"))
    tc = cases.LAPLACE_T_CASES["t_n1500"]
    c = cases.LAPLACE_CASES[tc["model"]]
    coords, y = cases.make_t_data(tc)
    kw = dict(gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(likelihood="t_fix_df", likelihood_additional_param=5.0, **kw)
    assert mdl.get_num_aux_pars() == 2
    mdl.set_optim_params(dict(cases.LAPLACE_TIGHT, init_aux_pars=np.array([0.5, 3.0])))       # the 3 is NOT taken over: the df stay at likelihood_additional_param
    v = mdl.neg_log_likelihood(cp, y)
    ref = float(g["df5_negll"])
    assert abs(v - ref) <= 1e-8 * abs(ref), (v, ref)
    np.testing.assert_allclose(mdl.get_aux_pars(), g["df5_aux_after_eval"], rtol=0, atol=0)
    for key, extra, cfg, rtol, ntol in (("df5", dict(likelihood_additional_param=5.0), dict(cases.LAPLACE_TIGHT), 1e-6, 1e-8), ("dfdef", {}, {}, 1e-3, 1e-7)):
        m2 = gpb.GPModel(likelihood="t_fix_df", **extra, **kw)
        m2.fit(y, params=dict(cfg))
        assert m2.get_num_optim_iter() == int(g[key + "_fit_num_it"]), (key, m2.get_num_optim_iter(), int(g[key + "_fit_num_it"]))
        np.testing.assert_allclose(m2.get_cov_pars(), g[key + "_fit_cov_pars"], rtol=rtol)
        aux = m2.get_aux_pars()
        assert aux[1] == g[key + "_fit_aux"][1], (aux, g[key + "_fit_aux"])             # the df did not move: 5, or the internal default 2 (likelihoods.h:391-393)
        np.testing.assert_allclose(aux[0], g[key + "_fit_aux"][0], rtol=rtol)
        nll = m2.get_current_neg_log_likelihood(); nref = float(g[key + "_fit_negll"])
        assert abs(nll - nref) <= ntol * abs(nref), (key, nll, nref)
    # "t" (df estimated) created with likelihood_additional_param = 5: the start value of the df (aux_pars_ = {1, 5}, likelihoods.h:397-399)
    m3 = gpb.GPModel(likelihood="t", likelihood_additional_param=5.0, **kw)
    m3.set_optim_params(dict(cases.LAPLACE_TIGHT, estimate_aux_pars=False))
    v3 = m3.neg_log_likelihood(cp, y)
    r3 = float(g["t_df5_negll"])
    assert abs(v3 - r3) <= 1e-8 * abs(r3), (v3, r3)
    np.testing.assert_allclose(m3.get_aux_pars(), g["t_df5_aux"], rtol=0, atol=0)
    # a parameter the likelihood does not take is refused, not dropped; a negative df as in the reference
    with pytest.raises(gpb.GPBoostError):
        gpb.GPModel(likelihood="gamma", likelihood_additional_param=2.0, **kw)
    with pytest.raises(gpb.GPBoostError):
        gpb.GPModel(likelihood="t", likelihood_additional_param=-1.0, **kw)


@pytest.mark.parametrize("name", ["gamma_n1500", "t_n1500", "t_fix_df5_n1500"])
def test_model_api_standard_deviations_of_auxiliary_parameters_follow_the_reference(gpb, name):
    """Round 6: GPB_GetAuxPars(calc_std_dev = true) and GPB_GetCovPar(calc_std_dev = true) of a model whose auxiliary parameters are estimated -- the joint numerical
    Hessian of CalcStdDevCovParAuxParsNonGaussian (re_model_template.h:11029-11117) on the device gradient -- after the lbfgs fit, against the unmodified reference
    (tests/golden/laplace_aux_se_ref.npz, oracle/make_golden.py laplace_aux_se).  The Hessian differences gradients that each carry the stochastic trace estimate and the CG's
    stopping error over a step of 1e-4, so the standard deviations agree to 1e-3 (estimates 1e-6); the df of "t_fix_df" have none (NaN), as in the reference."""
    g = np.load(os.path.join(GOLD, "laplace_aux_se_ref.npz"))
    if name.startswith("gamma"):
        cs = cases.LAPLACE_AUX_CASES[name]; coords, y = cases.make_aux_data(cs); lik, extra = "gamma", {}
    else:
        cs = cases.LAPLACE_T_CASES["t_n1500"]; coords, y = cases.make_t_data(cs)
        lik, extra = ("t_fix_df", dict(likelihood_additional_param=5.0)) if "fix" in name else ("t", {})
    c = cases.LAPLACE_CASES[cs["model"]]
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"],
                      vecchia_ordering=c["ordering"], seed=c["seed"], **extra)
    mdl.fit(y, params=dict(cases.LAPLACE_TIGHT))
    assert mdl.get_num_optim_iter() == int(g[name + "_num_it"])
    cov = mdl.get_cov_pars(std_err=True); aux = mdl.get_aux_pars(std_err=True)
    rc, ra = g[name + "_cov_pars"], g[name + "_aux"]
    k = len(ra) // 2
    np.testing.assert_allclose(cov[:2], rc[:2], rtol=1e-6)
    np.testing.assert_allclose(aux[:k], ra[:k], rtol=1e-6)
    np.testing.assert_allclose(cov[2:], rc[2:], rtol=1e-3)
    assert np.array_equal(np.isnan(aux[k:]), np.isnan(ra[k:])), (aux, ra)
    ok = ~np.isnan(ra[k:])
    np.testing.assert_allclose(aux[k:][ok], ra[k:][ok], rtol=1e-3)


def test_model_api_nelder_mead_with_an_estimated_auxiliary_parameter_follows_the_reference(gpb):
    """Round 6: optimizer_cov = "nelder_mead" for a likelihood whose auxiliary parameter is estimated -- OptimLib's simplex search over (log sigma1^2, log a, log shape),
    likelihood evaluations only (optim_utils.h:61-213, nm.hpp:95-372) -- on gamma_n1500 at cases.LAPLACE_TIGHT: the reference's iteration count, estimates 1e-6, value 1e-8."""
    g = np.load(os.path.join(GOLD, "laplace_aux_se_ref.npz"))
    ac = cases.LAPLACE_AUX_CASES["gamma_n1500"]; c = cases.LAPLACE_CASES[ac["model"]]
    coords, y = cases.make_aux_data(ac)
    mdl = gpb.GPModel(likelihood="gamma", gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"],
                      vecchia_ordering=c["ordering"], seed=c["seed"])
    mdl.fit(y, params=dict(cases.LAPLACE_TIGHT, optimizer_cov="nelder_mead"))
    assert mdl.get_num_optim_iter() == int(g["gamma_n1500_nm_num_it"]), (mdl.get_num_optim_iter(), int(g["gamma_n1500_nm_num_it"]))
    np.testing.assert_allclose(mdl.get_cov_pars(), g["gamma_n1500_nm_cov_pars"], rtol=1e-6)
    np.testing.assert_allclose(mdl.get_aux_pars(), g["gamma_n1500_nm_aux"], rtol=1e-6)
    nll = mdl.get_current_neg_log_likelihood(); ref = float(g["gamma_n1500_nm_negll"])
    assert abs(nll - ref) <= 1e-8 * abs(ref), (nll, ref)


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_AUX_GD_CASES))
def test_model_api_gradient_descent_with_estimated_auxiliary_parameters_follows_the_reference(gpb, name):
    """Round 6: optimizer_cov = "gradient_descent" for likelihoods whose auxiliary parameters are estimated -- the reference's internal loop with one learning rate for the
    covariance block and one for the auxiliary block, both Armijo conditions, joint halving, Nesterov momentum on the log scale (re_model_template.h:1514-1660, :8354-8375,
    :8420-8470, :8690-8850; gpb_optim.cpp run_gradient_descent_laplace_aux) -- against the unmodified reference at cases.LAPLACE_TIGHT
    (tests/golden/laplace_aux_gd_ref.npz): the iteration count, estimates 1e-6, value 1e-8."""
    g = np.load(os.path.join(GOLD, "laplace_aux_gd_ref.npz"))
    coords, y, c, lik, naux, cfg = cases.aux_gd_case(name)
    mdl = gpb.GPModel(likelihood=lik, gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"],
                      vecchia_ordering=c["ordering"], seed=c["seed"])
    params = dict(cases.LAPLACE_TIGHT, optimizer_cov="gradient_descent")
    for k, v in cfg.items():
        params[{"max_iter": "maxit"}.get(k, k)] = v
    mdl.fit(y, params=params)
    assert mdl.get_num_optim_iter() == int(g[name + "_num_it"]), (mdl.get_num_optim_iter(), int(g[name + "_num_it"]))
    np.testing.assert_allclose(mdl.get_cov_pars(), g[name + "_cov_pars"], rtol=1e-6)
    np.testing.assert_allclose(mdl.get_aux_pars()[:naux], g[name + "_aux"], rtol=1e-6)
    nll = mdl.get_current_neg_log_likelihood(); ref = float(g[name + "_negll"])
    assert abs(nll - ref) <= 1e-8 * abs(ref), (nll, ref)
