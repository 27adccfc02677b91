"""GPU (MI355X): likelihoods with an auxiliary parameter on the Vecchia-Laplace path -- gamma, negative_binomial and (second slice) beta (first slice of
round 5) -- through the C ABI against the UNMODIFIED reference (tests/golden/laplace_aux_ref.npz, oracle/make_golden.py laplace_aux):
  * value at the default thresholds, value + gradient wrt (log sigma1^2, log a, log shape) at cases.LAPLACE_TIGHT (the reference's own CalcGradPars ->
    CalcGradNegMargLikelihoodLaplaceApproxVecchia incl. its auxiliary-parameter branch, likelihoods.h:6743-6808): 1e-8 relative;
  * the oracle step by step (the parts of the auxiliary-parameter gradient);
  * fits with the shape estimated jointly with the covariance parameters (lbfgs) and held fixed: the reference's iteration counts and estimates.
(File name: sorts last -- added in round 5.)"""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RC = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}


@pytest.fixture(scope="module")
def gpb(lib_built):
    import gpboost_amd
    assert gpboost_amd.device_count() > 0, "no GPU visible: the -m gpu tests must run on the MI355X box"
    return gpboost_amd


def _state(orc, ac):
    from gpboost_amd import shim
    c = cases.LAPLACE_CASES[ac["model"]]
    coords, y = cases.make_aux_data(ac)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    st = shim.VecchiaState(co, c["m"])
    st.set_neighbors(nn)
    st.laplace_set_likelihood(ac["lik"])
    if ac["lik"] in ("gamma", "beta"):
        st.laplace_set_response_real(y[perm])
    else:
        st.laplace_set_labels(y[perm].astype(np.int32))
    st.laplace_set_aux(ac["aux"])
    return st, c, coords, y, perm, co, nn, ct


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_AUX_CASES))
def test_value_and_gradient_match_the_reference(gpb, orc, name):
    ac = cases.LAPLACE_AUX_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_aux_ref.npz"))
    st, c, coords, y, perm, co, nn, ct = _state(orc, ac)
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    negll, _ = st.laplace_logit(ct, cp[0], a)
    ref0 = float(g[name + "_negll_0"])
    assert abs(negll - ref0) <= 1e-8 * abs(ref0), (negll, ref0)
    for fe_key, fe in (("", None), ("_fe", cases.aux_fixed_effects(ac, coords)[perm])):
        st.laplace_set_fixed_effects(fe)
        nll_t, grad_t = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == (3,)
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-8 * abs(ref_v), (nll_t, ref_v)
    # another shape: the normalising constant and every likelihood term follow (against the oracle, itself pinned to the reference at 1e-9)
    st.laplace_set_fixed_effects(None)
    # (beta: a precision far below the data's makes the density bathtub-shaped, the information negative and the reference's own mode finding fail -- stay near the fixture's)
    for aux2 in ((0.6 * ac["aux"], 2.0 * ac["aux"]) if ac["lik"] == "beta" else (1.0, 0.37 * ac["aux"], 4.1 * ac["aux"])):
        st.laplace_set_aux(aux2)
        nll2, grad2 = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=ac["lik"], aux=aux2, cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"],
                                          delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
        assert abs(nll2 - on) <= 1e-8 * abs(on), (aux2, nll2, on)
        np.testing.assert_allclose(grad2, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
    st.close()


def test_errors_of_the_auxiliary_parameter_entry_points(gpb, orc):
    from gpboost_amd import shim
    ac = cases.LAPLACE_AUX_CASES["gamma_n1500"]
    st, c, coords, y, perm, co, nn, ct = _state(orc, ac)
    with pytest.raises(gpb.GPBoostError, match="not > 0"):
        st.laplace_set_aux(-1.0)
    with pytest.raises(gpb.GPBoostError, match="must be > 0"):
        st.laplace_set_response_real(-np.abs(y[perm]))
    with pytest.raises(gpb.GPBoostError, match="real-valued"):
        st.laplace_set_labels(np.ones(len(y), dtype=np.int32))
    st.laplace_set_likelihood("poisson")
    with pytest.raises(gpb.GPBoostError, match="no auxiliary"):
        st.laplace_set_aux(2.0)
    st.close()


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_AUX_CASES))
def test_model_api_evaluation_fit_and_prediction_follow_the_reference(gpb, name):
    """The same through the reference's model surface (GPModel -> GPB_CreateREModel / GPB_SetOptimConfig(init_aux_pars, estimate_aux_pars) /
    GPB_EvalNegLogLikelihood / GPB_OptimCovPar / GPB_GetAuxPars / GPB_PredictREModel): evaluation at a given shape, the fit with the shape estimated from
    Likelihood::FindInitialAuxPars' start (default and tight thresholds) and with the shape held fixed, latent and response predictions."""
    ac = cases.LAPLACE_AUX_CASES[name]
    c = cases.LAPLACE_CASES[ac["model"]]
    g = np.load(os.path.join(GOLD, "laplace_aux_ref.npz"))
    coords, y = cases.make_aux_data(ac)
    kw = dict(likelihood=ac["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
              num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(**kw)
    assert mdl.get_num_aux_pars() == 1
    v = mdl.neg_log_likelihood(cp, y, aux_pars=[ac["aux"]])
    ref0 = float(g[name + "_negll_0"])
    assert abs(v - ref0) <= 1e-8 * abs(ref0), (v, ref0)
    np.testing.assert_allclose(mdl.get_aux_pars(), [ac["aux"]], rtol=0, atol=0)
    # predictions at (cov_pars, aux)
    cases.check_predictions_against_reference(gpb, kw, g, name, y, cp, aux=[ac["aux"]])
    # fits with the shape estimated
    for key, cfg, rtol, ntol in (("_fit", {}, 2e-2 if ac.get("flat_default") else 1e-4, 1e-7), ("_fit_tight", dict(cases.LAPLACE_TIGHT), 1e-6, 1e-8)):
        m2 = gpb.GPModel(**kw)
        m2.fit(y, params=dict(cfg))
        assert m2.get_num_optim_iter() == int(g[name + key + "_num_it"]), (key, m2.get_num_optim_iter(), int(g[name + key + "_num_it"]))
        np.testing.assert_allclose(m2.get_cov_pars(), g[name + key + "_cov_pars"], rtol=rtol)
        np.testing.assert_allclose(m2.get_aux_pars(), g[name + key + "_aux"], rtol=rtol)
        nll = m2.get_current_neg_log_likelihood()
        assert abs(nll - float(g[name + key + "_negll"])) <= ntol * abs(nll)
        np.testing.assert_allclose(m2._get_init_cov_pars(), g[name + key + "_init_cov_pars"], rtol=1e-7)
    # ... and held at the given value
    m3 = gpb.GPModel(**kw)
    m3.fit(y, params={"init_aux_pars": [ac["aux"]], "estimate_aux_pars": False})
    assert m3.get_num_optim_iter() == int(g[name + "_fitfix_num_it"])
    np.testing.assert_allclose(m3.get_cov_pars(), g[name + "_fitfix_cov_pars"], rtol=1e-3 if ac.get("flat_default") else 1e-4)     # (default thresholds; cases.py on flat_default)
    np.testing.assert_allclose(m3.get_aux_pars(), [ac["aux"]], rtol=0)
    nll3 = m3.get_current_neg_log_likelihood()
    assert abs(nll3 - float(g[name + "_fitfix_negll"])) <= 1e-7 * abs(nll3)


def test_model_api_errors_for_the_auxiliary_parameter_likelihoods(gpb):
    ac = cases.LAPLACE_AUX_CASES["gamma_n1500"]
    c = cases.LAPLACE_CASES[ac["model"]]
    coords, y = cases.make_aux_data(ac)
    mdl = gpb.GPModel(likelihood="gamma", gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    with pytest.raises(gpb.GPBoostError, match="Must have y > 0"):
        mdl.neg_log_likelihood(np.array([1.0, 0.1]), -y)
    with pytest.raises(gpb.GPBoostError, match="not > 0"):
        mdl.set_optim_params({"init_aux_pars": [-2.0]})
    mdl.fit(y, params={"optimizer_cov": "gradient_descent", "maxit": 2})       # (an error until round 6; the fits against the reference: tests/test_zz_laplace_t_gpu.py)
    assert 1 <= mdl.get_num_optim_iter() <= 2 and np.all(np.isfinite(mdl.get_aux_pars()))
    m2 = gpb.GPModel(likelihood="negative_binomial", gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    with pytest.raises(gpb.GPBoostError, match="non-integer"):
        m2.neg_log_likelihood(np.array([1.0, 0.1]), y)
