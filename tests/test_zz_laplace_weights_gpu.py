"""GPU (MI355X): sample weights for the non-Gaussian likelihoods of the Vecchia-Laplace path (round 5; Likelihood::weights_, include/GPBoost/likelihoods.h:666-668
-- every per-datum term of the likelihood and of its derivatives is multiplied by w_d, the per-datum parts of the normalising constants and of the
auxiliary-parameter gradients too) through the C ABI against the UNMODIFIED reference created with weights (tests/golden/laplace_weights_ref.npz,
oracle/make_golden.py laplace_weights): value + gradient (incl. the shape's component for gamma / negative_binomial) 1e-8 at cases.LAPLACE_TIGHT, the boosting
gradient d(-mll)/dF, fits with the reference's iteration counts, latent and response predictions.  (File name: sorts last -- added in round 5.)"""
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


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_WEIGHT_CASES))
def test_weighted_value_gradient_and_boosting_gradient_match_the_reference(gpb, orc, name):
    from gpboost_amd import shim
    wc = cases.LAPLACE_WEIGHT_CASES[name]
    c = cases.LAPLACE_CASES[wc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_weights_ref.npz"))
    coords, y, w = cases.make_weight_data(wc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    st = shim.VecchiaState(co, c["m"])
    st.set_neighbors(nn)
    st.laplace_set_likelihood(wc["lik"])
    prop = wc["lik"] in cases.LAPLACE_PROP_CASES_LIKS
    if wc["lik"] == "gamma" or prop:
        st.laplace_set_response_real(y[perm])
    else:
        st.laplace_set_labels(y[perm].astype(np.int32))
    if "aux" in wc:
        st.laplace_set_aux(wc["aux"])
    if prop:
        st.laplace_set_binomial(wc["lik"].startswith("binomial"))
    st.laplace_set_weights(None if w is None else w[perm])
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
        st.laplace_set_fixed_effects(fe)
        nll_t, grad_t = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == ref.shape
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-8 * abs(ref_v), (nll_t, ref_v)
    if name + "_gradF" in g.files:          # (state of the evaluation with fixed effects just made)
        gF = st.laplace_grad_F()
        out = np.empty_like(gF); out[perm] = gF
        # (-W_d [(Sigma^-1 + W)^-1 d_mll_d_mode]_d carries the CG's 1e-8 stopping error times W_d = trials x information: up to 20 trials in the binomial cases)
        tolF = 1e-7 if wc["lik"].startswith("binomial") else 1e-8
        np.testing.assert_allclose(out, g[name + "_gradF"], rtol=0, atol=tolF * np.abs(g[name + "_gradF"]).max())
    if prop:
        st.close()
        return
    # the weights can be taken away again (the unweighted value: the oracle's, itself pinned to the reference) and unit weights change nothing
    st.laplace_set_fixed_effects(None)
    st.laplace_set_weights(None)
    nll_u, grad_u = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=wc["lik"], aux=wc.get("aux"), cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"],
                                      delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    assert abs(nll_u - on) <= 1e-8 * abs(on)
    st.laplace_set_weights(np.ones(len(y)))
    nll_1, grad_1 = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    assert nll_1 == nll_u and np.array_equal(grad_1, grad_u)
    with pytest.raises(gpb.GPBoostError, match="finite and >= 0"):
        st.laplace_set_weights(-np.ones(len(y)))
    st.close()


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_WEIGHT_CASES))
def test_weighted_model_api_fit_and_prediction_follow_the_reference(gpb, name):
    """GPModel(likelihood, weights) -> GPB_CreateREModel(has_weights, weights): the lbfgs fit (shape estimated where there is one, Likelihood::FindInitialAuxPars
    with the weighted moments) with the reference's iteration count, and the predictions at given parameters."""
    wc = cases.LAPLACE_WEIGHT_CASES[name]
    c = cases.LAPLACE_CASES[wc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_weights_ref.npz"))
    coords, y, w = cases.make_weight_data(wc)
    kw = dict(likelihood=wc["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
              num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"], weights=w)
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(**kw)
    mdl.set_optim_params(dict(cases.LAPLACE_TIGHT))
    aux_kw = {"aux_pars": [wc["aux"]]} if "aux" in wc else {}
    v = mdl.neg_log_likelihood(cp, y, **aux_kw)
    ref_v = float(g[name + "_negll_direct"])
    assert abs(v - ref_v) <= 1e-8 * abs(ref_v), (v, ref_v)
    cases.check_predictions_against_reference(gpb, kw, g, name, y, cp, aux=([wc["aux"]] if "aux" in wc else None))
    m2 = gpb.GPModel(**kw)
    m2.fit(y, params=dict(cases.LAPLACE_TIGHT))
    assert m2.get_num_optim_iter() == int(g[name + "_fit_tight_num_it"]), (m2.get_num_optim_iter(), int(g[name + "_fit_tight_num_it"]))
    np.testing.assert_allclose(m2.get_cov_pars(), g[name + "_fit_tight_cov_pars"], rtol=1e-6)
    if "aux" in wc:
        np.testing.assert_allclose(m2.get_aux_pars(), g[name + "_fit_tight_aux"], rtol=1e-6)
    nll = m2.get_current_neg_log_likelihood()
    assert abs(nll - float(g[name + "_fit_tight_negll"])) <= 1e-8 * abs(nll)
    np.testing.assert_allclose(m2._get_init_cov_pars(), g[name + "_fit_tight_init_cov_pars"], rtol=1e-7)


def test_weight_errors_of_the_model_api(gpb):
    wc = cases.LAPLACE_WEIGHT_CASES["w_poisson_n2000"]
    c = cases.LAPLACE_CASES[wc["model"]]
    coords, y, w = cases.make_weight_data(wc)
    kw = dict(likelihood="poisson", gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    w0 = w.copy(); w0[3] = 0.0
    with pytest.raises(gpb.GPBoostError, match="exactly zero"):
        gpb.GPModel(weights=w0, **kw)
    wn = w.copy(); wn[3] = -1.0
    with pytest.raises(gpb.GPBoostError, match="negative values"):
        gpb.GPModel(weights=wn, **kw)
    # (sample weights together with covariates were refused until round 6; the fits against the reference: tests/test_zz_laplace_train_re_gpu.py::test_fit_with_covariates_and_sample_weights)
    mdl = gpb.GPModel(weights=w, **kw)
    mdl.fit(y, X=np.column_stack([np.ones(len(y)), coords[:, 0]]), params={"maxit": 2})
    assert np.all(np.isfinite(mdl.get_coef())) and np.all(np.isfinite(mdl.get_cov_pars()))
