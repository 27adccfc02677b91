"""GPU (MI355X): cg_preconditioner_type = "vecchia_response" on the Vecchia-Laplace path (round 6; the fifth entry of the reference's
SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, re_model_template.h:5906) -- the (W^-1 + Sigma) solves preconditioned with the Vecchia approximation of W^-1 + Sigma itself:
P^-1 = B_p' D_p^-1 B_p, the factor renewed for every W by ONE launch of the Gaussian path's point kernel with the diagonal additions 1 / W_i
(gpb_laplace.inc pc_refresh_vr; reference: likelihoods.h:16315-16323, :16439-16450, :16471-16473, CG_utils.cpp:300-303, re_model_template.h:5473-5492) -- through the C ABI
against the UNMODIFIED reference (tests/golden/laplace_vresp_ref.npz, oracle/make_golden.py laplace_vresp):
  * the value at cases.LAPLACE_TIGHT 1e-8 relative without and with fixed effects, a second evaluation of the same model at other parameters, the value at the default thresholds;
  * the oracle's restatement at other parameters, iteration counts included;
  * the gradient is refused with the reference's message (likelihoods.h:6570-6572);
  * the model surface: GPB_SetOptimConfig(cg_preconditioner_type = "vecchia_response" / its aliases), evaluation, a Nelder-Mead fit with the reference's estimates, lbfgs refused;
  * together with sample weights and with repeated locations.
(File name: sorts last -- added in round 6.)"""
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


def _state(orc, pc):
    from gpboost_amd import shim
    c = cases.LAPLACE_CASES[pc["model"]]
    coords, y = cases.make_pivchol_data(pc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    st = shim.VecchiaState(co, c["m"])
    st.set_neighbors(nn)
    st.laplace_set_likelihood(pc["lik"])
    if pc["lik"] == "gamma":
        st.laplace_set_response_real(y[perm])
    else:
        st.laplace_set_labels(y[perm].astype(np.int32))
    if "aux" in pc:
        st.laplace_set_aux(pc["aux"])
    st.laplace_set_preconditioner("vecchia_response")
    return st, c, coords, y, perm, co, nn, ct


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_VRESP_CASES))
def test_value_matches_the_reference(gpb, orc, name):
    pc = cases.LAPLACE_VRESP_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_vresp_ref.npz"))
    st, c, coords, y, perm, co, nn, ct = _state(orc, pc)
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    negll, info = st.laplace_logit(ct, cp[0], a)
    ref_d = float(g[name + "_negll_default"])
    assert abs(negll - ref_d) <= 1e-6 * abs(ref_d), (negll, ref_d)       # default thresholds: two correct implementations stop one CG iteration apart
    # tight thresholds: first point, then a second point on the same handle (GPB_EvalNegLogLikelihood starts every mode finding at 0, re_model_template.h:3199-3201)
    v0, _ = st.laplace_logit(ct, cp[0], a, reset_mode=True, **cases.LAPLACE_TIGHT)
    ref0 = float(g[name + "_negll_tight"])
    assert abs(v0 - ref0) <= 1e-8 * abs(ref0), (v0, ref0)
    var1, rho1 = cases.LAPLACE_VRESP_SECOND_PARS
    v1, _ = st.laplace_logit(ct, var1, RC[ct] / rho1, reset_mode=True, **cases.LAPLACE_TIGHT)
    ref1 = float(g[name + "_negll_tight_1"])
    assert abs(v1 - ref1) <= 1e-8 * abs(ref1), (v1, ref1)
    st.laplace_set_fixed_effects(cases.laplace_fixed_effects(coords)[perm])
    vf, _ = st.laplace_logit(ct, cp[0], a, reset_mode=True, **cases.LAPLACE_TIGHT)
    reff = float(g[name + "_fe_negll_tight"])
    assert abs(vf - reff) <= 1e-8 * abs(reff), (vf, reff)
    # the gradient does not exist with this preconditioner (likelihoods.h:6570-6572)
    with pytest.raises(gpb.GPBoostError, match="not correctly implemented for the 'vecchia_response' preconditioner"):
        st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    # back to "vadu" on the same handle: the vadu oracle's value and gradient
    st.laplace_set_fixed_effects(None)
    st.laplace_set_preconditioner("vadu")
    nll_v, grad_v = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), **TIGHT_ORC)
    assert abs(nll_v - on) <= 1e-8 * abs(on), (nll_v, on)
    np.testing.assert_allclose(grad_v, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
    st.close()


@pytest.mark.parametrize("name", ["vr_logit_n2000", "vr_probit_u3d_n1200", "vr_gamma_n1500"])
def test_steps_match_the_oracle_at_other_parameters(gpb, orc, name):
    """Value, log-determinant, Newton / CG / Lanczos iteration counts and the mode against orc.vecchia_laplace_logit inside orc.vecchia_response_preconditioner at parameters
    the fixture does not hold (the oracle is pinned to the reference on the fixture's, tests/test_oracle_golden.py)."""
    pc = cases.LAPLACE_VRESP_CASES[name]
    st, c, coords, y, perm, co, nn, ct = _state(orc, pc)
    for var, rho in ((0.45, 0.3), (2.2, 0.08)):
        a = RC[ct] / rho
        v, info = st.laplace_logit(ct, var, a, reset_mode=True, want_mode=True, **cases.LAPLACE_TIGHT)
        with orc.vecchia_response_preconditioner(co, ct, var, a):
            ov, oi = orc.vecchia_laplace_logit(co, nn, ct, var, a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), **TIGHT_ORC)
        assert abs(v - ov) <= 1e-8 * abs(ov), (v, ov)
        assert abs(info["log_det"] - oi["log_det"]) <= 1e-8 * abs(oi["log_det"]) + 1e-8 * abs(ov)
        assert info["newton_it"] == oi["newton_it"] and abs(info["cg_it"] - oi["cg_it"]) <= 2 and abs(info["lanczos_it"] - oi["lanczos_it"]) <= 1, (info, oi)
        np.testing.assert_allclose(info["mode"], oi["mode"], rtol=0, atol=1e-7 * np.abs(oi["mode"]).max())
    st.close()


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_VRESP_CASES))
def test_model_api_evaluation_and_nelder_mead_fit_follow_the_reference(gpb, name):
    """GPModel -> GPB_SetOptimConfig(cg_preconditioner_type = "vecchia_response") / GPB_EvalNegLogLikelihood / GPB_OptimCovPar (Nelder-Mead: cases.LAPLACE_VRESP_NM) /
    GPB_GetCGPreconditionerType; a gradient-based optimiser ends with the reference's message."""
    pc = cases.LAPLACE_VRESP_CASES[name]
    c = cases.LAPLACE_CASES[pc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_vresp_ref.npz"))
    coords, y = cases.make_pivchol_data(pc)
    kw = dict(likelihood=pc["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
              num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(**kw)
    mdl.set_optim_params({"cg_preconditioner_type": "vecchia_observable"})          # ParsePreconditionerAlias (re_model_template.h:7507-7511)
    assert mdl.get_cg_preconditioner_type() == "vecchia_response"
    v = mdl.neg_log_likelihood(cp, y, aux_pars=[pc["aux"]]) if "aux" in pc else mdl.neg_log_likelihood(cp, y)
    ref_d = float(g[name + "_negll_default"])
    assert abs(v - ref_d) <= 1e-6 * abs(ref_d), (v, ref_d)
    nm = cases.LAPLACE_VRESP_NM
    m2 = gpb.GPModel(**kw)
    m2.fit(y, params=dict(cases.LAPLACE_TIGHT, cg_preconditioner_type="vecchia_response", optimizer_cov=nm["optimizer_cov"], maxit=nm["maxit"]))
    assert m2.get_num_optim_iter() == int(g[name + "_fit_num_it"]), (m2.get_num_optim_iter(), int(g[name + "_fit_num_it"]))
    np.testing.assert_allclose(m2.get_cov_pars(), g[name + "_fit_cov_pars"], rtol=1e-6)
    if "aux" in pc:
        np.testing.assert_allclose(m2.get_aux_pars(), g[name + "_fit_aux"], rtol=1e-6)
    nll = m2.get_current_neg_log_likelihood()
    assert abs(nll - float(g[name + "_fit_negll"])) <= 1e-8 * abs(nll)
    # predictions keep solving with "vadu" (same quantity, another solver): finite, variances positive
    cpred = np.random.default_rng(5).uniform(size=(7, coords.shape[1]))
    pr = m2.predict(y=y, gp_coords_pred=cpred, predict_var=True, predict_response=False)
    assert np.all(np.isfinite(pr["mu"])) and np.all(pr["var"] > 0)
    if name == "vr_logit_n2000":
        m3 = gpb.GPModel(**kw)
        with pytest.raises(gpb.GPBoostError, match="not correctly implemented for the 'vecchia_response' preconditioner"):
            m3.fit(y, params=dict(cg_preconditioner_type="vecchia_response", optimizer_cov="lbfgs", maxit=3))


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_VRESP_EXTRA_CASES))
def test_model_api_with_weights_and_repeated_locations(gpb, name):
    """vecchia_response together with sample weights (the information, and with it the pseudo nugget 1 / W, is weighted) and with repeated locations (the information of a
    random effect is the sum over its data; the factor lives on the unique locations) against the reference library at cases.LAPLACE_TIGHT: evaluation 1e-8, the Nelder-Mead fit."""
    ec = cases.LAPLACE_VRESP_EXTRA_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_vresp_ref.npz"))
    kw, y, cp, aux = cases.pc_extra_model(ec)
    nm = cases.LAPLACE_VRESP_NM
    mdl = gpb.GPModel(**kw)
    mdl.set_optim_params(dict(cases.LAPLACE_TIGHT, cg_preconditioner_type="vecchia_response"))
    v = mdl.neg_log_likelihood(cp, y)
    ref = float(g[name + "_negll"])
    assert abs(v - ref) <= 1e-8 * abs(ref), (v, ref)
    m2 = gpb.GPModel(**kw)
    m2.fit(y, params=dict(cases.LAPLACE_TIGHT, cg_preconditioner_type="vecchia_response", optimizer_cov=nm["optimizer_cov"], maxit=nm["maxit"]))
    assert m2.get_num_optim_iter() == int(g[name + "_fit_num_it"])
    np.testing.assert_allclose(m2.get_cov_pars(), g[name + "_fit_cov_pars"], rtol=1e-6)
    nll = m2.get_current_neg_log_likelihood()
    assert abs(nll - float(g[name + "_fit_negll"])) <= 1e-8 * abs(nll)


def test_vecchia_response_with_many_neighbours_and_in_four_dimensions(gpb, orc):
    """The factor launch of the preconditioner goes through the same dispatch as the model's own factor: m > 62 neighbours and d > 3 coordinates take the generality kernel
    (vecchia_big_kernels.hip) with the per-point diagonal additions.  Against the oracle."""
    from gpboost_amd import shim
    rng = np.random.default_rng(11)
    for n, d, m, ctn in ((700, 2, 70, 1), (600, 4, 12, 0)):
        coords = rng.uniform(size=(n, d))
        y = (rng.uniform(size=n) < 1 / (1 + np.exp(-1.5 * np.sin(4 * coords[:, 0])))).astype(np.float64)
        perm, co, nn = orc.vecchia_setup(coords, m, "random", 3)
        st = shim.VecchiaState(co, m)
        st.set_neighbors(nn)
        st.laplace_set_likelihood("bernoulli_logit")
        st.laplace_set_labels(y[perm].astype(np.int32))
        st.laplace_set_preconditioner("vecchia_response")
        var, a = 0.9, RC[ctn] / 0.2
        v, info = st.laplace_logit(ctn, var, a, **cases.LAPLACE_TIGHT)
        with orc.vecchia_response_preconditioner(co, ctn, var, a):
            ov, oi = orc.vecchia_laplace_logit(co, nn, ctn, var, a, y[perm], **TIGHT_ORC)
        assert abs(v - ov) <= 1e-8 * abs(ov), (n, d, m, v, ov)
        assert info["newton_it"] == oi["newton_it"]
        st.close()
