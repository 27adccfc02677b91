"""GPU (MI355X): cg_preconditioner_type = "pivoted_cholesky" and "fitc" on the Vecchia-Laplace path (the second entry of the reference's
SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, re_model_template.h:5906) -- the solves in the form (W^-1 + Sigma) u' = Sigma rhs preconditioned with
P = W^-1 + L_k L_k^T (pivchol_kernels.hip, gpb_laplace.inc) -- through the C ABI against the UNMODIFIED reference (tests/golden/laplace_pivchol_ref.npz,
oracle/make_golden.py laplace_pivchol):
  * the factor L_k itself against the oracle's restatement of PivotedCholsekyFactorizationSigma;
  * value + gradient wrt (log sigma1^2, log a[, log aux]) at cases.LAPLACE_TIGHT from the reference's own CalcGradPars: 1e-8 relative, without and with fixed
    effects; the boosting gradient 1e-8 of its scale; the value at the reference's default thresholds;
  * the oracle step by step (d logdet / d mode, the implicit solve, the per-parameter parts);
  * fits through the model surface (GPB_SetOptimConfig(cg_preconditioner_type, piv_chol_rank)): the reference's iteration counts and estimates.
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
    rank = cases.pivchol_rank(pc)
    st.laplace_set_preconditioner(pc.get("pc", "pivoted_cholesky"), -999 if pc["rank"] is None else pc["rank"])
    if pc.get("pc") == "fitc":       # the inducing points are the host's part (kmeans++ from the model's generator): here the oracle's restatement of that draw
        st.laplace_set_inducing_points(orc.vif_setup(coords, c["m"], rank, c["ordering"], c["seed"])[3])
    return st, c, coords, y, perm, co, nn, ct, rank


def _orc_context(orc, pc, c, coords, co, ct, var, a, rank):
    if pc.get("pc") == "fitc":
        return orc.fitc_preconditioner(co, orc.vif_setup(coords, c["m"], rank, c["ordering"], c["seed"])[3], ct, var, a)
    return orc.pivoted_cholesky_preconditioner(co, ct, var, a, rank=rank)


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_PIVCHOL_CASES))
def test_value_and_gradient_match_the_reference(gpb, orc, name):
    pc = cases.LAPLACE_PIVCHOL_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_pivchol_ref.npz"))
    st, c, coords, y, perm, co, nn, ct, rank = _state(orc, pc)
    cp = c["cov_pars"][0]
    a = RC[ct] / cp[1]
    negll, info = st.laplace_logit(ct, cp[0], a)
    ref_d = float(g[name + "_negll_default"])
    assert abs(negll - ref_d) <= 1e-6 * abs(ref_d), (negll, ref_d)       # default thresholds: two correct implementations stop one CG iteration apart
    for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
        st.laplace_set_fixed_effects(fe)
        nll_t, grad_t = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == ref.shape
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-8 * abs(ref_v), (nll_t, ref_v)
    if name + "_gradF" in g.files:
        st.laplace_set_fixed_effects(None)
        st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
        out = np.empty(len(y)); out[perm] = st.laplace_grad_F()
        np.testing.assert_allclose(out, g[name + "_gradF"], rtol=0, atol=1e-8 * np.abs(g[name + "_gradF"]).max())
    # back to "vadu" on the same handle: the values of the vadu fixtures' oracle
    st.laplace_set_fixed_effects(None)
    st.laplace_set_preconditioner("vadu")
    nll_v, grad_v = st.laplace_eval_grad(ct, cp[0], a, **cases.LAPLACE_TIGHT)
    on, og = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), **TIGHT_ORC)
    assert abs(nll_v - on) <= 1e-8 * abs(on), (nll_v, on)
    np.testing.assert_allclose(grad_v, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
    st.close()


@pytest.mark.parametrize("name", ["pc_logit_n2000", "pc_poisson_n1500_r20", "fitc_logit_n1500_r100"])
def test_steps_of_the_gradient_match_the_oracle(gpb, orc, name):
    """d logdet / d mode with the row-wise control variate of the pivoted_cholesky branch, the implicit solve in the (W^-1 + Sigma) form, and per parameter
    {mode' SigmaI_deriv mode, d logdet / d theta, implicit part} against orc_vecchia_laplace_grad inside orc.pivoted_cholesky_preconditioner; at other
    covariance parameters than the fixture's (the factor L_k is renewed by every evaluation), with a warm start."""
    pc = cases.LAPLACE_PIVCHOL_CASES[name]
    st, c, coords, y, perm, co, nn, ct, rank = _state(orc, pc)
    for var, rho, reset in ((c["cov_pars"][0][0], c["cov_pars"][0][1], True), (0.6, 0.22, False)):
        a = RC[ct] / rho
        nll, grad, parts = st.laplace_eval_grad(ct, var, a, reset_mode=reset, want_parts=True, **cases.LAPLACE_TIGHT)
        with _orc_context(orc, pc, c, coords, co, ct, var, a, rank):
            on, og, op = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=pc["lik"], want_parts=True, **TIGHT_ORC)
        assert abs(nll - on) <= 1e-8 * abs(on), (nll, on)
        np.testing.assert_allclose(grad, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
        np.testing.assert_allclose(parts["dlogdet_dmode"], op["dlogdet_dmode"], rtol=0, atol=1e-8 * np.abs(op["dlogdet_dmode"]).max())
        np.testing.assert_allclose(parts["implicit_solve"], op["implicit_solve"], rtol=0, atol=1e-8 * np.abs(op["implicit_solve"]).max())
        for j in range(2):
            np.testing.assert_allclose(parts["per_par"][j, [0, 1, 3]], op["per_par"][j, [0, 1, 3]], rtol=1e-7, atol=1e-8 * np.abs(op["per_par"][j]).max())
    st.close()


def test_errors_of_the_preconditioner_entry_point(gpb, orc):
    pc = cases.LAPLACE_PIVCHOL_CASES["pc_logit_n2000"]
    st, c, coords, y, perm, co, nn, ct, rank = _state(orc, pc)
    from gpboost_amd import shim
    with pytest.raises(gpb.GPBoostError, match="cannot be larger"):
        st.laplace_set_preconditioner("pivoted_cholesky", len(y) + 1)
    with pytest.raises(gpb.GPBoostError, match="not on this path"):
        shim._shim_call(shim._lib().gpb_hip_vecchia_laplace_set_preconditioner(st.h, 7, 10))
    st.laplace_set_preconditioner("fitc", 100)
    with pytest.raises(gpb.GPBoostError, match="inducing points"):
        st.laplace_logit(ct, 1.0, 5.0)                 # fitc without its inducing points
    with pytest.raises(gpb.GPBoostError, match="less inducing points"):
        st.laplace_set_inducing_points(np.zeros((len(y), 2)))
    st.close()


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_PIVCHOL_CASES))
def test_model_api_evaluation_and_fit_follow_the_reference(gpb, name):
    """GPModel -> GPB_SetOptimConfig(cg_preconditioner_type = "pivoted_cholesky", piv_chol_rank) / GPB_EvalNegLogLikelihood / GPB_OptimCovPar /
    GPB_GetCGPreconditionerType: evaluation at the default thresholds, the lbfgs fit at cases.LAPLACE_TIGHT with the reference's iteration count, predictions."""
    pc = cases.LAPLACE_PIVCHOL_CASES[name]
    c = cases.LAPLACE_CASES[pc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_pivchol_ref.npz"))
    coords, y = cases.make_pivchol_data(pc)
    kw = dict(likelihood=pc["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
              num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    pcp = {"cg_preconditioner_type": pc.get("pc", "pivoted_cholesky")}
    if pc["rank"] is not None:
        pcp["fitc_piv_chol_preconditioner_rank"] = pc["rank"]
    cp = np.asarray(c["cov_pars"][0], dtype=np.float64)
    mdl = gpb.GPModel(**kw)
    mdl.set_optim_params(dict(pcp))
    assert mdl.get_cg_preconditioner_type() == pc.get("pc", "pivoted_cholesky")
    v = mdl.neg_log_likelihood(cp, y, aux_pars=[pc["aux"]]) if "aux" in pc else mdl.neg_log_likelihood(cp, y)
    ref_d = float(g[name + "_negll_default"])
    assert abs(v - ref_d) <= 1e-6 * abs(ref_d), (v, ref_d)
    m2 = gpb.GPModel(**kw)
    m2.fit(y, params=dict(pcp, **cases.LAPLACE_TIGHT))
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


def test_model_api_errors_for_the_preconditioner(gpb):
    pc = cases.LAPLACE_PIVCHOL_CASES["pc_logit_n2000"]
    c = cases.LAPLACE_CASES[pc["model"]]
    coords, y = cases.make_pivchol_data(pc)
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords[:300], cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    with pytest.raises(gpb.GPBoostError, match="not on the MI355X hot path"):
        mdl.set_optim_params({"cg_preconditioner_type": "incomplete_cholesky"})
    with pytest.raises(gpb.GPBoostError, match="is not > 0"):
        mdl.set_optim_params({"cg_preconditioner_type": "pivoted_cholesky", "fitc_piv_chol_preconditioner_rank": 0})
    with pytest.raises(gpb.GPBoostError, match="cannot be larger"):
        mdl.set_optim_params({"cg_preconditioner_type": "pivoted_cholesky", "fitc_piv_chol_preconditioner_rank": 301})
    mdl.set_optim_params({"cg_preconditioner_type": "piv_chol_on_Sigma", "fitc_piv_chol_preconditioner_rank": 25})     # ParsePreconditionerAlias
    assert mdl.get_cg_preconditioner_type() == "pivoted_cholesky"
    assert np.isfinite(mdl.neg_log_likelihood(np.array([1.0, 0.1]), y[:300]))
    m2 = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords[:300], cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    m2.set_optim_params({"cg_preconditioner_type": "FITC", "fitc_piv_chol_preconditioner_rank": 300})
    assert m2.get_cg_preconditioner_type() == "fitc"
    with pytest.raises(gpb.GPBoostError, match="less inducing points"):
        m2.neg_log_likelihood(np.array([1.0, 0.1]), y[:300])
    m2.set_optim_params({"fitc_piv_chol_preconditioner_rank": 40})
    assert np.isfinite(m2.neg_log_likelihood(np.array([1.0, 0.1]), y[:300]))


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_PC_EXTRA_CASES))
def test_model_api_low_rank_preconditioners_with_weights_and_repeated_locations(gpb, name):
    """pivoted_cholesky / fitc together with sample weights (the information is weighted) and with repeated locations (the information of a random effect is the sum over its
    data; L_k / the inducing points live on the unique locations, the kmeans++ draw on their Vecchia-ordered coordinates) through the model surface against the reference
    library (tests/golden/laplace_pc_extra_ref.npz, oracle/make_golden.py laplace_pc_extra) at cases.LAPLACE_TIGHT: evaluation 1e-8, the lbfgs fit with the reference's iterations."""
    ec = cases.LAPLACE_PC_EXTRA_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_pc_extra_ref.npz"))
    kw, y, cp, aux = cases.pc_extra_model(ec)
    pcp = dict(cases.LAPLACE_TIGHT, cg_preconditioner_type=ec["pc"], fitc_piv_chol_preconditioner_rank=ec["rank"])
    mdl = gpb.GPModel(**kw)
    mdl.set_optim_params(dict(pcp))
    v = mdl.neg_log_likelihood(cp, y, aux_pars=[aux]) if aux is not None else mdl.neg_log_likelihood(cp, y)
    ref = float(g[name + "_negll"])
    assert abs(v - ref) <= 1e-8 * abs(ref), (v, ref)
    m2 = gpb.GPModel(**kw)
    m2.fit(y, params=dict(pcp))
    assert m2.get_num_optim_iter() == int(g[name + "_fit_num_it"]), (m2.get_num_optim_iter(), int(g[name + "_fit_num_it"]))
    np.testing.assert_allclose(m2.get_cov_pars(), g[name + "_fit_cov_pars"], rtol=1e-5)
    if aux is not None:
        np.testing.assert_allclose(m2.get_aux_pars(), g[name + "_fit_aux"], rtol=1e-5)
    nll = m2.get_current_neg_log_likelihood()
    assert abs(nll - float(g[name + "_fit_negll"])) <= 1e-8 * abs(nll)


def test_pivoted_cholesky_that_meets_its_error_bound_before_its_rank(gpb, orc):
    """A smooth kernel (Matern-2.5, range 10 on the unit square): PivotedCholsekyFactorizationSigma stops after 27 of the 60 columns asked for (trace of the Schur
    complement below PIV_CHOL_STOP_TOL = 1e-6, CG_utils.h:456), the remaining columns of L_k stay zero and the k x t normals still have 60 rows.  The value equals the
    oracle's (itself 5e-14 from the reference on this case); the gradient only to 1e-3: the per-point systems have condition numbers ~1e10 here and d A / d log(range)
    is good to ~1e-5 in ANY implementation (the oracle and the reference are 5e-5 apart on this case with either preconditioner)."""
    from gpboost_amd import shim
    rng = np.random.default_rng(1)
    n = 400
    coords = rng.uniform(size=(n, 2))
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-1.5 * np.sin(4 * coords[:, 0])))).astype(np.float64)
    perm, co, nn = orc.vecchia_setup(coords, 15, "none", 0)
    st = shim.VecchiaState(co, 15)
    st.set_neighbors(nn)
    st.laplace_set_likelihood("bernoulli_logit")
    st.laplace_set_labels(y[perm].astype(np.int32))
    st.laplace_set_preconditioner("pivoted_cholesky", 60)
    var, a = 1.0, np.sqrt(5.0) / 10.0
    L, k = orc.pivoted_cholesky_factor(co, 2, var, a, rank=60)
    assert k < 60 and np.all(L[:, k:] == 0.0)
    nll, grad = st.laplace_eval_grad(2, var, a, **cases.LAPLACE_TIGHT)
    with orc.pivoted_cholesky_preconditioner(co, 2, var, a, rank=60):
        on, og = orc.vecchia_laplace_grad(co, nn, 2, var, a, y[perm], likelihood="bernoulli_logit", **TIGHT_ORC)
    assert abs(nll - on) <= 1e-8 * abs(on), (nll, on)
    np.testing.assert_allclose(grad, og, rtol=1e-3)
    st.close()


@pytest.mark.parametrize("name,t,rank", [("pc_logit_n2000", 4, 37), ("pc_logit_n2000", 20, 50), ("pc_logit_n2000", 36, 23), ("pc_logit_n2000", 48, 50),
                                         ("fitc_logit_n1500_r100", 20, 100), ("fitc_logit_n1500_r100", 48, 100)])
def test_block_kernels_of_the_low_rank_part_at_other_probe_counts_and_ranks(gpb, orc, name, t, rank):
    """The block forms of L'(W X) and X - L x2 run on v_mfma_f64_16x16x4_f64 (pivchol_kernels.hip: pc_ltwx_mfma_kernel / pc_combine_mfma_kernel<CT>) with CT = 1, 2 or 4 tiles of
    16 block-vector columns per launch row -- chosen by the number of probe chunks -- and 16-column tiles of L.  The fixtures run t = 50 probes (13 chunks: CT = 4, one launch row,
    last tile a quarter full); here 1, 5, 9 and 12 chunks (CT = 1, 2, 4) and ranks that are no multiple of 16 or 4, value and gradient against the oracle with the same probes.
    Open (scripts/gpu_probe_counts.py, profiles/r06_probe_counts_*): with MORE than 50 probes device and oracle part ways for the low-rank preconditioners
    (5e-5 at 52 probes -- the same 13 chunks as the pinned 50 -- while "vadu" agrees to 6e-13 at every count), so neither side is pinned there."""
    pc = dict(cases.LAPLACE_PIVCHOL_CASES[name], rank=rank)
    st, c, coords, y, perm, co, nn, ct, rk = _state(orc, pc)
    assert rk == rank
    var, rho = c["cov_pars"][0][0], c["cov_pars"][0][1]
    a = RC[ct] / rho
    nll, grad = st.laplace_eval_grad(ct, var, a, num_rand_vec=t, **cases.LAPLACE_TIGHT)
    with _orc_context(orc, pc, c, coords, co, ct, var, a, rank):
        on, og = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=pc["lik"], num_rand_vec=t, **TIGHT_ORC)
    assert abs(nll - on) <= 1e-8 * abs(on), (nll, on)
    np.testing.assert_allclose(grad, og, rtol=1e-8, atol=1e-8 * np.abs(og).max())
    st.close()
