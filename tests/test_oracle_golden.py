"""CPU: the oracle (oracle/gpb_oracle.c) against (i) the golden numbers hard-coded in the reference's R
test-suite and (ii) the fixtures produced by the compiled reference (oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# R-package/tests/testthat/test_GPModel_gaussian_process.R:86-120 (exact GP) and :1104-1148 (Vecchia)
R_GOLDEN_EXACT = [("exponential", 0.5, 124.2549533), ("matern", 1.5, 141.3502172), ("matern", 2.5, 158.1111626)]
R_COV_PARS = np.array([0.1, 1.6, 0.2])
R_TOL = 1e-6   # the goldens carry 7 decimals; the R tests use TOLERANCE_STRICT = 1e-5


def test_r_golden_exact_and_vecchia_full(orc):
    coords, y = orc.r_fixture()
    for cf, sh, gold in R_GOLDEN_EXACT:
        ct = orc.cov_type_id(cf, sh)
        pt = orc.transform_cov_pars(ct, R_COV_PARS)
        assert abs(orc.exact_nll(coords, ct, pt, y)[2] - gold) < R_TOL
        # Vecchia with m = n-1 equals the exact GP (:1104-1111)
        assert abs(orc.gp_nll(coords, y, R_COV_PARS, cf, sh, m=99, ordering="none") - gold) < R_TOL


def test_r_golden_vecchia_m30(orc):
    coords, y = orc.r_fixture()
    nll = orc.gp_nll(coords, y, R_COV_PARS, "exponential", 0.5, m=30, ordering="none")
    assert abs(nll - 124.2252524) < R_TOL      # :1144-1148


@pytest.mark.parametrize("name", sorted(cases.GOLDEN_CASES))
def test_oracle_matches_reference_fixture(orc, name):
    c = cases.GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    coords, y = cases.make_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    assert np.array_equal(perm, g["perm"]), "Vecchia ordering differs from the reference"
    assert np.array_equal(nn, g["nn"]), "neighbour table differs from the reference (bit-exact requirement)"
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    for k, cp in enumerate(c["cov_pars"]):
        pt = orc.transform_cov_pars(ct, np.asarray(cp, dtype=np.float64))
        np.testing.assert_allclose(pt, g["pars_trans_%d" % k], rtol=1e-15)
        out, grad = orc.vecchia_nll_grad(co, nn, ct, pt, y[perm])
        assert abs(out[2] - g["nll_%d" % k]) <= 1e-10 * abs(g["nll_%d" % k])
        np.testing.assert_allclose(grad, g["grad_%d" % k], rtol=1e-8, atol=1e-9)
        A, D, bad = orc.vecchia_factor(co, nn, ct, pt[1], pt[2])
        assert bad == 0
        np.testing.assert_allclose(D, g["D_%d" % k], rtol=1e-10)
        rows = g["A_rows_%d" % k] if ("A_rows_%d" % k) in g else np.arange(len(D))
        np.testing.assert_allclose(A[rows], g["A_%d" % k], rtol=0, atol=1e-10)
        np.testing.assert_allclose(orc.vecchia_yaux(A, D, nn, y[perm]), g["yaux_%d" % k], rtol=1e-9, atol=1e-10)


def test_oracle_gradient_is_derivative_of_nll(orc):
    coords, y = cases.synthetic(400, 2, seed=3)
    perm, co, nn = orc.vecchia_setup(coords, 15, "random", 1)
    for ct in (0, 1, 2):
        pt = orc.transform_cov_pars(ct, np.array([0.2, 1.1, 0.15]))
        _, g = orc.vecchia_nll_grad(co, nn, ct, pt, y[perm])
        eps = 1e-6
        for k in range(3):
            pp = pt.copy(); pp[k] *= np.exp(eps)
            pm = pt.copy(); pm[k] *= np.exp(-eps)
            fd = (orc.vecchia_nll(co, nn, ct, pp, y[perm])[2] - orc.vecchia_nll(co, nn, ct, pm, y[perm])[2]) / (2 * eps)
            assert abs(fd - g[k]) < 1e-5 * max(1., abs(g[k]))


def test_oracle_histogram_against_numpy(orc):
    rng = np.random.default_rng(0)
    n, F = 5000, 7
    nb = np.array([255, 16, 64, 256, 3, 100, 200])
    bo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    bins = np.stack([rng.integers(0, nb[f], size=n) for f in range(F)]).astype(np.uint8)
    grad = rng.standard_normal(n); hess = rng.uniform(0.5, 2., size=n)
    idx = np.sort(rng.choice(n, size=1700, replace=False)).astype(np.int32)
    for di in (None, idx):
        rows = np.arange(n) if di is None else di
        hg, hc, hh = orc.hist_build(bins, bo, di, grad, hess)
        hg2, hc2, hh2 = orc.hist_build(bins, bo, di, grad, None, const_hess=1.0)
        for f in range(F):
            cnt = np.bincount(bins[f, rows], minlength=nb[f])
            assert np.array_equal(hc[bo[f]:bo[f + 1]], cnt.astype(np.uint64))
            assert np.array_equal(hc2[bo[f]:bo[f + 1]], cnt.astype(np.uint64))
            np.testing.assert_allclose(hg[bo[f]:bo[f + 1]], np.bincount(bins[f, rows], weights=grad[rows], minlength=nb[f]), atol=1e-10)
            np.testing.assert_allclose(hh[bo[f]:bo[f + 1]], np.bincount(bins[f, rows], weights=hess[rows], minlength=nb[f]), atol=1e-10)
            np.testing.assert_array_equal(hh2[bo[f]:bo[f + 1]], cnt.astype(np.float64))


def test_oracle_histogram_matches_reference_fixture(orc):
    """tests/golden/hist_ref.npz = the reference's own binning + Dataset::ConstructHistograms (dataset.cpp:1143-1245) through
    oracle/ref_driver.cpp:refdrv_hist; the oracle accumulates in the same per-feature row order, so even the fp64 sums are
    bit-identical."""
    g = np.load(os.path.join(GOLD, "hist_ref.npz"))
    X, grad, hess, leaf = cases.make_hist_data()
    bins, gnb = g["bins"], g["group_num_bin"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    for li, di in enumerate((None, leaf)):
        for hi, hs in enumerate((None, hess)):
            ref = g["hist_leaf%d_hess%d" % (li, hi)]
            hg, hc, hh = orc.hist_build(bins, bo, di, grad, hs, 1.0)
            assert np.array_equal(hg, ref[:, 0])
            assert np.array_equal(hh, ref[:, 1])
            if hs is None:
                assert np.array_equal(hc.astype(np.float64), ref[:, 1])     # count * 1.0 (dataset.cpp:1223-1226)


# ---- Vecchia-Laplace, Bernoulli-logit (BASELINE config 4) ------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_laplace_probit_matches_reference_fixture(orc, name):
    """Same stack with the Bernoulli-probit pieces (normalLogCDF, inverse Mills ratios) against the reference's
    GPB_EvalNegLogLikelihood(likelihood='bernoulli_probit')."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_ref.npz"))
    coords, y = cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    for k, cp in enumerate(c["cov_pars"]):
        a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
        negll, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood="bernoulli_probit")
        ref = float(g["%s_probit_negll_%d" % (name, k)])
        assert info["rc"] == 0
        assert abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref, info["newton_it"], info["cg_it"], info["lanczos_it"])


@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit"])
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_laplace_with_fixed_effects_matches_reference_fixture(orc, name, lik):
    """Location parameter = mode + fixed effects (likelihoods.h:3861-3870): how the GPBoost algorithm passes the ensemble's scores."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_ref.npz"))
    coords, y = cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    fe = cases.laplace_fixed_effects(coords)
    negll, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=lik, fixed_effects=fe[perm])
    ref = float(g["%s_fe_%snegll_0" % (name, "probit_" if lik == "bernoulli_probit" else "")])
    assert info["rc"] == 0 and abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref)


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_laplace_poisson_matches_reference_fixture(orc, name):
    """Poisson likelihood (log link; the normalising constant -sum log(y!) is part of every LogLikelihood value) against the
    reference's GPB_EvalNegLogLikelihood(likelihood='poisson'), without and with fixed effects."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_ref.npz"))
    coords, y = cases.make_count_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    for k, cp in enumerate(c["cov_pars"]):
        a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
        negll, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood="poisson")
        ref = float(g["%s_poisson_negll_%d" % (name, k)])
        assert info["rc"] == 0 and abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref, info["newton_it"], info["cg_it"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    fe = cases.laplace_fixed_effects(coords)
    negll, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood="poisson", fixed_effects=fe[perm])
    ref = float(g["%s_fe_poisson_negll_0" % name])
    assert abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref)


@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit", "poisson"])
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_laplace_gradient_matches_reference(orc, name, lik):
    """orc_vecchia_laplace_grad (stochastic trace estimators with the vadu control variates, implicit derivative through the mode)
    against the gradient the reference's own optimiser used for one plain gradient-descent step (tests/golden/laplace_grad_ref.npz,
    oracle/refdrv.py: ref_laplace_gradient).  The device path is checked against the same fixture (tests/test_z_laplace_grad_gpu.py).  The extraction is exact up to log / exp
    rounding over a step of 5e-4 (~1e-8); the gradient itself contains a CG solve that stops at |r| < 1e-2, whose iteration count can
    shift with rounding (seen: 2e-6 relative on one case), hence 1e-5."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_grad_ref.npz"))
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    negll, grad = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=lik)
    np.testing.assert_allclose(grad, g["%s_%s_grad" % (name, lik)], rtol=1e-5, atol=1e-5)
    # THE PIN (round 5): the reference's own CalcGradPars -> CalcGradNegMargLikelihoodLaplaceApproxVecchia (likelihoods.h:6521-6700) through
    # oracle/ref_driver.cpp: refdrv_laplace_nll_grad at cg_delta_conv 1e-8 / delta_conv_mode_finding 1e-13 (cases.LAPLACE_TIGHT), where no stopping rule
    # is left in the comparison: north_star's 1e-8 relative (observed <= 1e-9), value and gradient, without and with fixed effects
    for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
        negll_t, grad_t = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=lik, fixed_effects=fe,
                                                   cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
        ref = g["%s_%s%s_grad_direct" % (name, lik, fe_key)]
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
        ref_v = float(g["%s_%s%s_negll_direct" % (name, lik, fe_key)])
        assert abs(negll_t - ref_v) <= 1e-10 * abs(ref_v), (negll_t, ref_v)


def test_r_probit_fixture_oracle_vs_reference(orc):
    """R suite probit data, Vecchia on all predecessors (m = n - 1), iterative methods: oracle == reference (1e-8); both within the
    stochastic log-determinant's accuracy of the exact-GP golden 67.18342059 (test_GPModel_non_Gaussian_data.R:1405, :1426)."""
    g = np.load(os.path.join(GOLD, "laplace_ref.npz"))
    coords, y = orc.r_fixture_probit()
    perm, co, nn = orc.vecchia_setup(coords, 99, "none", 0)
    negll, info = orc.vecchia_laplace_logit(co, nn, 0, 1.0, 1.0 / 0.2, y, likelihood="bernoulli_probit")
    ref = float(g["r_probit_m99_negll"])
    assert abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref)
    assert abs(ref - 67.18342059) < 0.3


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_laplace_matches_reference_fixture(orc, name):
    """orc_vecchia_laplace_logit (Newton + vadu-CG + SLQ with the reference's probe vectors) against the reference's
    GPB_EvalNegLogLikelihood(likelihood='bernoulli_logit').  The CG stopping rules make the value a discontinuous function
    of rounding only at the 1e-2 residual threshold; agreement is ~1e-11 in practice, 1e-8 is the north_star bar."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_ref.npz"))
    coords, y = cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    for k, cp in enumerate(c["cov_pars"]):
        a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
        negll, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm])
        ref = float(g["%s_negll_%d" % (name, k)])
        assert info["rc"] == 0
        assert abs(negll - ref) <= 1e-8 * abs(ref), (negll, ref, info["newton_it"], info["cg_it"], info["lanczos_it"])


def test_oracle_probe_vectors_are_reproducible(orc):
    rv = orc.gen_rand_normal(1000, 3, seed=1, run_id=0)
    assert rv.shape == (1000, 3) and abs(rv.mean()) < 0.1 and abs(rv.std() - 1) < 0.1
    assert np.array_equal(rv, orc.gen_rand_normal(1000, 3, seed=1, run_id=0))
    assert not np.array_equal(rv[:, 0], rv[:, 1])


def test_r_golden_logit_mode_finding(orc):
    """test_GPModel_non_Gaussian_data.R:2537-2538: nll 66.299571 of the exact GP with a Bernoulli-logit likelihood at
    cov_pars (0.9, 0.2).  The reference value uses Cholesky; a Vecchia approximation conditioning on ALL predecessors
    (m = n - 1, no re-ordering) is that same GP, so the oracle's Newton / CG mode finding and objective are pinned by it once
    the log-determinant at the oracle's mode is computed densely (the SLQ estimate itself is stochastic: the reference's own
    tests allow 1e-1 for it, TOLERANCE_ITERATIVE) and the CG tolerance is tightened from the default 1e-2."""
    from scipy.spatial.distance import cdist
    coords, y = orc.r_fixture_logit()
    n = len(y)
    perm, co, nn = orc.vecchia_setup(coords, n - 1, "none", 0)
    negll, info = orc.vecchia_laplace_logit(co, nn, 0, 0.9, 1.0 / 0.2, y, cg_delta_conv=1e-8, delta_conv_mode=1e-12)
    p = 1.0 / (1.0 + np.exp(-info["mode"]))
    sw = np.sqrt(p * (1.0 - p))
    S = 0.9 * np.exp(-cdist(coords, coords) / 0.2)
    logdet = np.linalg.slogdet(np.eye(n) + sw[:, None] * S * sw[None, :])[1]
    assert abs(-(info["mll_no_det"] - 0.5 * logdet) - 66.299571) < R_TOL
    assert abs(info["log_det"] - logdet) < 0.5        # SLQ with 50 probes at n = 100
    # defaults (cg_delta_conv = 1e-2): the mode is only that sharp, the value moves in the 5th decimal
    negll_d, info_d = orc.vecchia_laplace_logit(co, nn, 0, 0.9, 1.0 / 0.2, y)
    assert abs(-(info_d["mll_no_det"] - 0.5 * logdet) - 66.299571) < 1e-3


# ---- Newton update of the leaf values (row a9) ---------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LEAF_CASES))
def test_oracle_newton_leaf_values_match_reference(orc, name):
    """orc_newton_leaf_values against the reference's own REModelTemplate::NewtonUpdateLeafValues (Vecchia branch)."""
    c = cases.GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    coords, y = cases.make_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    pt = orc.transform_cov_pars(ct, np.asarray(c["cov_pars"][0], dtype=np.float64))
    A, D, bad = orc.vecchia_factor(co, nn, ct, pt[1], pt[2])
    yaux = orc.vecchia_yaux(A, D, nn, y[perm])
    leaf, L = cases.make_leaf_index(name, len(y))
    vals = orc.newton_leaf_values(A, D, nn, yaux, leaf[perm], L)
    np.testing.assert_allclose(vals, g["leaf_values_0"], rtol=1e-9, atol=1e-11)


# ---- FixHistogram / histogram subtraction (row a12) -------------------------------------------------------------
def test_oracle_fix_histogram_matches_reference_fixture(orc):
    """orc_hist_fix against the reference's own Dataset::FixHistogram on its own histograms (bit-identical: same subtraction
    order), for features whose most frequent bin is > 0 (two of the six in the fixture); subtraction is exact by construction."""
    g = np.load(os.path.join(GOLD, "hist_ref.npz"))
    assert (g["fix_most_freq_bin"] > 0).sum() >= 2
    for li in (0, 1):
        for hi in (0, 1):
            key = "leaf%d_hess%d" % (li, hi)
            sums = g["fix_sums_" + key]
            fixed = orc.hist_fix(g["hist_" + key], g["fix_view_offset"], g["fix_num_bin"], g["fix_most_freq_bin"], sums[0], sums[1])
            assert np.array_equal(fixed, g["hist_fixed_" + key])
            assert not np.array_equal(fixed, g["hist_" + key])
    parent, small = g["hist_fixed_leaf0_hess1"], g["hist_fixed_leaf1_hess1"]
    assert np.array_equal(orc.hist_subtract(parent, small), parent - small)


# ---- Vecchia prediction, prediction points conditioning on observed points only ----------------------
R_PRED_COV_PARS = np.array([0.03297349, 1.07691542, 0.11378505])       # the fitted parameters of :1319-1321
R_PRED_COORDS = np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]])  # :1326
R_PRED_MU = np.array([0.06968068, 0.06967750, 0.44208925])               # :1329
R_PRED_VAR = np.array([0.6214955, 0.6215069, 0.4199531])                 # diagonal of :1330-1331 (response scale)


def test_r_golden_vecchia_prediction(orc):
    """test_GPModel_gaussian_process.R:1326-1333: vecchia_pred_type = 'order_obs_first_cond_obs_only', num_neighbors_pred = 30."""
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, R_PRED_COV_PARS)
    mu, var = orc.predict_obs_only(coords, y, R_PRED_COORDS, 0, pt, 30, predict_response=True)
    assert np.abs(mu - R_PRED_MU).sum() < R_TOL
    assert np.abs(var - R_PRED_VAR).sum() < R_TOL
    mu2, var_latent = orc.predict_obs_only(coords, y, R_PRED_COORDS, 0, pt, 30, predict_response=False)
    np.testing.assert_allclose(var - var_latent, R_PRED_COV_PARS[0], rtol=1e-12)


def test_r_golden_prediction_conditioning_on_all_observations(orc):
    """test_GPModel_gaussian_process.R:1241-1268: num_neighbors = n - 1, num_neighbors_pred = n + 2 -- every prediction point conditions on
    every observation, so means and variances are the exact GP's whichever prediction type is used (the suite uses
    'order_obs_first_cond_all'; its off-diagonal covariances, ~1e-5, are what 'cond_obs_only' does not model)."""
    coords, y = orc.r_fixture()
    cov_pars = np.array([0.02, 1.2, 0.9])
    pt = orc.transform_cov_pars(0, cov_pars)
    ct = np.array([[0.1, 0.9], [0.2, 0.4], [0.7, 0.55]])
    mu, var = orc.predict_obs_only(coords, y, ct, 0, pt, len(y) + 2, predict_response=True)
    assert np.abs(mu - [0.08704577, 1.63875604, 0.48513581]).sum() < R_TOL
    assert np.abs(var - [1.189093e-01, 7.427727e-02, 8.107455e-02]).sum() < R_TOL
    mu2, var_latent = orc.predict_obs_only(coords, y, ct, 0, pt, len(y) + 2, predict_response=False)
    assert np.abs(var_latent - (np.array([1.189093e-01, 7.427727e-02, 8.107455e-02]) - cov_pars[0])).sum() < R_TOL


def test_r_golden_predictions_at_given_parameters(orc):
    """test_GPModel_gaussian_process.R:1459-1495 (30 neighbours, ordering none, cov_pars (0.02, 1.2, 0.9), two of the three prediction
    points 1.4e-5 apart): 'order_obs_first_cond_obs_only' and 'order_obs_first_cond_all' (the latter correlates the two close points:
    covariance 0.09889262, and moves the second mean)."""
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, np.array([0.02, 1.2, 0.9]))
    mu, var = orc.predict_obs_only(coords, y, R_PRED_COORDS, 0, pt, 30, predict_response=True)
    assert np.abs(mu - [0.08665472, 0.08664854, 0.49011216]).sum() < R_TOL
    assert np.abs(var - [0.11891, 0.1189129, 0.08108126]).sum() < R_TOL
    mu, cov = orc.predict_cond_all(coords, y, R_PRED_COORDS, 0, pt, 30, predict_response=True)
    assert np.abs(mu - [0.08665472, 0.08661259, 0.49011216]).sum() < R_TOL
    assert np.abs(cov.ravel() - [0.11891004, 0.09889262, 0., 0.09889262, 0.11891291, 0., 0., 0., 0.08108126]).sum() < R_TOL
    mu2, cov_latent = orc.predict_cond_all(coords, y, R_PRED_COORDS, 0, pt, 30, predict_response=False)
    np.testing.assert_allclose(np.diag(cov) - np.diag(cov_latent), 0.02, rtol=1e-10)


def test_r_golden_cond_all_prediction_on_all_observations(orc):
    """test_GPModel_gaussian_process.R:1241-1258: num_neighbors_pred = n + 2 with 'order_obs_first_cond_all' = the exact GP's joint
    predictive distribution, off-diagonal covariances (1e-5 .. 1e-7) included."""
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, np.array([0.02, 1.2, 0.9]))
    ct = np.array([[0.1, 0.9], [0.2, 0.4], [0.7, 0.55]])
    mu, cov = orc.predict_cond_all(coords, y, ct, 0, pt, len(y) + 2, predict_response=True)
    assert np.abs(mu - [0.08704577, 1.63875604, 0.48513581]).sum() < R_TOL
    assert np.abs(cov.ravel() - [1.189093e-01, 1.171632e-05, -4.172444e-07, 1.171632e-05, 7.427727e-02, 1.492859e-06, -4.172444e-07,
                                 1.492859e-06, 8.107455e-02]).sum() < R_TOL


# ---- several clusters (independent realisations of the GP) ---------------------------------------------------------------------
def test_r_golden_vecchia_cluster_ids(orc):
    """test_GPModel_gaussian_process.R:1638-1648: cluster_ids = 40 x 1, 60 x 2; nll 129.3761486 at the fitted parameters
    (stationary point of the fit: the 8-digit rounding of the parameters does not move the value at that precision)."""
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, np.array([0.05870373, 1.05572659, 0.12775754]))
    ids = np.r_[np.ones(40), 2 * np.ones(60)]
    quad = logdet = 0.0
    for c in (1, 2):
        sel = ids == c
        perm, co, nn = orc.vecchia_setup(coords[sel], 30, "none", 0)
        out = orc.vecchia_nll(co, nn, 0, pt, y[sel][perm])
        quad += out[0]; logdet += out[1]
    nll = quad / 2 / pt[0] + logdet / 2 + len(y) / 2 * (np.log(pt[0]) + np.log(2 * np.pi))
    assert abs(nll - 129.3761486) < R_TOL


def test_r_golden_vecchia_cluster_ids_prediction(orc):
    """test_GPModel_gaussian_process.R:1660-1672: prediction with cluster_ids_pred = (1, 3, 1), 'order_obs_first_cond_all', num_neighbors_pred = 30,
    cov_pars (0.1, 1, 0.15), predict_response (GPModel.predict's default): the two points of cluster 1 condition on the 40 observations of that
    cluster and on each other; cluster 3 has no observations -- prior mean 0 and variance sigma2 + sigma1_2 = 1.1, no covariance with the others
    (REModelTemplate::Predict, re_model_template.h:3750-3936)."""
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, np.array([0.1, 1.0, 0.15]))
    ids = np.r_[np.ones(40), 2 * np.ones(60)]
    ct = np.array([[0.1, 0.9], [0.2, 0.4], [0.1001, 0.9001]])
    ids_pred = np.array([1, 3, 1])
    mu = np.zeros(3); cov = np.zeros((3, 3))
    for c in np.unique(ids_pred):
        ip = np.where(ids_pred == c)[0]
        sel = ids == c
        if not sel.any():
            cov[np.ix_(ip, ip)] = pt[0] * (pt[1] + 1.0) * np.eye(len(ip))       # one point: the prior variance of the response
            continue
        perm, co, nn = orc.vecchia_setup(coords[sel], 30, "none", 0)
        m_c, c_c = orc.predict_cond_all(co, y[sel][perm], ct[ip], 0, pt, 30, predict_response=True)
        mu[ip] = m_c; cov[np.ix_(ip, ip)] = c_c
    assert np.abs(mu - [-0.01438585, 0.0, -0.01500132]).sum() < R_TOL
    assert np.abs(cov.ravel() - [0.7430552, 0.0, 0.6423148, 0.0, 1.1, 0.0, 0.6423148, 0.0, 0.7434589]).sum() < R_TOL


@pytest.mark.parametrize("name", sorted(cases.SPLIT_DATA_UNIT))
def test_oracle_split_search_regularisation_paths_match_reference_fixture(orc, name):
    """The same with lambda_l1 / max_delta_step / path_smooth (and a given parent_output): the reference's USE_L1 / USE_MAX_OUTPUT /
    USE_SMOOTHING instances of FindBestThresholdSequentially (feature_histogram.hpp:137-161), every field of SplitInfo bit-identical."""
    g = np.load(os.path.join(GOLD, "split_ref.npz"))
    meta3 = g[name + "_meta3"]
    n_all = g[name + "_bins"].shape[1]
    checked = 0
    for ci, cfg in enumerate(cases.SPLIT_CFGS_REG):
        for li in (0, 1):
            for hi in (0, 1):
                key = "%s_cfgr%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums = g[key + "_sums"]
                num_data = n_all if li == 0 else 2500
                best, out, dl = orc.find_best_split(g[key + "_hist_fixed"], g[name + "_view_offset"], g[name + "_num_bin"], meta3[:, 0],
                                                    meta3[:, 1], meta3[:, 2], sums[0], sums[1], num_data, *cfg)
                ref = g[key + "_split"]
                assert np.array_equal(out, ref), (key, np.abs(out - ref).max())
                assert np.array_equal(dl, g[key + "_default_left"])
                checked += int(np.isfinite(ref[:, 0]).sum())
    assert checked >= 40


# ---- split search on a leaf histogram ----------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.SPLIT_DATA_UNIT))
def test_oracle_split_search_matches_reference_fixture(orc, name):
    """orc_find_best_split against the reference's own FeatureHistogram::FindBestThreshold (all three missing-value types, two
    regularisation settings, root and leaf, constant and per-row hessians): every field of SplitInfo bit-identical."""
    g = np.load(os.path.join(GOLD, "split_ref.npz"))
    meta3 = g[name + "_meta3"]
    n_all = g[name + "_bins"].shape[1]
    checked = 0
    for ci, cfg in enumerate(cases.SPLIT_CFGS):
        for li in (0, 1):
            for hi in (0, 1):
                key = "%s_cfg%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums = g[key + "_sums"]
                num_data = n_all if li == 0 else 2500
                best, out, dl = orc.find_best_split(g[key + "_hist_fixed"], g[name + "_view_offset"], g[name + "_num_bin"], meta3[:, 0],
                                                    meta3[:, 1], meta3[:, 2], sums[0], sums[1], num_data, *cfg)
                ref = g[key + "_split"]
                assert np.array_equal(out, ref), (key, np.abs(out - ref).max())
                assert np.array_equal(dl, g[key + "_default_left"])
                gains = ref[:, 0]
                assert best == int(np.argmax(gains)) and np.isfinite(gains.max())
                checked += int(np.isfinite(gains).sum())
    assert checked >= 30        # most features are splittable in every setting


@pytest.mark.parametrize("name", sorted(cases.SPLIT_DATA_UNIT))
def test_oracle_leaf_partition_matches_reference_fixture(orc, name):
    """orc_split_leaf against the reference's own Dataset::Split (DenseBin::SplitInner, all missing-value variants): the lists of
    rows going left / right are identical, order included."""
    g = np.load(os.path.join(GOLD, "split_ref.npz"))
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3, mfb = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"], g[name + "_most_freq_bin"]
    req, cnts, flat = g[name + "_part_req"], g[name + "_part_lte_count"], g[name + "_part_lte"]
    pos = 0
    sides = set()
    for (f, th, dl), nl in zip(req, cnts):
        lte, gt = orc.split_leaf(bins[f], gnb[f] - 1, meta3[f, 1], mfb[f], meta3[f, 2], dl, th, leaf)
        assert np.array_equal(lte, flat[pos:pos + nl]), (name, f, th, dl)
        assert np.array_equal(np.sort(np.concatenate([lte, gt])), leaf)
        assert np.all(np.diff(gt) > 0) and np.all(np.diff(lte) > 0)           # both sides keep the (sorted) input order
        sides.add((len(lte) > 0, len(gt) > 0))
        pos += nl
    assert pos == flat.size and (True, True) in sides


# ---- a whole tree: the hot-path primitives composed the way SerialTreeLearner::Train composes them -------------------------------
@pytest.mark.parametrize("name", sorted(cases.TREE_CASES))
@pytest.mark.parametrize("hi", [0, 1])
def test_oracle_primitives_grow_the_reference_tree(orc, name, hi):
    """Leaf histogram + FixHistogram + subtraction + split search + partition (all oracle restatements), driven by the control flow of
    SerialTreeLearner::Train (tests/tree_harness.py), reproduce the tree the reference's own SerialTreeLearner grows on its own Dataset
    (tests/golden/tree_ref.npz): structure, thresholds, default directions, counts exactly; leaf values and gains bit for bit."""
    from tests import tree_harness as th
    r5 = name in cases.TREE_CASES_R5       # round 5: categorical columns / bundled groups (bins = the unbundled per-feature columns), tree_ref_r5.npz
    g = np.load(os.path.join(GOLD, "tree_ref_r5.npz" if r5 else "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    k = "%s_hess%d_" % (name, hi)
    hs = hess if hi else None
    be = th.OracleBackend(orc, g[k + "bins"], g[k + "group_num_bin"], g[k + "view_offset"], g[k + "num_bin"], g[k + "most_freq_bin"],
                          g[k + "meta3"], grad, hs, is_cat=g[k + "layout"][:, 3] if r5 else None, cat_cfg=cases.tree_cat_cfg(name))
    t = th.grow_tree(be, grad, hs, X.shape[0], L, cfg, max_depth=cases.tree_max_depth(name))
    assert t["num_leaves"] == int(g[k + "num_leaves"])
    for key in ("split_feature_inner", "threshold_in_bin", "default_left", "left_child", "right_child", "internal_count", "leaf_count"):
        assert np.array_equal(t[key], g[k + key]), key
    assert np.array_equal(t["leaf_value"], g[k + "leaf_value"])
    assert np.array_equal(t["split_gain"], g[k + "split_gain"])
    if r5:       # categorical nodes: the same nodes, the same sets of bins going left (Tree::SplitCategorical's cat_threshold_inner_)
        assert np.array_equal(t["node_is_cat"], g[k + "node_is_cat"])
        assert np.array_equal(np.asarray(t["node_cat_bits"]).reshape(-1, 8), g[k + "node_cat_bits"])
        if data == "cat":
            assert int(g[k + "node_is_cat"].sum()) >= 5
        if data == "efb":
            assert len(set(t["split_feature_inner"].tolist()) & set(range(2, 10))) >= 2        # splits on bundled columns


def test_oracle_categorical_split_search_matches_reference_fixture(orc):
    """orc_find_best_split_cat against the reference's own FeatureHistogram::FindBestThreshold for categorical features
    (FindBestThresholdCategoricalInner: one-hot and sorted many-vs-many, L1 / max_delta_step / path smoothing, cat_smooth / cat_l2 /
    min_data_per_group / max_cat_threshold varied; root and leaf, constant and per-row hessians): every field of SplitInfo and the set of bins going
    left bit-identical (tests/golden/split_cat_ref.npz, oracle/make_golden.py split_cat)."""
    g = np.load(os.path.join(GOLD, "split_cat_ref.npz"))
    name = "cat"
    meta3, is_cat = g[name + "_meta3"], g[name + "_is_categorical"]
    assert is_cat.sum() == 2
    n_all = g[name + "_bins"].shape[1]
    splittable = onehot = sorted_sets = 0
    for ci, (cfg, cc) in enumerate(cases.SPLIT_CAT_CFGS):
        for li in (0, 1):
            for hi in (0, 1):
                key = "%s_cfg%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums = g[key + "_sums"]
                num_data = n_all if li == 0 else 2500
                for f in np.flatnonzero(is_cat):
                    row, fl, bits = orc.find_best_split_cat(g[key + "_hist_fixed"], g[name + "_view_offset"][f], g[name + "_num_bin"][f], meta3[f, 0],
                                                            sums[0], sums[1], num_data, *cfg, cat_cfg=cc)
                    ref = g[key + "_split"][f]
                    assert np.array_equal(row, ref), (key, f, row, ref)
                    assert (fl & 1) == int(g[key + "_default_left"][f])
                    assert np.array_equal(bits, g[key + "_cat_bits"][f]), (key, f)
                    if np.isfinite(ref[0]):
                        splittable += 1
                        assert int(ref[1]) == sum(bin(int(w)).count("1") for w in bits)
                        onehot += int(g[name + "_num_bin"][f] <= cc[0]); sorted_sets += int(ref[1] > 1)
                # the numerical columns of the same histograms through the threshold scans
                best, out, dl = orc.find_best_split(g[key + "_hist_fixed"], g[name + "_view_offset"], g[name + "_num_bin"], meta3[:, 0], meta3[:, 1],
                                                    meta3[:, 2], sums[0], sums[1], num_data, *cfg)
                num = np.flatnonzero(is_cat == 0)
                assert np.array_equal(out[num], g[key + "_split"][num]) and np.array_equal(dl[num], g[key + "_default_left"][num])
    assert splittable >= 30 and onehot >= 8 and sorted_sets >= 8


def test_oracle_categorical_partition_matches_reference_fixture(orc):
    """orc_split_leaf_layout (categorical) against the reference's Dataset::Split with a bitset over bins (DenseBin::SplitCategoricalInner): the rows going
    left, order included."""
    g = np.load(os.path.join(GOLD, "split_cat_ref.npz"))
    name = "cat"
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3, mfb = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"], g[name + "_most_freq_bin"]
    pos = 0
    two_sided = 0
    for (f, th, dl), w, nl in zip(g[name + "_part_req"], g[name + "_part_bits"], g[name + "_part_lte_count"]):
        lte, gt = orc.split_leaf_layout(bins[f], 1, gnb[f] - 1, False, meta3[f, 1], mfb[f], meta3[f, 2], dl, th, True, w, leaf)
        assert np.array_equal(lte, g[name + "_part_lte"][pos:pos + nl]), (f, w)
        assert np.array_equal(np.sort(np.concatenate([lte, gt])), leaf)
        two_sided += int(len(lte) > 0 and len(gt) > 0)
        pos += nl
    assert pos == g[name + "_part_lte"].size and two_sided >= 6


# ---- standard errors (Fisher information) -- checker only, no device path yet -----------------------------
@pytest.mark.parametrize("name", ["r_gd_nesterov_parcrit", "r_mat15_lbfgs", "u1d_n1000_mat15_lbfgs"])
def test_oracle_standard_errors_match_the_reference(orc, name):
    """orc.fisher_std_errors (Hutchinson estimate of the Fisher information on the original scale with the reference's probe vectors)
    against GPB_GetCovPar(calc_std_dev = true) after the reference's own fit (tests/golden/fisher_ref.npz, oracle/make_golden.py fisher).
    The R suite's own standard errors (test_GPModel_gaussian_process.R:1320) use 1000 probes of a later run id and are pinned there to 1e-2."""
    g = np.load(os.path.join(GOLD, "fisher_ref.npz"))
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    ct = orc.cov_type_id(mc["cov_function"], mc["shape"])
    se = orc.fisher_std_errors(co, nn, ct, g[name + "_cov_pars"])
    np.testing.assert_allclose(se, g[name + "_std"], rtol=1e-7)
    if name == "r_gd_nesterov_parcrit":
        assert np.abs(se - [0.07545639, 0.24785457, 0.03493878]).sum() < 1e-2        # the R suite's tolerance for these


@pytest.mark.parametrize("lik", ["bernoulli_logit", "poisson"])
@pytest.mark.parametrize("n,d,m,ct", [(3000, 2, 30, 0), (2000, 2, 10, 1)])
def test_stage_tolerances_admit_one_block_cg_iteration(orc, n, d, m, ct, lik):
    """The stage-by-stage GPU test (tests/test_z_laplace_grad_gpu.py) compares device and oracle with tolerances that must admit a block CG
    that stops one iteration earlier (the stopping test is a rounded norm against 1e-2): here the oracle with its iteration count capped at
    k - 1 plays the other implementation."""
    from tests.laplace_grad_harness import check_stages
    coords, y = cases.synthetic_binary(n, d, seed=600 + n)
    if lik == "poisson":
        y = np.random.default_rng(7).poisson(1.0 + y).astype(np.float64)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 4)
    var, a = 0.9, {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / 0.15
    ref, gref, oparts = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=lik, want_parts=True)
    _, info = orc.vecchia_laplace_logit(co, nn, ct, var, a, y[perm], likelihood=lik)
    k = info["lanczos_it"]
    assert k > 2
    _, g2, parts2 = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=lik, want_parts=True, cg_max_num_it_tridiag=k - 1)
    assert not np.array_equal(g2, gref)
    check_stages(g2, parts2, gref, oparts)


# ---- boosting gradient for non-Gaussian data (d(-mll)/dF) -- checker only, no device path yet ----------------------------------
@pytest.mark.parametrize("lik", ["bernoulli_logit", "bernoulli_probit", "poisson"])
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_CASES))
def test_oracle_boosting_gradient_matches_the_reference(orc, name, lik):
    """orc.vecchia_laplace_grad_F against REModel::CalcGradient of the reference (tests/golden/laplace_gradF_ref.npz,
    oracle/make_golden.py laplace_grad_F): -d log p / d loc + 0.5 d logdet / d mode - W .* (implicit solve), with fixed effects.  The
    implicit solve is a CG that stops at |r| < 1e-2 (seen: 3e-7 of the gradient's scale)."""
    c = cases.LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "laplace_gradF_ref.npz"))
    coords, y = cases.make_count_data(c) if lik == "poisson" else cases.make_binary_data(c)
    fe = cases.laplace_fixed_effects(coords)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    gF = orc.vecchia_laplace_grad_F(co, nn, ct, cp[0], a, y[perm], likelihood=lik, fixed_effects=fe[perm])
    out = np.empty_like(gF); out[perm] = gF
    ref = g["%s_%s_gradF" % (name, lik)]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5 * np.abs(ref).max())
    # the pin (round 5): both sides at cases.LAPLACE_TIGHT -> 1e-8 of the gradient's scale
    gFt = orc.vecchia_laplace_grad_F(co, nn, ct, cp[0], a, y[perm], likelihood=lik, fixed_effects=fe[perm],
                                     cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    out_t = np.empty_like(gFt); out_t[perm] = gFt
    ref_t = g["%s_%s_gradF_tight" % (name, lik)]
    np.testing.assert_allclose(out_t, ref_t, rtol=0, atol=1e-8 * np.abs(ref_t).max())


# ---- likelihoods with an auxiliary parameter: gamma, negative_binomial -----------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_AUX_CASES))
def test_oracle_gamma_negbin_value_and_gradient_match_the_reference(orc, name):
    """orc_vecchia_laplace_grad with link 3 (gamma) / 4 (negative_binomial): value and gradient wrt (log sigma1^2, log a, log shape) against the reference's
    own CalcGradPars -> CalcGradNegMargLikelihoodLaplaceApproxVecchia incl. its auxiliary-parameter branch (likelihoods.h:6743-6808; fixture
    tests/golden/laplace_aux_ref.npz from oracle/make_golden.py laplace_aux) at cases.LAPLACE_TIGHT: 1e-8 relative, without and with fixed effects; and the
    reference's GPB_EvalNegLogLikelihood at its default thresholds."""
    ac = cases.LAPLACE_AUX_CASES[name]
    c = cases.LAPLACE_CASES[ac["model"]]
    g = np.load(os.path.join(GOLD, "laplace_aux_ref.npz"))
    coords, y = cases.make_aux_data(ac)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    negll, _ = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=ac["lik"], aux=ac["aux"])
    ref0 = float(g[name + "_negll_0"])
    assert abs(negll - ref0) <= 1e-8 * abs(ref0), (negll, ref0)
    for fe_key, fe in (("", None), ("_fe", cases.aux_fixed_effects(ac, coords)[perm])):
        nll_t, grad_t = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=ac["lik"], fixed_effects=fe, aux=ac["aux"],
                                                 cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == (3,)
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-10 * abs(ref_v), (nll_t, ref_v)


# ---- sample weights for non-Gaussian likelihoods (round 5) ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_WEIGHT_CASES))
def test_oracle_weighted_non_gaussian_value_and_gradient_match_the_reference(orc, name):
    """Likelihood::weights_ (likelihoods.h:666-668): the oracle with orc.sample_weights -- every per-datum term weighted, the closed-form parts of the
    normalising constants and of CalcGradNegLogLikAuxPars that the reference multiplies by num_data_ not -- against the reference's own CalcGradPars on a model
    created with weights (tests/golden/laplace_weights_ref.npz, oracle/make_golden.py laplace_weights): value 1e-10, gradient (incl. the auxiliary
    parameter's component for gamma / negative_binomial) 1e-8, without and with fixed effects; the boosting gradient d(-mll)/dF 1e-8 of its scale."""
    wc = cases.LAPLACE_WEIGHT_CASES[name]
    c = cases.LAPLACE_CASES[wc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_weights_ref.npz"))
    coords, y, w = cases.make_weight_data(wc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    tight = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    wv = None if w is None else w[perm]          # (quasi_bernoulli_* may come without weights; binomial_*: the trials)
    with orc.sample_weights(wv):
        for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
            nll_t, grad_t = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=wc["lik"], fixed_effects=fe, aux=wc.get("aux"), **tight)
            ref = g[name + fe_key + "_grad_direct"]
            assert grad_t.shape == ref.shape
            np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
            ref_v = float(g[name + fe_key + "_negll_direct"])
            assert abs(nll_t - ref_v) <= 1e-10 * abs(ref_v), (nll_t, ref_v)
        if name + "_gradF" in g.files:
            gF = orc.vecchia_laplace_grad_F(co, nn, ct, cp[0], a, y[perm], likelihood=wc["lik"], fixed_effects=cases.laplace_fixed_effects(coords)[perm],
                                            weights=wv, **tight)
            out = np.empty_like(gF); out[perm] = gF
            # (-W_d [(Sigma^-1 + W)^-1 d_mll_d_mode]_d carries the CG's 1e-8 stopping error times W_d = trials x information: up to 20 trials here)
            tolF = 1e-7 if wc["lik"].startswith("binomial") else 1e-8
            np.testing.assert_allclose(out, g[name + "_gradF"], rtol=0, atol=tolF * np.abs(g[name + "_gradF"]).max())
    # without the context the weights are gone again
    if w is not None and not wc["lik"].startswith("binomial"):
        nll_u, _ = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=wc["lik"], aux=wc.get("aux"), **tight)
        assert abs(nll_u - float(g[name + "_negll_direct"])) > 1.0


# ---- cg_preconditioner_type = "pivoted_cholesky" (round 5; the second preconditioner of SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_) ----------------------
def _pivchol_setup(orc, name):
    pc = cases.LAPLACE_PIVCHOL_CASES[name]
    c = cases.LAPLACE_CASES[pc["model"]]
    coords, y = cases.make_pivchol_data(pc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    return pc, c, coords, y, perm, co, nn, ct, cp, a, cases.pivchol_rank(pc)


def _preconditioner_context(orc, pc, c, coords, co, ct, var, a, rank):
    """orc.pivoted_cholesky_preconditioner, or -- pc = "fitc" -- orc.fitc_preconditioner with the inducing points the reference's generator draws (orc.vif_setup)."""
    if pc.get("pc") == "fitc":
        ip = orc.vif_setup(coords, c["m"], rank, c["ordering"], c["seed"])[3]
        return orc.fitc_preconditioner(co, ip, ct, var, a)
    return orc.pivoted_cholesky_preconditioner(co, ct, var, a, rank=rank)


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_PIVCHOL_CASES))
def test_oracle_pivoted_cholesky_preconditioner_matches_the_reference(orc, name):
    """(fitc_* cases: the same with cg_preconditioner_type = "fitc" -- orc.fitc_preconditioner, the fitc branches of the same reference functions.)
    The (W^-1 + Sigma) form of the Vecchia-Laplace solves with P = W^-1 + L_k L_k^T (orc.pivoted_cholesky_preconditioner: PivotedCholsekyFactorizationSigma,
    CGVecchiaLaplace_Version_SigmaPlusWinvVec, CGTridiagVecchiaLaplace_Version_SigmaPlusWinv, the pivoted_cholesky branches of CalcLogDetStochVecchia /
    CalcLogDetStochDerivModeVecchia / CalcLogDetStochDerivCovParVecchia) against the reference's own CalcGradPars at cases.LAPLACE_TIGHT
    (tests/golden/laplace_pivchol_ref.npz, oracle/make_golden.py laplace_pivchol): value 1e-9, gradient (incl. the auxiliary parameter's component) 1e-8,
    without and with fixed effects; the boosting gradient 1e-8 of its scale; GPB_EvalNegLogLikelihood at the default thresholds 1e-6 (stopping-rule noise)."""
    g = np.load(os.path.join(GOLD, "laplace_pivchol_ref.npz"))
    pc, c, coords, y, perm, co, nn, ct, cp, a, rank = _pivchol_setup(orc, name)
    tight = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    with _preconditioner_context(orc, pc, c, coords, co, ct, cp[0], a, rank) as ctx:
        assert ctx.k >= 1
        for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
            nll_t, grad_t = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], fixed_effects=fe, aux=pc.get("aux"), **tight)
            ref = g[name + fe_key + "_grad_direct"]
            assert grad_t.shape == ref.shape
            np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
            ref_v = float(g[name + fe_key + "_negll_direct"])
            # (1e-9, not the 1e-10 of the vadu tests: the residual norm the CG stops on is the one of the (W^-1 + Sigma) system -- seen 2e-10 on the gamma case with fixed effects)
            assert abs(nll_t - ref_v) <= 1e-9 * abs(ref_v), (nll_t, ref_v)
        negll_d, _ = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"))
        ref_d = float(g[name + "_negll_default"])
        assert abs(negll_d - ref_d) <= 1e-6 * abs(ref_d), (negll_d, ref_d)
        if name + "_gradF" in g.files:
            gF = orc.vecchia_laplace_grad_F(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], **tight)
            out = np.empty_like(gF); out[perm] = gF
            np.testing.assert_allclose(out, g[name + "_gradF"], rtol=0, atol=1e-8 * np.abs(g[name + "_gradF"]).max())
    # the preconditioner is gone with the context: the vadu value differs in the stochastic part only
    nll_v, _ = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), **tight)
    assert abs(nll_v - float(g[name + "_negll_direct"])) <= 2e-2 * abs(nll_v) and nll_v != float(g[name + "_negll_direct"])


@pytest.mark.parametrize("name", sorted(cases.LAPLACE_VRESP_CASES))
def test_oracle_vecchia_response_preconditioner_matches_the_reference(orc, name):
    """cg_preconditioner_type = "vecchia_response" (orc.vecchia_response_preconditioner: the (W^-1 + Sigma) solves preconditioned with the Vecchia approximation of
    W^-1 + Sigma -- CalcVecchiaApproxLatentAddDiagonal with the pseudo nugget 1 / W, renewed for every W; probes B_p^-1 D_p^1/2 r; log|P| = sum log D_p) against the
    reference's GPB_EvalNegLogLikelihood (tests/golden/laplace_vresp_ref.npz, oracle/make_golden.py laplace_vresp): at cases.LAPLACE_TIGHT 1e-9 without and with fixed
    effects, at the default thresholds 1e-6 (stopping-rule noise).  The gradient does not exist with this preconditioner (likelihoods.h:6570-6572): NaN."""
    g = np.load(os.path.join(GOLD, "laplace_vresp_ref.npz"))
    pc = cases.LAPLACE_VRESP_CASES[name]
    c = cases.LAPLACE_CASES[pc["model"]]
    coords, y = cases.make_pivchol_data(pc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    tight = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    with orc.vecchia_response_preconditioner(co, ct, cp[0], a):
        for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
            v, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), fixed_effects=fe, **tight)
            ref = float(g[name + fe_key + "_negll_tight"])
            assert info["rc"] == 0 and abs(v - ref) <= 1e-9 * abs(ref), (v, ref)
        vd, _ = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"))
        ref_d = float(g[name + "_negll_default"])
        assert abs(vd - ref_d) <= 1e-6 * abs(ref_d), (vd, ref_d)
        if name == "vr_logit_n2000":
            nll_g, grad = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], **tight)
            assert abs(nll_g - float(g[name + "_negll_tight"])) <= 1e-9 * abs(nll_g) and np.all(np.isnan(grad))
    # another preconditioner, another stochastic estimate of the same log-determinant
    v_vadu, _ = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=pc["lik"], aux=pc.get("aux"), **tight)
    assert abs(v_vadu - float(g[name + "_negll_tight"])) <= 2e-2 * abs(v_vadu) and v_vadu != float(g[name + "_negll_tight"])


def test_oracle_pivoted_cholesky_factor_properties(orc):
    """PivotedCholsekyFactorizationSigma: L_k L_k^T reproduces the pivot rows / columns of the covariance matrix exactly, the residual diagonal is >= 0 and its trace
    decreases with the rank; rank n reproduces the whole matrix."""
    rng = np.random.default_rng(3)
    n = 300
    co = rng.uniform(size=(n, 2))
    var, a = 1.3, 1.0 / 0.2
    d = np.sqrt(((co[:, None, :] - co[None, :, :]) ** 2).sum(-1))
    S = var * np.exp(-a * d)
    tr = []
    for rank in (5, 20, 60):
        L, k = orc.pivoted_cholesky_factor(co, 0, var, a, rank=rank)
        assert k == rank and L.shape == (n, rank)
        R = S - L @ L.T
        assert np.diag(R).min() > -1e-12
        piv = [int(np.argmax(np.abs(L[:, q]) * (np.count_nonzero(L[:, :q], axis=1) == 0))) for q in range(1)]      # the first pivot: the first point (all diagonals equal)
        assert piv[0] == 0
        tr.append(np.trace(R))
    assert tr[0] > tr[1] > tr[2] > 0
    Lf, kf = orc.pivoted_cholesky_factor(co[:40], 0, var, a, rank=40, err_tol=0.0)
    np.testing.assert_allclose(Lf @ Lf.T, S[:40, :40], atol=1e-10)


# ---- Student-t likelihood with the Fisher-Laplace approximation (round 5, third slice) ---------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_T_CASES))
def test_oracle_student_t_value_and_gradient_match_the_reference(orc, name):
    """orc_vecchia_laplace_grad with link 6 (likelihoods.h:384-423: the information is the constant Fisher information (df + 1) / (df + 3) / scale^2, its derivative wrt the mode
    zero): value and gradient wrt (log sigma1^2, log a, log scale, log df) -- the auxiliary components = CalcGradNegLogLikAuxPars (:14241-14262) + 0.5 tr((Sigma^-1 + W)^-1 dW / d log aux)
    by CalcLogDetStochDerivAuxParVecchia's vadu branch (:16838-16856), no implicit part -- against the reference's own CalcGradPars at cases.LAPLACE_TIGHT
    (tests/golden/laplace_t_ref.npz, oracle/make_golden.py laplace_t): value 1e-10, gradient 1e-8, without and with fixed effects; GPB_EvalNegLogLikelihood at the default thresholds."""
    tc = cases.LAPLACE_T_CASES[name]
    lik = tc.get("lik", "t")          # lognormal (link 7, round 5 fourth slice): one auxiliary parameter, constant information 1 / aux (likelihoods.h:505-513); the same trace, :14275-14286, :14891-14900
    c = cases.LAPLACE_CASES[tc["model"]]
    g = np.load(os.path.join(GOLD, "laplace_t_ref.npz"))
    coords, y = cases.make_t_data(tc)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    cp = c["cov_pars"][0]
    a = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[1]
    negll, _ = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=lik, aux=tc["aux"])
    ref0 = float(g[name + "_negll_0"])
    assert abs(negll - ref0) <= 1e-8 * abs(ref0), (negll, ref0)
    for fe_key, fe in (("", None), ("_fe", cases.laplace_fixed_effects(coords)[perm])):
        nll_t, grad_t = orc.vecchia_laplace_grad(co, nn, ct, cp[0], a, y[perm], likelihood=lik, fixed_effects=fe, aux=tc["aux"],
                                                 cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
        ref = g[name + fe_key + "_grad_direct"]
        assert grad_t.shape == (2 + len(tc["aux"]),) and ref.shape == grad_t.shape
        np.testing.assert_allclose(grad_t, ref, rtol=1e-8, atol=3e-8 * np.abs(ref).max())      # (the scale's trace term multiplies the block CG's 1e-8 stopping error by dW / d log scale = -2 W: seen 1.6e-8 of the gradient's scale on the d = 3 case with fixed effects)
        ref_v = float(g[name + fe_key + "_negll_direct"])
        assert abs(nll_t - ref_v) <= 1e-10 * abs(ref_v), (nll_t, ref_v)


RC_VIFL = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}


@pytest.mark.parametrize("name", sorted(cases.VIF_LAPLACE_CASES))
def test_oracle_vif_non_gaussian_matches_the_reference(orc, name):
    """Full-scale Vecchia with a non-Gaussian likelihood (round 6; FindModePostRandEffCalcMLLFSVA, likelihoods.h:3379-3750): the oracle (gpb_oracle.c orc_set_vif + the
    numpy gradient orc.vif_laplace_grad) against the unmodified reference's values with the "fitc" / "vifdu" / "none" preconditioners and its own CalcGradPars
    (tests/golden/vif_laplace_ref.npz)."""
    c = cases.VIF_LAPLACE_CASES[name]
    g = np.load(os.path.join(GOLD, "vif_laplace_ref.npz"))
    coords, y = cases.vif_laplace_data(name)
    rank = 200 if c["rank"] is None else c["rank"]
    perm, co, nn, ip, ip2 = orc.vif_setup(coords, c["m"], c["k"], c["ordering"], c["seed"], num_ind_points_preconditioner=rank)
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    tight = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
    tight_fitc = dict(cg_delta_conv=cases.VIF_LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.VIF_LAPLACE_TIGHT["delta_conv_mode_finding"])
    wts = cases.vif_laplace_weights(name)
    wctx = orc.sample_weights(None if wts is None else wts[perm])
    wctx.__enter__()
    for pc in ("fitc", "vifdu") + (("none",) if name.endswith("logit") else ()):       # ("none" needs hundreds of CG iterations: one case; missing keys: the reference aborts there)
        for j, cp in enumerate(c["cov_pars"]):
            key = "%s_%s_negll_%d" % (name, pc, j)
            if key not in g.files:
                continue
            a = RC_VIFL[ct] / cp[1]
            tl = tight_fitc if pc == "fitc" else tight
            with orc.vif_laplace(co, nn, ip, ct, cp[0], a, pc, ip2) as ctx:
                f = ctx.factor
                v, info = orc.vecchia_laplace_logit(co, nn, ct, cp[0], a, y[perm], likelihood=c["lik"], factor=(f["A"], f["D"]), aux=c["aux"], **tl)
            assert abs(v - float(g[key])) <= 1e-9 * abs(v), (key, v, float(g[key]))
    cp = c["cov_pars"][0]
    # (two mode findings, the second from the first one's mode: what the fixture's driver does -- EvalNegLogLikelihood, then CalcCovFactorOrModeAndNegLL, oracle/ref_driver.cpp:438-446;
    #  cases.py: VIF_LAPLACE_TIGHT says why it matters)
    v0, g0, p0 = orc.vif_laplace_grad(co, nn, ip, ip2, ct, cp[0], RC_VIFL[ct] / cp[1], y[perm], likelihood=c["lik"], aux=c["aux"], want_parts=True, **tight_fitc)
    v, gr = orc.vif_laplace_grad(co, nn, ip, ip2, ct, cp[0], RC_VIFL[ct] / cp[1], y[perm], likelihood=c["lik"], aux=c["aux"], mode_init=p0["mode"], **tight_fitc)
    ref = g[name + "_fitc_grad_0"]
    assert abs(v - float(g[name + "_fitc_negll_direct_0"])) <= 1e-9 * abs(v)
    wctx.__exit__()
    assert gr.shape == ref.shape
    np.testing.assert_allclose(gr, ref, rtol=0, atol=c.get("grad_rtol", 1e-8) * np.abs(ref).max())
