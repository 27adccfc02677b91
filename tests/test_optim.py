"""Parameter estimation (GPB_OptimCovPar -- the direct caller of the hot path).

Pins: tests/golden/optim_ref.npz = the reference's own GPB_SetOptimConfig + GPB_OptimCovPar on tests/cases.py:OPTIM_CASES
(oracle/make_golden.py optim), and the R suite's golden fit (378 iterations, test_GPModel_gaussian_process.R:1316-1324).
CPU tests drive the product's host optimiser with the oracle's likelihood / gradient; GPU tests run the whole path on the MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "optim_ref.npz")
CPU_CASES = [k for k, c in cases.OPTIM_CASES.items() if c["cpu"]]


def _cfg_kwargs(cfg):
    m = dict(cfg)
    if "optimizer_cov" in m:
        m["optimizer"] = m.pop("optimizer_cov")
    return m


def _check(name, g, cov_pars, num_it, negll):
    # identical control flow: same number of iterations; parameters / likelihood to the accuracy of the evaluations (1e-8 relative
    # per evaluation, accumulated over the trajectory)
    assert num_it == int(g[name + "_num_it"]), (num_it, int(g[name + "_num_it"]))
    np.testing.assert_allclose(cov_pars, g[name + "_cov_pars"], rtol=2e-6)
    np.testing.assert_allclose(negll, float(g[name + "_negll"]), rtol=1e-9)


def test_r_suite_golden_is_in_the_fixture():
    """test_GPModel_gaussian_process.R:1316-1324 (fit of the Vecchia model with 30 neighbours)."""
    g = np.load(GOLDEN)
    assert int(g["r_gd_nesterov_parcrit_num_it"]) == 378
    np.testing.assert_allclose(g["r_gd_nesterov_parcrit_cov_pars"], [0.03297349, 1.07691542, 0.11378505], atol=1e-6)
    assert abs(float(g["r_gd_nesterov_parcrit_negll"]) - 122.7680889) < 1e-6


@pytest.mark.parametrize("name", CPU_CASES)
def test_host_optimiser_follows_the_reference_trajectory(lib_built, name):
    from oracle import orc
    from tests import optim_harness as oh
    g = np.load(GOLDEN)
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    assert ids is None
    ct = orc.cov_type_id(mc["cov_function"], mc["shape"])
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    th0 = orc.transform_cov_pars(ct, g[name + "_init_cov_pars"])          # FindInitCovPar itself is checked on the GPU (needs a model)
    if init is not None:
        np.testing.assert_allclose(g[name + "_init_cov_pars"], init, rtol=1e-12)
    cb, calls = oh.oracle_terms(orc, co, nn, ct, y[perm])
    lib = C.CDLL(lib_built)
    th, nit, nll, ne = oh.optimize(lib, coords.shape[0], th0, cb, range_const=[1.0, np.sqrt(3.0), np.sqrt(5.0)][ct], **_cfg_kwargs(cfg))
    cov_pars = np.array([th[0], th[1] * th[0], [1.0, np.sqrt(3.0), np.sqrt(5.0)][ct] / th[2]])
    _check(name, g, cov_pars, nit, nll)
    assert ne[0] + ne[1] == len(calls)


# test_GPModel_gaussian_process.R:1364-1398 ("Holding some parameters fix"): lbfgs from init_cov_pars with estimate_cov_par_index
R_FIXED_PAR_GOLDENS = {
    (1, 0, 0): ([0.4585860589, 0.5170731356, 0.1786480774], 127.8100465),
    (1, 1, 0): ([0.10238832994, 1.23364920496, 0.17864807736], 123.4597106),
    (0, 1, 0): ([0.5170731356, 0.6109062004, 0.1786480774], 128.005439),
}
R_FIXED_PAR_CFG = dict(optimizer="lbfgs", lr_cov=0.1, acc_rate_cov=0.5, delta_rel_conv=1e-6)     # params_vecchia with optimizer_cov = "lbfgs"


@pytest.mark.parametrize("est", sorted(R_FIXED_PAR_GOLDENS))
def test_holding_parameters_fixed_reproduces_the_r_suite_goldens(lib_built, est):
    """estimate_cov_par_index in the host optimiser (zero gradient entries, ProfileOutSigma2 only when the nugget is estimated,
    MaybeKeepVarianceConstant when the nugget is estimated but the GP variance is not) against the R suite's golden estimates and
    likelihood values (TOLERANCE_STRICT = 1e-6 there)."""
    from oracle import orc
    from tests import optim_harness as oh
    coords, y, ids, mc, init, cfg = cases.optim_case("r_gd_nesterov_parcrit")
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    cb, calls = oh.oracle_terms(orc, co, nn, 0, y[perm])
    th, nit, nll, ne = oh.optimize(C.CDLL(lib_built), len(y), orc.transform_cov_pars(0, init), cb, estimate_cov_par_index=est, **R_FIXED_PAR_CFG)
    cp, nll_ref = R_FIXED_PAR_GOLDENS[est]
    out = np.array([th[0], th[1] * th[0], 1.0 / th[2]])
    assert np.abs(out - cp).sum() < 1e-8
    assert abs(nll - nll_ref) < 1e-6
    for i, e in enumerate(est):                       # held parameters come back as they went in (original scale)
        if not e:
            assert abs(out[i] - init[i]) < 1e-12


@pytest.mark.parametrize("ordering", ["none", "random"])
def test_r_suite_fit_with_the_maximal_number_of_neighbours(lib_built, ordering):
    """test_GPModel_gaussian_process.R:1175-1192, :1231-1239 (random ordering: the same numbers): Vecchia on ALL predecessors (m = n - 1 = 99: exact GP), gradient descent + Nesterov,
    parameter criterion: 382 iterations, estimates (0.03276547, 1.07617676, 0.11352557), nll 122.7752664.  Host optimiser + oracle
    (m = 99 is beyond the device kernels' 62 neighbours; the m = 30 sibling, 378 iterations, runs on the device)."""
    from oracle import orc
    from tests import optim_harness as oh
    coords, y, ids, mc, init, cfg = cases.optim_case("r_gd_nesterov_parcrit")
    perm, co, nn = orc.vecchia_setup(coords, len(y) - 1, ordering, 0)
    cb, calls = oh.oracle_terms(orc, co, nn, 0, y[perm])
    th, nit, nll, ne = oh.optimize(C.CDLL(lib_built), len(y), orc.transform_cov_pars(0, init), cb, **_cfg_kwargs(cfg))
    assert nit == 382
    assert np.abs(np.array([th[0], th[1] * th[0], 1.0 / th[2]]) - [0.03276547, 1.07617676, 0.11352557]).sum() < 1e-6      # TOLERANCE_STRICT
    assert abs(nll - 122.7752664) < 1e-6


def test_r_suite_fit_with_multiple_observations_per_location(lib_built):
    """test_GPModel_gaussian_process.R:1676-1687: Vecchia on all predecessors, every location observed four times, lbfgs from the suite's
    initial values: estimates (0.03713823078, 1.15342626349, 0.19206772520), nll 33.43573582 (the suite allows 1e-2; 1e-7 here).  Host
    optimiser + oracle: m = n - 1 = 99 is beyond the device kernels' 62 neighbours."""
    from oracle import orc
    from tests import optim_harness as oh
    coords, y, init = orc.r_fixture_multiple()
    perm, co, nn = orc.vecchia_setup(coords, len(y) - 1, "none", 0)
    cb, calls = oh.oracle_terms(orc, co, nn, 0, y[perm])
    th, nit, nll, ne = oh.optimize(C.CDLL(lib_built), len(y), orc.transform_cov_pars(0, init), cb, optimizer="lbfgs", max_iter=1000)
    np.testing.assert_allclose([th[0], th[1] * th[0], 1.0 / th[2]], [0.03713823078, 1.15342626349, 0.19206772520], rtol=0, atol=1e-7)
    assert abs(nll - 33.43573582) < 1e-7


@pytest.mark.parametrize("name", [k for k, c in cases.OPTIM_CASES.items() if c["init"] is None and c["model"] != "clusters"])
def test_initial_values_match_the_reference(lib_built, name):
    """FindInitCovPar (var(y)/2, ratio 1, range from the median pairwise distance; 1000 points drawn from the model's generator AFTER
    the ordering shuffle when n > 1000) through the host-only seam against the reference's GPB_GetInitCovPar."""
    from oracle import orc
    g = np.load(GOLDEN)
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    ct = orc.cov_type_id(mc["cov_function"], mc["shape"])
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    cm = np.asfortranarray(co)
    th = np.empty(3)
    lib = C.CDLL(lib_built)
    lib.LGBM_GetLastError.restype = C.c_char_p
    n = len(y)
    rc = lib.GPB_HIP_FindInitCovParHost(C.c_int(n), y.ctypes.data_as(C.c_void_p), None, C.c_int(n), C.c_int(coords.shape[1]),
                                        cm.ctypes.data_as(C.c_void_p), C.c_int(ct), C.c_int(mc["seed"]),
                                        C.c_int(n if mc["ordering"] == "random" else 0), th.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.LGBM_GetLastError()
    rc_ = [1.0, np.sqrt(3.0), np.sqrt(5.0)][ct]
    np.testing.assert_allclose([th[0], th[1] * th[0], rc_ / th[2]], g[name + "_init_cov_pars"], rtol=1e-10)


@pytest.mark.parametrize("name", ["logit_n1500_lbfgs", "poisson_n1500_lbfgs", "logit_u3d_n1200_lbfgs"])
def test_find_init_cov_par_for_non_gaussian_likelihoods(lib_built, name):
    """Non-Gaussian models start from marginal variance 1 (re_model_template.h:4865, :4904-4913) and the same range heuristic with the same
    generator state; reference: GPB_GetInitCovPar after its own fit (tests/golden/optim_laplace_ref.npz)."""
    from oracle import orc
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "optim_laplace_ref.npz"))
    oc = cases.OPTIM_LAPLACE_CASES[name]
    c = cases.LAPLACE_CASES[oc["model"]]
    coords, y = cases.make_count_data(c) if oc["lik"] == "poisson" else cases.make_binary_data(c)
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    cm = np.asfortranarray(co)
    th = np.empty(3)
    lib = C.CDLL(lib_built)
    n = len(y)
    rc = lib.GPB_HIP_FindInitCovParHost(C.c_int(n), y.ctypes.data_as(C.c_void_p), None, C.c_int(n), C.c_int(coords.shape[1]),
                                        cm.ctypes.data_as(C.c_void_p), C.c_int(ct), C.c_int(c["seed"]),
                                        C.c_int(n if c["ordering"] == "random" else 0), th.ctypes.data_as(C.c_void_p))
    assert rc == 0
    rc_ = [1.0, np.sqrt(3.0), np.sqrt(5.0)][ct]
    ref = g[name + "_init_cov_pars"]
    assert ref[0] == 1.0
    np.testing.assert_allclose(rc_ / th[2], ref[1], rtol=1e-7)


@pytest.mark.parametrize("name", list(cases.OPTIM_LAPLACE_CASES))
def test_host_optimiser_for_non_gaussian_likelihoods_follows_the_reference(lib_built, name):
    """The non-Gaussian branch of the product's host optimiser (lbfgs / gradient descent on (sigma1_2, a), warm-started mode finding,
    mode reset on rejected steps) driven by the oracle's Laplace approximation and gradient, against the reference's own fits of the
    Vecchia-Laplace model (iterative methods, vadu).  The device half is tests/test_z_laplace_grad_gpu.py; this pins the host half."""
    from oracle import orc
    from tests import optim_harness as oh
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "optim_laplace_ref.npz"))
    oc = cases.OPTIM_LAPLACE_CASES[name]
    c = cases.LAPLACE_CASES[oc["model"]]
    coords, y = cases.make_count_data(c) if oc["lik"] == "poisson" else cases.make_binary_data(c)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    rc_ = [1.0, np.sqrt(3.0), np.sqrt(5.0)][ct]
    init = g[name + "_init_cov_pars"]
    fe = cases.laplace_fixed_effects(coords)[perm] if oc.get("fe") else None
    solver = {k: oc["cfg"][k] for k in ("cg_delta_conv", "delta_conv_mode_finding") if k in oc["cfg"]}      # thresholds of the evaluator, not of the optimiser
    ev = oh.OracleLaplaceEvaluator(orc, co, nn, ct, y[perm], oc["lik"], fixed_effects_ord=fe, cg_delta_conv=solver.get("cg_delta_conv", 1e-2),
                                   delta_conv_mode=solver.get("delta_conv_mode_finding", 1e-8))
    th, nit, nll, ne = oh.optimize_laplace(C.CDLL(lib_built), [init[0], rc_ / init[1]], ev,
                                           **_cfg_kwargs({k: v for k, v in oc["cfg"].items() if k not in solver}))
    ref_it = int(g[name + "_num_it"])
    if oc.get("tight"):      # tight solver thresholds (cases.LAPLACE_TIGHT): the fit is reproducible to the accuracy of its evaluations
        assert nit == ref_it, (nit, ref_it)
        np.testing.assert_allclose([th[0], rc_ / th[1]], g[name + "_cov_pars"], rtol=1e-6)
        assert abs(nll - float(g[name + "_negll"])) <= 1e-8 * abs(nll)
    elif oc["exact_it"]:
        assert nit == ref_it, (nit, ref_it)
        np.testing.assert_allclose([th[0], rc_ / th[1]], g[name + "_cov_pars"], rtol=1e-4)     # flat optimum: 2e-5 seen at nll agreement 1e-8
        assert abs(nll - float(g[name + "_negll"])) <= 1e-7 * abs(nll)
    else:
        assert abs(nit - ref_it) <= 2, (nit, ref_it)
        np.testing.assert_allclose([th[0], rc_ / th[1]], g[name + "_cov_pars"], rtol=2e-2)
        assert abs(nll - float(g[name + "_negll"])) <= 1e-5 * abs(nll)
    assert ne == sum(1 for op in ev.calls if op[0] in (0, 1))


def test_r_suite_probit_fit_within_the_iterative_tolerance(lib_built):
    """test_GPModel_non_Gaussian_data.R:1428-1435, :1641-1649: probit GP without covariates, Vecchia on all predecessors (random ordering),
    gradient descent + Nesterov from (1, mean(dist)/3), 500 probe vectors, cg_delta_conv 1e-3.  Golden of the Cholesky-based fit:
    (0.6875476, 0.1062862); the suite accepts the iterative methods within TOLERANCE_ITERATIVE = 0.1 (sum of absolute differences).
    Host optimiser + oracle (m = 99)."""
    from scipy.spatial.distance import pdist
    from oracle import orc
    from tests import optim_harness as oh
    coords, y = orc.r_fixture_probit()
    init = [1.0, pdist(coords).mean() / 3]
    perm, co, nn = orc.vecchia_setup(coords, len(y) - 1, "random", 0)
    ev = oh.OracleLaplaceEvaluator(orc, co, nn, 0, y[perm], "bernoulli_probit", num_rand_vec=500, cg_delta_conv=1e-3)
    th, nit, nll, ne = oh.optimize_laplace(C.CDLL(lib_built), [init[0], 1.0 / init[1]], ev, optimizer="gradient_descent", lr_cov=0.1,
                                           acc_rate_cov=0.5, max_iter=1000, use_nesterov_acc=True)
    assert np.abs(np.array([th[0], 1.0 / th[1]]) - [0.6875476, 0.1062862]).sum() < 0.1
    assert np.abs(np.array([th[0], 1.0 / th[1]]) - [0.6875476, 0.1062862]).sum() < 0.02       # seen: 0.007


def test_cond_all_prediction_host_half_reproduces_the_r_goldens(lib_built):
    """GPB_HIP_PredictCondAllHost (forward substitution with Bp, rows of Bp^-1) fed with the oracle's factor rows of the appended prediction
    points: test_GPModel_gaussian_process.R:1478-1485 (30 neighbours: the two points 1.4e-5 apart get covariance 0.09889262) and
    :1241-1258 (all observations: the exact GP's joint predictive distribution)."""
    from oracle import orc
    lib = C.CDLL(lib_built)
    coords, y = orc.r_fixture()
    pt = orc.transform_cov_pars(0, np.array([0.02, 1.2, 0.9]))
    cases_ = [(np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]]), 30, [0.08665472, 0.08661259, 0.49011216],
               [0.11891004, 0.09889262, 0., 0.09889262, 0.11891291, 0., 0., 0., 0.08108126]),
              (np.array([[0.1, 0.9], [0.2, 0.4], [0.7, 0.55]]), len(y) + 2, [0.08704577, 1.63875604, 0.48513581],
               [1.189093e-01, 1.171632e-05, -4.172444e-07, 1.171632e-05, 7.427727e-02, 1.492859e-06, -4.172444e-07, 1.492859e-06, 8.107455e-02])]
    for ct_, m_pred, mu_ref, cov_ref in cases_:
        n_obs, n_pred = len(y), len(ct_)
        call = np.vstack([coords, ct_])
        nn = orc.neighbors_range(call, m_pred, n_obs, -1)
        A, D, bad = orc.vecchia_factor(call, nn, 0, pt[1], pt[2], gauss=True)
        nnp = np.ascontiguousarray(nn[n_obs:], dtype=np.int32); Ap = np.ascontiguousarray(A[n_obs:]); Dp = np.ascontiguousarray(D[n_obs:])
        mu = np.empty(n_pred); var = np.empty(n_pred); cov = np.empty((n_pred, n_pred))
        for resp in (True, False):
            rc = lib.GPB_HIP_PredictCondAllHost(C.c_int(n_obs), C.c_int(n_pred), C.c_int(nnp.shape[1]), nnp.ctypes.data_as(C.c_void_p),
                                                Ap.ctypes.data_as(C.c_void_p), Dp.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                                C.c_double(pt[0]), C.c_bool(resp), mu.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p),
                                                cov.ctypes.data_as(C.c_void_p))
            assert rc == 0
            assert np.abs(mu - mu_ref).sum() < 1e-6
            expect = np.array(cov_ref).reshape(n_pred, n_pred) - (0. if resp else 0.02) * np.eye(n_pred)
            assert np.abs(cov - expect).sum() < 1e-6
            np.testing.assert_allclose(var, np.diag(cov), rtol=0, atol=0)
        om, oc = orc.predict_cond_all(coords, y, ct_, 0, pt, m_pred, predict_response=False)
        np.testing.assert_allclose(mu, om, rtol=1e-12); np.testing.assert_allclose(cov, oc, rtol=1e-10, atol=1e-14)
    # variances only; a neighbour that does not precede its row is refused
    rc = lib.GPB_HIP_PredictCondAllHost(C.c_int(n_obs), C.c_int(n_pred), C.c_int(nnp.shape[1]), nnp.ctypes.data_as(C.c_void_p),
                                        Ap.ctypes.data_as(C.c_void_p), Dp.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                        C.c_double(pt[0]), C.c_bool(False), mu.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), None)
    assert rc == 0
    np.testing.assert_allclose(var, np.diag(cov), rtol=1e-14)
    bad_nn = nnp.copy(); bad_nn[0, 0] = n_obs + 1
    rc = lib.GPB_HIP_PredictCondAllHost(C.c_int(n_obs), C.c_int(n_pred), C.c_int(nnp.shape[1]), bad_nn.ctypes.data_as(C.c_void_p),
                                        Ap.ctypes.data_as(C.c_void_p), Dp.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                        C.c_double(pt[0]), C.c_bool(False), mu.ctypes.data_as(C.c_void_p), None, None)
    assert rc == -1


def test_host_optimiser_errors(lib_built):
    from tests import optim_harness as oh
    lib = C.CDLL(lib_built)
    cb = oh.TERMS_FN(lambda ctx, r, a, wg, t7: -1)
    with pytest.raises(RuntimeError, match="not on the MI355X path"):
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], cb, optimizer="fisher_scoring")
    with pytest.raises(RuntimeError):                      # failing evaluation callback under the simplex search too
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], cb, optimizer="nelder_mead")
    with pytest.raises(RuntimeError, match="estimate_cov_par_index"):
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], cb, optimizer="nelder_mead", estimate_cov_par_index=[1, 0, 1])
    with pytest.raises(RuntimeError, match="nesterov_schedule_version = 1"):
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], cb, optimizer="gradient_descent", nesterov_schedule_version=1)
    with pytest.raises(RuntimeError, match="positive"):
        oh.optimize(lib, 10, [1.0, -1.0, 1.0], cb)
    with pytest.raises(RuntimeError):                      # failing evaluation callback -> -1, no crash
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], cb)

    def nan_terms(ctx, r, a, wg, t7):
        for q in range(7):
            t7[q] = float("nan")
        return 0
    with pytest.raises(RuntimeError, match="NaN occurred in initial"):
        oh.optimize(lib, 10, [1.0, 1.0, 1.0], oh.TERMS_FN(nan_terms), optimizer="gradient_descent")


def test_nelder_mead_default_tolerance_is_1e_8(lib_built):
    """R-package/tests/testthat/test_GPModel_gaussian_process.R:200-208: without delta_rel_conv the simplex search stops at 1e-8 (the other
    optimisers at 1e-6, SetInitialValueDeltaRelConv re_model_template.h:8338-8347): the default fit equals the 1e-8 fit and differs from 1e-6."""
    from oracle import orc
    from tests import optim_harness as oh
    g = np.load(GOLDEN)
    name = "r_nelder_mead"
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    th0 = orc.transform_cov_pars(0, g[name + "_init_cov_pars"])
    cb, _ = oh.oracle_terms(orc, co, nn, 0, y[perm])
    lib = C.CDLL(lib_built)
    d = oh.optimize(lib, len(y), th0, cb, optimizer="nelder_mead")
    d8 = oh.optimize(lib, len(y), th0, cb, optimizer="nelder_mead", delta_rel_conv=1e-8)
    d6 = oh.optimize(lib, len(y), th0, cb, optimizer="nelder_mead", delta_rel_conv=1e-6)
    assert np.array_equal(d[0], d8[0]) and d[1] == d8[1] and d[2] == d8[2]
    assert d6[1] < d[1] and not np.array_equal(d6[0], d[0])


def test_nan_in_a_gradient_based_fit_restarts_with_the_simplex_search(lib_built, capfd):
    """re_model_template.h:1706-1731: when NaN / Inf occurs with a gradient-based optimiser the reference starts the optimisation a second time
    from the initial values with 'nelder_mead' (tolerance of the first optimiser).  Here the oracle's gradient terms turn into NaN after a few
    gradient evaluations; the result must be that of a direct 'nelder_mead' run with delta_rel_conv = 1e-6 from the same initial values."""
    from oracle import orc
    from tests import optim_harness as oh
    g = np.load(GOLDEN)
    name = "r_lbfgs_default"
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    ct = orc.cov_type_id(mc["cov_function"], mc["shape"])
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    th0 = orc.transform_cov_pars(ct, g[name + "_init_cov_pars"])
    good, _ = oh.oracle_terms(orc, co, nn, ct, y[perm])
    lib = C.CDLL(lib_built)
    for optimizer in ("gradient_descent",):      # (lbfgs backtracks out of a NaN step and stops at the last finite point, as the reference's copy does)
        ngrad = [0]

        def poisoned(ctx, ratio, a, with_grad, t7):
            rc = good(ctx, ratio, a, with_grad, t7)
            if with_grad:
                ngrad[0] += 1
                if ngrad[0] > 2:
                    for q in range(3, 7):
                        t7[q] = float("nan")
            return rc
        th, nit, nll, ne = oh.optimize(lib, coords.shape[0], th0, oh.TERMS_FN(poisoned), optimizer=optimizer)
        assert "started a second time using 'nelder_mead'" in capfd.readouterr().err
        th2, nit2, nll2, ne2 = oh.optimize(lib, coords.shape[0], th0, good, optimizer="nelder_mead", delta_rel_conv=1e-6)
        assert nit == nit2 and np.array_equal(th, th2) and nll == nll2
        assert np.all(np.isfinite(th)) and nit > 10


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.OPTIM_CASES))
def test_fit_on_device_matches_the_reference(lib_built, name):
    import gpboost_amd
    g = np.load(GOLDEN)
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=mc["cov_function"], cov_fct_shape=mc["shape"], gp_approx="vecchia",
                              num_neighbors=mc["m"], vecchia_ordering=mc["ordering"], seed=mc["seed"], cluster_ids=ids)
    params = {("maxit" if k == "max_iter" else k): v for k, v in cfg.items()}
    if init is not None:
        params["init_cov_pars"] = init
    assert np.all(mdl._get_init_cov_pars() == -1.0) or init is not None
    mdl.fit(y, params=params if params else None)
    np.testing.assert_allclose(mdl._get_init_cov_pars(), g[name + "_init_cov_pars"], rtol=1e-10)    # FindInitCovPar (or the given values)
    _check(name, g, mdl.get_cov_pars(), mdl.get_num_optim_iter(), mdl.get_current_neg_log_likelihood())
    info = mdl.optim_info()
    if cfg.get("optimizer_cov") == "nelder_mead":       # derivative-free: likelihood evaluations only (at least one per iteration + the initial simplex)
        assert info["num_grad_evals"] == 0 and info["num_ll_evals"] >= mdl.get_num_optim_iter() + 3
    else:
        assert info["num_grad_evals"] >= mdl.get_num_optim_iter()
    # the stored parameters are the estimates: evaluating the likelihood there reproduces the optimum
    nll = mdl.neg_log_likelihood(cov_pars=mdl.get_cov_pars(), y=y)
    assert abs(nll - mdl.get_current_neg_log_likelihood()) <= 1e-9 * abs(nll)
    assert abs(mdl.neg_log_likelihood(y=y) - nll) <= 1e-9 * abs(nll)


@pytest.mark.gpu
def test_r_suite_fit_then_predict_end_to_end(lib_built):
    """test_GPModel_gaussian_process.R:1310-1334 as one chain on the device, through the reference's own entry points: fit (378
    iterations) -> set_prediction_data -> predict from the FITTED model (cov_pars and y not passed again) with predict_cov_mat."""
    import gpboost_amd
    from oracle import orc
    coords, y, ids, mc, init, cfg = cases.optim_case("r_gd_nesterov_parcrit")
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none")
    mdl.fit(y, params=dict(cfg, init_cov_pars=init))
    assert mdl.get_num_optim_iter() == 378
    assert np.abs(mdl.get_cov_pars() - np.array([0.03297349, 1.07691542, 0.11378505])).sum() < 1e-6
    assert abs(mdl.get_current_neg_log_likelihood() - 122.7680889) < 1e-6
    coord_test = np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]])
    mdl.set_prediction_data(vecchia_pred_type="order_obs_first_cond_obs_only", num_neighbors_pred=30)
    pred = mdl.predict(y=y, gp_coords_pred=coord_test, predict_cov_mat=True)
    assert np.abs(pred["mu"] - np.array([0.06968068, 0.06967750, 0.44208925])).sum() < 1e-6
    exp_cov = np.array([0.6214955, 0, 0, 0, 0.6215069, 0, 0, 0, 0.4199531]).reshape(3, 3)
    assert np.abs(pred["cov"] - exp_cov).sum() < 1e-6
    again = mdl.predict(gp_coords_pred=coord_test, predict_var=True)          # y of the fit is still resident
    np.testing.assert_allclose(again["var"], np.diag(pred["cov"]), rtol=1e-12)
    lat = mdl.predict(gp_coords_pred=coord_test, vecchia_pred_type="latent_order_obs_first_cond_all", predict_var=True)    # (tests/test_predtypes.py holds the values)
    assert np.all(np.isfinite(lat["mu"])) and np.all(lat["var"] > 0.)
    with pytest.raises(gpboost_amd.GPBoostError, match="not supported for the Veccia"):
        mdl.set_prediction_data(vecchia_pred_type="nonsense")


@pytest.mark.gpu
def test_fit_at_the_metric_size_reaches_a_stationary_point(lib_built):
    """n = 1e6, m = 30 (BASELINE.json's metric size; no oracle can follow there): the lbfgs fit decreases the likelihood from the
    initial values, ends where the gradient wrt (log sigma1_2/sigma2, log a) is small against the likelihood's scale, the profiled
    nugget satisfies its closed form sigma2 = y' Psi^-1 y / n, and a second fit started at the optimum stops at once."""
    import gpboost_amd
    n = 1000000
    rng = np.random.default_rng(7)
    coords = rng.uniform(size=(n, 2))
    y = np.sin(4 * coords[:, 0]) + 0.5 * rng.standard_normal(n)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="random", seed=1)
    nll_init = mdl.neg_log_likelihood(y=y)                      # at the reference's initial values (FindInitCovPar)
    init = mdl._get_init_cov_pars()
    mdl.fit(y)
    cp, nll = mdl.get_cov_pars(), mdl.get_current_neg_log_likelihood()
    assert nll < nll_init and 1 < mdl.get_num_optim_iter() < 60
    v, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    assert abs(v - nll) <= 1e-9 * abs(nll)
    assert np.abs(grad).max() < 1e-4 * abs(nll)                 # all three log-scale derivatives (the nugget's through the profile)
    assert abs(grad[0]) < 1e-6 * n                              # d nll / d log sigma2 = n/2 - y'Psi^-1 y / (2 sigma2) = 0 at the profile
    mdl.fit(y, params={"init_cov_pars": cp})
    assert mdl.get_num_optim_iter() <= 2                        # relative change of the likelihood below 1e-6 at once
    assert abs(mdl.get_current_neg_log_likelihood() - nll) <= 2e-6 * abs(nll)
    np.testing.assert_allclose(mdl.get_cov_pars(), cp, rtol=0.05)   # the likelihood is flat along the range at that tolerance
    assert not np.allclose(init, cp)


@pytest.mark.gpu
def test_fit_through_the_rccl_path_single_rank(lib_built):
    """A sharded fit is the same host loop on every rank over all-reduced sums: with a 1-rank communicator on the model's handle
    GPB_OptimCovPar evaluates through kernel + reduction + ncclAllReduce and must reproduce the plain fit exactly."""
    import gpboost_amd
    from gpboost_amd import shim
    g = np.load(GOLDEN)
    name = "u2d_n3000_lbfgs"
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=mc["cov_function"], cov_fct_shape=mc["shape"], gp_approx="vecchia",
                              num_neighbors=mc["m"], vecchia_ordering=mc["ordering"], seed=mc["seed"])
    st = shim.VecchiaState.from_handle(mdl.vecchia_handle(), len(y), coords.shape[1], mc["m"])
    st.comm_init(shim.comm_unique_id(), 0, 1)
    mdl.fit(y)
    _check(name, g, mdl.get_cov_pars(), mdl.get_num_optim_iter(), mdl.get_current_neg_log_likelihood())


@pytest.mark.gpu
def test_fit_errors_on_device(lib_built):
    import gpboost_amd
    coords, y = cases.make_data(cases.GOLDEN_CASES["r_exp_m30_none"])
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10, vecchia_ordering="none")
    with pytest.raises(gpboost_amd.GPBoostError, match="not on the MI355X path"):
        mdl.fit(y, params={"optimizer_cov": "fisher_scoring"})
    se = mdl.fit(y, params={"optimizer_cov": "lbfgs"}).get_cov_pars(std_err=True)     # standard errors: now on the device (Fisher information)
    assert se.shape == (6,) and np.all(se[3:] > 0)
    yb = y.copy(); yb[3] = np.nan
    with pytest.raises(gpboost_amd.GPBoostError, match="NaN or Inf in response"):
        mdl.fit(yb)
    vf = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="full_scale_vecchia", num_neighbors=10, num_ind_points=20)
    with pytest.raises(gpboost_amd.GPBoostError, match="standard deviations"):              # VIF: estimates yes (Nelder-Mead), standard errors not (nor in the reference, :10056-10058)
        vf.fit(y, params={"optimizer_cov": "nelder_mead", "maxit": 5}).get_cov_pars(std_err=True)


@pytest.mark.gpu
def test_cond_all_prediction_on_device_reproduces_the_r_goldens_and_the_oracle(lib_built):
    """vecchia_pred_type = 'order_obs_first_cond_all' through the reference's entry points (GPB_SetPredictionData + GPB_PredictREModel):
    neighbour search among observed AND preceding prediction points + factor of the appended rows on the device, forward substitution with
    Bp on the host (CalcPredVecchiaObservedFirstOrder, CondObsOnly = false, Vecchia_utils.cpp:1701-2093).  R goldens
    test_GPModel_gaussian_process.R:1478-1485 (30 neighbours; the two points 1.4e-5 apart get covariance 0.09889262), and the oracle on a
    seeded larger case (means / covariance matrix / variances, response and latent)."""
    import gpboost_amd
    from oracle import orc
    coords, y = orc.r_fixture()
    cp = np.array([0.02, 1.2, 0.9])
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none")
    ct = np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]])
    mdl.set_prediction_data(vecchia_pred_type="order_obs_first_cond_all", num_neighbors_pred=30)
    pred = mdl.predict(y=y, gp_coords_pred=ct, cov_pars=cp, predict_cov_mat=True, predict_response=True)
    assert np.abs(pred["mu"] - [0.08665472, 0.08661259, 0.49011216]).sum() < 1e-6
    assert np.abs(pred["cov"].ravel() - [0.11891004, 0.09889262, 0., 0.09889262, 0.11891291, 0., 0., 0., 0.08108126]).sum() < 1e-6
    lat = mdl.predict(y=y, gp_coords_pred=ct, cov_pars=cp, predict_var=True, predict_response=False)
    np.testing.assert_allclose(np.diag(pred["cov"]) - lat["var"], 0.02, rtol=1e-9)
    # seeded larger case against the oracle: random ordering, Matern-1.5, d = 3, prediction points that neighbour each other
    cases_ = [(4000, 3, 20, "matern", 1.5, 300, 25), (3000, 2, 30, "exponential", 0.5, 500, 40)]
    for n, d, m, cf, sh, npred, mpred in cases_:
        c2, y2 = cases.synthetic(n, d, seed=n + 1)
        rng = np.random.default_rng(3)
        cpred = rng.uniform(0.3, 0.5, size=(npred, d))
        cp2 = np.array([0.1, 1.0, 0.15])
        md = gpboost_amd.GPModel(gp_coords=c2, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=2)
        md.set_prediction_data(vecchia_pred_type="order_obs_first_cond_all", num_neighbors_pred=mpred)
        pr = md.predict(y=y2, gp_coords_pred=cpred, cov_pars=cp2, predict_cov_mat=True, predict_response=False)
        perm, _ = md.vecchia_structure()
        ctid = orc.cov_type_id(cf, sh)
        om, oc = orc.predict_cond_all(c2[perm], y2[perm], cpred, ctid, orc.transform_cov_pars(ctid, cp2), mpred, predict_response=False)
        np.testing.assert_allclose(pr["mu"], om, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(pr["cov"], oc, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_exact_gp_gradient_and_fit_reproduce_the_r_goldens(lib_built):
    """gp_approx = "none" (SURVEY.md 8 row a10): the gradient of the exact GP -- CalcPsiInv + trace + quadratic forms
    (re_model_template.h:6586-6614, 2016-2040), here one partial factorisation of [[Psi, .], [I, 0]] on the device -- and GPB_OptimCovPar on
    top of it.  Pins: central differences of the oracle's exact likelihood (1e-6), and the R suite's own exact-GP fits
    (test_GPModel_gaussian_process.R:130-182): gradient descent + Nesterov 59 iterations, without Nesterov 97, lr_cov = 1 -> 49,
    relative_change_in_parameters 382, lbfgs to the suite's own tolerance."""
    import gpboost_amd
    from oracle import orc
    from scipy.spatial.distance import pdist
    coords, y = orc.r_fixture()
    init = np.array([np.var(y, ddof=1) / 2, np.var(y, ddof=1) / 2, pdist(coords).mean() / 3])
    # gradient against central differences of the oracle's exact likelihood, on the optimiser's scale (log sigma2, log ratio, log a)
    for (n, d, cf, sh, cp) in [(100, 2, "exponential", 0.5, np.array([0.1, 1.6, 0.2])), (700, 3, "matern", 2.5, np.array([0.2, 0.9, 0.3])),
                               (1300, 2, "matern", 1.5, np.array([0.05, 1.2, 0.15]))]:
        c2, y2 = (coords, y) if n == 100 else cases.synthetic(n, d, seed=n)
        mdl = gpboost_amd.GPModel(gp_coords=c2, cov_function=cf, cov_fct_shape=sh)
        nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y2)
        ct = orc.cov_type_id(cf, sh)
        pt = orc.transform_cov_pars(ct, cp)
        assert abs(nll - orc.exact_nll(c2, ct, pt, y2)[2]) <= 1e-8 * abs(nll)
        fd = np.empty(3)
        for k in range(3):
            h = 1e-5
            pp, pm = pt.copy(), pt.copy()
            pp[k] *= np.exp(h); pm[k] *= np.exp(-h)
            fd[k] = (orc.exact_nll(c2, ct, pp, y2)[2] - orc.exact_nll(c2, ct, pm, y2)[2]) / (2 * h)
        np.testing.assert_allclose(grad, fd, rtol=2e-6, atol=2e-6 * np.abs(fd).max())
    gd = dict(optimizer_cov="gradient_descent", lr_cov=0.1, acc_rate_cov=0.5, delta_rel_conv=1e-6, use_nesterov_acc=True, init_cov_pars=init)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential")
    mdl.fit(y, params=gd)
    assert mdl.get_num_optim_iter() == 59
    assert np.abs(mdl.get_cov_pars() - np.array([0.03784221, 1.07390943, 0.11451432])).sum() < 1e-6
    assert abs(mdl.get_current_neg_log_likelihood() - 122.7771373) < 1e-6
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential")
    mdl.fit(y, params=dict(gd, use_nesterov_acc=False))
    assert mdl.get_num_optim_iter() == 97
    assert np.abs(mdl.get_cov_pars() - np.array([0.04040441, 1.06926607, 0.11502362])).sum() < 5e-6
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential")
    mdl.fit(y, params=dict(gd, lr_cov=1.0))
    assert mdl.get_num_optim_iter() == 49
    assert np.abs(mdl.get_cov_pars() - np.array([0.03738147, 1.07520000, 0.11441031])).sum() < 1e-6
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential")
    mdl.fit(y, params=dict(gd, convergence_criterion="relative_change_in_parameters"))
    assert mdl.get_num_optim_iter() == 382
    assert np.abs(mdl.get_cov_pars() - np.array([0.03276547, 1.07617676, 0.11352557])).sum() < 1e-6
    assert abs(mdl.neg_log_likelihood(mdl.get_cov_pars(), y) - 122.7752664) < 1e-6
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential")
    mdl.fit(y, params=dict(optimizer_cov="lbfgs", init_cov_pars=init))
    assert np.abs(mdl.get_cov_pars() - np.array([0.03784221, 1.07390943, 0.11451432])).sum() < 0.02
    assert abs(mdl.get_current_neg_log_likelihood() - 122.7771373) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["r_gd_nesterov_parcrit", "r_mat15_lbfgs", "u1d_n1000_mat15_lbfgs"])
def test_standard_errors_on_device_match_the_reference(lib_built, name):
    """GPB_GetCovPar(calc_std_dev = true) after GPB_OptimCovPar: the stochastic Fisher information of the Gaussian Vecchia model
    (CalcFisherInformation_Vecchia, re_model_template.h:10137-10230) on the device -- level-scheduled solves with B and B' for the whole
    probe block, per-point derivative kernel for dA / dD -- against the reference's own standard errors after its own fit
    (tests/golden/fisher_ref.npz, 1e-5) and the oracle's estimate at the same parameters (1e-7).  R suite: standard errors of the Vecchia fit
    (0.07545639, 0.24785457, 0.03493878), test_GPModel_gaussian_process.R:1318-1322, a different probe set: the suite's own 1e-2."""
    import gpboost_amd
    from oracle import orc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fisher_ref.npz"))
    coords, y, ids, mc, init, cfg = cases.optim_case(name)
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function=mc["cov_function"], cov_fct_shape=mc["shape"], gp_approx="vecchia",
                              num_neighbors=mc["m"], vecchia_ordering=mc["ordering"], seed=mc["seed"])
    mdl.fit(y, params=dict(cfg, init_cov_pars=init))
    out = mdl.get_cov_pars(std_err=True)
    np.testing.assert_allclose(out[:3], g[name + "_cov_pars"], rtol=2e-6)
    np.testing.assert_allclose(out[3:], g[name + "_std"], rtol=1e-5)
    perm, co, nn = orc.vecchia_setup(coords, mc["m"], mc["ordering"], mc["seed"])
    se_o = orc.fisher_std_errors(co, nn, orc.cov_type_id(mc["cov_function"], mc["shape"]), out[:3])
    np.testing.assert_allclose(out[3:], se_o, rtol=1e-7)
    if name == "r_gd_nesterov_parcrit":     # the R suite's own values use 1000 probes of a later run id and are pinned there to 1e-2
        assert np.abs(out[3:] - np.array([0.07545639, 0.24785457, 0.03493878])).sum() < 1e-2


@pytest.mark.gpu
def test_standard_errors_with_more_than_62_neighbours(lib_built):
    """The per-point derivative kernel's 128-lane form (62 < m <= 126).  The R suite's Vecchia model with num_neighbors = n - 1 = 99
    (test_GPModel_gaussian_process.R:1207-1229): its standard errors are those of the exact GP up to the stochastic trace
    (0.07943467, 0.25351519, 0.03840236 at the suite's 1e-2); and the oracle's estimate with the same probes at the same parameters, m = 99
    and a seeded m = 70 case (1e-7)."""
    import gpboost_amd
    from oracle import orc
    coords, y = orc.r_fixture()
    cp = np.array([0.03784221, 1.07390943, 0.11451432])
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=99, vecchia_ordering="none")
    mdl.fit(y, params=dict(optimizer_cov="lbfgs", init_cov_pars=cp, maxit=1))
    out = mdl.get_cov_pars(std_err=True)
    perm, co, nn = orc.vecchia_setup(coords, 99, "none", 1)
    np.testing.assert_allclose(out[3:], orc.fisher_std_errors(co, nn, 0, out[:3]), rtol=1e-7)
    assert np.abs(out[3:] - np.array([0.07943467, 0.25351519, 0.03840236])).sum() < 2e-2
    c2, y2 = cases.synthetic(1200, 2, seed=77)
    md = gpboost_amd.GPModel(gp_coords=c2, cov_function="matern", cov_fct_shape=1.5, gp_approx="vecchia", num_neighbors=70, vecchia_ordering="random", seed=2)
    md.fit(y2, params=dict(optimizer_cov="lbfgs", init_cov_pars=np.array([0.5, 0.5, 0.1]), maxit=3))
    o2 = md.get_cov_pars(std_err=True)
    perm, co, nn = orc.vecchia_setup(c2, 70, "random", 2)
    np.testing.assert_allclose(o2[3:], orc.fisher_std_errors(co, nn, 1, o2[:3]), rtol=1e-7)


@pytest.mark.parametrize("tight", [False, True])
@pytest.mark.parametrize("name", sorted(cases.LAPLACE_AUX_CASES))
def test_host_optimiser_with_an_estimated_shape_parameter_follows_the_reference(lib_built, name, tight):
    """gamma / negative_binomial: the shape is the third entry of the lbfgs vector (log scale; optim_utils.h:256-283), started at
    Likelihood::FindInitialAuxPars (GPB_HIP_FindInitialAuxParsHost) -- the product's host optimiser driven by the oracle against the reference's own
    GPB_OptimCovPar with estimate_aux_pars = true (tests/golden/laplace_aux_ref.npz): same iteration count, estimates 1e-4 (default thresholds) /
    1e-6 (cases.LAPLACE_TIGHT)."""
    from oracle import orc
    from tests import optim_harness as oh
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "laplace_aux_ref.npz"))
    ac = cases.LAPLACE_AUX_CASES[name]
    c = cases.LAPLACE_CASES[ac["model"]]
    coords, y = cases.make_aux_data(ac)
    perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
    ct = orc.cov_type_id(c["cov_function"], c["shape"])
    rc_ = [1.0, np.sqrt(3.0), np.sqrt(5.0)][ct]
    key = name + ("_fit_tight" if tight else "_fit")
    init = g[key + "_init_cov_pars"]
    lib = C.CDLL(lib_built)
    aux0 = np.empty(1)
    yc = np.ascontiguousarray(y)
    assert lib.GPB_HIP_FindInitialAuxParsHost(ac["lik"].encode(), C.c_int(len(y)), yc.ctypes.data_as(C.c_void_p), None, aux0.ctypes.data_as(C.c_void_p)) == 0
    thr = cases.LAPLACE_TIGHT if tight else dict(cg_delta_conv=1e-2, delta_conv_mode_finding=1e-8)
    ev = oh.OracleLaplaceAuxEvaluator(orc, co, nn, ct, y[perm], ac["lik"], cg_delta_conv=thr["cg_delta_conv"], delta_conv_mode=thr["delta_conv_mode_finding"])
    th, aux, nit, nll, ne = oh.optimize_laplace_aux(lib, [init[0], rc_ / init[1]], aux0, ev)
    assert nit == int(g[key + "_num_it"]), (nit, int(g[key + "_num_it"]))
    rtol = 1e-6 if tight else (2e-2 if ac.get("flat_default") else 1e-4)
    np.testing.assert_allclose([th[0], rc_ / th[1]], g[key + "_cov_pars"], rtol=rtol)
    np.testing.assert_allclose(aux, g[key + "_aux"], rtol=rtol)
    assert abs(nll - float(g[key + "_negll"])) <= (1e-8 if tight else 1e-7) * abs(nll)
