"""GPU (MI355X): VALUE parity at BASELINE.json's own sizes.

Three Gaussian configurations -- config 2 (n = 1e5, d = 2, exponential, m = 30), the metric configuration (n = 1e6, d = 2,
exponential, m = 30) and config 5 (n = 1e6, d = 3, Matern-2.5, m = 40) -- are compared
  (i)  with the UNMODIFIED REFERENCE: tests/golden/atsize_ref.npz holds its nll, its gradient, SHA-256 digests of its Vecchia
       ordering and of its WHOLE n x m neighbour table, and 1000 sampled rows of D / y_aux / the table
       (oracle/make_golden.py atsize; src/GPBoost/Vecchia_utils.cpp:733-985, 1367-1699, re_model_template.h:1988-2011);
  (ii) with the CPU oracle on the same inputs in the same run (whole table array_equal, nll and gradient),
both at north_star's tolerances: indices bit-exact, nll and gradient 1e-8 relative.
Config 4 (Bernoulli-logit Vecchia-Laplace, n = 1e5) is compared with ONE evaluation of the reference at that size
(tests/golden/config4_ref.npz; likelihoods.h:3773-4059, CG_utils.cpp:21-229)."""
import hashlib
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-8      # north_star tolerance on nll / gradients

ATSIZE = {
    "config2_n1e5_exp_m30": (100000, 2, 30, "exponential", 0.5, (0.1, 1.0, 0.1)),
    "metric_n1e6_exp_m30": (1000000, 2, 30, "exponential", 0.5, (0.1, 1.0, 0.1)),
    "config5_n1e6_d3_mat25_m40": (1000000, 3, 40, "matern", 2.5, (0.1, 1.0, 0.1)),
}


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gpb(lib_built):
    import gpboost_amd
    assert gpboost_amd.device_count() > 0, "no GPU visible: the -m gpu tests must run on the MI355X box"
    return gpboost_amd


@pytest.mark.parametrize("name", sorted(ATSIZE))
def test_gaussian_configs_at_baseline_size(gpb, orc, name):
    n, d, m, cf, sh, cp = ATSIZE[name]
    cp = np.asarray(cp, dtype=np.float64)
    g = np.load(os.path.join(GOLD, "atsize_ref.npz"))
    coords, y = cases.synthetic(n, d, seed=1)
    mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering="random", seed=1)
    perm, nn = mdl.vecchia_structure()
    # (i) the reference itself
    assert _sha(perm.astype(np.int32)) == str(g[name + "_perm_sha256"]), "Vecchia ordering differs from the reference"
    assert _sha(nn.astype(np.int32)) == str(g[name + "_nn_sha256"]), "the n x m neighbour table is not bit-identical to the reference's"
    rows = g[name + "_rows"]
    assert np.array_equal(nn[rows], g[name + "_nn_rows"])
    nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    ref_nll, ref_grad = float(g[name + "_nll"]), g[name + "_grad"]
    assert abs(nll - ref_nll) <= RTOL * abs(ref_nll), (nll, ref_nll)
    np.testing.assert_allclose(grad, ref_grad, rtol=RTOL, atol=RTOL * np.abs(ref_grad).max())
    nll0 = mdl.neg_log_likelihood(cp, y)                  # the likelihood-only launch (MODE_NLL) as well
    assert abs(nll0 - ref_nll) <= RTOL * abs(ref_nll), (nll0, ref_nll)
    ya = mdl.y_aux(cp, y)
    np.testing.assert_allclose(ya[perm][rows], g[name + "_yaux_rows"], rtol=1e-7, atol=1e-9)
    # (ii) the oracle, same inputs, same run: WHOLE table, nll, gradient
    perm_o, co, nn_o = orc.vecchia_setup(coords, m, "random", 1)
    assert np.array_equal(perm, perm_o)
    assert np.array_equal(nn, nn_o), "neighbour indices must be bit-exact (whole table)"
    ct = orc.cov_type_id(cf, sh)
    out, grad_o = orc.vecchia_nll_grad(co, nn_o, ct, orc.transform_cov_pars(ct, cp), y[perm])
    assert abs(nll - out[2]) <= RTOL * abs(out[2]), (nll, out[2])
    np.testing.assert_allclose(grad, grad_o, rtol=RTOL, atol=RTOL * np.abs(grad_o).max())


@pytest.mark.parametrize("name", ["config5_n1e6_d3_mat25_m40", "metric_n1e6_exp_m30"])
def test_eight_shards_summed_in_rank_order_equal_the_reference(gpb, orc, name):
    """BASELINE config 5 AS WRITTEN ("points sharded across 8 x MI355X with all-reduce") and the metric's N = 8 case, value parity: the
    eight contiguous shards of the Vecchia ordering (parallel.shard_range -- what rank r of an 8-rank job evaluates, gpb_hip_vecchia_set_shard) are evaluated one
    after the other on this device, their 7 sums are added IN RANK ORDER (what the mailbox / the all-reduce does), and likelihood and gradient formed from the
    total (re_model_template.h:3132, :1994-2004) are held to the unmodified reference's values at 1e-8 (tests/golden/atsize_ref.npz: *_nll, *_grad).
    The likelihood-only launch (3 sums) of every shard is checked the same way."""
    from gpboost_amd import shim, parallel
    n, d, m, cf, sh, cp = ATSIZE[name]
    cp = np.asarray(cp, dtype=np.float64)
    g = np.load(os.path.join(GOLD, "atsize_ref.npz"))
    coords, y = cases.synthetic(n, d, seed=1)
    mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering="random", seed=1)
    perm, nn = mdl.vecchia_structure()
    assert _sha(nn.astype(np.int32)) == str(g[name + "_nn_sha256"])
    del mdl
    ct = orc.cov_type_id(cf, sh)
    sigma2, var, a = orc.transform_cov_pars(ct, cp)
    st = shim.VecchiaState(np.ascontiguousarray(coords[perm]), m)
    st.set_neighbors(nn)
    st.set_y(y[perm])
    world = 8
    t7 = np.zeros(7); t3 = np.zeros(3); covered = 0
    for r in range(world):
        i0, i1 = parallel.shard_range(n, r, world)
        assert i0 == covered and i1 > i0
        covered = i1
        st.set_shard(i0, i1)
        t7 += st.grad_terms(ct, var, a)
        t3 += st.nll_terms(ct, var, a)
    assert covered == n
    st.set_shard(0, n)
    ref_nll, ref_grad = float(g[name + "_nll"]), g[name + "_grad"]
    assert t7[2] == 0 and t3[2] == 0
    nll = shim.nll_from_terms(n, t7[0], t7[1], sigma2)
    assert abs(nll - ref_nll) <= RTOL * abs(ref_nll), (nll, ref_nll)
    nll3 = shim.nll_from_terms(n, t3[0], t3[1], sigma2)
    assert abs(nll3 - ref_nll) <= RTOL * abs(ref_nll), (nll3, ref_nll)
    grad = shim.grad_from_terms(n, t7, sigma2)
    np.testing.assert_allclose(grad, ref_grad, rtol=RTOL, atol=RTOL * np.abs(ref_grad).max())
    # and the one-launch evaluation of the same handle: the shards add up to it (1e-12: only the order of the partial sums differs)
    full = st.grad_terms(ct, var, a)
    np.testing.assert_allclose(t7, full, rtol=1e-12, atol=1e-9)


def test_config4_n1e5_against_the_reference(gpb):
    """BASELINE config 4 at its full size against the unmodified reference (tests/golden/config4_ref.npz, oracle/make_golden.py config4).

    Two evaluations of the reference are pinned:
    * cg_delta_conv = 1e-6 (negll_tight_0, 57 s on 8 cores): with the CG residual threshold four orders below the default, the value no
      longer depends on WHICH iteration a rounded residual norm crosses the threshold -- this is the comparison of the arithmetic
      (factor, Newton, CG, Lanczos quadrature with the reference's probe vectors) and it is held to north_star's 1e-8.
    * the defaults (cg_delta_conv = 1e-2, negll_0, 22 s): every Newton system is solved until ||r|| < 1e-2 (CG_utils.cpp:74-82) and the
      log-determinant's block CG stops on the MEAN residual norm of the 50 probes (:196-204), so the value is only defined up to one CG /
      Lanczos iteration: the reference, the C oracle and three versions of this path have given 62930.4025 / .4010 / .4023 / .4016
      (spread 2.4e-8 relative) on these inputs, each moved by kernel values that differ in the last bit.  Admitted here: 5e-8."""
    g = np.load(os.path.join(GOLD, "config4_ref.npz"))
    n, m = 100000, 30
    coords, y = cases.synthetic_binary(n, 2, seed=1)
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia",
                      num_neighbors=m, vecchia_ordering="random", seed=1)
    v = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    ref = float(g["negll_0"])
    assert abs(v - ref) <= 5e-8 * abs(ref), (v, ref, mdl.laplace_info())
    mdl.set_optim_params({"cg_delta_conv": 1e-6})
    vt = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    reft = float(g["negll_tight_0"])
    assert abs(vt - reft) <= RTOL * abs(reft), (vt, reft, mdl.laplace_info())


def test_config4_n1e5_pivoted_cholesky_against_the_reference(gpb):
    """Round 5: the same data with cg_preconditioner_type = "pivoted_cholesky" (rank 50; the (W^-1 + Sigma) form of the solves, pivchol_kernels.hip) against ONE evaluation
    of the unmodified reference with that preconditioner (tests/golden/config4_pivchol_ref.npz, oracle/make_golden.py config4_pivchol): cg_delta_conv = 1e-6 -> 1e-8;
    the defaults -> 1e-6 (the value is defined up to one CG / Lanczos iteration, as above; the (W^-1 + Sigma) residual norms are on another scale than vadu's)."""
    path = os.path.join(GOLD, "config4_pivchol_ref.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/config4_pivchol_ref.npz has not been generated")
    g = np.load(path)
    n, m = 100000, 30
    coords, y = cases.synthetic_binary(n, 2, seed=1)
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia",
                      num_neighbors=m, vecchia_ordering="random", seed=1)
    mdl.set_optim_params({"cg_preconditioner_type": "pivoted_cholesky"})
    v = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    ref = float(g["negll_0"])
    assert abs(v - ref) <= 1e-6 * abs(ref), (v, ref, mdl.laplace_info())
    mdl.set_optim_params({"cg_delta_conv": 1e-6})
    vt = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    reft = float(g["negll_tight_0"])
    assert abs(vt - reft) <= RTOL * abs(reft), (vt, reft, mdl.laplace_info())


def test_config4_n1e5_vecchia_response_against_the_reference(gpb):
    """Round 6: the same data with cg_preconditioner_type = "vecchia_response" (the factor of W^-1 + Sigma renewed by one point-kernel launch per Newton step; gpb_laplace.inc
    pc_refresh_vr) against ONE evaluation of the unmodified reference with that preconditioner per threshold set (tests/golden/config4_vresp_ref.npz, oracle/make_golden.py
    config4_vresp): cg_delta_conv = 1e-6 -> 1e-8; the defaults -> 1e-6 (defined up to one CG / Lanczos iteration)."""
    path = os.path.join(GOLD, "config4_vresp_ref.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/config4_vresp_ref.npz has not been generated")
    g = np.load(path)
    n, m = 100000, 30
    coords, y = cases.synthetic_binary(n, 2, seed=1)
    mdl = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=coords, cov_function="exponential", gp_approx="vecchia",
                      num_neighbors=m, vecchia_ordering="random", seed=1)
    mdl.set_optim_params({"cg_preconditioner_type": "vecchia_response"})
    v = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    ref = float(g["negll_0"])
    assert abs(v - ref) <= 1e-6 * abs(ref), (v, ref, mdl.laplace_info())
    mdl.set_optim_params({"cg_delta_conv": 1e-6})
    vt = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    reft = float(g["negll_tight_0"])
    assert abs(vt - reft) <= RTOL * abs(reft), (vt, reft, mdl.laplace_info())
    print("config 4 with vecchia_response: %.6f (reference %.6f, %.1f s on 8 host cores); device: %s" % (v, ref, float(g["seconds_negll_0"]), mdl.laplace_info()))


@pytest.mark.parametrize("lik", ["lognormal", "gamma", "t"])
def test_auxiliary_parameter_likelihoods_at_config4_size_against_the_reference(gpb, lik):
    """Round 6: the likelihoods with auxiliary parameters at BASELINE config 4's size (n = 1e5, m = 30; smooth latent surface, response drawn from the
    likelihood -- the data of scripts/gpu_r6_targets.py aux:<lik>) against ONE fresh evaluation of the unmodified reference per threshold set
    (tests/golden/config4_size_aux_ref.json, scripts/ref_newton_counts.py: value and Likelihood::num_it_mode_finding_): the value at cg_delta_conv = 1e-6 to 1e-8, at the
    defaults to 1e-6 (defined up to one CG / Lanczos iteration, as config 4 itself), and the NEWTON ITERATION COUNT of the mode finding equal to the reference's -- the 16-18
    Newton steps of the t likelihood (2.2 s per evaluation) are the reference's own on these data, not a different warm start or step capping."""
    import json
    path = os.path.join(GOLD, "config4_size_aux_ref.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/config4_size_aux_ref.json has not been generated")
    ref = json.load(open(path))
    if lik + "_tight" not in ref:
        pytest.skip("no reference values for " + lik)
    n, m = 100000, 30
    rng = np.random.default_rng(7)
    cc = rng.uniform(size=(n, 2))
    eta = np.sin(4 * cc[:, 0]) + np.cos(3 * cc[:, 1])
    if lik == "t":
        y = 0.8 * eta + 0.35 * rng.standard_t(4.0, size=n)
    elif lik == "gamma":
        y = rng.gamma(2.0, np.exp(0.5 * eta) / 2.0)
    else:
        y = np.exp(0.5 * eta + np.sqrt(0.2) * rng.standard_normal(n))
    for key, tol in (("default", 1e-6), ("tight", RTOL)):
        mdl = gpb.GPModel(likelihood=lik, gp_coords=cc, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
        if ref[lik + "_" + key]["thresholds"]:
            mdl.set_optim_params(dict(ref[lik + "_" + key]["thresholds"]))
        v = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
        info = mdl.laplace_info()
        r = ref[lik + "_" + key]
        assert abs(v - r["negll"]) <= tol * abs(r["negll"]), (key, v, r["negll"], info)
        assert int(info["newton_it"]) == int(r["newton_it"]), (key, info["newton_it"], r["newton_it"])


def test_config3_shape_tree_with_categorical_columns_equals_the_oracle_tree(gpb, orc):
    """Round 5, at BASELINE config 3's shape (n = 1e5 rows, 50 columns, 255 bins, 31 leaves) with 6 of the columns categorical (12 / 100 / 250 categories): the
    device's whole-tree grower (categorical search, bitset partitions, resident row lists) against the oracle's primitives driven by tests/tree_harness.py -- the
    oracle is pinned bit for bit to the reference's FeatureHistogram / Dataset::Split on the small fixtures (tests/golden/split_cat_ref.npz, tree_ref_r5.npz) --:
    the same splits, thresholds, sets of bins and counts; leaf values to the summation order of the histogram build; every row in the leaf the tree sends it to."""
    from gpboost_amd import shim
    from tests import tree_harness as th
    n, F, NB, L = 100000, 50, 255, 31
    rng = np.random.default_rng(17)
    X = rng.uniform(size=(n, F))
    cat_cols = {3: 12, 11: 100, 19: 250, 27: 100, 35: 12, 43: 250}
    bins = np.empty((F, n), dtype=np.uint8)
    gnb = np.empty(F, dtype=np.int32); nbin = np.empty(F, dtype=np.int32); mfb = np.zeros(F, dtype=np.int32)
    meta3 = np.zeros((F, 3), dtype=np.int32); is_cat = np.zeros(F, dtype=np.int32)
    signal = np.sin(4 * X[:, 0]) + X[:, 1] ** 2
    for f in range(F):
        if f in cat_cols:
            # a categorical column as the reference stores it: bin 0 = its most frequent category (stored value 0 = most frequent bin), the others 1 .. K - 1
            K = cat_cols[f]
            pr = rng.dirichlet(np.full(K, 0.7)); pr = np.sort(pr)[::-1]
            cat = rng.choice(K, size=n, p=pr)
            bins[f] = cat.astype(np.uint8); gnb[f] = K; nbin[f] = K; is_cat[f] = 1
            meta3[f] = (1, 0, 0)                      # offset 1 (most_freq_bin 0), default bin 0, no missing type
            signal = signal + rng.standard_normal(K)[cat] * (0.6 if K <= 100 else 0.3)
        else:
            bins[f] = np.minimum((X[:, f] * (NB - 1)).astype(np.int64) + 1, NB - 1).astype(np.uint8)
            gnb[f] = NB; nbin[f] = NB; meta3[f] = (1, 0, 0)
    grad = signal + 0.5 * rng.standard_normal(n)
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    voff = (bo[:-1] + 1).astype(np.int32)
    cfg = (0.5, 20, 1e-3, 0.0)
    cat_cfg = (4, 32, 10.0, 10.0, 100)
    be = th.OracleBackend(orc, bins, gnb, voff, nbin, mfb, meta3, grad, None, is_cat=is_cat, cat_cfg=cat_cfg)
    to = th.grow_tree(be, grad, None, n, L, cfg)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(L + 1)
    hb.set_fix_info(voff, nbin, mfb)
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    hb.set_categorical(is_cat, *cat_cfg)
    hb.set_gradients(grad, None)
    sg = float(np.cumsum(grad)[-1])
    t = hb.grow_tree(L, sg, float(n), *cfg)
    assert t["num_leaves"] == to["num_leaves"] == L
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "node_is_cat"):
        assert np.array_equal(t[key], np.asarray(to[key])), key
    assert np.array_equal(t["node_cat_bits"], np.asarray(to["node_cat_bits"]).reshape(-1, 8))
    assert int(t["node_is_cat"].sum()) >= 3 and int((1 - t["node_is_cat"]).sum()) >= 3          # both kinds of splits in the tree
    np.testing.assert_allclose(t["leaf_value"], to["leaf_value"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t["split_gain"], to["split_gain"], rtol=1e-6)
    # size-independent properties: the leaves partition the rows, the counts are the leaf sizes
    dli = t["data_leaf_index"]
    assert dli.min() == 0 and dli.max() == L - 1 and np.array_equal(np.bincount(dli, minlength=L), t["leaf_count"]) and t["leaf_count"].sum() == n
    hb.close()
