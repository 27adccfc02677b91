"""GPU (MI355X): parity of the HIP Vecchia path, through the C ABI, against the CPU oracle and the
reference-generated golden fixtures.  Tolerances follow BASELINE.json's north_star: neighbour indices
bit-exact; fp64 nll and covariance-parameter gradients within 1e-8 relative."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-8      # north_star tolerance on nll / gradients


@pytest.fixture(scope="module")
def gpb(lib_built):
    import gpboost_amd
    assert gpboost_amd.device_count() > 0, "no GPU visible: the -m gpu tests must run on the MI355X box"
    return gpboost_amd


def test_device_is_gfx950_and_dpp_selftest(gpb):
    import ctypes
    yes = ctypes.c_int(0)
    from gpboost_amd.basic import _lib
    assert _lib().gpb_hip_device_is_gfx950(ctypes.byref(yes)) == 0
    assert yes.value == 1
    gpb.selftest()


# ---- golden fixtures (reference outputs) ---------------------------------------------------------------
GPU_GOLDEN = [n for n, c in sorted(cases.GOLDEN_CASES.items()) if c["m"] <= 126]


@pytest.mark.parametrize("name", GPU_GOLDEN)
def test_against_reference_fixture(gpb, name):
    c = cases.GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    coords, y = cases.make_data(c)
    mdl = gpb.GPModel(gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    perm, nn = mdl.vecchia_structure()
    assert np.array_equal(perm, g["perm"]), "Vecchia ordering differs from the reference"
    assert np.array_equal(nn, g["nn"]), "neighbour table is not bit-identical to the reference"
    for k, cp in enumerate(c["cov_pars"]):
        cp = np.asarray(cp, dtype=np.float64)
        nll = mdl.neg_log_likelihood(cov_pars=cp, y=y)
        assert abs(nll - g["nll_%d" % k]) <= RTOL * abs(g["nll_%d" % k]), (nll, g["nll_%d" % k])
        assert mdl.get_current_neg_log_likelihood() == nll
        nll2, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
        assert abs(nll2 - g["nll_%d" % k]) <= RTOL * abs(g["nll_%d" % k])
        np.testing.assert_allclose(grad, g["grad_%d" % k], rtol=RTOL, atol=RTOL * np.abs(g["grad_%d" % k]).max())
        ya = mdl.y_aux(cp, y)
        np.testing.assert_allclose(ya[perm], g["yaux_%d" % k], rtol=1e-7, atol=1e-9)


def test_r_suite_golden_values(gpb, orc):
    """R-package/tests/testthat/test_GPModel_gaussian_process.R:1144-1148 (Vecchia m=30, ordering none: 124.2252524)."""
    coords, y = orc.r_fixture()
    mdl = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30,
                      vecchia_ordering="none")
    assert abs(mdl.neg_log_likelihood(cov_pars=np.array([0.1, 1.6, 0.2]), y=y) - 124.2252524) < 1e-6
    # fixed effects are subtracted from y (EvalNegLogLikelihoodGauss :2905-2916)
    fe = 0.3 * np.cos(np.arange(100.))
    a = mdl.neg_log_likelihood(cov_pars=np.array([0.1, 1.6, 0.2]), y=y + fe, fixed_effects=fe)
    assert abs(a - 124.2252524) < 1e-6


# ---- oracle on seeded inputs ---------------------------------------------------------------------------
ORACLE_CASES = [
    # n, d, m, cov_function, shape, ordering
    (20000, 2, 30, "exponential", 0.5, "random"),
    (20000, 3, 40, "matern", 2.5, "random"),
    (5000, 2, 20, "matern", 1.5, "none"),
    (3000, 1, 5, "exponential", 0.5, "random"),
    (4000, 3, 10, "matern", 1.5, "random"),
    (2000, 2, 50, "exponential", 0.5, "random"),
    (2000, 2, 62, "matern", 2.5, "random"),
    (1500, 2, 63, "exponential", 0.5, "random"),  # first size of the LDS-resident generality kernel (vecchia_big_kernels.hip)
    (1200, 3, 100, "matern", 1.5, "random"),
    (900, 2, 126, "matern", 2.5, "random"),       # its maximum
    (100, 2, 99, "exponential", 0.5, "none"),     # m = n - 1: the Vecchia approximation is exact (R suite, test_GPModel_gaussian_process.R:1104-1111)
    (1500, 2, 1, "exponential", 0.5, "random"),
    (777, 2, 33, "matern", 1.5, "random"),      # m not an instantiated size: padded to 40
    (40, 2, 30, "exponential", 0.5, "random"),   # mostly short rows
    (2, 2, 30, "exponential", 0.5, "none"),      # smallest model
]


@pytest.mark.parametrize("n,d,m,cf,sh,ordering", ORACLE_CASES)
def test_against_oracle(gpb, orc, n, d, m, cf, sh, ordering):
    coords, y = cases.synthetic(n, d, seed=n + m)
    cp = np.array([0.1, 1.0, 0.1])
    mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering=ordering, seed=1)
    perm, nn = mdl.vecchia_structure()
    perm_o, co, nn_o = orc.vecchia_setup(coords, m, ordering, 1)
    assert np.array_equal(perm, perm_o)
    assert np.array_equal(nn, nn_o), "neighbour indices must be bit-exact"
    ct = orc.cov_type_id(cf, sh)
    pt = orc.transform_cov_pars(ct, cp)
    out, grad_o = orc.vecchia_nll_grad(co, nn_o, ct, pt, y[perm])
    nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    assert abs(nll - out[2]) <= RTOL * abs(out[2])
    assert abs(mdl.neg_log_likelihood(cp, y) - out[2]) <= RTOL * abs(out[2])
    np.testing.assert_allclose(grad, grad_o, rtol=RTOL, atol=RTOL * np.abs(grad_o).max())


def test_factor_A_D_u_and_yaux_against_oracle(gpb, orc):
    from gpboost_amd import shim
    coords, y = cases.synthetic(6000, 2, seed=9)
    perm, co, nn = orc.vecchia_setup(coords, 30, "random", 2)
    st = shim.VecchiaState(co, 30)
    st.set_neighbors(nn)
    st.set_y(y[perm])
    for ct, gauss in ((0, True), (2, True), (1, False)):
        var, a = 7.5, 11.0
        st.factor(ct, var, a, gauss=gauss)
        A, D, u = st.get_factor()
        Ao, Do, bad = orc.vecchia_factor(co, nn, ct, var, a, gauss=gauss)
        # without a nugget (gauss=False) D_i = var - c^T C^-1 c cancels to ~1e-3 * var and C_nn is ill-conditioned
        # (jitter 1e-10): the comparison is absolute there
        tol = 1e-10 if gauss else 1e-6
        np.testing.assert_allclose(D, Do, rtol=1e-10 if gauss else 0, atol=0 if gauss else 1e-11 * var)
        np.testing.assert_allclose(A, Ao, rtol=0, atol=tol)
        uo = y[perm] - np.einsum("ij,ij->i", Ao, np.where(nn >= 0, y[perm][np.maximum(nn, 0)], 0.))
        np.testing.assert_allclose(u, uo, rtol=0, atol=tol)
        yo = orc.vecchia_yaux(Ao, Do, nn, y[perm])
        np.testing.assert_allclose(st.yaux(), yo, rtol=10 * tol, atol=tol * np.abs(yo).max())
        t = st.nll_terms(ct, var, a, gauss=gauss)
        assert abs(t[0] - np.sum(uo ** 2 / Do)) <= tol * abs(t[0])
        assert abs(t[1] - np.sum(np.log(Do))) <= tol * max(1., abs(t[1]))
        assert t[2] == 0


@pytest.mark.parametrize("n,d,m,ct", [(6000, 2, 30, 0), (5000, 3, 40, 2), (4000, 1, 10, 1), (3000, 2, 62, 0)])
def test_spatially_sorted_gather_changes_no_bit(gpb, orc, n, d, m, ct):
    """Round 5: the neighbour gathers of the point kernel read a Morton-sorted copy of the records through a rewritten neighbour table
    (gpb_hip_vecchia_set_sorted_gather; default for n >= 32768 from the third evaluation).  Forced on for these small cases: likelihood terms, gradient
    terms, factor (A, D, u) and y_aux are bit-identical to the plain gather -- before and after a new response (the copy follows), and for a shard."""
    from gpboost_amd import shim
    coords, y = cases.synthetic(n, d, seed=31 + d)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 2)
    res = {}
    for mode in (0, 1):
        st = shim.VecchiaState(co, m)
        st.set_neighbors(nn)
        st.set_sorted_gather(mode)
        st.set_y(y[perm])
        out = [st.nll_terms(ct, 3.0, 9.0), st.grad_terms(ct, 3.0, 9.0), st.nll_terms(ct, 0.7, 4.0, gauss=False)]
        st.factor(ct, 3.0, 9.0)
        out += list(st.get_factor()) + [st.yaux()]
        st.set_y(np.cos(3 * y[perm]))                      # a new response: the sorted copy is renewed before the next launch
        out += [st.nll_terms(ct, 3.0, 9.0), st.grad_terms(ct, 2.0, 7.0)]
        i0, i1 = 16 * (n // 48), 16 * (n // 24)
        st.set_shard(i0, i1)
        out += [st.nll_terms(ct, 3.0, 9.0)]
        res[mode] = out
        st.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_new_response_keeps_the_factor_and_renews_u(gpb, orc):
    """The GPBoost algorithm hands over a new response (F - y) every boosting iteration at unchanged covariance parameters (CalcGradientF,
    re_model_template.h:3313-3316: SetY, CalcYAux): A, D stay on the device, u = B y is renewed by one pass over A -- no refactorisation."""
    from gpboost_amd import shim
    coords, y = cases.synthetic(5000, 2, seed=4)
    perm, co, nn = orc.vecchia_setup(coords, 20, "random", 2)
    st = shim.VecchiaState(co, 20)
    st.set_neighbors(nn)
    st.set_y(y[perm])
    st.factor(2, 3.0, 9.0)
    A0, D0, _ = st.get_factor()
    Ao, Do, _ = orc.vecchia_factor(co, nn, 2, 3.0, 9.0)
    rng = np.random.default_rng(3)
    for _ in range(2):
        y2 = rng.standard_normal(len(y))
        st.set_y(y2)
        yo = orc.vecchia_yaux(Ao, Do, nn, y2)
        np.testing.assert_allclose(st.yaux(), yo, rtol=1e-9, atol=1e-10 * np.abs(yo).max())
        A, D, u = st.get_factor()
        assert np.array_equal(A, A0) and np.array_equal(D, D0)
        uo = y2 - np.einsum("ij,ij->i", Ao, np.where(nn >= 0, y2[np.maximum(nn, 0)], 0.))
        np.testing.assert_allclose(u, uo, rtol=0, atol=1e-10)


def test_device_neighbor_search_equals_host_table(gpb, orc):
    """find_neighbors (device) == set_neighbors(oracle table): both routes give identical likelihood terms."""
    from gpboost_amd import shim
    coords, y = cases.synthetic(9000, 3, seed=21)
    perm, co, nn = orc.vecchia_setup(coords, 25, "random", 4)
    a = shim.VecchiaState(co, 25); dup = a.find_neighbors()
    assert not dup
    assert np.array_equal(a.get_neighbors(), nn)
    b = shim.VecchiaState(co, 25); b.set_neighbors(nn)
    a.set_y(y[perm]); b.set_y(y[perm])
    assert np.array_equal(a.nll_terms(2, 3.0, 9.0), b.nll_terms(2, 3.0, 9.0))   # deterministic reduction: bitwise


def test_duplicates_flag_and_ties(gpb, orc):
    from gpboost_amd import shim
    c = cases.GOLDEN_CASES["dup2d_n600_exp_m15"]
    coords, y = cases.make_data(c)
    perm, co, nn = orc.vecchia_setup(coords, 15, "random", 5)
    st = shim.VecchiaState(co, 15)
    assert st.find_neighbors() is True
    assert np.array_equal(st.get_neighbors(), nn)


def test_shards_add_up_and_results_are_reproducible(gpb, orc):
    from gpboost_amd import shim, parallel
    coords, y = cases.synthetic(30011, 2, seed=33)
    perm, co, nn = orc.vecchia_setup(coords, 30, "random", 1)
    st = shim.VecchiaState(co, 30); st.set_neighbors(nn); st.set_y(y[perm])
    full = st.grad_terms(0, 10.0, 10.0)
    again = st.grad_terms(0, 10.0, 10.0)
    assert np.array_equal(full, again), "fixed-order reductions must be bit-reproducible"
    for world in (2, 3, 8):
        acc = np.zeros(7)
        for r in range(world):
            i0, i1 = parallel.shard_range(st.n, r, world)
            st.set_shard(i0, i1)
            acc += st.grad_terms(0, 10.0, 10.0)
        st.set_shard(0, st.n)
        np.testing.assert_allclose(acc, full, rtol=1e-12, atol=1e-9)


def test_full_size_properties_n1e6(gpb, orc):
    """BASELINE.json's metric size (n = 1e6, m = 30, d = 2, exponential): properties that need no oracle pass --
    the quadratic form is quadratic in y, the log-determinant does not depend on y, shards add up, and a
    strided sample of rows of the factor matches the oracle evaluated on those rows only."""
    from gpboost_amd import shim, parallel
    n, m = 1_000_000, 30
    coords, y = cases.synthetic(n, 2, seed=1)
    st = shim.VecchiaState(coords, m)          # ordering "none": coords already in (random) order
    assert st.find_neighbors() is False
    nn = st.get_neighbors()
    # neighbour rows: strictly earlier indices, sorted by distance, and exact on a sample (brute force)
    assert (nn[m + 1:] < np.arange(m + 1, n)[:, None]).all() and (nn[m + 1:] >= 0).all()
    rng = np.random.default_rng(0)
    for i in rng.integers(m + 1, n, size=25):
        d2 = ((coords[:i] - coords[i]) ** 2).sum(1)
        assert set(np.argsort(d2, kind="stable")[:m]) == set(nn[i]), i
        assert (np.diff(d2[nn[i]]) >= 0).all()
    var, a = 10.0, 10.0
    st.set_y(y)
    t1 = st.nll_terms(0, var, a)
    st.set_y(2.0 * y)
    t2 = st.nll_terms(0, var, a)
    assert t1[2] == 0 and t2[2] == 0
    assert abs(t2[0] - 4.0 * t1[0]) <= 1e-12 * abs(t2[0])
    assert t2[1] == t1[1]
    st.set_y(y)
    acc = np.zeros(3)
    for r in range(8):
        st.set_shard(*parallel.shard_range(n, r, 8)); acc += st.nll_terms(0, var, a)
    st.set_shard(0, n)
    np.testing.assert_allclose(acc[:2], t1[:2], rtol=1e-12)
    # sample rows against the oracle (oracle evaluates only the sampled rows: sub-problem with remapped indices)
    rows = rng.integers(0, n, size=300)
    st.factor(0, var, a)
    A, D, u = st.get_factor()
    for i in rows:
        idx = nn[i][nn[i] >= 0]
        sub = np.concatenate([coords[idx], coords[i:i + 1]])
        k = len(idx)
        nn_sub = np.full((k + 1, max(k, 1)), -1, dtype=np.int32)
        nn_sub[k, :k] = np.arange(k)
        Ao, Do, _ = orc.vecchia_factor(sub, nn_sub, 0, var, a)
        assert abs(D[i] - Do[k]) <= 1e-10 * Do[k]
        np.testing.assert_allclose(A[i, :k], Ao[k, :k], rtol=0, atol=1e-10)


def test_errors_are_loud(gpb):
    from gpboost_amd import shim
    coords, y = cases.synthetic(300, 2, seed=2)
    with pytest.raises(gpb.GPBoostError):
        gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=200)   # > 126
    with pytest.raises(gpb.GPBoostError):
        gpb.GPModel(gp_coords=np.random.default_rng(0).uniform(size=(100, 11)), cov_function="exponential",
                    gp_approx="vecchia", num_neighbors=10)                                            # d > 10
    st = shim.VecchiaState(coords, 10)
    with pytest.raises(gpb.GPBoostError):
        st.nll_terms(0, 1.0, 1.0)        # neighbours not set
    st.find_neighbors()
    with pytest.raises(gpb.GPBoostError):
        st.nll_terms(0, 1.0, 1.0)        # y not set
    st.set_y(y)
    with pytest.raises(gpb.GPBoostError):
        st.nll_terms(0, -1.0, 1.0)
    with pytest.raises(gpb.GPBoostError):
        st.yaux()                        # factor not computed
    mdl = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    with pytest.raises(gpb.GPBoostError):
        mdl.neg_log_likelihood(cov_pars=np.array([0.1, -1.0, 0.1]), y=y)


# ---- exact GP (BASELINE config 1) -------------------------------------------------------------------------
def test_exact_gp_r_suite_golden_values(gpb, orc):
    """R-package/tests/testthat/test_GPModel_gaussian_process.R:86-120: 124.2549533, 141.3502172, 158.1111626."""
    coords, y = orc.r_fixture()
    for cf, sh, gold in (("exponential", 0.5, 124.2549533), ("matern", 1.5, 141.3502172), ("matern", 2.5, 158.1111626)):
        mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="none")
        assert abs(mdl.neg_log_likelihood(cov_pars=np.array([0.1, 1.6, 0.2]), y=y) - gold) < 1e-6


@pytest.mark.parametrize("n,d,ct", [(1500, 2, 1), (777, 3, 2), (64, 1, 0), (130, 2, 0)])
def test_exact_gp_against_oracle(gpb, orc, n, d, ct):
    from gpboost_amd import shim
    coords, y = cases.synthetic(n, d, seed=n)
    var, a = 10.0, 10.0 * [1., 3. ** .5, 5. ** .5][ct]
    st = shim.ExactState(coords); st.set_y(y)
    out, ya, ms = st.nll_terms(ct, var, a, want_yaux=True)
    o, yo = orc.exact_nll(coords, ct, np.array([0.1, var, a]), y, want_yaux=True)
    assert abs(out[0] - o[0]) <= RTOL * abs(o[0]) and abs(out[1] - o[1]) <= RTOL * max(1., abs(o[1]))
    np.testing.assert_allclose(ya, yo, rtol=1e-7, atol=1e-9 * np.abs(yo).max())
    mdl = gpb.GPModel(gp_coords=coords, cov_function="matern", cov_fct_shape=[0.5, 1.5, 2.5][ct], gp_approx="none")
    cp = np.array([0.1, 1.0, [1., 3. ** .5, 5. ** .5][ct] / a])
    assert abs(mdl.neg_log_likelihood(cp, y) - o[2]) <= RTOL * abs(o[2])
    np.testing.assert_allclose(mdl.y_aux(cp, y), yo, rtol=1e-7, atol=1e-9 * np.abs(yo).max())


def test_several_models_alive_and_interleaved(gpb, orc):
    """REModel has no locking but several handles may be alive (e.g. the temporary model of
    InitCoefAuxParsFromIidModel, re_model.cpp:402-409): interleaved calls must not disturb each other."""
    c1, y1 = cases.synthetic(3000, 2, seed=101)
    c2, y2 = cases.synthetic(2500, 3, seed=102)
    m1 = gpb.GPModel(gp_coords=c1, cov_function="exponential", gp_approx="vecchia", num_neighbors=20, seed=3)
    m2 = gpb.GPModel(gp_coords=c2, cov_function="matern", cov_fct_shape=2.5, gp_approx="vecchia", num_neighbors=35, seed=4)
    m3 = gpb.GPModel(gp_coords=c1[:500], cov_function="matern", cov_fct_shape=1.5, gp_approx="none")
    cp = np.array([0.2, 1.0, 0.15])
    a1 = m1.neg_log_likelihood(cp, y1); a2 = m2.neg_log_likelihood(cp, y2); a3 = m3.neg_log_likelihood(cp, y1[:500])
    for _ in range(3):
        assert m2.neg_log_likelihood(cp, y2) == a2
        assert m1.neg_log_likelihood(cp, y1) == a1
        assert m3.neg_log_likelihood(cp, y1[:500]) == a3
    assert abs(a1 - orc.gp_nll(c1, y1, cp, "exponential", 0.5, 20, "random", 3)) <= RTOL * abs(a1)
    assert abs(a2 - orc.gp_nll(c2, y2, cp, "matern", 2.5, 35, "random", 4)) <= RTOL * abs(a2)
    del m1
    assert m2.neg_log_likelihood(cp, y2) == a2


def test_sharded_yaux_and_dev_entry_points(gpb, orc):
    """Multi-GPU composition on one device: per-shard partial y_aux vectors add up to y_aux; the *_dev entry points
    (device output buffers, asynchronous) agree with the host-returning ones."""
    from gpboost_amd import shim, parallel
    coords, y = cases.synthetic(20011, 2, seed=77)
    perm, co, nn = orc.vecchia_setup(coords, 30, "random", 1)
    st = shim.VecchiaState(co, 30); st.set_neighbors(nn); st.set_y(y[perm])
    var, a = 10.0, 10.0
    st.factor(0, var, a)
    full = st.yaux()
    Ao, Do, _ = orc.vecchia_factor(co, nn, 0, var, a)
    np.testing.assert_allclose(full, orc.vecchia_yaux(Ao, Do, nn, y[perm]), rtol=1e-9, atol=1e-10)
    buf = shim.DeviceBuffer(st.n)
    acc = np.zeros(st.n)
    for r in range(4):
        st.set_shard(*parallel.shard_range(st.n, r, 4))
        st.factor(0, var, a)
        st.yaux_partial_dev(buf.ptr)
        acc += buf.to_host()
    st.set_shard(0, st.n)
    np.testing.assert_allclose(acc, full, rtol=1e-11, atol=1e-12)
    t3 = shim.DeviceBuffer(3); t7 = shim.DeviceBuffer(7)
    st.nll_terms_dev(0, var, a, t3.ptr); st.grad_terms_dev(0, var, a, t7.ptr)
    assert np.array_equal(t3.to_host(), st.nll_terms(0, var, a))
    assert np.array_equal(t7.to_host(), st.grad_terms(0, var, a))


@pytest.mark.parametrize("offset,scale", [(5.0e6, 1.0e3), (0.0, 1.0e-6), (-3.0e4, 1.0), (1.0e9, 1.0e5)])
def test_coordinate_offsets_and_scales(gpb, orc, offset, scale):
    """Projected coordinates (large common offset, metre-scale differences) and tiny/huge units: the kernel centres every
    neighbourhood on its point before scaling, so accuracy follows the neighbourhood size, not the absolute coordinates."""
    coords, y = cases.synthetic(4000, 2, seed=5)
    coords = offset + scale * coords
    rho = 0.1 * scale
    cp = np.array([0.1, 1.0, rho])
    mdl = gpb.GPModel(gp_coords=coords, cov_function="matern", cov_fct_shape=1.5, gp_approx="vecchia", num_neighbors=30, seed=2)
    perm, nn = mdl.vecchia_structure()
    perm_o, co, nn_o = orc.vecchia_setup(coords, 30, "random", 2)
    assert np.array_equal(perm, perm_o) and np.array_equal(nn, nn_o)
    out, grad_o = orc.vecchia_nll_grad(co, nn_o, 1, orc.transform_cov_pars(1, cp), y[perm])
    nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    assert abs(nll - out[2]) <= RTOL * abs(out[2])
    np.testing.assert_allclose(grad, grad_o, rtol=RTOL, atol=RTOL * np.abs(grad_o).max())


@pytest.mark.parametrize("cp", [(1e-3, 5.0, 0.5), (10.0, 0.01, 0.001), (0.5, 1.0, 50.0), (1e-6, 1.0, 0.05)])
def test_extreme_covariance_parameters(gpb, orc, cp):
    """Tiny nugget (ill-conditioned blocks), tiny signal, very long and very short ranges."""
    coords, y = cases.synthetic(3000, 2, seed=8)
    cp = np.array(cp)
    for cf, sh, ct in (("exponential", 0.5, 0), ("matern", 2.5, 2)):
        mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=20, seed=1)
        perm, nn = mdl.vecchia_structure()
        out, grad_o = orc.vecchia_nll_grad(coords[perm], nn, ct, orc.transform_cov_pars(ct, cp), y[perm])
        nll, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
        # conditioning of the m x m blocks scales with sigma1^2/sigma^2: allow 1e-8 relative up to ratio 1e4, looser beyond
        tol = RTOL * max(1.0, cp[1] / cp[0] / 1e4)
        assert abs(nll - out[2]) <= tol * abs(out[2]), (cf, nll, out[2])
        np.testing.assert_allclose(grad, grad_o, rtol=tol, atol=tol * np.abs(grad_o).max())


def test_in_library_rccl_allreduce_single_rank(gpb, orc):
    """The RCCL path with a 1-rank communicator: unique id -> ncclCommInitRank -> kernel + reduction + ncclAllReduce on one
    stream; must reproduce the plain evaluation bit for bit (multi-rank composition: tests/test_distributed_cpu.py + bench.py)."""
    from gpboost_amd import shim
    coords, y = cases.synthetic(5000, 2, seed=12)
    perm, co, nn = orc.vecchia_setup(coords, 30, "random", 1)
    st = shim.VecchiaState(co, 30); st.set_neighbors(nn); st.set_y(y[perm])
    uid = shim.comm_unique_id()
    assert len(uid) == 128
    st.comm_init(uid, 0, 1)
    assert np.array_equal(st.nll_terms_allreduce(0, 10.0, 10.0), st.nll_terms(0, 10.0, 10.0))
    assert np.array_equal(st.grad_terms_allreduce(0, 10.0, 10.0), st.grad_terms(0, 10.0, 10.0))
    with pytest.raises(gpb.GPBoostError):
        shim.VecchiaState(co, 30).nll_terms_allreduce(0, 10.0, 10.0)      # no neighbours / no communicator


# ---- Newton update of the leaf values (row a9) ---------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.LEAF_CASES))
def test_newton_leaf_values_against_reference_fixture(gpb, name):
    """GPB_HIP_NewtonUpdateLeafValues (factor + y_aux + H^T Psi^-1 H on the device, L x L solve on the host) against the
    reference's own REModelTemplate::NewtonUpdateLeafValues (tests/golden, oracle/ref_driver.cpp:refdrv_newton_leaf)."""
    c = cases.GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    coords, y = cases.make_data(c)
    leaf, L = cases.make_leaf_index(name, len(y))
    mdl = gpb.GPModel(gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    vals = mdl.newton_update_leaf_values(np.asarray(c["cov_pars"][0], dtype=np.float64), y, leaf, L)
    np.testing.assert_allclose(vals, g["leaf_values_0"], rtol=1e-8, atol=1e-10)
    # the reference's calling sequence: the gradient call (y_aux) leaves factor and y_aux behind, the leaf update reuses them
    with pytest.raises(gpb.GPBoostError, match="y_aux has not been calculated"):
        mdl.neg_log_likelihood(np.asarray(c["cov_pars"][0], dtype=np.float64), y)
        mdl.newton_update_leaf_values(None, None, leaf, L)
    mdl.y_aux(np.asarray(c["cov_pars"][0], dtype=np.float64), y)
    assert np.array_equal(mdl.newton_update_leaf_values(None, None, leaf, L), vals)


@pytest.mark.parametrize("n,m,L", [(20000, 30, 31), (5000, 20, 64), (3000, 10, 3)])
def test_newton_leaf_values_against_oracle(gpb, orc, n, m, L):
    from gpboost_amd import shim
    coords, y = cases.synthetic(n, 2, seed=7 + L)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 2)
    rng = np.random.default_rng(L)
    leaf = rng.integers(0, L, size=n).astype(np.int32)
    var, a = 8.0, 1.0 / 0.07
    st = shim.VecchiaState(co, m)
    st.set_neighbors(nn)
    st.set_y(y[perm])
    with pytest.raises(gpb.GPBoostError):
        st.newton_leaf_values(leaf, L)               # factor / y_aux missing
    st.factor(0, var, a)
    ya = st.yaux()
    vals = st.newton_leaf_values(leaf, L)
    A, D, bad = orc.vecchia_factor(co, nn, 0, var, a)
    ref = orc.newton_leaf_values(A, D, nn, orc.vecchia_yaux(A, D, nn, y[perm]), leaf, L)
    np.testing.assert_allclose(vals, ref, rtol=1e-8, atol=1e-10)
    assert np.array_equal(vals, st.newton_leaf_values(leaf, L))          # reproducible
    with pytest.raises(gpb.GPBoostError):
        st.newton_leaf_values(leaf, 65)
    with pytest.raises(gpb.GPBoostError):
        st.newton_leaf_values(leaf + 1, L)           # an index == L
    st.close()


# ---- multi-GPU pieces on one device: parts of the neighbour search, 1-rank RCCL merges -------------------
def test_neighbor_search_parts_and_rccl_merges_single_rank(gpb, orc):
    from gpboost_amd import shim
    n, m = 30000, 20
    coords, y = cases.synthetic(n, 2, seed=21)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 3)
    st = shim.VecchiaState(co, m)
    merged = np.full((n, m), np.iinfo(np.int32).min, dtype=np.int32)
    for part in range(3):                                  # what three ranks would compute, merged as the max-all-reduce does
        st.find_neighbors_part(part, 3)
        tab = st.get_neighbors()
        merged = np.maximum(merged, tab)
        searched = (tab >= -1).all(axis=1)
        assert n / 3 - m - 2 <= searched[m + 1:].sum() <= n / 3 + 1      # equal blocks of positions (minus head rows)
        with pytest.raises(gpb.GPBoostError):
            st.nll_terms(0, 10.0, 10.0)                    # a partial table is not usable
    assert np.array_equal(merged, nn)
    # 1-rank communicator: part 0 of 1 + max-all-reduce == plain search; y_aux all-reduce == y_aux
    st.comm_init(shim.comm_unique_id(), 0, 1)
    st.find_neighbors_part(0, 1)
    st.neighbors_allreduce()
    assert np.array_equal(st.get_neighbors(), nn)
    st.set_y(y[perm]); st.factor(0, 10.0, 10.0)
    assert np.array_equal(st.yaux_allreduce(), st.yaux())
    st.close()


# ---- prediction at new locations, conditioning on the observed points only --------------------------
def test_prediction_r_suite_golden_values(gpb):
    """test_GPModel_gaussian_process.R:1326-1333 (mean and response variance at three locations, two of them 1e-5 apart)."""
    from tests.test_oracle_golden import R_PRED_COV_PARS, R_PRED_COORDS, R_PRED_MU, R_PRED_VAR
    from oracle import orc
    coords, y = orc.r_fixture()
    mdl = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none")
    pred = mdl.predict(y, R_PRED_COORDS, R_PRED_COV_PARS, predict_var=True, predict_response=True, num_neighbors_pred=30)
    assert np.abs(pred["mu"] - R_PRED_MU).sum() < 1e-6
    assert np.abs(pred["var"] - R_PRED_VAR).sum() < 1e-6
    lat = mdl.predict(y, R_PRED_COORDS, R_PRED_COV_PARS, predict_var=True, predict_response=False)
    np.testing.assert_allclose(pred["var"] - lat["var"], R_PRED_COV_PARS[0], rtol=1e-10)


@pytest.mark.parametrize("n,npred,d,m,ct,ordering", [(20000, 5000, 2, 30, 0, "random"), (3000, 700, 3, 15, 2, "random"), (500, 33, 1, 10, 1, "none"),
                                                     (1200, 150, 5, 20, 1, "random"), (300, 40, 10, 70, 0, "none")])      # d > 3: the generality path
def test_prediction_against_oracle(gpb, orc, n, npred, d, m, ct, ordering):
    cf, sh = {0: ("exponential", 0.5), 1: ("matern", 1.5), 2: ("matern", 2.5)}[ct]
    coords, y = cases.synthetic(n, d, seed=31 + n)
    rng = np.random.default_rng(n)
    cpred = rng.uniform(size=(npred, d))
    cpred[:3] = coords[:3]                                   # a few prediction points ON observed points
    cp = np.array([0.2, 1.1, 0.15])
    mdl = gpb.GPModel(gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m,
                      vecchia_ordering=ordering, seed=4)
    pred = mdl.predict(y, cpred, cp, predict_var=True, predict_response=False, num_neighbors_pred=m)
    perm, _ = mdl.vecchia_structure()
    mu, var = orc.predict_obs_only(coords[perm], y[perm], cpred, ct, orc.transform_cov_pars(ct, cp), m, predict_response=False)
    np.testing.assert_allclose(pred["mu"], mu, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(pred["var"], var, rtol=1e-8, atol=1e-10)
    # neighbour tables of the prediction rows are bit-identical to the reference's search (through the shim)
    from gpboost_amd import shim
    st = shim.VecchiaState.from_handle(mdl.vecchia_handle(), n, d, m)
    mu2, Dp, dup = st.predict_obs_only(cpred, m, ct, cp[1] / cp[0], {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[ct] / cp[2])
    assert dup                                               # the three coinciding points
    np.testing.assert_allclose(mu2, mu, rtol=1e-8, atol=1e-10)


def test_dimension_limits_fail_loudly(gpb):
    """d = 4 .. 10 runs (Gaussian Vecchia: search, likelihood, gradient, fit, prediction -- the fixtures above); what does not is rejected by name."""
    rng = np.random.default_rng(5)
    with pytest.raises(gpb.GPBoostError, match="coordinate dimension 11"):
        gpb.GPModel(gp_coords=rng.uniform(size=(50, 11)), cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    c5 = rng.uniform(size=(400, 5)); yb = (rng.uniform(size=400) < 0.5).astype(np.float64)
    mb = gpb.GPModel(likelihood="bernoulli_logit", gp_coords=c5, cov_function="exponential", gp_approx="vecchia", num_neighbors=15, seed=1)
    assert np.isfinite(mb.neg_log_likelihood(np.array([1.0, 0.5]), yb))       # the Laplace value needs only the factor: any d
    with pytest.raises(gpb.GPBoostError, match="coordinate dimensions 1..3"):
        mb.fit(yb)                                                             # its gradient uses the per-point derivative kernel
    mg = gpb.GPModel(gp_coords=c5, cov_function="exponential", gp_approx="vecchia", num_neighbors=15, seed=1)
    mg.fit(rng.standard_normal(400))
    with pytest.raises(gpb.GPBoostError, match="coordinate dimensions 1..3"):
        mg.get_cov_pars(std_err=True)


# ---- several clusters (independent realisations of the GP, cluster_ids) -------------------------------------------------------
def test_cluster_ids_r_golden_and_reference_fixture(gpb, orc):
    coords, y = orc.r_fixture()
    ids = np.r_[np.ones(40), 2 * np.ones(60)].astype(np.int32)
    mdl = gpb.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="none",
                      cluster_ids=ids)
    nll = mdl.neg_log_likelihood(np.array([0.05870373, 1.05572659, 0.12775754]), y)
    assert abs(nll - 129.3761486) < 1e-6                      # test_GPModel_gaussian_process.R:1638-1648
    # random ordering: clusters in order of first appearance, shuffled one after the other by ONE std::mt19937(seed)
    c = cases.CLUSTER_CASE
    coords, y, ids = cases.make_cluster_data()
    ref = float(np.load(os.path.join(GOLD, "clusters_ref.npz"))["nll"])
    mdl = gpb.GPModel(gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"], cluster_ids=ids)
    cp = np.asarray(c["cov_pars"], dtype=np.float64)
    nll = mdl.neg_log_likelihood(cp, y)
    assert abs(nll - ref) <= RTOL * abs(ref)
    # gradient and y_aux add up over clusters too: finite-difference check of the gradient, y_aux . y = quadratic form
    nll2, grad = mdl.neg_log_likelihood_and_gradient(cp, y)
    assert nll2 == nll
    eps = 1e-6
    lp = np.log(np.array([cp[0], cp[1] / cp[0], 1.0 / cp[2]]))
    def f(l):
        s2, v, a = np.exp(l)
        return mdl.neg_log_likelihood(np.array([s2, v * s2, 1.0 / a]), y)
    for k in range(3):
        e = np.zeros(3); e[k] = eps
        assert abs((f(lp + e) - f(lp - e)) / (2 * eps) - grad[k]) <= 1e-5 * max(1.0, abs(grad[k]))
    with pytest.raises(gpb.GPBoostError):
        mdl.vecchia_structure()


def test_batched_evaluations_equal_single_evaluations(gpb):
    """GPB_HIP_EvalNegLogLikelihoodBatch (K parameter sets, one synchronisation) against K calls of GPB_EvalNegLogLikelihood on the resident
    response (y_data = NULL): bitwise equal -- same kernels, same fixed-order reductions."""
    coords, y = cases.synthetic(30000, 2, seed=77)
    mdl = gpb.GPModel(gp_coords=coords, cov_function="matern", cov_fct_shape=1.5, gp_approx="vecchia", num_neighbors=25, vecchia_ordering="random", seed=3)
    base = np.array([0.2, 0.9, 0.12])
    first = mdl.neg_log_likelihood(base, y)                 # uploads y once
    cps = np.stack([base * (1 + 0.01 * k) for k in range(9)])
    single = np.array([mdl.neg_log_likelihood(cp) for cp in cps])
    assert single[0] == first
    batch = mdl.neg_log_likelihood_batch(cps)
    assert np.array_equal(batch, single)
    with pytest.raises(gpb.GPBoostError):
        gpb.GPModel(gp_coords=coords[:100], cov_function="exponential", gp_approx="vecchia", num_neighbors=5).neg_log_likelihood_batch(cps)   # no response set
