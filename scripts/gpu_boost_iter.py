"""BASELINE config 3 on the MI355X: the device work of ONE GPBoost iteration (Gaussian likelihood, Vecchia GP, tree of 31 leaves on
F = 50 features with 255 bins), every primitive through the C ABI:
  1. gradient           y_aux = Psi^-1 (F - y)                 GPB_HIP_CalcYAux            (regression_objective.hpp:153-201)
  2. tree growth        histograms / fix / subtract / split search / partition with resident row lists: gpb_hip_hist_grow_tree
                        (SerialTreeLearner::Train's control flow; 2b = the same through the single-step calls from tests/tree_harness.py)
  3. Newton leaf values (H' Psi^-1 H)^-1 (-H' y_aux)           GPB_HIP_NewtonUpdateLeafValues
  4. covariance parameters: one accelerated gradient step      GPB_OptimCovPar, maxit = 1   (gbdt.cpp: re-estimation every iteration)
Bins are synthetic in the layout the reference's Dataset uses for a feature whose most frequent bin is 0 (stored bin = feature bin,
histogram view starts at stored bin 1: tests/golden/tree_ref.npz, features 0 / 4 / 5)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd                      # noqa: E402
from gpboost_amd import shim            # noqa: E402
from tests import tree_harness as th    # noqa: E402


def synthetic_dataset(n, F, nb, rng):
    X = rng.uniform(size=(n, F))
    bins = np.minimum((X * (nb - 1)).astype(np.int64) + 1, nb - 1).astype(np.uint8).T.copy()      # stored bins 1 .. nb-1, (F, n)
    gnb = np.full(F, nb, dtype=np.int32)
    start = np.concatenate([[0], np.cumsum(gnb)[:-1]]).astype(np.int32)
    return X, bins, gnb, (start + 1).astype(np.int32), np.full(F, nb, dtype=np.int32), np.zeros(F, dtype=np.int32), \
        np.tile(np.array([1, 0, 0], dtype=np.int32), (F, 1))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    F, nb, L, m = 50, 255, 31, 30
    gpboost_amd.set_device(0)
    rng = np.random.default_rng(1)
    coords = rng.uniform(size=(n, 2))
    X, bins, gnb, voff, num_bin, mfb, meta3 = synthetic_dataset(n, F, nb, rng)
    y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + 0.5 * rng.standard_normal(n)
    cov_pars = np.array([0.25, 0.1, 0.1])
    mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
    mdl.set_optim_params({"optimizer_cov": "gradient_descent", "maxit": 1, "init_cov_pars": cov_pars})
    cfg = (0.0, 20, 1e-3, 0.0)                     # lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split
    score = np.zeros(n)
    out = dict(n=n, F=F, bins=nb, num_leaves=L, m=m)
    hb = None
    niter = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for it in range(niter):                        # iteration 0 warms clocks / allocations; report the last
        t = {}
        t0 = time.perf_counter()
        grad = mdl.y_aux(mdl.get_cov_pars() if it else cov_pars, score - y)        # gradient of the Gaussian GPBoost objective
        t["1_gradient_yaux_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        if hb is None:
            bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
            hb = shim.HistBuilder(bins, bo)
            hb.pool_resize(L + 1); hb.set_fix_info(voff, num_bin, mfb); hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
        hb.set_gradients(grad, None)
        sg = float(np.cumsum(grad)[-1])
        t1 = time.perf_counter()
        tree = hb.grow_tree(L, sg, float(n), *cfg)
        t["2a_grow_tree_call_only_ms"] = (time.perf_counter() - t1) * 1e3
        t["2_tree_growth_ms"] = (time.perf_counter() - t0) * 1e3
        leaf_of, nleaves = tree["data_leaf_index"], tree["num_leaves"]
        if it == niter - 1:                          # the same tree through the single-step entry points driven from Python (harness)
            t0 = time.perf_counter()
            be = th.GpuBackend(shim, bins, gnb, voff, num_bin, mfb, meta3, grad, None, L)
            t0 = time.perf_counter()
            tree_h = th.grow_tree(be, grad, None, n, L, cfg)
            t["2b_tree_growth_single_step_calls_from_python_ms"] = (time.perf_counter() - t0) * 1e3
            be.close()
            assert np.array_equal(tree_h["threshold_in_bin"], tree["threshold_in_bin"]) and np.array_equal(tree_h["leaf_count"], tree["leaf_count"])
        t0 = time.perf_counter()
        vals = mdl.newton_update_leaf_values(None, None, leaf_of, nleaves)      # reuses the factor / y_aux of step 1, as the reference does
        t["3_newton_leaf_values_ms"] = (time.perf_counter() - t0) * 1e3
        score = score + 0.1 * vals[leaf_of]
        t0 = time.perf_counter()
        mdl.fit(y - score)
        t["4_cov_par_step_ms"] = (time.perf_counter() - t0) * 1e3
        t["device_path_total_ms"] = t["1_gradient_yaux_ms"] + t["2_tree_growth_ms"] + t["3_newton_leaf_values_ms"] + t["4_cov_par_step_ms"]
        out["iteration_%d" % it] = {k: round(v, 3) for k, v in t.items()}
        out["iteration_%d" % it]["num_leaves"] = int(nleaves)
        out["iteration_%d" % it]["cov_pars"] = [float(v) for v in mdl.get_cov_pars()]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
