"""Workloads of the round-5 kernels that had no rocprofv3 evidence (VERDICT r05 #1 i): run ONE of them under `rocprofv3 --kernel-trace --stats` / `--pmc`.

    python scripts/gpu_r6_targets.py config4:<vadu|pivoted_cholesky|fitc>     one Bernoulli-logit Vecchia-Laplace evaluation at BASELINE config 4's size (after a set-up one)
    python scripts/gpu_r6_targets.py aux:<t|gamma|lognormal>                   one evaluation of an auxiliary-parameter likelihood at config 4's size (smooth latent surface)
    python scripts/gpu_r6_targets.py cattree                                   one 31-leaf tree at config 3's shape with 6 categorical columns (best_split_feature_cat, bitset partitions)
    python scripts/gpu_r6_targets.py blockxchg                                 the same tree data-parallel with 2 ranks on this device, feature-block exchange (hist_limbs_pack_kernel)
Prints what the library's own timers say (iteration counts, ms per phase) so that the kernel times of the trace can be set against them."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd          # noqa: E402
from gpboost_amd import shim   # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "config4:vadu"
kind, _, arg = what.partition(":")
gpboost_amd.set_device(0)
n, m = 100000, 30

if kind == "config4":
    rng = np.random.default_rng(1)
    c4 = rng.uniform(size=(n, 2)); y4 = (rng.uniform(size=n) < 0.5).astype(np.float64)
    mdl = gpboost_amd.GPModel(likelihood="bernoulli_logit", gp_coords=c4, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
    mdl.set_optim_params({"cg_preconditioner_type": arg or "vadu"})
    mdl.neg_log_likelihood(np.array([1.0, 0.1]), y4)
    t0 = time.perf_counter()
    v = mdl.neg_log_likelihood(np.array([1.01, 0.1]), y4)
    print("config 4, %s: negll %.6f, %.3f s, %s" % (arg, v, time.perf_counter() - t0, mdl.laplace_info()), flush=True)
elif kind == "aux":
    rng = np.random.default_rng(7)
    cc = rng.uniform(size=(n, 2))
    eta = np.sin(4 * cc[:, 0]) + np.cos(3 * cc[:, 1])
    if arg == "t":
        y = 0.8 * eta + 0.35 * rng.standard_t(4.0, size=n)
    elif arg == "gamma":
        y = rng.gamma(2.0, np.exp(0.5 * eta) / 2.0)
    else:
        y = np.exp(0.5 * eta + np.sqrt(0.2) * rng.standard_normal(n))
    mdl = gpboost_amd.GPModel(likelihood=arg, gp_coords=cc, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
    mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
    t0 = time.perf_counter()
    v = mdl.neg_log_likelihood(np.array([1.01, 0.1]), y)
    print("%s at config 4's size: negll %.6f, %.3f s, %s" % (arg, v, time.perf_counter() - t0, mdl.laplace_info()), flush=True)
elif kind in ("cattree", "blockxchg"):
    F, NB, L = 50, 255, 31
    rng = np.random.default_rng(17)
    X = rng.uniform(size=(n, F))
    cat_cols = {3: 12, 11: 100, 19: 250, 27: 100, 35: 12, 43: 250}
    bins = np.empty((F, n), dtype=np.uint8); gnb = np.empty(F, dtype=np.int32); is_cat = np.zeros(F, dtype=np.int32)
    signal = np.sin(4 * X[:, 0]) + X[:, 1] ** 2
    for f in range(F):
        if f in cat_cols:
            K = cat_cols[f]
            pr = np.sort(rng.dirichlet(np.full(K, 0.7)))[::-1]
            cat = rng.choice(K, size=n, p=pr)
            bins[f] = cat.astype(np.uint8); gnb[f] = K; is_cat[f] = 1
            signal = signal + rng.standard_normal(K)[cat] * (0.6 if K <= 100 else 0.3)
        else:
            bins[f] = np.minimum((X[:, f] * (NB - 1)).astype(np.int64) + 1, NB - 1).astype(np.uint8); gnb[f] = NB
    grad = signal + 0.5 * rng.standard_normal(n)
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    voff = (bo[:-1] + 1).astype(np.int32); mfb = np.zeros(F, dtype=np.int32)

    def make(rows):
        hb = shim.HistBuilder(np.ascontiguousarray(bins[:, rows]), bo)
        hb.pool_resize(L + 1)
        hb.set_fix_info(voff, gnb, mfb)
        hb.set_split_info(np.ones(F, dtype=np.int32), np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32))
        hb.set_categorical(is_cat, 4, 32, 10.0, 10.0, 100)
        return hb
    if kind == "cattree":
        hb = make(np.arange(n))
        hb.set_gradients(grad, None)
        for rep in range(4):
            t0 = time.perf_counter()
            t = hb.grow_tree(L, float(np.cumsum(grad)[-1]), float(n), 0.5, 20, 1e-3, 0.0)
            print("categorical tree %d: %d leaves, %d categorical nodes, %.3f ms" % (rep, t["num_leaves"], int(t["node_is_cat"].sum()), 1e3 * (time.perf_counter() - t0)), flush=True)
        hb.close()
    else:
        W = 2
        grp = shim.LocalGroup(W)
        parts = [np.arange(r, n, W) for r in range(W)]

        def rank(r):
            hb = make(parts[r])
            hb.comm_init_local(grp, r)
            hb.set_feature_block_exchange(True)
            ts = []
            for rep in range(3):
                hb.set_gradients(grad[parts[r]], None)
                t0 = time.perf_counter()
                t = hb.grow_tree(L, float("nan"), float("nan"), 0.5, 20, 1e-3, 0.0)
                ts.append(1e3 * (time.perf_counter() - t0))
            hb.close()
            return t["num_leaves"], ts
        for r, (nl, ts) in enumerate(grp.run(rank)):
            print("feature-block exchange, rank %d of %d on one device: %d leaves, ms per tree %s" % (r, W, nl, ["%.3f" % v for v in ts]), flush=True)
        grp.close()
else:
    raise SystemExit("unknown workload " + what)
