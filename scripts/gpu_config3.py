"""BASELINE config 3 AS WRITTEN, timed as a whole: GPBoost boosting loop, 100 trees, 255 bins, 50 features + Vecchia GP (m = 30, exponential), n = 1e5.

 (A) route B: the reference's OWN Booster / GBDT / REModel host code (integration/_build/lib_gpboost_hip.so) with GPU_use = true --
     100 x LGBM_BoosterUpdateOneIter, each = one covariance-parameter step, the gradient Psi^-1 (F - y), the tree, the Newton leaf values;
 (B) the same library with GPU_use = false (the reference's CPU path) for the first K iterations only (it takes ~6 s per iteration): the ensemble
     predictions of (A) truncated to its first K trees must reproduce it (<= 1e-6);
 (C) natively: the same iteration through this library's own C ABI (GPModel.y_aux -> HistBuilder.grow_tree -> newton_update_leaf_values -> one
     gradient-descent step of the covariance parameters), 100 iterations, timed as a whole.
Prints one JSON line: {"config3_100_trees_route_b_s", "config3_100_trees_native_s", ...}."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, F, NB, LEAVES, ROUNDS, K_CPU = 100000, 50, 255, 31, 100, int(os.environ.get("GPB_CONFIG3_CPU_ITERS", "4"))
rng = np.random.default_rng(1)
coords = rng.uniform(size=(N, 2))
X = np.ascontiguousarray(rng.uniform(size=(N, F)))
y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + np.sin(5 * coords[:, 0]) * np.cos(4 * coords[:, 1]) + 0.5 * rng.standard_normal(N)
out = {"workload": "BASELINE config 3: GPBoost boosting loop, %d trees, %d bins, %d features, %d leaves + Vecchia GP (m = 30, exponential), n = %d" % (ROUNDS, NB, F, LEAVES, N)}

# ---- (A), (B): route B ---------------------------------------------------------------------------------------------------------------
LIBP = os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so")
if os.path.isfile(LIBP) and "--native-only" not in sys.argv:
    from oracle import refdrv          # (a ctypes wrapper of the reference's own C API; checker infrastructure, used here to drive the route-B library)
    LB = C.CDLL(LIBP)
    LB.LGBM_GetLastError.restype = C.c_char_p

    def okb(rc):
        if rc != 0:
            raise RuntimeError(LB.LGBM_GetLastError().decode())

    yf = y.astype(np.float32)
    params = ("objective=regression num_leaves=%d learning_rate=0.1 min_data_in_leaf=20 verbosity=-1 num_threads=16 max_bin=%d leaves_newton_update=true "
              "train_gp_model_cov_pars=true" % (LEAVES, NB))

    def run(gpu, rounds, device_trees=False):
        t0 = time.perf_counter()
        mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, 30, "random", 1, threads=-1, lib_path=LIBP, gpu_use=gpu)
        ds = C.c_void_p()
        okb(LB.LGBM_DatasetCreateFromMat(X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(N), C.c_int32(F), C.c_int(1),
                                         C.c_char_p(("verbosity=-1 max_bin=%d" % NB).encode()), C.c_void_p(), C.byref(ds)))
        okb(LB.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yf.ctypes.data_as(C.c_void_p), C.c_int(N), C.c_int(0)))
        bst = C.c_void_p()
        okb(LB.LGBM_GPBoosterCreate(ds, C.c_char_p((params + (" device_type=gpu" if device_trees else "")).encode()), mdl.h, C.byref(bst)))
        t_setup = time.perf_counter() - t0
        fin = C.c_int(0)
        ts = []
        t1 = time.perf_counter()
        for _ in range(rounds):
            tt = time.perf_counter()
            okb(LB.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
            ts.append(time.perf_counter() - tt)
        t_loop = time.perf_counter() - t1

        def predict(num_iteration):
            o = np.empty(N); olen = C.c_int64(0)
            okb(LB.LGBM_BoosterPredictForMat(bst, X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(N), C.c_int32(F), C.c_int(1), C.c_int(1), C.c_int(0),
                                             C.c_int(num_iteration), C.c_char_p(b""), C.byref(olen), o.ctypes.data_as(C.POINTER(C.c_double))))
            return o
        res = dict(t_setup=t_setup, t_loop=t_loop, ts=ts, pred_all=predict(-1), pred_k=predict(K_CPU), cov=mdl.get_cov_par(3))
        okb(LB.LGBM_BoosterFree(bst)); okb(LB.LGBM_DatasetFree(ds))
        return res

    a = run(True, ROUNDS)
    print("route B, GPU_use=true: setup %.2f s, %d iterations %.3f s (median %.2f ms, first %.1f ms), cov pars %s" %
          (a["t_setup"], ROUNDS, a["t_loop"], 1e3 * float(np.median(a["ts"])), 1e3 * a["ts"][0], a["cov"]), file=sys.stderr, flush=True)
    out.update(config3_100_trees_route_b_s=a["t_loop"], route_b_setup_s=a["t_setup"], route_b_ms_per_iteration_median=1e3 * float(np.median(a["ts"])),
               route_b_cov_pars_after_100=[float(v) for v in a["cov"]])
    # the whole MI355X configuration of route B: GPU_use = true for the GP AND device_type = gpu for the trees (HIPTreeLearner: whole trees grown by
    # gpb_hip_hist_grow_tree) -- the reference's Booster / REModel host code around both
    a2 = run(True, ROUNDS, device_trees=True)
    err2 = float(np.abs(a2["pred_all"] - a["pred_all"]).max())
    print("route B, GPU_use=true + device_type=gpu: setup %.2f s, %d iterations %.3f s (median %.2f ms, first %.1f ms), cov pars %s; ensemble of all %d trees: "
          "max |device trees - host trees| = %.2e" % (a2["t_setup"], ROUNDS, a2["t_loop"], 1e3 * float(np.median(a2["ts"])), 1e3 * a2["ts"][0], a2["cov"], ROUNDS, err2),
          file=sys.stderr, flush=True)
    out.update(config3_100_trees_route_b_device_trees_s=a2["t_loop"], route_b_device_trees_ms_per_iteration_median=1e3 * float(np.median(a2["ts"])),
               route_b_device_trees_vs_host_trees_max_abs_diff=err2)
    if K_CPU > 0:
        b = run(False, K_CPU)
        err = float(np.abs(a["pred_k"] - b["pred_all"]).max())
        print("route B, GPU_use=false: %d iterations %.1f s; ensemble of the first %d trees: max |GPU_use=true - GPU_use=false| = %.2e" %
              (K_CPU, b["t_loop"], K_CPU, err), file=sys.stderr, flush=True)
        assert err <= 1e-6, err
        out.update(route_b_cpu_path_s_per_iteration=b["t_loop"] / K_CPU, route_b_prediction_parity_first_k_trees=dict(k=K_CPU, max_abs_diff=err),
                   route_b_speedup_per_iteration=(b["t_loop"] / K_CPU) / (a["t_loop"] / ROUNDS))
else:
    out["route_b"] = "integration/_build/lib_gpboost_hip.so not built (make -C oracle routeB)"

# ---- (C): natively through this library's C ABI ---------------------------------------------------------------------------------------
if "--routeb-only" not in sys.argv:
    import gpboost_amd
    from gpboost_amd import shim
    gpboost_amd.set_device(0)
    t0 = time.perf_counter()
    bins = np.minimum((X * (NB - 1)).astype(np.int64) + 1, NB - 1).astype(np.uint8).T.copy()      # synthetic equal-width bins in the reference's layout
    gnb = np.full(F, NB, dtype=np.int32)
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    m3 = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=30, vecchia_ordering="random", seed=1)
    cp0 = np.array([0.25, 0.1, 0.1])
    m3.set_optim_params({"optimizer_cov": "gradient_descent", "maxit": 1, "init_cov_pars": cp0})
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(LEAVES + 1)
    hb.set_fix_info((bo[:-1] + 1).astype(np.int32), np.full(F, NB, dtype=np.int32), np.zeros(F, dtype=np.int32))
    hb.set_split_info(np.ones(F, dtype=np.int32), np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32))
    t_setup = time.perf_counter() - t0
    score = np.zeros(N)
    ts = []
    t1 = time.perf_counter()
    for it in range(ROUNDS):
        tt = time.perf_counter()
        grad = m3.y_aux(m3.get_cov_pars() if it else cp0, score - y)
        hb.set_gradients(grad, None)
        tree = hb.grow_tree(LEAVES, float(grad.sum()), float(N), 0.0, 20, 1e-3, 0.0)
        vals = m3.newton_update_leaf_values(None, None, tree["data_leaf_index"], tree["num_leaves"])
        score = score + 0.1 * vals[tree["data_leaf_index"]]
        m3.fit(y - score)
        ts.append(time.perf_counter() - tt)
    t_loop = time.perf_counter() - t1
    rmse = float(np.sqrt(np.mean((y - score) ** 2)))
    print("native: setup %.2f s, %d iterations %.3f s (median %.2f ms), training RMSE of the tree ensemble %.4f, cov pars %s" %
          (t_setup, ROUNDS, t_loop, 1e3 * float(np.median(ts)), rmse, m3.get_cov_pars()), file=sys.stderr, flush=True)
    out.update(config3_100_trees_native_s=t_loop, native_setup_s=t_setup, native_ms_per_iteration_median=1e3 * float(np.median(ts)), native_train_rmse=rmse)
print(json.dumps(out), flush=True)
