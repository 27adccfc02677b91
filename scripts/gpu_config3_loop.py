"""BASELINE config 3 as a WHOLE LOOP under test: the first K = 10 boosting iterations of the GPBoost algorithm at config 3's size
(n = 1e5, 50 features, 255 bins, 31 leaves, learning rate 0.1, Vecchia GP m = 30 exponential, covariance parameters trained inside the loop) through the
reference's own Booster / GBDT / REModel host code of route B (integration/_build/lib_gpboost_hip.so):

  --make-ref   (CPU, in the build container): GPU_use = false, device_type = cpu -- the reference's CPU path of the same build -- K iterations; stores the
               ensemble's predictions on every 20th row, their sums over all rows and the covariance parameters in tests/golden/config3_loop_ref.npz
  (default)    (MI355X): leg 1 GPU_use = true (GP on the device, the reference's CPU tree learner); leg 2 GPU_use = true AND device_type = gpu (whole trees on
               the device as well: every kernel of the iteration is this library's) -- both against the stored CPU values: predictions 1e-8 of their scale,
               covariance parameters 1e-6.
Each iteration = one covariance-parameter step, the gradient Psi^-1 (F - y), the tree, the Newton leaf values (gbdt.cpp:411-567, regression_objective.hpp:153-201,
re_model_template.h:5002-5062).  Nothing here reads /root/reference."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refdrv  # noqa: E402   (a ctypes wrapper of the reference's C API: drives the route-B library; checker infrastructure)

N, F, NB, LEAVES, K = 100000, 50, 255, 31, 10
REF_PATH = os.path.join(ROOT, "tests", "golden", "config3_loop_ref.npz")
LIBP = os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so")
MAKE_REF = "--make-ref" in sys.argv

rng = np.random.default_rng(1)
coords = rng.uniform(size=(N, 2))
X = np.ascontiguousarray(rng.uniform(size=(N, F)))
y = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + np.sin(5 * coords[:, 0]) * np.cos(4 * coords[:, 1]) + 0.5 * rng.standard_normal(N)
yf = y.astype(np.float32)
LB = C.CDLL(LIBP)
LB.LGBM_GetLastError.restype = C.c_char_p


def okb(rc):
    if rc != 0:
        raise RuntimeError(LB.LGBM_GetLastError().decode())


def run(gpu, device_trees):
    t0 = time.perf_counter()
    mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, 30, "random", 1, threads=-1, lib_path=LIBP, gpu_use=gpu)
    ds = C.c_void_p()
    okb(LB.LGBM_DatasetCreateFromMat(X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(N), C.c_int32(F), C.c_int(1),
                                     C.c_char_p(("verbosity=-1 max_bin=%d" % NB).encode()), C.c_void_p(), C.byref(ds)))
    okb(LB.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yf.ctypes.data_as(C.c_void_p), C.c_int(N), C.c_int(0)))
    params = ("objective=regression num_leaves=%d learning_rate=0.1 min_data_in_leaf=20 verbosity=-1 num_threads=16 max_bin=%d leaves_newton_update=true "
              "train_gp_model_cov_pars=true" % (LEAVES, NB)) + (" device_type=gpu" if device_trees else "")
    bst = C.c_void_p()
    okb(LB.LGBM_GPBoosterCreate(ds, C.c_char_p(params.encode()), mdl.h, C.byref(bst)))
    t_setup = time.perf_counter() - t0
    fin = C.c_int(0)
    t1 = time.perf_counter()
    for _ in range(K):
        okb(LB.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
    t_loop = time.perf_counter() - t1
    pred = np.empty(N); olen = C.c_int64(0)
    okb(LB.LGBM_BoosterPredictForMat(bst, X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(N), C.c_int32(F), C.c_int(1), C.c_int(1), C.c_int(0),
                                     C.c_int(-1), C.c_char_p(b""), C.byref(olen), pred.ctypes.data_as(C.POINTER(C.c_double))))
    cov = np.asarray(mdl.get_cov_par(3), dtype=np.float64)
    okb(LB.LGBM_BoosterFree(bst)); okb(LB.LGBM_DatasetFree(ds))
    del mdl
    return pred, cov, t_setup, t_loop


if MAKE_REF:
    pred, cov, t_setup, t_loop = run(False, False)
    np.savez(REF_PATH, pred_rows=np.arange(0, N, 20), pred=pred[::20], pred_sum=pred.sum(), pred_sumsq=(pred ** 2).sum(), cov_pars=cov, k=K,
             cpu_s_per_iteration=t_loop / K)
    print("config 3, %d iterations on the reference's CPU path (GPU_use=false, device_type=cpu): %.1f s per iteration; cov pars %s -> %s" % (K, t_loop / K, cov, REF_PATH))
    sys.exit(0)

g = np.load(REF_PATH)
assert int(g["k"]) == K
scale = float(np.abs(g["pred"]).max())
for name, device_trees in (("GPU_use=true", False), ("GPU_use=true device_type=gpu", True)):
    pred, cov, t_setup, t_loop = run(True, device_trees)
    err = float(np.abs(pred[g["pred_rows"]] - g["pred"]).max())
    print("config 3 whole loop, %s: %d iterations %.3f s (%.2f ms per iteration; set-up %.2f s); max |prediction - CPU path| = %.2e (scale %.2f); cov pars %s (CPU path %s)"
          % (name, K, t_loop, 1e3 * t_loop / K, t_setup, err, scale, cov, g["cov_pars"]), flush=True)
    assert err <= 1e-8 * scale, err
    assert abs(pred.sum() - float(g["pred_sum"])) <= 1e-8 * scale * N
    assert abs((pred ** 2).sum() - float(g["pred_sumsq"])) <= 1e-8 * scale * scale * N
    np.testing.assert_allclose(cov, g["cov_pars"], rtol=1e-6)
print("CONFIG 3 WHOLE LOOP ON MI355X: OK (the reference's CPU path took %.1f s per iteration)" % float(g["cpu_s_per_iteration"]))
