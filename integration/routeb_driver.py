"""Route B from Python without the reference's package: a minimal ctypes driver of integration/_build/lib_gpboost_hip.so -- the reference's own host code
(REModel, Booster / GBDT, SerialTreeLearner) compiled with integration/reference_hip_seams.patch against this library.

Only the calls bench.py needs to TIME the drop-in a user would load: GPB_CreateREModel (include/LightGBM/c_api.h:1359-1391, 32 positional inputs + handle) with
GPU_use on / off, LGBM_DatasetCreateFromMat, LGBM_GPBoosterCreate (:437), LGBM_BoosterUpdateOneIter (:533), GPB_GetCovPar (:1534).  No dependency on oracle/."""
import ctypes as C
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so")


def available():
    return os.path.isfile(LIB_PATH)


def _P(a):
    return a.ctypes.data_as(C.c_void_p)


class RouteB(object):
    def __init__(self, lib_path=LIB_PATH):
        self.L = C.CDLL(lib_path)
        self.L.LGBM_GetLastError.restype = C.c_char_p

    def ok(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())

    def create_vecchia_model(self, coords, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=1, gpu_use=True):
        """One Gaussian Vecchia GP (the model of BASELINE configs 2 / 3); returns the REModelHandle."""
        cm = np.asfortranarray(coords, dtype=np.float64)
        n, d = cm.shape
        h = C.c_void_p()
        s = lambda x: C.c_char_p(x.encode())     # noqa: E731
        self.ok(self.L.GPB_CreateREModel(
            C.c_int(n), C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_void_p(),
            C.c_int(1), _P(cm), C.c_int(d), C.c_void_p(), C.c_int(0), s(cov_function), C.c_double(shape), s("vecchia"),
            C.c_double(1.), C.c_double(0.), C.c_int(m), s(ordering), C.c_int(500), C.c_double(1.), s("kmeans++"),
            s("gaussian"), C.c_double(-999.), s("default"), C.c_int(seed), C.c_int(-1), C.c_bool(bool(gpu_use)),
            C.c_bool(False), C.c_void_p(), C.c_double(1.), C.byref(h)))
        return h

    def get_cov_pars(self, h, num=3):
        out = np.zeros(num)
        self.ok(self.L.GPB_GetCovPar(h, _P(out), C.c_bool(False)))
        return out

    def free_model(self, h):
        self.ok(self.L.GPB_REModelFree(h))

    def boosting_loop(self, coords, X, y, rounds, gpu_use=True, device_trees=False, num_leaves=31, max_bin=255, learning_rate=0.1, m=30):
        """`rounds` x LGBM_BoosterUpdateOneIter of the GPBoost algorithm (each: one covariance-parameter step, the gradient Psi^-1 (F - y), the tree, the Newton
        leaf values).  -> dict(setup_s, loop_s, per_iteration_s, cov_pars, pred)"""
        L = self.L
        n, F = X.shape
        X = np.ascontiguousarray(X, dtype=np.float64)
        yf = np.ascontiguousarray(y, dtype=np.float32)
        t0 = time.perf_counter()
        h = self.create_vecchia_model(coords, m=m, gpu_use=gpu_use)
        ds = C.c_void_p()
        self.ok(L.LGBM_DatasetCreateFromMat(_P(X), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1),
                                            C.c_char_p(("verbosity=-1 max_bin=%d" % max_bin).encode()), C.c_void_p(), C.byref(ds)))
        self.ok(L.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), _P(yf), C.c_int(n), C.c_int(0)))
        params = ("objective=regression num_leaves=%d learning_rate=%g min_data_in_leaf=20 verbosity=-1 num_threads=16 max_bin=%d leaves_newton_update=true "
                  "train_gp_model_cov_pars=true" % (num_leaves, learning_rate, max_bin)) + (" device_type=gpu" if device_trees else "")
        bst = C.c_void_p()
        self.ok(L.LGBM_GPBoosterCreate(ds, C.c_char_p(params.encode()), h, C.byref(bst)))
        setup_s = time.perf_counter() - t0
        fin = C.c_int(0)
        ts = []
        t1 = time.perf_counter()
        for _ in range(rounds):
            tt = time.perf_counter()
            self.ok(L.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
            ts.append(time.perf_counter() - tt)
        loop_s = time.perf_counter() - t1
        pred = np.empty(n); olen = C.c_int64(0)
        self.ok(L.LGBM_BoosterPredictForMat(bst, _P(X), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1), C.c_int(1), C.c_int(0), C.c_int(-1),
                                            C.c_char_p(b""), C.byref(olen), pred.ctypes.data_as(C.POINTER(C.c_double))))
        cov = self.get_cov_pars(h)
        self.ok(L.LGBM_BoosterFree(bst)); self.ok(L.LGBM_DatasetFree(ds)); self.free_model(h)
        return dict(setup_s=setup_s, loop_s=loop_s, per_iteration_s=ts, cov_pars=cov, pred=pred)
