"""Evidence run on the MI355X for INTEGRATION.md route B (scripts run by gpurun; needs integration/_build/lib_gpboost_hip.so built by
`make -f integration/Makefile.routeB`): the reference's OWN host code -- REModel, its optimiser, Booster / GBDT / SerialTreeLearner, reached
through its unchanged C API -- with the patched seams calling lib_gpboost_amd.so.

 (1) Gaussian Vecchia model: GPB_EvalNegLogLikelihood and GPB_OptimCovPar with GPU_use = true against GPU_use = false of the same library
     (same process, same inputs): likelihood to 1e-8 relative, same number of optimiser iterations, estimates to 1e-6, and predictions
     after the fit (the optimiser's evaluations run fused on the device; the host factor is built once at the final parameters).
 (2) GPBoost-free LightGBM boosting: LGBM_BoosterUpdateOneIter with device_type = gpu (-> HIPTreeLearner, histograms on the device)
     against device_type = cpu: predictions after 20 iterations agree to 1e-9 -- once with whole trees grown on the device
     (HIPTreeLearner::Train -> gpb_hip_hist_grow_tree; also with max_depth, feature_fraction and lambda_l1 / max_delta_step / path_smooth) and once
     with feature_fraction_bynode < 1 (not restated: SerialTreeLearner::Train + device histograms)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refdrv   # noqa: E402
from tests import cases     # noqa: E402

LIBP = os.environ.get("GPB_ROUTEB_LIB", os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so"))       # (GPB_ROUTEB_LIB: another build of route B, e.g. one with timeline markers)
print("library:", LIBP, flush=True)

# ---- (1) GP ---------------------------------------------------------------------------------------------------------------------
# --test (tests/test_routes_gpu.py): the reduced sizes of the -m gpu test; --gp-only / --trees-only: one half
# --cpu-mock (tests/test_routeB_seams_cpu.py, no GPU): the same seams with tests/mock_shim's CPU restatement of gpb_hip_* preloaded -- small sizes, no trees
TEST = "--test" in sys.argv
MOCK = "--cpu-mock" in sys.argv
GP_CASES = ((20000, 30, 2), (100000, 30, 2), (5000, 70, 5))   # the last one: d = 5, m = 70 -> the library's generality kernels
if TEST:
    GP_CASES = ((20000, 30, 2), (5000, 70, 5))
if MOCK:
    GP_CASES = ((1500, 12, 2),)
if "--trees-only" in sys.argv or "--gpboost-only" in sys.argv or "--laplace-only" in sys.argv:
    GP_CASES = ()
for n, m, d in GP_CASES:
    coords, _ = cases.synthetic(n, d, seed=3)
    rng = np.random.default_rng(5)
    y = np.sin(4 * coords[:, 0]) + 0.5 * rng.standard_normal(n)
    cp = np.array([0.2, 0.8, 0.15])
    res = {}
    for gpu in (False, True):
        t0 = time.perf_counter()
        mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, m, "random", 1, threads=-1, lib_path=LIBP, gpu_use=gpu)
        t_create = time.perf_counter() - t0
        mdl.neg_log_likelihood(cp, y)
        t0 = time.perf_counter()
        nll = mdl.neg_log_likelihood(cp * 1.01, y)
        t_eval = time.perf_counter() - t0
        t0 = time.perf_counter()
        mdl.optim_cov_par(y)
        t_fit = time.perf_counter() - t0
        # prediction after the fit: the factor / y_aux at the final parameters must be there (the optimiser's evaluations used the fused kernel)
        mu, var = mdl.predict(np.random.default_rng(9).uniform(size=(50, d)))
        res[gpu] = dict(nll=nll, cov=mdl.get_cov_par(3), it=mdl.get_num_it(), t_create=t_create, t_eval=t_eval, t_fit=t_fit, mu=mu, var=var,
                        negll_fit=mdl.current_neg_log_likelihood())
        print("n=%d GPU_use=%s: nll %.10f | fit: %d iterations, cov pars %s | create %.2f s, eval %.3f s, fit %.2f s" %
              (n, gpu, nll, res[gpu]["it"], res[gpu]["cov"], t_create, t_eval, t_fit), flush=True)
    a, b = res[False], res[True]
    assert abs(a["nll"] - b["nll"]) <= 1e-8 * abs(a["nll"]), (a["nll"], b["nll"])
    assert a["it"] == b["it"], (a["it"], b["it"])
    np.testing.assert_allclose(a["cov"], b["cov"], rtol=1e-6)
    assert abs(a["negll_fit"] - b["negll_fit"]) <= 1e-8 * abs(a["negll_fit"])
    np.testing.assert_allclose(b["mu"], a["mu"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(b["var"], a["var"], rtol=1e-6, atol=1e-8)
    print("n=%d: GPU_use=true reproduces the CPU path of the same build; likelihood evaluation %.1fx, fit %.1fx faster" % (n, a["t_eval"] / b["t_eval"], a["t_fit"] / b["t_fit"]), flush=True)

# ---- (1c) non-Gaussian likelihoods through the Laplace seams (round 4): Likelihood::FindModePostRandEffCalcMLLVecchia, its gradient and
#      ResetModeToPreviousValue call gpb_hip_vecchia_laplace_eval / _grad_current / _reset_mode_to_previous when the model is one Vecchia GP with
#      matrix_inversion_method = "iterative" and the "vadu" preconditioner -- GPU_use = true against GPU_use = false of the same build -------------
# The reference's own iterative CPU path is slow (n = 20000: 112 s per evaluation, 1063 s per fit on the GPU box's 256 threads, profiles/r04_l_*): its values
# are computed ONCE (--make-laplace-ref, CPU only, GPU_use = false of the same build) and kept in tests/golden/routeB_laplace_ref.json; the GPU run compares
# GPU_use = true against them.  The CPU-mock test runs both legs at n = 800.
import json  # noqa: E402
LAP_REF_PATH = os.path.join(ROOT, "tests", "golden", "routeB_laplace_ref.json")
LAP_REF = json.load(open(LAP_REF_PATH)) if os.path.exists(LAP_REF_PATH) else {}
MAKE_LAP_REF = "--make-laplace-ref" in sys.argv
# ("<likelihood>:pivoted_cholesky": round 5 -- the same seams with cg_preconditioner_type = "pivoted_cholesky"; the host's PivotedCholsekyFactorizationSigma is skipped, the device forms the factor)
# (round 5, second widening: the likelihoods with auxiliary parameters -- gamma (shape), negative_binomial (shape), beta (precision), t (scale and df, Fisher-Laplace) -- through the
#  same seams: real-valued response by gpb_hip_vecchia_laplace_set_response_real, the parameters of every evaluation by gpb_hip_vecchia_laplace_set_aux_pars, their gradient by
#  gpb_hip_vecchia_laplace_grad_aux_current; the fits estimate them together with the covariance parameters)
AUX_LIKS = ("gamma", "negative_binomial", "beta", "t")
LAP_CASES = ((800, 10, ("bernoulli_logit", "bernoulli_logit:pivoted_cholesky", "gamma", "t")),) if MOCK else (((5000, 20, ("bernoulli_logit", "poisson", "bernoulli_logit:pivoted_cholesky") + AUX_LIKS + ("gamma:pivoted_cholesky",)),) if (TEST or MAKE_LAP_REF) else ((5000, 20, ("bernoulli_logit", "poisson", "bernoulli_logit:pivoted_cholesky") + AUX_LIKS + ("gamma:pivoted_cholesky",)), (20000, 30, ("bernoulli_logit",)), (100000, 30, ("bernoulli_logit", "bernoulli_logit:pivoted_cholesky"))))
if "--trees-only" in sys.argv or "--gpboost-only" in sys.argv:
    LAP_CASES = ()
for n, m, liks in LAP_CASES:
    rng = np.random.default_rng(21)
    coords = rng.uniform(size=(n, 2))
    eta = np.sin(4 * coords[:, 0]) + np.cos(3 * coords[:, 1])
    for lik_pc in liks:
        lik, _, precond = lik_pc.partition(":")
        precond = precond or "vadu"
        if lik == "poisson":
            yl = rng.poisson(np.exp(0.5 * eta)).astype(np.float64)
        elif lik == "gamma":
            yl = rng.gamma(2.0, np.exp(0.5 * eta) / 2.0)
        elif lik == "negative_binomial":
            yl = rng.negative_binomial(3.0, 3.0 / (3.0 + np.exp(0.5 * eta))).astype(np.float64)
        elif lik == "beta":
            mu_b = 1.0 / (1.0 + np.exp(-0.8 * eta))
            yl = np.clip(rng.beta(mu_b * 8.0, (1.0 - mu_b) * 8.0), 1e-6, 1.0 - 1e-6)
        elif lik == "t":
            yl = 0.8 * eta + 0.3 * rng.standard_t(4.0, size=n)
        else:
            yl = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-1.5 * eta))).astype(np.float64)
        cp = np.array([1.0, 0.1])
        key = "%s_n%d_m%d" % (lik, n, m) + ("" if precond == "vadu" else "_" + precond)
        if MAKE_LAP_REF and key in LAP_REF and "--redo" not in sys.argv:
            continue
        res = {}
        legs = (False,) if MAKE_LAP_REF else ((False, True) if MOCK else (True,))
        for gpu in legs:
            mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, m, "random", 1, threads=-1, likelihood=lik, lib_path=LIBP, gpu_use=gpu,
                                      matrix_inversion_method="iterative")
            mdl.set_optim_config(init_cov_pars=cp, optimizer_cov="lbfgs", cg_delta_conv=1e-6, cg_preconditioner_type=precond, estimate_aux_pars=lik in AUX_LIKS)
            t0 = time.perf_counter()
            nll0 = mdl.neg_log_likelihood(cp, yl)
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            nll1 = mdl.neg_log_likelihood(cp * np.array([1.1, 0.9]), yl)
            t_eval = time.perf_counter() - t0
            out = dict(nll0=nll0, nll1=nll1, t_first=t_first, t_eval=t_eval)
            if n <= 20000:
                t0 = time.perf_counter()
                mdl.optim_cov_par(yl)
                out.update(t_fit=time.perf_counter() - t0, cov=[float(v) for v in mdl.get_cov_par(2)], it=int(mdl.get_num_it()), negll_fit=mdl.current_neg_log_likelihood())
                if lik in AUX_LIKS:
                    out["aux"] = [float(v) for v in mdl.get_aux_pars(2 if lik == "t" else 1)]
            res[gpu] = out
            print("Laplace %s n=%d GPU_use=%s: nll %.10f / %.10f, evaluation %.3f s%s" % (
                lik_pc, n, gpu, nll0, nll1, t_eval, "" if "cov" not in out else "; fit: %d iterations, cov pars %s%s, negll %.8f, %.2f s" % (out["it"], out["cov"], "" if "aux" not in out else ", aux pars %s" % out["aux"], out["negll_fit"], out["t_fit"])), flush=True)
            del mdl
        if MAKE_LAP_REF:
            LAP_REF[key] = res[False]
            json.dump(LAP_REF, open(LAP_REF_PATH, "w"), indent=1, sort_keys=True)
            continue
        a = res[False] if False in res else LAP_REF.get(key)
        b = res[True]
        if a is None:
            print("Laplace %s n=%d: GPU_use=true evaluation %.3f s (no stored CPU values at this size: the reference's CPU path takes minutes per evaluation)" % (lik_pc, n, b["t_eval"]), flush=True)
            continue
        assert abs(a["nll0"] - b["nll0"]) <= 1e-7 * abs(a["nll0"]), (a["nll0"], b["nll0"])
        assert abs(a["nll1"] - b["nll1"]) <= 1e-7 * abs(a["nll1"]), (a["nll1"], b["nll1"])
        msg = "evaluation %.1fx" % (a["t_eval"] / b["t_eval"])
        if "cov" in a and "cov" in b:
            np.testing.assert_allclose(b["cov"], a["cov"], rtol=1e-3)
            if "aux" in a:
                np.testing.assert_allclose(b["aux"], a["aux"], rtol=1e-3)
            assert abs(a["negll_fit"] - b["negll_fit"]) <= 1e-6 * abs(a["negll_fit"]), (a["negll_fit"], b["negll_fit"])
            msg += ", fit %.1fx (%d / %d iterations)" % (a["t_fit"] / b["t_fit"], a["it"], b["it"])
        print("Laplace %s n=%d: GPU_use=true (mode finding, stochastic log-determinant and gradient on the device) reproduces the CPU path of the same build%s; %s faster" % (
            lik_pc, n, "" if False in res else " (its values: tests/golden/routeB_laplace_ref.json)", msg), flush=True)
# ---- (1d) the GPBoost algorithm for binary classification: every boosting iteration finds the mode at the current scores (fixed effects of the
#      Laplace state), takes one covariance-parameter step (device mode finding + device gradient) and hands d(-mll)/dF to the tree as the gradient
#      (CalcGradNegMargLikelihoodLaplaceApproxVecchia with calc_F_grad -> gpb_hip_vecchia_laplace_grad_F_current) -- GPU_use = true against false ----
CLS_CASES = ((1200, 6, 3, 10),) if MOCK else ((5000, 10, 4, 20),)
if "--trees-only" in sys.argv or "--gpboost-only" in sys.argv:
    CLS_CASES = ()
LC = C.CDLL(LIBP)
LC.LGBM_GetLastError.restype = C.c_char_p
for n, F, nit, m in CLS_CASES:
    rng = np.random.default_rng(31)
    coords = rng.uniform(size=(n, 2))
    X = np.ascontiguousarray(rng.uniform(size=(n, F)))
    eta = 2.0 * np.sin(4 * X[:, 0]) + X[:, 1] - 0.5 + np.sin(5 * coords[:, 0]) * np.cos(4 * coords[:, 1])
    yc = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float32)
    key = "gpboost_binary_n%d_F%d_m%d_it%d" % (n, F, m, nit)
    res = {}
    legs = (False,) if MAKE_LAP_REF else ((False, True) if MOCK else (True,))
    for gpu in legs:
        mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, m, "random", 1, threads=-1, likelihood="bernoulli_logit", lib_path=LIBP, gpu_use=gpu,
                                  matrix_inversion_method="iterative")
        mdl.set_optim_config(init_cov_pars=np.array([1.0, 0.1]), optimizer_cov="gradient_descent", cg_delta_conv=1e-6)
        ds = C.c_void_p()
        rc = LC.LGBM_DatasetCreateFromMat(X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1),
                                          C.c_char_p(b"verbosity=-1 max_bin=63"), C.c_void_p(), C.byref(ds))
        assert rc == 0, LC.LGBM_GetLastError().decode()
        assert LC.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yc.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(0)) == 0
        bst = C.c_void_p()
        params = "objective=binary num_leaves=15 learning_rate=0.1 min_data_in_leaf=20 verbosity=-1 num_threads=8 max_bin=63 train_gp_model_cov_pars=true"
        rc = LC.LGBM_GPBoosterCreate(ds, C.c_char_p(params.encode()), mdl.h, C.byref(bst))
        assert rc == 0, LC.LGBM_GetLastError().decode()
        fin = C.c_int(0)
        ts = []
        for _ in range(nit):
            t0 = time.perf_counter()
            rc = LC.LGBM_BoosterUpdateOneIter(bst, C.byref(fin))
            assert rc == 0, LC.LGBM_GetLastError().decode()
            ts.append(time.perf_counter() - t0)
        out = np.empty(n); olen = C.c_int64(0)
        rc = LC.LGBM_BoosterPredictForMat(bst, X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1), C.c_int(1), C.c_int(0),
                                          C.c_int(-1), C.c_char_p(b""), C.byref(olen), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0, LC.LGBM_GetLastError().decode()
        res[gpu] = dict(pred=[float(v) for v in out[:200]], cov=[float(v) for v in mdl.get_cov_par(2)], t_iter=float(np.median(ts)))
        print("GPBoost binary n=%d GPU_use=%s: median %.1f ms per boosting iteration; cov pars %s; tree ensemble[:3] = %s" % (n, gpu, 1e3 * res[gpu]["t_iter"], res[gpu]["cov"], out[:3]), flush=True)
        LC.LGBM_BoosterFree(bst); LC.LGBM_DatasetFree(ds)
        del mdl
    if MAKE_LAP_REF:
        LAP_REF[key] = res[False]
        json.dump(LAP_REF, open(LAP_REF_PATH, "w"), indent=1, sort_keys=True)
        continue
    a = res[False] if False in res else LAP_REF.get(key)
    b = res[True]
    if a is None:
        print("GPBoost binary n=%d: no stored CPU values" % n, flush=True)
        continue
    np.testing.assert_allclose(b["pred"], a["pred"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b["cov"], a["cov"], rtol=1e-5)
    print("GPBoost binary n=%d: GPU_use=true (mode finding at the scores, covariance step and the boosting gradient d(-mll)/dF on the device) reproduces the CPU path%s; "
          "boosting iteration %.1fx faster" % (n, "" if False in res else " (its values: tests/golden/routeB_laplace_ref.json)", a["t_iter"] / b["t_iter"]), flush=True)
if MAKE_LAP_REF:
    sys.exit(0)

# ---- (1b) GPBoost iterations through the round-4 seams: device neighbour search at model creation, y_aux = Psi^-1 (F - y) from the resident
#      factor (CalcYAux), Newton leaf values (NewtonUpdateLeafValues) -- GPU_use = true against GPU_use = false of the same build -------------
GPB_CASES = ((20000, 20, 3),) if TEST else ((100000, 50, 5), (1000000, 50, 2))
if "--laplace-only" in sys.argv:
    GPB_CASES = ()
if MOCK:
    GPB_CASES = ((int(os.environ.get("GPB_ROUTEB_MOCK_N", "2500")), 6, 4),)
if "--trees-only" in sys.argv:
    GPB_CASES = ()
LB = C.CDLL(LIBP)
LB.LGBM_GetLastError.restype = C.c_char_p
EXTRA_IT = 0 if (MOCK or TEST) else int(os.environ.get("GPB_ROUTEB_EXTRA_IT", "8"))
API_TIMING = os.environ.get("GPB_HIP_API_TIMING", "") not in ("", "0")
AMD = C.CDLL(os.path.join(ROOT, "gpboost_amd", "lib_gpboost_amd.so")) if API_TIMING else None


def okb(rc):
    if rc != 0:
        raise RuntimeError(LB.LGBM_GetLastError().decode())


for n, F, nit in GPB_CASES:
    rng = np.random.default_rng(11)
    coords = rng.uniform(size=(n, 2))
    X = np.ascontiguousarray(rng.uniform(size=(n, F)))
    yb = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + np.sin(5 * coords[:, 0]) * np.cos(4 * coords[:, 1]) + 0.3 * rng.standard_normal(n)
    yf = yb.astype(np.float32)
    res = {}
    # --skip-cpu-1e6: the CPU leg at n = 1e6 takes 236 s per boosting iteration (profiles/r04_d_routeB.log: measured once); the GPU leg is then checked
    # against the values that run printed
    skip_cpu = n == 1000000 and "--skip-cpu-1e6" in sys.argv
    # legs: False = the reference's CPU path; True = GPU_use = true (GP on the device, the reference's CPU tree learner); "trees" = GPU_use = true AND
    # device_type = gpu (HIPTreeLearner: whole trees on the device too)
    for gpu in ((True, "trees") if skip_cpu else (False, True) if (MOCK or TEST) else (False, True, "trees")):
        t0 = time.perf_counter()
        mdl = refdrv.RefCAPIModel(coords, "exponential", 0.5, 12 if MOCK else 30, "random", 1, threads=-1, lib_path=LIBP, gpu_use=bool(gpu))
        t_create = time.perf_counter() - t0
        ds = C.c_void_p()
        okb(LB.LGBM_DatasetCreateFromMat(X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1),
                                         C.c_char_p(b"verbosity=-1 max_bin=255"), C.c_void_p(), C.byref(ds)))
        okb(LB.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yf.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(0)))
        bst = C.c_void_p()
        params = "objective=regression num_leaves=31 learning_rate=0.1 min_data_in_leaf=20 verbosity=-1 num_threads=16 max_bin=255 leaves_newton_update=true train_gp_model_cov_pars=true"
        okb(LB.LGBM_GPBoosterCreate(ds, C.c_char_p((params + (" device_type=gpu" if gpu == "trees" else "")).encode()), mdl.h, C.byref(bst)))
        fin = C.c_int(0)
        ts = []
        for _ in range(nit):
            t0 = time.perf_counter()
            okb(LB.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
            ts.append(time.perf_counter() - t0)
        out = np.empty(n); olen = C.c_int64(0)
        okb(LB.LGBM_BoosterPredictForMat(bst, X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1), C.c_int(1), C.c_int(0),
                                         C.c_int(-1), C.c_char_p(b""), C.byref(olen), out.ctypes.data_as(C.POINTER(C.c_double))))
        cov_checked = mdl.get_cov_par(3)
        if gpu and EXTRA_IT > 0:
            # (the values above are those the CPU leg is compared on; the iterations below only add timing samples: the first iteration of a Booster
            #  carries BoostFromAverage and the learner's one-time set-up, so a median over 2 is not a per-iteration cost)
            if API_TIMING:
                AMD.gpb_hip_api_timing_report(1)
            for _ in range(EXTRA_IT):
                t0 = time.perf_counter()
                okb(LB.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
                ts.append(time.perf_counter() - t0)
            print("GPBoost n=%d GPU_use=%s: per-iteration wall ms %s" % (n, gpu, " ".join("%.1f" % (1e3 * t) for t in ts)), flush=True)
            if API_TIMING:
                print("time inside lib_gpboost_amd.so over the last %d iterations (%.1f ms wall):" % (EXTRA_IT, 1e3 * sum(ts[-EXTRA_IT:])), file=sys.stderr, flush=True)
                AMD.gpb_hip_api_timing_report(1)
            ts = ts[nit:]
        res[gpu] = dict(pred=out, cov=cov_checked, t_create=t_create, t_iter=float(np.median(ts)))
        print("GPBoost n=%d GPU_use=%s: model creation %.2f s, median %.1f ms per boosting iteration (gradient Psi^-1(F-y), tree, Newton leaf values, one "
              "covariance-parameter step); cov pars %s; tree ensemble[:3] = %s" % (n, gpu, t_create, 1e3 * res[gpu]["t_iter"], res[gpu]["cov"], out[:3]), flush=True)
        okb(LB.LGBM_BoosterFree(bst)); okb(LB.LGBM_DatasetFree(ds))
        del mdl
    if "trees" in res:
        np.testing.assert_allclose(res["trees"]["pred"], res[True]["pred"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(res["trees"]["cov"], res[True]["cov"], rtol=1e-6)
        print("GPBoost n=%d: device_type=gpu trees under GPU_use=true give the ensemble and covariance parameters of the host trees; %.1f ms against %.1f ms per iteration"
              % (n, 1e3 * res["trees"]["t_iter"], 1e3 * res[True]["t_iter"]), flush=True)
    if skip_cpu:
        np.testing.assert_allclose(res[True]["cov"], [0.32799919, 0.29574258, 1.36517975], rtol=1e-6)
        np.testing.assert_allclose(res[True]["pred"][:3], [0.71453864, 0.51212445, 0.9169032], rtol=1e-6)
        print("GPBoost n=%d: GPU_use=true reproduces the CPU values of profiles/r04_d_routeB.log (model creation 14.92 s, 236254.6 ms per iteration there): "
              "model creation %.1fx, boosting iteration %.1fx faster" % (n, 14.92 / res[True]["t_create"], 236.2546 / res[True]["t_iter"]), flush=True)
        continue
    a, b = res[False], res[True]
    np.testing.assert_allclose(b["pred"], a["pred"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(b["cov"], a["cov"], rtol=1e-6)
    print("GPBoost n=%d: GPU_use=true (device neighbour search, y_aux and Newton leaf values from the resident factor) reproduces the CPU path; "
          "model creation %.1fx, boosting iteration %.1fx faster" % (n, a["t_create"] / b["t_create"], a["t_iter"] / b["t_iter"]), flush=True)

if MOCK:
    print("ROUTE B SEAMS ON THE CPU RESTATEMENT: OK")
    sys.exit(0)
if "--gpboost-only" in sys.argv or "--laplace-only" in sys.argv:
    print("ROUTE B (%s) ON MI355X: OK" % ("GPBoost iterations" if "--gpboost-only" in sys.argv else "Laplace seams"))
    sys.exit(0)

# ---- (2) trees ------------------------------------------------------------------------------------------------------------------
L = C.CDLL(LIBP)
L.LGBM_GetLastError.restype = C.c_char_p


def ok(rc):
    if rc != 0:
        raise RuntimeError(L.LGBM_GetLastError().decode())


NIT = 20
REG = " lambda_l1=2 lambda_l2=1 max_delta_step=0.5 path_smooth=20 min_gain_to_split=0.01"      # the other regularisation paths of the split search
SIZES = ((100000, 50, (("cpu", "cpu", ""), ("gpu", "gpu", ""), ("gpu_maxdepth", "gpu", " max_depth=8"), ("cpu_maxdepth", "cpu", " max_depth=8"),
                       ("cpu_reg", "cpu", REG), ("gpu_reg", "gpu", REG),
                       ("cpu_colsample", "cpu", " feature_fraction=0.8"), ("gpu_colsample", "gpu", " feature_fraction=0.8"),
                       ("cpu_bynode", "cpu", " feature_fraction_bynode=0.8"), ("gpu_bynode", "gpu", " feature_fraction_bynode=0.8"),
                       ("cpu_bag", "cpu", " bagging_fraction=0.7 bagging_freq=1 bagging_seed=3"), ("gpu_bag", "gpu", " bagging_fraction=0.7 bagging_freq=1 bagging_seed=3"),
                       ("cpu_bagsub", "cpu", " bagging_fraction=0.6 bagging_freq=3 bagging_seed=3"), ("gpu_bagsub", "gpu", " bagging_fraction=0.6 bagging_freq=3 bagging_seed=3"),
                       # the reference's other bin storages: row-wise multi-value bins for the histograms, 4-bit dense bins --
                       # CreateDeviceBins reads the bins through the Dataset's own group iterators, whatever the storage
                       ("cpu_rowwise", "cpu", " force_row_wise=true"), ("gpu_rowwise", "gpu", " force_row_wise=true"),
                       ("cpu_4bit", "cpu", "", " max_bin=15"), ("gpu_4bit", "gpu", "", " max_bin=15"))),
         (1000000, 50, (("cpu", "cpu", ""), ("gpu", "gpu", ""))))
if "--trees-only" in sys.argv or TEST:
    SIZES = SIZES[:1]
if "--gp-only" in sys.argv:
    SIZES = ()
for n, F, variants in SIZES:
  rng = np.random.default_rng(1)
  X = np.ascontiguousarray(rng.uniform(size=(n, F)))
  yb = (np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + 0.5 * rng.standard_normal(n)).astype(np.float32)
  pred = {}
  print("---- trees, n = %d, F = %d ----" % (n, F), flush=True)
  for var in variants:
      tag, dev, extra = var[:3]
      dsp = var[3] if len(var) > 3 else " max_bin=255"
      ds = C.c_void_p()
      ok(L.LGBM_DatasetCreateFromMat(X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1),
                                     C.c_char_p(("verbosity=-1 device_type=%s%s%s" % (dev, dsp, extra)).encode()), C.c_void_p(), C.byref(ds)))
      ok(L.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yb.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(0)))
      bst = C.c_void_p()
      params = "objective=regression num_leaves=31 learning_rate=0.1 min_data_in_leaf=20 verbosity=1 device_type=%s num_threads=16%s%s" % (dev, extra, dsp)
      ok(L.LGBM_BoosterCreate(ds, C.c_char_p(params.encode()), C.byref(bst)))
      fin = C.c_int(0)
      ok(L.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))          # first iteration: allocations, uploads
      t0 = time.perf_counter()
      for _ in range(NIT - 1):
          ok(L.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
      dt = time.perf_counter() - t0
      out = np.empty(n)
      olen = C.c_int64(0)
      ok(L.LGBM_BoosterPredictForMat(bst, X.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(F), C.c_int(1), C.c_int(0), C.c_int(0),
                                     C.c_int(-1), C.c_char_p(b""), C.byref(olen), out.ctypes.data_as(C.POINTER(C.c_double))))
      pred[tag] = out
      print("%s (device_type=%s%s): %.2f ms per LGBM_BoosterUpdateOneIter over %d iterations; prediction[:3] = %s" % (tag, dev, extra, dt / (NIT - 1) * 1e3, NIT - 1, out[:3]), flush=True)
      ok(L.LGBM_BoosterFree(bst)); ok(L.LGBM_DatasetFree(ds))
  # whole trees on the device (num_leaves, max_depth) and the reference's own Train with device histograms (column sampling) against the CPU learner
  np.testing.assert_allclose(pred["gpu"], pred["cpu"], rtol=0, atol=1e-9)
  if "gpu_maxdepth" in pred:
    np.testing.assert_allclose(pred["gpu_maxdepth"], pred["cpu_maxdepth"], rtol=0, atol=1e-9)
  if "gpu_colsample" in pred:      # the per-tree column sample of the reference's ColSampler is handed to the device grower
    np.testing.assert_allclose(pred["gpu_colsample"], pred["cpu_colsample"], rtol=0, atol=1e-9)
  if "gpu_bag" in pred:            # bagging on the full Dataset: the device grower starts from the bag's rows
    np.testing.assert_allclose(pred["gpu_bag"], pred["cpu_bag"], rtol=0, atol=1e-9)
  for t in ("rowwise", "4bit"):
    if "gpu_" + t in pred:
      np.testing.assert_allclose(pred["gpu_" + t], pred["cpu_" + t], rtol=0, atol=1e-9)
  if "gpu_bagsub" in pred:         # small bags: the reference copies a subset Dataset; the device keeps the full data's bins and starts from the bag's rows
    np.testing.assert_allclose(pred["gpu_bagsub"], pred["cpu_bagsub"], rtol=0, atol=1e-9)
  if "gpu_bynode" in pred:         # per-node column sampling is not restated by the device grower: SerialTreeLearner::Train + device histograms
    np.testing.assert_allclose(pred["gpu_bynode"], pred["cpu_bynode"], rtol=0, atol=1e-9)
  if "gpu_reg" in pred:
    np.testing.assert_allclose(pred["gpu_reg"], pred["cpu_reg"], rtol=0, atol=1e-9)
    print("trees with lambda_l1 / max_delta_step / path_smooth: whole trees on the device reproduce device_type=cpu, max |diff| = %.2e" % np.abs(pred["gpu_reg"] - pred["cpu_reg"]).max(), flush=True)
  print("trees (n = %d): device_type=gpu (HIPTreeLearner, whole trees) reproduces device_type=cpu, max |diff| = %.2e" % (n, np.abs(pred["gpu"] - pred["cpu"]).max()), flush=True)

# ---- (2b) round 5: categorical columns and exclusive sparse columns (the reference bundles them, EFB) -- HIPTreeLearner keeps one column per FEATURE on the
#      device and grows whole trees with the categorical search (gpb_hip_hist_set_categorical); no CPU-histogram fallback ------------------------------
if "--gp-only" not in sys.argv:
    n = 20000 if TEST else 100000
    rng = np.random.default_rng(5)
    Xc = np.zeros((n, 14))
    Xc[:, :4] = rng.uniform(size=(n, 4))
    Xc[:, 4] = rng.integers(0, 5, size=n)
    Xc[:, 5] = rng.integers(0, 60, size=n) * (rng.uniform(size=n) < 0.85)
    c12 = rng.integers(0, 12, size=n)
    for k in range(8):
        Xc[:, 6 + k] = (c12 == k) * rng.uniform(0.5, 2.0, size=n)
    eff = rng.standard_normal(60)
    yc = (np.sin(4 * Xc[:, 0]) + Xc[:, 1] ** 2 + 0.7 * (Xc[:, 4] == 2) + eff[Xc[:, 5].astype(int)] + Xc[:, 7] - 0.5 * Xc[:, 9] + 0.4 * rng.standard_normal(n)).astype(np.float32)
    Xc = np.ascontiguousarray(Xc)
    predc = {}
    print("---- trees with categorical and bundled columns, n = %d, F = 14 ----" % n, flush=True)
    for tag, dev, extra in (("cpu", "cpu", ""), ("gpu", "gpu", ""), ("cpu_cfg", "cpu", " max_cat_to_onehot=8 cat_smooth=3 cat_l2=1 min_data_per_group=40 lambda_l1=0.5 path_smooth=5"),
                            ("gpu_cfg", "gpu", " max_cat_to_onehot=8 cat_smooth=3 cat_l2=1 min_data_per_group=40 lambda_l1=0.5 path_smooth=5")):
        ds = C.c_void_p()
        ok(L.LGBM_DatasetCreateFromMat(Xc.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(14), C.c_int(1),
                                       C.c_char_p(("verbosity=1 device_type=%s max_bin=63 categorical_feature=4,5" % dev).encode()), C.c_void_p(), C.byref(ds)))
        ok(L.LGBM_DatasetSetField(ds, C.c_char_p(b"label"), yc.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(0)))
        bst = C.c_void_p()
        params = "objective=regression num_leaves=31 learning_rate=0.1 min_data_in_leaf=20 verbosity=1 device_type=%s num_threads=16 max_bin=63 categorical_feature=4,5%s" % (dev, extra)
        ok(L.LGBM_BoosterCreate(ds, C.c_char_p(params.encode()), C.byref(bst)))
        fin = C.c_int(0)
        ok(L.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
        t0 = time.perf_counter()
        for _ in range(NIT - 1):
            ok(L.LGBM_BoosterUpdateOneIter(bst, C.byref(fin)))
        dt = time.perf_counter() - t0
        out = np.empty(n); olen = C.c_int64(0)
        ok(L.LGBM_BoosterPredictForMat(bst, Xc.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int32(n), C.c_int32(14), C.c_int(1), C.c_int(0), C.c_int(0),
                                       C.c_int(-1), C.c_char_p(b""), C.byref(olen), out.ctypes.data_as(C.POINTER(C.c_double))))
        # the model text names the categorical nodes (decision_type bit 0, num_cat)
        blen = C.c_int64(0); buf = C.create_string_buffer(1 << 22)
        ok(L.LGBM_BoosterSaveModelToString(bst, C.c_int(0), C.c_int(-1), C.c_int(0), C.c_int64(len(buf)), C.byref(blen), buf))
        ncat = sum(int(ln.split("=")[1]) for ln in buf.value.decode().splitlines() if ln.startswith("num_cat="))
        predc[tag] = (out, ncat)
        print("%s (device_type=%s%s): %.2f ms per LGBM_BoosterUpdateOneIter over %d iterations; categorical nodes in the model: %d; prediction[:3] = %s"
              % (tag, dev, extra, dt / (NIT - 1) * 1e3, NIT - 1, ncat, out[:3]), flush=True)
        ok(L.LGBM_BoosterFree(bst)); ok(L.LGBM_DatasetFree(ds))
    for a, b in (("gpu", "cpu"), ("gpu_cfg", "cpu_cfg")):
        np.testing.assert_allclose(predc[a][0], predc[b][0], rtol=0, atol=1e-9)
        assert predc[a][1] == predc[b][1] and predc[a][1] > 0
    print("trees with categorical + bundled columns (n = %d): device_type=gpu (HIPTreeLearner, whole trees, per-feature columns) reproduces device_type=cpu, "
          "max |diff| = %.2e / %.2e, %d / %d categorical nodes" % (n, np.abs(predc["gpu"][0] - predc["cpu"][0]).max(), np.abs(predc["gpu_cfg"][0] - predc["cpu_cfg"][0]).max(),
                                                                   predc["gpu"][1], predc["gpu_cfg"][1]), flush=True)
print("ROUTE B ON MI355X: OK", flush=True)
