"""Test harness: the product's host optimiser (gpboost_amd/csrc/gpb_optim.cpp, entry GPB_HIP_OptimizeGaussianWithCallback) driven by
the CPU oracle's likelihood / gradient instead of the device kernels, so that its control flow can be compared with the reference's
own optimisation trajectories without a GPU.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

TERMS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double))


def oracle_terms(orc, coords_ord, nn, cov_type, y_ord):
    """Callback returning the seven shard sums from the oracle: grad_k(s) = t_k1 / s + t_k2 evaluated at s = 1 and s = 2."""
    calls = []

    def fn(ctx, ratio, a, with_grad, t7):
        calls.append((ratio, a, with_grad))
        if with_grad:
            o1, g1 = orc.vecchia_nll_grad(coords_ord, nn, cov_type, (1.0, ratio, a), y_ord)
            o2, g2 = orc.vecchia_nll_grad(coords_ord, nn, cov_type, (2.0, ratio, a), y_ord)
            t7[0], t7[1], t7[2] = o1[0], o1[1], 0.0
            for k in (1, 2):
                ta = 2.0 * (g1[k] - g2[k])
                t7[1 + 2 * k], t7[2 + 2 * k] = ta, g1[k] - ta
        else:
            o1 = orc.vecchia_nll(coords_ord, nn, cov_type, (1.0, ratio, a), y_ord)
            t7[0], t7[1], t7[2] = o1[0], o1[1], 0.0
        return 0
    return TERMS_FN(fn), calls


def optimize(lib, n, init_theta, terms_cb, optimizer="lbfgs", lr_cov=-999., acc_rate_cov=-999., max_iter=-999, delta_rel_conv=-999.,
             use_nesterov_acc=True, nesterov_schedule_version=-999, momentum_offset=-999, convergence_criterion="default", m_lbfgs=-999,
             range_const=1.0, estimate_cov_par_index=None):
    th0 = np.ascontiguousarray(init_theta, dtype=np.float64)
    out = np.empty(3); nit = C.c_int(0); nll = C.c_double(0); ne = np.zeros(2, dtype=np.int32)
    lib.GPB_HIP_OptimizeGaussianWithCallback.argtypes = [
        C.c_int32, C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_bool, C.c_int, C.c_int, C.c_char_p, C.c_int,
        C.c_double, TERMS_FN, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
    est = None if estimate_cov_par_index is None else np.ascontiguousarray(estimate_cov_par_index, dtype=np.int32)
    lib.LGBM_GetLastError.restype = C.c_char_p
    rc = lib.GPB_HIP_OptimizeGaussianWithCallback(
        n, th0.ctypes.data, optimizer.encode(), lr_cov, acc_rate_cov, max_iter, delta_rel_conv, use_nesterov_acc, nesterov_schedule_version,
        momentum_offset, convergence_criterion.encode(), m_lbfgs, range_const, terms_cb, None, out.ctypes.data, C.byref(nit),
        C.byref(nll), ne.ctypes.data, None if est is None else est.ctypes.data)
    if rc != 0:
        raise RuntimeError(lib.LGBM_GetLastError().decode())
    return out, nit.value, nll.value, ne


LAPLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double))


class OracleLaplaceEvaluator(object):
    """Stateful evaluator of the Laplace approximation for GPB_HIP_OptimizeLaplaceWithCallback with the oracle behind it: keeps the
    mode (warm start), its previous value (op 3) and the gradient of the current state (op 2)."""

    def __init__(self, orc, coords_ord, nn, cov_type, y_ord, likelihood, cg_max_num_it=1000, cg_max_num_it_tridiag=1000, num_rand_vec=50,
                 cg_delta_conv=1e-2, fixed_effects_ord=None, delta_conv_mode=1e-8):
        self.orc, self.co, self.nn, self.ct, self.y, self.lik = orc, coords_ord, nn, cov_type, y_ord, likelihood
        self.cg, self.cgt, self.nrv, self.cgd = cg_max_num_it, cg_max_num_it_tridiag, num_rand_vec, cg_delta_conv
        self.dcm = delta_conv_mode
        self.fe = fixed_effects_ord
        self.mode = None; self.mode_prev = None; self.grad = None
        self.calls = []
        self.cb = LAPLACE_FN(self._fn)

    def _fn(self, ctx, op, var, a, out3):
        first_update = bool(op & 16); op &= 15
        self.calls.append((op, first_update, var, a))
        if op == 3:
            self.mode = None if self.mode_prev is None else self.mode_prev.copy()
            return 0
        if op == 4:
            self.mode = None; self.mode_prev = None
            return 0
        if op == 2:
            out3[1], out3[2] = self.grad
            return 0
        div = 3.0 if first_update else 1.0
        self.mode_prev = None if self.mode is None else self.mode.copy()
        nll, g, mode = self.orc.vecchia_laplace_grad(self.co, self.nn, self.ct, var, a, self.y, likelihood=self.lik, mode_init=self.mode,
                                                     want_mode=True, cg_max_num_it=int(round(self.cg / div)),
                                                     cg_max_num_it_tridiag=int(round(self.cgt / div)), num_rand_vec=self.nrv,
                                                     cg_delta_conv=self.cgd, fixed_effects=self.fe, delta_conv_mode=self.dcm)
        if self.mode_prev is None:
            self.mode_prev = np.zeros_like(mode)            # InitializeModeAvec: mode and its previous value start at 0
        self.mode, self.grad = mode, (g[0], g[1])
        out3[0], out3[1], out3[2] = nll, g[0], g[1]
        return 0


def optimize_laplace(lib, init_theta2, evaluator, optimizer="lbfgs", lr_cov=-999., acc_rate_cov=-999., max_iter=-999, delta_rel_conv=-999.,
                     use_nesterov_acc=True, nesterov_schedule_version=-999, momentum_offset=-999, convergence_criterion="default", m_lbfgs=-999):
    th0 = np.ascontiguousarray(init_theta2, dtype=np.float64)
    out = np.empty(2); nit = C.c_int(0); nll = C.c_double(0); ne = C.c_int(0)
    lib.GPB_HIP_OptimizeLaplaceWithCallback.argtypes = [
        C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_bool, C.c_int, C.c_int, C.c_char_p, C.c_int, LAPLACE_FN,
        C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.LGBM_GetLastError.restype = C.c_char_p
    rc = lib.GPB_HIP_OptimizeLaplaceWithCallback(
        th0.ctypes.data, optimizer.encode(), lr_cov, acc_rate_cov, max_iter, delta_rel_conv, use_nesterov_acc, nesterov_schedule_version,
        momentum_offset, convergence_criterion.encode(), m_lbfgs, evaluator.cb, None, out.ctypes.data, C.byref(nit), C.byref(nll), C.byref(ne))
    if rc != 0:
        raise RuntimeError(lib.LGBM_GetLastError().decode())
    return out, nit.value, nll.value, ne.value


LAPLACE_FE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


class OracleLaplaceFEEvaluator(object):
    """gpb_laplace_fe_fn (gpboost_amd/csrc/gpb_optim.h) with the oracle behind it: the linear predictor arrives as fixed effects (data order), the
    boosting gradient d(-mll)/dF goes back (data order).  Keeps the mode (warm start), its previous value (op 3) and the gradients of the current
    state (op 2)."""

    def __init__(self, orc, coords_ord, nn, cov_type, y_ord, likelihood, perm):
        self.orc, self.co, self.nn, self.ct, self.y, self.lik, self.perm = orc, coords_ord, nn, cov_type, y_ord, likelihood, perm
        self.n = len(y_ord)
        self.mode = None; self.mode_prev = None; self.grad = None; self.gF = None
        self.calls = []
        self.cb = LAPLACE_FE_FN(self._fn)

    def _fn(self, ctx, op, var, a, fe, out3, gradF):
        self.calls.append((op, var, a))
        if op == 3:
            self.mode = None if self.mode_prev is None else self.mode_prev.copy()
            return 0
        if op == 4:
            self.mode = None; self.mode_prev = None
            return 0
        if op in (0, 1):
            fe_data = np.ctypeslib.as_array(fe, shape=(self.n,)).copy()
            fe_ord = fe_data[self.perm]
            self.mode_prev = None if self.mode is None else self.mode.copy()
            nll, g, parts = self.orc.vecchia_laplace_grad(self.co, self.nn, self.ct, var, a, self.y, likelihood=self.lik, mode_init=self.mode,
                                                          fixed_effects=fe_ord, want_parts=True)
            if self.mode_prev is None:
                self.mode_prev = np.zeros(self.n)
            self.mode = parts["mode"].copy()
            first, info, _ = self.orc._lik_terms(self.lik, self.y, self.mode + fe_ord)
            gF_ord = -first + 0.5 * parts["dlogdet_dmode"] - info * parts["implicit_solve"]
            self.gF = np.empty(self.n); self.gF[self.perm] = gF_ord
            self.grad = (g[0], g[1])
            out3[0] = nll
            if op == 0:
                return 0
        out3[1], out3[2] = self.grad
        for i in range(self.n):
            gradF[i] = self.gF[i]
        return 0


def optimize_laplace_coef(lib, likelihood, X, y, init_theta2, evaluator, fixed_effects=None, init_coef=None, lr_cov=-999., max_iter=-999,
                          delta_rel_conv=-999., m_lbfgs=-999, init_from_iid_model=False, want_init_coef=False):
    """GPB_HIP_OptimizeLaplaceCoefWithCallback -> ((sigma1_2, a), coefficients on the original scale, iterations, negll)."""
    Xf = np.asfortranarray(X, dtype=np.float64)
    n, p = Xf.shape
    yv = np.ascontiguousarray(y, dtype=np.float64)
    th0 = np.ascontiguousarray(init_theta2, dtype=np.float64)
    fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
    ic = None if init_coef is None else np.ascontiguousarray(init_coef, dtype=np.float64)
    out = np.empty(2); coef = np.empty(p); nit = C.c_int(0); nll = C.c_double(0)
    lib.GPB_HIP_OptimizeLaplaceCoefWithCallback.argtypes = [
        C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_bool, C.c_double, C.c_int, C.c_double, C.c_int,
        LAPLACE_FE_FN, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p]
    ic_out = np.full(p, np.nan)
    lib.LGBM_GetLastError.restype = C.c_char_p
    rc = lib.GPB_HIP_OptimizeLaplaceCoefWithCallback(
        likelihood.encode(), n, p, Xf.ctypes.data, yv.ctypes.data, None if fe is None else fe.ctypes.data, th0.ctypes.data,
        None if ic is None else ic.ctypes.data, bool(init_from_iid_model), lr_cov, max_iter, delta_rel_conv, m_lbfgs, evaluator.cb, None,
        out.ctypes.data, coef.ctypes.data, C.byref(nit), C.byref(nll), ic_out.ctypes.data)
    if rc != 0:
        raise RuntimeError(lib.LGBM_GetLastError().decode())
    if want_init_coef:
        return out, coef, nit.value, nll.value, ic_out
    return out, coef, nit.value, nll.value


LAPLACE_AUX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))


class OracleLaplaceAuxEvaluator(object):
    """gpb_laplace_aux_fn (gpboost_amd/csrc/gpb_optim.h) with the oracle behind it: likelihoods whose auxiliary parameter (gamma / negative_binomial: the
    shape) is estimated with the covariance parameters.  Keeps the mode (warm start), its previous value (op 3) and the gradient of the current state."""

    def __init__(self, orc, coords_ord, nn, cov_type, y_ord, likelihood, cg_delta_conv=1e-2, delta_conv_mode=1e-8, fixed_effects_ord=None):
        self.orc, self.co, self.nn, self.ct, self.y, self.lik = orc, coords_ord, nn, cov_type, y_ord, likelihood
        self.cgd, self.dcm, self.fe = cg_delta_conv, delta_conv_mode, fixed_effects_ord
        self.mode = None; self.mode_prev = None; self.grad = None
        self.calls = []
        self.cb = LAPLACE_AUX_FN(self._fn)

    def _fn(self, ctx, op, var, a, aux, naux, out):
        self.calls.append((op, var, a, aux[0] if (naux and aux) else None))
        if op == 3:
            self.mode = None if self.mode_prev is None else self.mode_prev.copy()
            return 0
        if op == 4:
            self.mode = None; self.mode_prev = None
            return 0
        if op == 2:
            out[1], out[2], out[3] = self.grad
            return 0
        self.mode_prev = None if self.mode is None else self.mode.copy()
        nll, g, mode = self.orc.vecchia_laplace_grad(self.co, self.nn, self.ct, var, a, self.y, likelihood=self.lik, mode_init=self.mode, want_mode=True,
                                                     cg_delta_conv=self.cgd, delta_conv_mode=self.dcm, fixed_effects=self.fe, aux=aux[0])
        if self.mode_prev is None:
            self.mode_prev = np.zeros_like(mode)
        self.mode, self.grad = mode, (g[0], g[1], g[2])
        out[0] = nll
        if op == 1:
            out[1], out[2], out[3] = self.grad
        return 0


def optimize_laplace_aux(lib, init_theta2, init_aux, evaluator, lr_cov=-999., max_iter=-999, delta_rel_conv=-999., m_lbfgs=-999):
    """GPB_HIP_OptimizeLaplaceAuxWithCallback -> ((sigma1_2, a), aux, iterations, negll, evaluations)."""
    th0 = np.ascontiguousarray(init_theta2, dtype=np.float64)
    a0 = np.ascontiguousarray(np.atleast_1d(init_aux), dtype=np.float64)
    out = np.empty(2); aout = np.empty(a0.size); nit = C.c_int(0); nll = C.c_double(0); ne = C.c_int(0)
    lib.GPB_HIP_OptimizeLaplaceAuxWithCallback.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, LAPLACE_AUX_FN,
                                                           C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.LGBM_GetLastError.restype = C.c_char_p
    rc = lib.GPB_HIP_OptimizeLaplaceAuxWithCallback(th0.ctypes.data, a0.ctypes.data, a0.size, lr_cov, max_iter, delta_rel_conv, m_lbfgs, evaluator.cb, None,
                                                    out.ctypes.data, aout.ctypes.data, C.byref(nit), C.byref(nll), C.byref(ne))
    if rc != 0:
        raise RuntimeError(lib.LGBM_GetLastError().decode())
    return out, aout, nit.value, nll.value, ne.value
