"""ctypes access to oracle/_ref/libref_driver.so (the unmodified reference behind oracle/ref_driver.cpp).
TEST INFRASTRUCTURE ONLY.  ``available()`` is False on machines without a prebuilt oracle/_ref."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref_driver.so")
_L = None


def available():
    return os.path.exists(_PATH) and os.path.exists(os.path.join(_HERE, "_ref", "lib_gpboost_ref.so"))


def _lib():
    global _L
    if _L is None:
        C.CDLL(os.path.join(_HERE, "_ref", "lib_gpboost_ref.so"), mode=C.RTLD_GLOBAL)
        _L = C.CDLL(_PATH)
        _L.refdrv_create.restype = C.c_void_p
        _L.refdrv_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_double, C.c_int, C.c_char_p,
                                     C.c_int, C.c_char_p, C.c_int]
    return _L


def _P(a):
    return a.ctypes.data_as(C.c_void_p)


class RefModel(object):
    """One Gaussian Vecchia GP built by the reference's own REModel constructor."""

    def __init__(self, coords, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=0, threads=8):
        cm = np.asfortranarray(coords, dtype=np.float64)
        self.n, self.d = cm.shape
        self.m = min(m, self.n - 1)
        h = _lib().refdrv_create(self.n, _P(cm), self.d, cov_function.encode(), float(shape), int(m),
                                 ordering.encode(), int(seed), b"gaussian", int(threads))
        if not h:
            raise RuntimeError("reference model creation failed")
        self.h = C.c_void_p(h)

    def __del__(self):
        try:
            _lib().refdrv_free(self.h)
        except Exception:
            pass

    def perm(self):
        p = np.empty(self.n, dtype=np.int32)
        _lib().refdrv_get_perm(self.h, _P(p))
        return p

    def neighbors(self):
        nn = np.empty((self.n, self.m), dtype=np.int32)
        _lib().refdrv_get_neighbors(self.h, C.c_int(self.m), _P(nn))
        return nn

    def nll_grad(self, y, cov_pars):
        y = np.ascontiguousarray(y, dtype=np.float64)
        cp = np.ascontiguousarray(cov_pars, dtype=np.float64)
        nll = C.c_double(0); g = np.empty(3); pt = np.empty(3)
        rc = _lib().refdrv_nll_grad(self.h, _P(y), _P(cp), C.byref(nll), _P(g), _P(pt))
        if rc != 0:
            raise RuntimeError("refdrv_nll_grad failed")
        return nll.value, g, pt

    def factor(self):
        """(A, D, y_aux_vecchia_order) of the last nll_grad call."""
        A = np.empty((self.n, self.m)); Di = np.empty(self.n); ya = np.empty(self.n)
        rc = _lib().refdrv_get_factor(self.h, C.c_int(self.m), _P(A), _P(Di), _P(ya))
        if rc != 0:
            raise RuntimeError("refdrv_get_factor failed")
        return A, 1. / Di, ya

    def newton_leaf_values(self, data_leaf_index, num_leaves, marg_variance):
        """The reference's NewtonUpdateLeafValues for the state of the last nll_grad call (y passed there = F - y)."""
        lf = np.ascontiguousarray(data_leaf_index, dtype=np.int32)
        out = np.empty(num_leaves)
        rc = _lib().refdrv_newton_leaf(self.h, _P(lf), C.c_int(num_leaves), C.c_double(marg_variance), _P(out))
        if rc != 0:
            raise RuntimeError("refdrv_newton_leaf failed")
        return out


class RefVifModel(RefModel):
    """One Gaussian full-scale Vecchia (VIF) model built by the reference's own REModel constructor (gp_approx = "full_scale_vecchia"):
    nll_grad / perm / neighbors as RefModel; grad_factor() = the derivative factors of the residual process of the last nll_grad call."""

    def __init__(self, coords, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=0, num_ind_points=200, threads=8):
        cm = np.asfortranarray(coords, dtype=np.float64)
        self.n, self.d = cm.shape
        self.m = min(m, self.n - 1)
        L = _lib()
        L.refdrv_create_vif.restype = C.c_void_p
        L.refdrv_create_vif.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_double, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        h = L.refdrv_create_vif(self.n, _P(cm), self.d, cov_function.encode(), float(shape), int(m), ordering.encode(), int(seed),
                                int(num_ind_points), int(threads))
        if not h:
            raise RuntimeError("reference VIF model creation failed")
        self.h = C.c_void_p(h)

    def grad_factor(self, ipar):
        """(dA, dD) of parameter ipar (0: variance, 1: range) wrt the log of the transformed parameter, aligned with neighbors()."""
        dA = np.empty((self.n, self.m)); dD = np.empty(self.n)
        if _lib().refdrv_get_grad_factor(self.h, C.c_int(self.m), C.c_int(ipar), _P(dA), _P(dD)) != 0:
            raise RuntimeError("refdrv_get_grad_factor failed")
        return dA, dD

    def yaux(self):
        ya = np.empty(self.n)
        if _lib().refdrv_get_yaux(self.h, _P(ya)) < 0:
            raise RuntimeError("refdrv_get_yaux failed")
        return ya


# ---------------------------------------------------------------------------------------------
# The reference's own public C API (lib_gpboost_ref.so), bound the way python-package/gpboost/basic.py:5206-5240
# binds it: used as the "reference" CPU baseline of bench.py.
# ---------------------------------------------------------------------------------------------
class RefCAPIModel(object):
    def __init__(self, coords, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=1, threads=-1,
                 likelihood="gaussian", cluster_ids=None, lib_path=None, gpu_use=False, gp_approx="vecchia", num_ind_points=500, weights=None,
                 matrix_inversion_method="default", likelihood_additional_param=-999.):
        # lib_path / gpu_use: the route-B build (integration/Makefile.routeB -> integration/_build/lib_gpboost_hip.so), the same C API with GPU_use = true
        self.L = C.CDLL(lib_path or os.path.join(_HERE, "_ref", "lib_gpboost_ref.so"))
        self.L.LGBM_GetLastError.restype = C.c_char_p
        cm = np.asfortranarray(coords, dtype=np.float64)
        self.n, self.d = cm.shape
        self.h = C.c_void_p()
        s = lambda x: C.c_char_p(x.encode())
        cid = None if cluster_ids is None else np.ascontiguousarray(cluster_ids, dtype=np.int32)
        rc = self.L.GPB_CreateREModel(
            C.c_int(self.n), C.c_void_p() if cid is None else _P(cid), C.c_void_p(), C.c_int(0), C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_void_p(),
            C.c_int(1), _P(cm), C.c_int(self.d), C.c_void_p(), C.c_int(0), s(cov_function), C.c_double(shape), s(gp_approx),
            C.c_double(1.), C.c_double(0.), C.c_int(m), s(ordering), C.c_int(int(num_ind_points)), C.c_double(1.), s("kmeans++"),
            s(likelihood), C.c_double(float(likelihood_additional_param)), s(matrix_inversion_method), C.c_int(seed), C.c_int(threads), C.c_bool(bool(gpu_use)),
            C.c_bool(weights is not None), C.c_void_p() if weights is None else _P(self._w(weights)), C.c_double(1.), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())

    def _w(self, weights):
        self._weights = np.ascontiguousarray(weights, dtype=np.float64)      # kept alive for the constructor call
        return self._weights

    def neg_log_likelihood(self, cov_pars, y, fixed_effects=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        cp = np.ascontiguousarray(cov_pars, dtype=np.float64)
        fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
        out = C.c_double(0)
        rc = self.L.GPB_EvalNegLogLikelihood(self.h, _P(y), _P(cp), C.c_void_p() if fe is None else _P(fe), C.byref(out))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out.value

    def set_optim_config(self, init_cov_pars=None, lr_cov=-999., acc_rate_cov=-999., max_iter=-999, delta_rel_conv=-999.,
                         use_nesterov_acc=True, nesterov_schedule_version=-999, trace=False, optimizer_cov="", momentum_offset=-999,
                         convergence_criterion="default", m_lbfgs=-999, cg_delta_conv=-999., delta_conv_mode_finding=-999.,
                         init_coef_aux_pars_from_iid_model=False, estimate_cov_par_index=None, init_aux_pars=None, estimate_aux_pars=False,
                         cg_preconditioner_type="vadu", piv_chol_rank=-999):
        """GPB_SetOptimConfig with the argument order of include/LightGBM/c_api.h:1437-1467 (basic.py:5460-5496 binds it the same way)."""
        s = lambda x: C.c_char_p(x.encode())
        ic = None if init_cov_pars is None else np.ascontiguousarray(init_cov_pars, dtype=np.float64)
        est = np.array([-1], dtype=np.int32) if estimate_cov_par_index is None else np.ascontiguousarray(estimate_cov_par_index, dtype=np.int32)
        ia = None if init_aux_pars is None else np.ascontiguousarray(np.atleast_1d(init_aux_pars), dtype=np.float64)
        rc = self.L.GPB_SetOptimConfig(
            self.h, C.c_void_p() if ic is None else _P(ic), C.c_double(lr_cov), C.c_double(acc_rate_cov), C.c_int(max_iter),
            C.c_double(delta_rel_conv), C.c_bool(use_nesterov_acc), C.c_int(nesterov_schedule_version), C.c_bool(trace), s(optimizer_cov),
            C.c_int(momentum_offset), s(convergence_criterion), C.c_int(0), C.c_void_p(), C.c_double(-999.), C.c_double(-999.), s(""),
            C.c_int(-999), C.c_int(-999), C.c_double(cg_delta_conv), C.c_int(-999), C.c_bool(True), s(cg_preconditioner_type), C.c_int(1), C.c_int(piv_chol_rank),
            C.c_void_p() if ia is None else _P(ia), C.c_bool(bool(estimate_aux_pars)), C.c_bool(bool(init_coef_aux_pars_from_iid_model)), _P(est), C.c_int(m_lbfgs),
            C.c_double(delta_conv_mode_finding))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())

    def get_aux_pars(self, num=1, std_dev=False):
        """GPB_GetAuxPars (c_api.h:1804-1807) -> the first `num` auxiliary parameters (original scale); std_dev: followed by their `num` standard deviations."""
        out = np.zeros(max(num, 1) * 2); name = C.create_string_buffer(256)
        rc = self.L.GPB_GetAuxPars(self.h, _P(out), name, C.c_bool(bool(std_dev)))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out[:num * (2 if std_dev else 1)].copy()

    def get_init_aux_pars(self, num=1):
        """GPB_GetInitAuxPars (c_api.h:1824-1825)."""
        out = np.zeros(max(num, 1))
        rc = self.L.GPB_GetInitAuxPars(self.h, _P(out))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out[:num].copy()

    def optim_cov_par(self, y, fixed_effects=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
        rc = self.L.GPB_OptimCovPar(self.h, _P(y), C.c_void_p() if fe is None else _P(fe))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())

    def optim_lin_regr_coef_cov_par(self, y, X, fixed_effects=None):
        """GPB_OptimLinRegrCoefCovPar (c_api.h:1490-1494); X: n x p."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        Xf = np.asfortranarray(X, dtype=np.float64)
        fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
        self.p = Xf.shape[1]
        rc = self.L.GPB_OptimLinRegrCoefCovPar(self.h, _P(y), _P(Xf), C.c_int(self.p), C.c_void_p() if fe is None else _P(fe))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())

    def get_coef(self, std_dev=False):
        out = np.empty(self.p * (2 if std_dev else 1))
        rc = self.L.GPB_GetCoef(self.h, _P(out), C.c_bool(bool(std_dev)))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out

    def predict(self, coords_pred, X_pred=None, predict_var=True, predict_response=True, vecchia_pred_type=None, num_neighbors_pred=-1,
                y=None, cov_pars=None, predict_cov_mat=False):
        """GPB_SetPredictionData + GPB_PredictREModel (c_api.h:1594-1660) with the response / parameters of the last fit, or with the given
        y / cov_pars.  predict_cov_mat: the second return value is the n_pred x n_pred predictive covariance matrix."""
        s = lambda x: C.c_char_p(x.encode())
        cp = np.asfortranarray(coords_pred, dtype=np.float64)
        npred = cp.shape[0]
        if vecchia_pred_type is not None or num_neighbors_pred > 0:
            rc = self.L.GPB_SetPredictionData(self.h, C.c_int(0), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(),
                                              C.c_void_p() if vecchia_pred_type is None else s(vecchia_pred_type), C.c_int(num_neighbors_pred),
                                              C.c_double(-1.), C.c_int(-1), C.c_int(-1))
            if rc != 0:
                raise RuntimeError(self.L.LGBM_GetLastError().decode())
        Xp = None if X_pred is None else np.asfortranarray(X_pred, dtype=np.float64)
        if predict_cov_mat:
            predict_var = False
        out = np.empty(npred * (1 + npred) if predict_cov_mat else npred * (2 if predict_var else 1))
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
        cv = None if cov_pars is None else np.ascontiguousarray(cov_pars, dtype=np.float64)
        rc = self.L.GPB_PredictREModel(self.h, C.c_void_p() if yv is None else _P(yv), C.c_int(npred), _P(out), C.c_bool(bool(predict_cov_mat)),
                                       C.c_bool(bool(predict_var)),
                                       C.c_bool(bool(predict_response)), C.c_bool(False), C.c_bool(False), C.c_int(0), C.c_int(0), C.c_void_p(), C.c_void_p(),
                                       C.c_void_p(), _P(cp), C.c_void_p(), C.c_void_p() if cv is None else _P(cv), C.c_void_p() if Xp is None else _P(Xp),
                                       C.c_bool(False), C.c_void_p(), C.c_void_p())
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        if predict_cov_mat:
            return out[:npred].copy(), out[npred:].reshape(npred, npred).copy()
        return out[:npred].copy(), (out[npred:].copy() if predict_var else None)

    def predict_training_data_random_effects(self, y, cov_pars, calc_var=True):
        """GPB_PredictREModelTrainingDataRandomEffects (c_api.h:1672-1680): (mean, var) of the latent GP at the training locations."""
        yv = np.ascontiguousarray(y, dtype=np.float64); cp = np.ascontiguousarray(cov_pars, dtype=np.float64)
        out = np.empty(self.n * (2 if calc_var else 1))
        rc = self.L.GPB_PredictREModelTrainingDataRandomEffects(self.h, _P(cp), _P(yv), _P(out), C.c_void_p(), C.c_bool(bool(calc_var)))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out[:self.n].copy(), (out[self.n:].copy() if calc_var else None)

    def get_cov_par(self, num_cov_pars=3, std_dev=False):
        out = np.empty(num_cov_pars * (2 if std_dev else 1))
        rc = self.L.GPB_GetCovPar(self.h, _P(out), C.c_bool(bool(std_dev)))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out

    def get_init_cov_par(self):
        out = np.empty(3)
        rc = self.L.GPB_GetInitCovPar(self.h, _P(out))
        if rc != 0:
            raise RuntimeError(self.L.LGBM_GetLastError().decode())
        return out

    def get_num_it(self):
        out = C.c_int(0)
        self.L.GPB_GetNumIt(self.h, C.byref(out))
        return out.value

    def current_neg_log_likelihood(self):
        out = C.c_double(0)
        self.L.GPB_GetCurrentNegLogLikelihood(self.h, C.byref(out))
        return out.value

    def __del__(self):
        try:
            self.L.GPB_REModelFree(self.h)
        except Exception:
            pass


def ref_laplace_gradient(coords, y, cov_pars, likelihood, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=1, lr=1e-3,
                         threads=8, cg_delta_conv=-999.):
    """Gradient of the reference's approximate negative marginal log-likelihood wrt (log sigma1^2, log a) at cov_pars = (sigma1^2, rho),
    read off ONE plain gradient-descent step of its own optimiser: theta_1 = exp(log theta_0 - lr * grad) (re_model_template.h:8737-8742),
    so grad = -(log theta_1 - log theta_0) / lr exactly, provided the step was not halved -- checked by repeating with lr / 2."""
    res = []
    for step in (lr, lr / 2):
        mdl = RefCAPIModel(coords, cov_function, shape, m, ordering, seed, threads=threads, likelihood=likelihood)
        mdl.set_optim_config(init_cov_pars=np.asarray(cov_pars, dtype=np.float64), lr_cov=step, max_iter=1, use_nesterov_acc=False,
                             optimizer_cov="gradient_descent", cg_delta_conv=cg_delta_conv)
        mdl.optim_cov_par(y)
        th1 = mdl.get_cov_par(2)
        res.append(np.array([-(np.log(th1[0]) - np.log(cov_pars[0])) / step, (np.log(th1[1]) - np.log(cov_pars[1])) / step]))
    if not np.allclose(res[0], res[1], rtol=1e-6):
        raise RuntimeError("the reference halved its step: %s vs %s" % (res[0], res[1]))
    return res[1]


def ref_laplace_nll_grad(coords, y, cov_pars, likelihood, fixed_effects=None, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=1,
                         threads=8, cg_delta_conv=-999., delta_conv_mode_finding=-999., aux_pars=None, estimate_aux=False, weights=None,
                         cg_preconditioner_type="vadu", piv_chol_rank=-999, gp_approx="vecchia", num_ind_points=500):
    """(negll, grad): the reference's approximate negative marginal log-likelihood and its gradient wrt (log sigma1^2, log a[, log aux...]) at
    cov_pars = (sigma1^2, rho), from the reference's OWN CalcGradPars -> CalcGradNegMargLikelihoodLaplaceApproxVecchia
    (ref_driver.cpp: refdrv_laplace_nll_grad) with the solver thresholds given -- the pin of orc_vecchia_laplace_grad and of the device gradient."""
    mdl = RefCAPIModel(coords, cov_function, shape, m, ordering, seed, threads=threads, likelihood=likelihood, weights=weights, gp_approx=gp_approx,
                       num_ind_points=num_ind_points)
    mdl.set_optim_config(cg_delta_conv=cg_delta_conv, delta_conv_mode_finding=delta_conv_mode_finding, init_aux_pars=aux_pars, estimate_aux_pars=estimate_aux,
                         cg_preconditioner_type=cg_preconditioner_type, piv_chol_rank=piv_chol_rank)
    y = np.ascontiguousarray(y, dtype=np.float64)
    cp = np.ascontiguousarray(cov_pars, dtype=np.float64)
    fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
    nll_pub, nll_fun, ng = C.c_double(0), C.c_double(0), C.c_int(0)
    g = np.zeros(8)
    fn = _lib().refdrv_laplace_nll_grad
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if fn(mdl.h, y.ctypes.data, cp.ctypes.data, None if fe is None else fe.ctypes.data, 1 if estimate_aux else 0, C.addressof(nll_pub), C.addressof(nll_fun),
          g.ctypes.data, C.addressof(ng)) != 0:
        raise RuntimeError("refdrv_laplace_nll_grad failed")
    return nll_fun.value, g[:ng.value].copy(), nll_pub.value


def ref_laplace_grad_F(coords, y, cov_pars, likelihood, fixed_effects=None, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=1,
                       threads=8, cg_delta_conv=-999., delta_conv_mode_finding=-999., weights=None, cg_preconditioner_type="vadu", piv_chol_rank=-999,
                       gp_approx="vecchia", num_ind_points=500):
    """The reference's boosting gradient for non-Gaussian data, d(-approximate marginal log-likelihood) / dF in data order, at cov_pars =
    (sigma1^2, rho) and the fixed effects F (zero if None): REModel::CalcGradient on a model of the reference's own C API (ref_driver.cpp:
    refdrv_laplace_grad_F)."""
    mdl = RefCAPIModel(coords, cov_function, shape, m, ordering, seed, threads=threads, likelihood=likelihood, weights=weights, gp_approx=gp_approx, num_ind_points=num_ind_points)
    mdl.set_optim_config(init_cov_pars=np.asarray(cov_pars, dtype=np.float64), cg_delta_conv=cg_delta_conv, delta_conv_mode_finding=delta_conv_mode_finding,
                         cg_preconditioner_type=cg_preconditioner_type, piv_chol_rank=piv_chol_rank)
    y = np.ascontiguousarray(y, dtype=np.float64)
    fe = np.zeros_like(y) if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
    out = np.empty_like(y)
    fn = _lib().refdrv_laplace_grad_F
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if fn(mdl.h, y.ctypes.data, fe.ctypes.data, out.ctypes.data) != 0:
        raise RuntimeError("refdrv_laplace_grad_F failed")
    return out


def ref_histogram(X, max_bin, data_indices, grad, hess=None, const_hess=1.0, with_fix=False, extra_params="", split_cfg=None,
                  partitions=None, cat_cfg=None, partition_cat_bits=None):
    """The reference's own binning + Dataset::ConstructHistograms for one leaf.
    Returns (bins uint8 (G, n) = the reference's stored group bins, group_num_bin (G,), hist (sum bins, 2)); with_fix=True adds
    a dict with the per-feature view offsets / num_bin / most_freq_bin, the leaf sums and the histogram after
    Dataset::FixHistogram on every feature; split_cfg = (lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split)
    also runs the reference's FeatureHistogram::FindBestThreshold per feature (dict keys meta3, split, split_default_left);
    partitions = [(feature, threshold, default_left), ...] runs Dataset::Split of the leaf for each (dict key part_lte: list of arrays)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, F = X.shape
    di = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
    nd = n if di is None else di.size
    g = np.ascontiguousarray(grad, dtype=np.float64)
    h = None if hess is None else np.ascontiguousarray(hess, dtype=np.float64)
    ng = C.c_int(0)
    gnb = np.zeros(F, dtype=np.int32)
    bins = np.zeros((F, n), dtype=np.uint8)
    hist = np.zeros((F * (max_bin + 3), 2))
    voff = np.zeros(F, dtype=np.int32); nbin = np.zeros(F, dtype=np.int32); mfb = np.zeros(F, dtype=np.int32)
    sums = np.zeros(2); hfix = np.zeros((F * (max_bin + 3), 2))
    cfg = None
    if split_cfg is not None:      # (lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split[, lambda_l1, max_delta_step, path_smooth[, parent_output]])
        cfg = np.zeros(13); cfg[7:] = np.nan
        cfg[:len(split_cfg)] = split_cfg
        if cat_cfg is not None:    # (max_cat_to_onehot, max_cat_threshold, cat_smooth, cat_l2, min_data_per_group); categorical columns: extra_params "categorical_feature=..."
            cfg[8:13] = cat_cfg
    is_cat = np.zeros(F, dtype=np.int32); cat_bits = np.zeros((F, 8), dtype=np.uint32)
    meta3 = np.zeros((F, 3), dtype=np.int32); sp = np.zeros((F, 10)); sdl = np.zeros(F, dtype=np.int32)
    npart = 0 if partitions is None else len(partitions)
    ftd = np.ascontiguousarray(partitions if npart else np.zeros((1, 3)), dtype=np.int32)
    plte = np.zeros((max(npart, 1), nd), dtype=np.int32); pcnt = np.zeros(max(npart, 1), dtype=np.int32)
    rc = _lib().refdrv_hist(C.c_int(n), C.c_int(F), _P(X), C.c_int(max_bin), None if di is None else _P(di), C.c_int(nd), _P(g),
                            None if h is None else _P(h), C.c_double(const_hess), C.byref(ng), _P(gnb), _P(bins), _P(hist),
                            _P(voff), _P(nbin), _P(mfb), _P(sums), _P(hfix) if (with_fix or cfg is not None) else None,
                            C.c_char_p(extra_params.encode()), None if cfg is None else _P(cfg), _P(meta3),
                            None if cfg is None else _P(sp), _P(sdl), C.c_int(npart), _P(ftd), _P(plte), _P(pcnt), _P(is_cat), _P(cat_bits),
                            None if partition_cat_bits is None else _P(np.ascontiguousarray(partition_cat_bits, dtype=np.uint32)))
    if rc != 0:
        raise RuntimeError("refdrv_hist failed")
    G = ng.value
    tot = int(gnb[:G].sum())
    if with_fix or cfg is not None:
        out = dict(view_offset=voff, num_bin=nbin, most_freq_bin=mfb, sums=sums, hist_fixed=hfix[:tot].copy())
        if cfg is not None:
            out.update(meta3=meta3, split=sp, split_default_left=sdl, is_categorical=is_cat, split_cat_bits=cat_bits)
        if npart:
            out["part_lte"] = [plte[p, :pcnt[p]].copy() for p in range(npart)]
        return bins[:G].copy(), gnb[:G].copy(), hist[:tot].copy(), out
    return bins[:G].copy(), gnb[:G].copy(), hist[:tot].copy()


def ref_train_tree(X, params, grad, hess=None, max_leaves=64, unbundle=False):
    """One tree grown by the reference's own SerialTreeLearner::Train (single OpenMP thread) on its own Dataset.
    Returns a dict: bins (G, n), group_num_bin, view_offset / num_bin / most_freq_bin / meta3 per feature, num_leaves, and the tree
    arrays split_feature_inner, threshold_in_bin, default_left, left_child, right_child, split_gain, internal_count (num_leaves - 1)
    and leaf_value, leaf_count (num_leaves)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, F = X.shape
    g = np.ascontiguousarray(grad, dtype=np.float64)
    h = None if hess is None else np.ascontiguousarray(hess, dtype=np.float64)
    ng = C.c_int(0); nl = C.c_int(0)
    gnb = np.zeros(F, dtype=np.int32); bins = np.zeros((F, n), dtype=np.uint8)
    voff = np.zeros(F, dtype=np.int32); nbin = np.zeros(F, dtype=np.int32); mfb = np.zeros(F, dtype=np.int32); meta3 = np.zeros((F, 3), dtype=np.int32)
    L = int(max_leaves)
    ia = {k: np.zeros(L, dtype=np.int32) for k in ("split_feature_inner", "threshold_in_bin", "default_left", "left_child", "right_child",
                                                    "internal_count", "leaf_count")}
    gain = np.zeros(L); lv = np.zeros(L)
    layout = np.zeros((F, 4), dtype=np.int32); node_cat = np.zeros(L, dtype=np.int32); node_bits = np.zeros((L, 8), dtype=np.uint32)
    rc = _lib().refdrv_train_tree(C.c_int(n), C.c_int(F), _P(X), C.c_char_p(params.encode()), _P(g), None if h is None else _P(h),
                                  C.byref(ng), _P(gnb), _P(bins), _P(voff), _P(nbin), _P(mfb), _P(meta3), C.byref(nl),
                                  _P(ia["split_feature_inner"]), _P(ia["threshold_in_bin"]), _P(ia["default_left"]), _P(ia["left_child"]),
                                  _P(ia["right_child"]), _P(gain), _P(ia["internal_count"]), _P(lv), _P(ia["leaf_count"]), _P(layout), _P(node_cat),
                                  _P(node_bits), C.c_int(-1 if unbundle else F))
    if rc != 0:
        raise RuntimeError("refdrv_train_tree failed")
    G, nlv = ng.value, nl.value
    out = dict(bins=bins[:G].copy(), group_num_bin=gnb[:G].copy(), view_offset=voff, num_bin=nbin, most_freq_bin=mfb, meta3=meta3,
               num_leaves=nlv, split_gain=gain[:nlv - 1].copy(), leaf_value=lv[:nlv].copy(), leaf_count=ia["leaf_count"][:nlv].copy())
    for k in ("split_feature_inner", "threshold_in_bin", "default_left", "left_child", "right_child", "internal_count"):
        out[k] = ia[k][:nlv - 1].copy()
    # round 5: per feature (column, min_bin, max_bin, bin type) and per node (is categorical, bitset over the feature's bins of the categories going left)
    out.update(layout=layout, node_is_cat=node_cat[:nlv - 1].copy(), node_cat_bits=node_bits[:nlv - 1].copy())
    return out
