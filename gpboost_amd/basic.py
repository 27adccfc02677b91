"""Host-side mirror of the reference's ``gpboost.GPModel`` for the MI355X hot path.

Same constructor argument names, same ``neg_log_likelihood(cov_pars, y, fixed_effects)`` call and the
same error type as ``python-package/gpboost/basic.py`` (class GPModel :4172, ``neg_log_likelihood``
:5640-5700, ``_safe_call`` :136-145), bound with ctypes to the SAME C symbols
(``GPB_CreateREModel`` / ``GPB_EvalNegLogLikelihood`` / ``GPB_REModelFree`` / ``LGBM_GetLastError``), so the
parity tests read like the reference's own.  Only the model slice the hot path covers is accepted
(see include/gpboost_c_api_subset.h); everything else raises ``GPBoostError`` -- there is no fallback.
"""
import ctypes

import numpy as np

from .libpath import load_lib

_LIB = None


class GPBoostError(Exception):
    """Error thrown by the library (same name as the reference's, basic.py:131)."""


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = load_lib()
        _LIB.LGBM_GetLastError.restype = ctypes.c_char_p
        _LIB.gpb_hip_get_last_error.restype = ctypes.c_char_p
        _LIB.GPB_HIP_GetVecchiaHandle.restype = ctypes.c_void_p
        _LIB.GPB_HIP_GetVecchiaHandle.argtypes = [ctypes.c_void_p]
    return _LIB


def _safe_call(ret):
    if ret != 0:
        raise GPBoostError(_lib().LGBM_GetLastError().decode("utf-8"))


def _shim_call(ret):
    if ret != 0:
        raise GPBoostError(_lib().gpb_hip_get_last_error().decode("utf-8"))


def c_str(string):
    return ctypes.c_char_p(string.encode("utf-8"))


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def device_count():
    n = ctypes.c_int(0)
    _lib().gpb_hip_device_count(ctypes.byref(n))
    return n.value


def set_device(device):
    _shim_call(_lib().gpb_hip_set_device(ctypes.c_int(int(device))))


def selftest():
    """Runs the on-device check of the fp64 DPP primitives."""
    _shim_call(_lib().gpb_hip_selftest())


class GPModel(object):
    """Gaussian-process model evaluated on an MI355X (Vecchia approximation or exact GP with a Gaussian likelihood;
    Vecchia-Laplace approximation with likelihood="bernoulli_logit")."""

    def __init__(self, likelihood="gaussian", group_data=None, group_rand_coef_data=None,
                 ind_effect_group_rand_coef=None, drop_intercept_group_rand_effect=None, gp_coords=None,
                 gp_rand_coef_data=None, cov_function="matern", cov_fct_shape=1.5, gp_approx="none",
                 num_parallel_threads=None, GPU_use=True, matrix_inversion_method="default", weights=None,
                 likelihood_learning_rate=1., cov_fct_taper_range=1., cov_fct_taper_shape=1., num_neighbors=None,
                 vecchia_ordering="random", ind_points_selection="kmeans++", num_ind_points=None,
                 cover_tree_radius=1., seed=0, cluster_ids=None, num_data=None, likelihood_additional_param=None):
        self.handle = ctypes.c_void_p()
        if group_data is not None or group_rand_coef_data is not None:
            raise GPBoostError("grouped random effects are not on the MI355X hot path of this library")
        if gp_coords is None:
            raise ValueError("'gp_coords' is required")
        if gp_rand_coef_data is not None:
            raise GPBoostError("GP random coefficients are not on the MI355X hot path of this library")
        gp_coords = np.asarray(gp_coords, dtype=np.float64)
        if gp_coords.ndim == 1:
            gp_coords = gp_coords.reshape(-1, 1)
        self.num_data = gp_coords.shape[0]
        self.dim_coords = gp_coords.shape[1]
        self.cov_function = cov_function
        self.cov_fct_shape = float(cov_fct_shape)
        self.gp_approx = gp_approx
        self.vecchia_ordering = vecchia_ordering
        self.seed = int(seed)
        self.likelihood = likelihood
        # Gaussian: error variance, GP variance, range; non-Gaussian: GP variance, range (basic.py:4618-4624)
        self.num_cov_pars = 3 if likelihood == "gaussian" else 2
        # None / <= 0: the LIBRARY resolves the default as the reference's does (20 for "vecchia", 30 for "full_scale_vecchia",
        # re_model_template.h:287-298; the reference's package passes -1, basic.py:4590-4600)
        self.num_neighbors = -1 if (num_neighbors is None or num_neighbors <= 0) else int(num_neighbors)
        coords_c = np.asfortranarray(gp_coords)   # column-major, basic.py:5075-5081
        cluster_c = ctypes.c_void_p()
        if cluster_ids is not None:
            cid = np.ascontiguousarray(cluster_ids, dtype=np.int32)
            cluster_c = cid.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        weights_c = ctypes.c_void_p()
        if weights is not None:      # sample weights (reference: GPModel(weights=...), basic.py:4440-4470): one positive value per data point
            self._weights = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
            if self._weights.shape[0] != self.num_data:
                raise ValueError("Incorrect number of data points in 'weights'")
            weights_c = _dptr(self._weights)
        lap = -999. if likelihood_additional_param is None else float(likelihood_additional_param)
        _safe_call(_lib().GPB_CreateREModel(
            ctypes.c_int(self.num_data), cluster_c, ctypes.c_void_p(), ctypes.c_int(0), ctypes.c_void_p(),
            ctypes.c_void_p(), ctypes.c_int(0), ctypes.c_void_p(), ctypes.c_int(1), _dptr(coords_c),
            ctypes.c_int(self.dim_coords), ctypes.c_void_p(), ctypes.c_int(0), c_str(cov_function),
            ctypes.c_double(self.cov_fct_shape), c_str(gp_approx), ctypes.c_double(cov_fct_taper_range),
            ctypes.c_double(cov_fct_taper_shape), ctypes.c_int(self.num_neighbors), c_str(vecchia_ordering),
            ctypes.c_int(-1 if num_ind_points is None else int(num_ind_points)), ctypes.c_double(cover_tree_radius),
            c_str(ind_points_selection), c_str(likelihood), ctypes.c_double(lap), c_str(matrix_inversion_method),
            ctypes.c_int(self.seed), ctypes.c_int(-1 if num_parallel_threads is None else int(num_parallel_threads)),
            ctypes.c_bool(bool(GPU_use)), ctypes.c_bool(weights is not None), weights_c,
            ctypes.c_double(likelihood_learning_rate), ctypes.byref(self.handle)))
        if self.num_neighbors <= 0 and gp_approx != "none":      # the default the library resolved
            m = ctypes.c_int(0)
            if _lib().GPB_HIP_GetVecchiaStructure(self.handle, None, None, ctypes.byref(m)) == 0:
                self.num_neighbors = int(m.value)
            else:      # several clusters: one table per cluster, the defaults are the library's (re_model_template.h:287-298)
                self.num_neighbors = 30 if gp_approx in ("full_scale_vecchia", "vif", "VIF") else 20

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value is not None:
                _safe_call(_lib().GPB_REModelFree(self.handle))
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    # --- reference surface -------------------------------------------------------------------
    def neg_log_likelihood(self, cov_pars=None, y=None, fixed_effects=None, aux_pars=None):
        """Evaluate the negative log-likelihood (reference: basic.py:5640-5700); aux_pars: the likelihood's auxiliary parameters ("gamma" /
        "negative_binomial": the shape), set as the reference's package does it -- set_optim_params({"init_aux_pars": aux_pars}) (basic.py:5688-5690)."""
        if aux_pars is not None:
            self.set_optim_params({"init_aux_pars": aux_pars})
        y_c = ctypes.c_void_p()        # None -> NULL: the response already resident on the device is used (C-level semantics of
        if y is not None:              # GPB_EvalNegLogLikelihood, re_model_template.h:2905-2921: SetY only for a non-NULL y_data)
            y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
            if y.shape[0] != self.num_data:
                raise ValueError("Incorrect number of data points in 'y'")
            y_c = _dptr(y)
        cp_c = ctypes.c_void_p()       # None: the stored (initial or estimated) parameters, as in the reference
        if cov_pars is not None:
            cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
            if cov_pars.shape[0] != self.num_cov_pars:
                raise ValueError("'cov_pars' does not contain the correct number of parameters")
            cp_c = _dptr(cov_pars)
        fe_c = ctypes.c_void_p()
        if fixed_effects is not None:
            fixed_effects = np.ascontiguousarray(fixed_effects, dtype=np.float64).reshape(-1)
            if fixed_effects.shape[0] != self.num_data:
                raise ValueError("Length of 'fixed_effects' is not correct ")
            fe_c = _dptr(fixed_effects)
        negll = ctypes.c_double(0)
        _safe_call(_lib().GPB_EvalNegLogLikelihood(self.handle, y_c, cp_c, fe_c, ctypes.byref(negll)))
        return negll.value

    _OPTIM_DEFAULTS = {   # basic.py:4528-4570 (self.params) -> GPB_SetOptimConfig; -999 / "" / "default" = the library's default
        "init_cov_pars": None, "lr_cov": -999., "acc_rate_cov": -999., "maxit": -999, "delta_rel_conv": -999.,
        "use_nesterov_acc": True, "nesterov_schedule_version": -999, "trace": False, "optimizer_cov": "", "momentum_offset": -999,
        "convergence_criterion": "default", "m_lbfgs": -999, "estimate_cov_par_index": None,
        "cg_max_num_it": -999, "cg_max_num_it_tridiag": -999, "cg_delta_conv": -999., "num_rand_vec_trace": -999,
        "seed_rand_vec_trace": 1, "delta_conv_mode_finding": -999., "cg_preconditioner_type": "",
        "fitc_piv_chol_preconditioner_rank": -999,   # rank of the "pivoted_cholesky" preconditioner (basic.py:4767, :5510-5511; -999: the library's default 50)
        # non-Gaussian models with covariates: start of the coefficients in the lbfgs vector; default as in the reference's packages: the fit of the
        # same likelihood without the Gaussian process (init_coef_aux_pars_from_iid_model, re_model.cpp:380-470)
        "init_coef": None, "init_coef_aux_pars_from_iid_model": True,
        # likelihoods with auxiliary parameters (gamma / negative_binomial: the shape): initial values and whether they are estimated (basic.py:5420-5440)
        "init_aux_pars": None, "estimate_aux_pars": True}

    def set_optim_params(self, params):
        """Optimiser and iterative-method settings (reference: GPModel.set_optim_params, basic.py:5238-5420 -> GPB_SetOptimConfig).
        Estimation: 'optimizer_cov' ("lbfgs" | "gradient_descent" | "nelder_mead"), 'init_cov_pars', 'lr_cov', 'acc_rate_cov', 'maxit',
        'delta_rel_conv', 'use_nesterov_acc', 'nesterov_schedule_version', 'momentum_offset', 'convergence_criterion', 'm_lbfgs',
        'estimate_cov_par_index' (0 = hold a covariance parameter at its initial value: (error variance, GP variance, range) for Gaussian models, (GP variance, range) for non-Gaussian ones with 'lbfgs'), 'trace', 'init_coef', 'init_coef_aux_pars_from_iid_model'; non-Gaussian likelihoods: 'cg_max_num_it', 'cg_max_num_it_tridiag', 'cg_delta_conv', 'num_rand_vec_trace',
        'seed_rand_vec_trace', 'delta_conv_mode_finding', 'cg_preconditioner_type' ("vadu" | "pivoted_cholesky" | "fitc" | "vecchia_response" -- the last one for evaluations and Nelder-Mead fits only), 'fitc_piv_chol_preconditioner_rank'; likelihoods with auxiliary parameters ("gamma", "negative_binomial"):
        'init_aux_pars', 'estimate_aux_pars'.  Anything else raises: no silent ignore."""
        if not hasattr(self, "_optim_params"):
            self._optim_params = dict(self._OPTIM_DEFAULTS)
        for k in params:
            if k not in self._optim_params:
                raise GPBoostError("set_optim_params: '%s' is not a setting of the MI355X path of this library" % k)
        o = dict(self._optim_params)
        o.update(params)
        init_c = ctypes.c_void_p()
        if o["init_cov_pars"] is not None:
            init = np.ascontiguousarray(o["init_cov_pars"], dtype=np.float64).reshape(-1)
            if init.shape[0] != self.num_cov_pars:
                raise ValueError("'init_cov_pars' does not contain the correct number of parameters")
            init_c = _dptr(init)
        est = np.array([-1], dtype=np.int32)
        if o["estimate_cov_par_index"] is not None:
            est = np.ascontiguousarray(o["estimate_cov_par_index"], dtype=np.int32).reshape(-1)
            if est.shape[0] != self.num_cov_pars or np.any(est < 0):
                raise ValueError("'estimate_cov_par_index' needs one entry (1 = estimate, 0 = hold fixed) per covariance parameter")
        ncov_c, icoef_c = 0, ctypes.c_void_p()
        if o["init_coef"] is not None:
            icoef = np.ascontiguousarray(o["init_coef"], dtype=np.float64).reshape(-1)
            ncov_c, icoef_c = icoef.shape[0], _dptr(icoef)
        iaux_c = ctypes.c_void_p()
        if o["init_aux_pars"] is not None:
            iaux = np.ascontiguousarray(np.atleast_1d(o["init_aux_pars"]), dtype=np.float64).reshape(-1)
            if iaux.shape[0] != self.get_num_aux_pars():
                raise ValueError("params['init_aux_pars'] does not contain the correct number of parameters")
            iaux_c = _dptr(iaux)
        _safe_call(_lib().GPB_SetOptimConfig(
            self.handle, init_c, ctypes.c_double(float(o["lr_cov"])), ctypes.c_double(float(o["acc_rate_cov"])), ctypes.c_int(int(o["maxit"])),
            ctypes.c_double(float(o["delta_rel_conv"])), ctypes.c_bool(bool(o["use_nesterov_acc"])),
            ctypes.c_int(int(o["nesterov_schedule_version"])), ctypes.c_bool(bool(o["trace"])), c_str(o["optimizer_cov"]),
            ctypes.c_int(int(o["momentum_offset"])), c_str(o["convergence_criterion"]), ctypes.c_int(ncov_c), icoef_c,
            ctypes.c_double(-999.), ctypes.c_double(-999.), c_str(""),
            ctypes.c_int(int(o["cg_max_num_it"])), ctypes.c_int(int(o["cg_max_num_it_tridiag"])),
            ctypes.c_double(float(o["cg_delta_conv"])), ctypes.c_int(int(o["num_rand_vec_trace"])), ctypes.c_bool(True),
            c_str(o["cg_preconditioner_type"]), ctypes.c_int(int(o["seed_rand_vec_trace"])), ctypes.c_int(int(o["fitc_piv_chol_preconditioner_rank"])),
            iaux_c, ctypes.c_bool(bool(o["estimate_aux_pars"])), ctypes.c_bool(bool(o["init_coef_aux_pars_from_iid_model"])), est.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int(int(o["m_lbfgs"])),
            ctypes.c_double(float(o["delta_conv_mode_finding"]))))
        self._optim_params = o          # only settings the library accepted are remembered
        return self

    def fit(self, y, X=None, params=None, fixed_effects=None):
        """Maximum-likelihood estimation of the covariance parameters (reference: GPModel.fit, basic.py:5422-5560 ->
        GPB_OptimCovPar).  y is uploaded once; every likelihood / gradient evaluation of the fit runs on the device."""
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if y.shape[0] != self.num_data:
            raise ValueError("Incorrect number of data points in 'y'")
        if params is not None:
            self.set_optim_params(params)
        elif X is not None and not hasattr(self, "_optim_params"):
            self.set_optim_params({})          # hand the library this package's defaults (they are the reference's)
        fe_c = ctypes.c_void_p()
        if fixed_effects is not None:
            fixed_effects = np.ascontiguousarray(fixed_effects, dtype=np.float64).reshape(-1)
            if fixed_effects.shape[0] != self.num_data:
                raise ValueError("Length of 'fixed_effects' is not correct ")
            fe_c = _dptr(fixed_effects)
        if X is not None:
            # linear regression term X beta: GPB_OptimLinRegrCoefCovPar (basic.py:5519-5540); Gaussian models: beta is profiled out by GLS on the
            # device; non-Gaussian models: beta is part of the lbfgs vector, its gradient X' grad_F with grad_F from the device
            X = np.asarray(X, dtype=np.float64)
            if X.ndim == 1:
                X = X.reshape(-1, 1)
            if X.shape[0] != self.num_data:
                raise ValueError("Incorrect number of data points in 'X'")
            Xf = np.asfortranarray(X)
            self.num_coef = X.shape[1]
            _safe_call(_lib().GPB_OptimLinRegrCoefCovPar(self.handle, _dptr(y), _dptr(Xf), ctypes.c_int(self.num_coef), fe_c))
            return self
        _safe_call(_lib().GPB_OptimCovPar(self.handle, _dptr(y), fe_c))
        return self

    def get_cg_preconditioner_type(self):
        """The preconditioner of the iterative methods as the library resolved it (GPB_GetCGPreconditionerType; reference: GPModel.get_optim_params,
        basic.py:5942-5946): "vadu", "pivoted_cholesky", "fitc" or "vecchia_response"."""
        buf = ctypes.create_string_buffer(256)
        k = ctypes.c_int(0)
        _safe_call(_lib().GPB_GetCGPreconditionerType(self.handle, buf, ctypes.byref(k)))
        return buf.value.decode()

    def get_num_aux_pars(self):
        """Number of auxiliary parameters of the likelihood (GPB_GetNumAuxPars): 1 (the shape) for "gamma" / "negative_binomial", else 0."""
        k = ctypes.c_int(0)
        _safe_call(_lib().GPB_GetNumAuxPars(self.handle, ctypes.byref(k)))
        return k.value

    def get_aux_pars(self, std_err=False):
        """Auxiliary parameters of the likelihood on the original scale (reference: GPModel.get_aux_pars, basic.py:6372-6400 -> GPB_GetAuxPars);
        None for likelihoods without any.  std_err: their standard deviations behind the values (2 k entries; NaN for a parameter that is not estimated)."""
        k = self.get_num_aux_pars()
        if k == 0:
            return None
        out = np.empty(k * (2 if std_err else 1)); name = ctypes.create_string_buffer(256)
        _safe_call(_lib().GPB_GetAuxPars(self.handle, _dptr(out), name, ctypes.c_bool(bool(std_err))))
        return out

    def get_coef(self, std_err=False):
        """Estimated linear regression coefficients (reference: GPModel.get_coef, basic.py:6332-6370 -> GPB_GetCoef)."""
        p = int(getattr(self, "num_coef", 0))
        out = np.empty(max(p, 1) * (2 if std_err else 1))
        _safe_call(_lib().GPB_GetCoef(self.handle, _dptr(out), ctypes.c_bool(bool(std_err))))
        return out[:p * (2 if std_err else 1)]

    def get_cov_pars(self, std_err=False):
        """(error variance, GP variance, range) on the original scale (reference: GPModel.get_cov_pars, basic.py:6290-6330)."""
        out = np.empty(self.num_cov_pars * (2 if std_err else 1))
        _safe_call(_lib().GPB_GetCovPar(self.handle, _dptr(out), ctypes.c_bool(bool(std_err))))
        return out

    def _get_init_cov_pars(self):
        out = np.empty(self.num_cov_pars)
        _safe_call(_lib().GPB_GetInitCovPar(self.handle, _dptr(out)))
        return out

    def get_num_optim_iter(self):
        n = ctypes.c_int(0)
        _safe_call(_lib().GPB_GetNumIt(self.handle, ctypes.byref(n)))
        return n.value

    def optim_info(self):
        """Device launches of the last fit: likelihood-only, with gradient sums; final learning rate (GPB_HIP_GetOptimInfo)."""
        a = ctypes.c_int(0); b = ctypes.c_int(0); lr = ctypes.c_double(0)
        _safe_call(_lib().GPB_HIP_GetOptimInfo(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(lr)))
        return dict(num_ll_evals=a.value, num_grad_evals=b.value, lr_cov_final=lr.value)

    def laplace_info(self):
        """Diagnostics of the last Laplace evaluation (GPB_HIP_GetLaplaceInfo)."""
        o = np.empty(9)
        _safe_call(_lib().GPB_HIP_GetLaplaceInfo(self.handle, _dptr(o)))
        return dict(mll=o[0], newton_it=int(o[1]), cg_it=int(o[2]), log_det=o[3], lanczos_it=int(o[4]), mll_no_det=o[5],
                    ms_factor=o[6], ms_mode=o[7], ms_logdet=o[8])

    def get_current_neg_log_likelihood(self):
        negll = ctypes.c_double(0)
        _safe_call(_lib().GPB_GetCurrentNegLogLikelihood(self.handle, ctypes.byref(negll)))
        return negll.value

    def _get_likelihood_name(self):
        buf = ctypes.create_string_buffer(256)
        num = ctypes.c_int(0)
        _safe_call(_lib().GPB_GetLikelihoodName(self.handle, buf, ctypes.byref(num)))
        return buf.value.decode("utf-8")

    # --- additions used by tests / bench (no reference C entry point exists for these) --------
    def neg_log_likelihood_batch(self, cov_pars_K3):
        """K likelihood evaluations at K parameter sets (K x 3) on the resident response with one synchronisation (GPB_HIP_EvalNegLogLikelihoodBatch)."""
        cp = np.ascontiguousarray(cov_pars_K3, dtype=np.float64).reshape(-1, 3)
        out = np.empty(cp.shape[0])
        _safe_call(_lib().GPB_HIP_EvalNegLogLikelihoodBatch(self.handle, ctypes.c_int32(cp.shape[0]), _dptr(cp), _dptr(out)))
        return out


    def neg_log_likelihood_and_gradient(self, cov_pars, y, fixed_effects=None):
        """(nll, gradient wrt log(sigma2), log(sigma1_2/sigma2), log(transformed range)): what
        ``CalcGradPars`` hands to the reference's optimisers (re_model_template.h:1988-2011)."""
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
        fe_c = ctypes.c_void_p()
        if fixed_effects is not None:
            fixed_effects = np.ascontiguousarray(fixed_effects, dtype=np.float64).reshape(-1)
            fe_c = _dptr(fixed_effects)
        negll = ctypes.c_double(0)
        grad = np.empty(3)
        _safe_call(_lib().GPB_HIP_EvalNegLogLikelihoodAndGrad(self.handle, _dptr(y), _dptr(cov_pars), fe_c,
                                                              ctypes.byref(negll), _dptr(grad)))
        return negll.value, grad

    def y_aux(self, cov_pars, y):
        """Psi^-1 y in data order (the boosting gradient, re_model_template.h:3298-3321)."""
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
        out = np.empty(self.num_data)
        _safe_call(_lib().GPB_HIP_CalcYAux(self.handle, _dptr(y), _dptr(cov_pars), _dptr(out)))
        return out

    def newton_update_leaf_values(self, cov_pars, y, data_leaf_index, num_leaves):
        """Leaf values of the Newton step of the GPBoost algorithm (REModel::NewtonUpdateLeafValues); y = F - y, data order.
        cov_pars = y = None reuses the factor and y_aux of the preceding y_aux() call, as the reference does."""
        leaf = np.ascontiguousarray(data_leaf_index, dtype=np.int32).reshape(-1)
        if leaf.shape[0] != self.num_data:
            raise ValueError("Incorrect number of data points")
        out = np.empty(int(num_leaves))
        if y is None and cov_pars is None:
            _safe_call(_lib().GPB_HIP_NewtonUpdateLeafValues(self.handle, None, None, leaf.ctypes.data_as(ctypes.c_void_p),
                                                             ctypes.c_int(int(num_leaves)), _dptr(out)))
            return out
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
        if y.shape[0] != self.num_data:
            raise ValueError("Incorrect number of data points")
        _safe_call(_lib().GPB_HIP_NewtonUpdateLeafValues(self.handle, _dptr(y), _dptr(cov_pars),
                                                         leaf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(num_leaves)), _dptr(out)))
        return out

    def set_prediction_data(self, gp_coords_pred=None, vecchia_pred_type=None, num_neighbors_pred=None):
        """Reference: GPModel.set_prediction_data (basic.py:6052-6200) -> GPB_SetPredictionData."""
        cp_c = ctypes.c_void_p(); npred = 0
        if gp_coords_pred is not None:
            cp = np.asarray(gp_coords_pred, dtype=np.float64)
            if cp.ndim == 1:
                cp = cp.reshape(-1, 1)
            if cp.shape[1] != self.dim_coords:
                raise ValueError("Incorrect dimension of 'gp_coords_pred'")
            cpc = np.asfortranarray(cp); cp_c = _dptr(cpc); npred = cp.shape[0]
            self._num_data_pred_saved = npred
        _safe_call(_lib().GPB_SetPredictionData(
            self.handle, ctypes.c_int(npred), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), cp_c, ctypes.c_void_p(),
            ctypes.c_void_p(), ctypes.c_void_p() if vecchia_pred_type is None else c_str(vecchia_pred_type),
            ctypes.c_int(-1 if num_neighbors_pred is None else int(num_neighbors_pred)), ctypes.c_double(-1.), ctypes.c_int(-1), ctypes.c_int(-1)))
        return self

    def predict_training_data_random_effects(self, y=None, cov_pars=None, predict_var=False, fixed_effects=None):
        """Posterior mean (and variance) of the latent GP at the training locations (reference: GPModel.predict_training_data_random_effects,
        python-package/gpboost/basic.py; GPB_PredictREModelTrainingDataRandomEffects).  Returns an (n,) array, or (n, 2) with predict_var."""
        n = self.num_data
        out = np.empty(n * (2 if predict_var else 1))
        yv = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
        cp = None if cov_pars is None else np.ascontiguousarray(cov_pars, dtype=np.float64)
        fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
        _safe_call(_lib().GPB_PredictREModelTrainingDataRandomEffects(
            self.handle, ctypes.c_void_p() if cp is None else _dptr(cp), ctypes.c_void_p() if yv is None else _dptr(yv), _dptr(out),
            ctypes.c_void_p() if fe is None else _dptr(fe), ctypes.c_bool(bool(predict_var))))
        return out.reshape(2, n).T.copy() if predict_var else out

    def predict(self, y=None, gp_coords_pred=None, cov_pars=None, predict_var=False, predict_cov_mat=False, predict_response=True,
                num_neighbors_pred=None, vecchia_pred_type=None, use_saved_data=False, X_pred=None, cluster_ids_pred=None, offset=None, offset_pred=None):
        """Predictive mean / variances / covariance matrix at new locations (reference: GPModel.predict, basic.py:5702-6050 ->
        GPB_PredictREModel -> CalcPredVecchiaObservedFirstOrder); vecchia_pred_type "order_obs_first_cond_obs_only".  cov_pars=None
        uses the estimated parameters, y=None the response of the last fit / evaluation.  Returns {'mu', 'var', 'cov'}.
        Non-Gaussian likelihoods ('latent_order_obs_first_cond_obs_only'): latent mean -Bpo mode, latent variances / covariance matrix
        Dp + Bpo (Sigma^-1 + W)^-1 Bpo' (PredictLaplaceApproxVecchia, likelihoods.h:8563-8824: the exact value, which the reference's iterative
        branch estimates with random vectors) and, predict_response=True, the response mean / variance (PredictResponse, :9626-9672)."""
        if num_neighbors_pred is not None or vecchia_pred_type is not None:
            self.set_prediction_data(vecchia_pred_type=vecchia_pred_type, num_neighbors_pred=num_neighbors_pred)
        y_c = ctypes.c_void_p()
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
            if y.shape[0] != self.num_data:
                raise ValueError("Incorrect number of data points in 'y'")
            y_c = _dptr(y)
        cp_c = ctypes.c_void_p()
        if cov_pars is not None:
            cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
            if cov_pars.shape[0] != self.num_cov_pars:
                raise ValueError("'cov_pars' does not contain the correct number of parameters")
            cp_c = _dptr(cov_pars)
        crd_c = ctypes.c_void_p(); npred = 0
        if not use_saved_data:
            if gp_coords_pred is None:
                raise ValueError("'gp_coords_pred' is required")
            cp = np.asarray(gp_coords_pred, dtype=np.float64)
            if cp.ndim == 1:
                cp = cp.reshape(-1, 1)
            if cp.shape[1] != self.dim_coords:
                raise ValueError("Incorrect dimension of 'gp_coords_pred'")
            cpc = np.asfortranarray(cp); crd_c = _dptr(cpc); npred = cp.shape[0]
        else:
            npred = int(getattr(self, "_num_data_pred_saved", 0))
        xp_c = ctypes.c_void_p()
        if X_pred is not None:
            X_pred = np.asarray(X_pred, dtype=np.float64)
            if X_pred.ndim == 1:
                X_pred = X_pred.reshape(-1, 1)
            if X_pred.shape[0] != npred or X_pred.shape[1] != int(getattr(self, "num_coef", 0)):
                raise ValueError("Incorrect dimensions of 'X_pred'")
            Xpf = np.asfortranarray(X_pred); xp_c = _dptr(Xpf)
        cid_c = ctypes.c_void_p()
        if cluster_ids_pred is not None:       # independent realisations of the GP (reference: GPModel.predict(cluster_ids_pred=...), basic.py:5702-6050)
            cidp = np.ascontiguousarray(cluster_ids_pred, dtype=np.int32).reshape(-1)
            if cidp.shape[0] != npred:
                raise ValueError("Incorrect number of data points in 'cluster_ids_pred'")
            cid_c = cidp.ctypes.data_as(ctypes.c_void_p)
        # offset / offset_pred: fixed effects of the observed data (the location parameter the mode is found at) and of the prediction points (added to the mean)
        fe_c = ctypes.c_void_p(); fep_c = ctypes.c_void_p()
        if offset is not None:
            offset = np.ascontiguousarray(offset, dtype=np.float64).reshape(-1)
            if offset.shape[0] != self.num_data:
                raise ValueError("Incorrect number of data points in 'offset'")
            fe_c = _dptr(offset)
        if offset_pred is not None:
            offset_pred = np.ascontiguousarray(offset_pred, dtype=np.float64).reshape(-1)
            if offset_pred.shape[0] != npred:
                raise ValueError("Incorrect number of data points in 'offset_pred'")
            fep_c = _dptr(offset_pred)
        n_out = npred * (1 + npred) if predict_cov_mat else (2 * npred if predict_var else npred)
        out = np.empty(max(n_out, 1))
        _safe_call(_lib().GPB_PredictREModel(
            self.handle, y_c, ctypes.c_int(npred), _dptr(out), ctypes.c_bool(bool(predict_cov_mat)), ctypes.c_bool(bool(predict_var)),
            ctypes.c_bool(bool(predict_response)), ctypes.c_bool(False), ctypes.c_bool(False), ctypes.c_int(0), ctypes.c_int(0),
            cid_c, ctypes.c_void_p(), ctypes.c_void_p(), crd_c, ctypes.c_void_p(), cp_c, xp_c,
            ctypes.c_bool(bool(use_saved_data)), fe_c, fep_c))
        res = {"mu": out[:npred].copy(), "var": None, "cov": None}
        if predict_var:
            res["var"] = out[npred:2 * npred].copy()
        if predict_cov_mat:
            res["cov"] = out[npred:].reshape(npred, npred, order="F").copy()
        return res

    def vecchia_structure(self):
        """(perm, nn): Vecchia ordering and the (n, m) neighbour table, -1 padded."""
        m = ctypes.c_int(0)
        perm = np.empty(self.num_data, dtype=np.int32)
        _safe_call(_lib().GPB_HIP_GetVecchiaStructure(self.handle, perm.ctypes.data_as(ctypes.c_void_p), None,
                                                      ctypes.byref(m)))
        nn = np.empty((self.num_data, max(m.value, 1)), dtype=np.int32)
        _safe_call(_lib().GPB_HIP_GetVecchiaStructure(self.handle, None, nn.ctypes.data_as(ctypes.c_void_p), None))
        return perm, nn

    def vif_grad_factor(self, cov_pars, p):
        """Full-scale Vecchia (VIF) models, test seam: (dA, dD) of parameter p (0: variance, 1: range; log of the transformed parameter) of the
        residual-process factor on the resident response -- the reference's -B_grad / D_grad (src/GPBoost/Vecchia_utils.cpp:1640-1656)."""
        cov_pars = np.ascontiguousarray(cov_pars, dtype=np.float64).reshape(-1)
        m = ctypes.c_int(0)
        _safe_call(_lib().GPB_HIP_GetVecchiaStructure(self.handle, None, None, ctypes.byref(m)))
        dA = np.empty((self.num_data, max(m.value, 1))); dD = np.empty(self.num_data)
        _safe_call(_lib().GPB_HIP_VifGradFactor(self.handle, _dptr(cov_pars), ctypes.c_int(int(p)), _dptr(dA), _dptr(dD)))
        return dA, dD

    def vecchia_handle(self):
        return ctypes.c_void_p(_lib().GPB_HIP_GetVecchiaHandle(self.handle))
