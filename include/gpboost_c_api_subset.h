/*
 * include/gpboost_c_api_subset.h -- the reference C API entry points that sit ON the hot path,
 * re-exported by libgpboost_amd.so with byte-identical signatures, so that a ctypes/.Call client
 * written against the reference (python-package/gpboost/basic.py, R-package/src/gpboost_R.cpp)
 * can evaluate the Gaussian Vecchia likelihood on an MI355X without changing its binding.
 *
 * Source of every declaration: include/LightGBM/c_api.h of fabsig/GPBoost v1.7.3 (line cited).
 * Everything else of that header (optimiser, prediction, Booster, Dataset, Network: 100 further
 * functions) is host orchestration that stays in the reference's own C++ and is out of scope
 * here; INTEGRATION.md shows how the reference links this library instead.
 *
 * Supported model slice (anything else returns -1 with a message, never a silent fallback; DESIGN.md section 8 has the full list):
 *   one GP, no grouped effects / random coefficients; cov_fct "exponential" or "matern" with shape 0.5 / 1.5 / 2.5;
 *   likelihood "gaussian" with
 *     gp_approx "vecchia" (num_neighbors <= 126, vecchia_ordering "none" | "random", d <= 10, any number of clusters = independent realisations through
 *       cluster_ids_data, sample weights): likelihood, gradient, y_aux, parameter estimation (GPB_OptimCovPar / GPB_OptimLinRegrCoefCovPar with "lbfgs",
 *       "gradient_descent", "nelder_mead"; with covariates "lbfgs" and "gradient_descent", coefficients by generalised least squares), standard errors, all five vecchia_pred_type values of GPB_PredictREModel, training-data random effects;
 *     gp_approx "none" (exact GP, dense MFMA Cholesky): the same calls;
 *     gp_approx "full_scale_vecchia" (<= 256 inducing points by kmeans++, d <= 3): likelihood, fits, prediction "order_obs_first_cond_obs_only";
 *   likelihood "bernoulli_logit", "bernoulli_probit" (aliases "binary", "binary_logit", "binary_probit") or "poisson" with gp_approx "vecchia" and
 *     matrix_inversion_method "default" | "iterative" (Vecchia-Laplace approximation, "vadu"-preconditioned CG + stochastic Lanczos quadrature: the
 *     reference's defaults for that model), cov_pars = (sigma1_2, rho), repeated locations allowed (the reference's unique-location mapping):
 *     likelihood, its gradient, fits (GPB_OptimCovPar; GPB_OptimLinRegrCoefCovPar with the coefficients in the lbfgs vector, initial coefficients
 *     given, from the data, or from the model without the GP), standard errors, fixed effects / offset, training-data random effects, and
 *     GPB_PredictREModel "latent_order_obs_first_cond_obs_only" / "latent_order_obs_first_cond_all" -- latent mean, variances, covariance matrix,
 *     and the response mean / variance.
 */
#ifndef GPBOOST_C_API_SUBSET_H_
#define GPBOOST_C_API_SUBSET_H_

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPBOOST_C_EXPORT __attribute__((visibility("default")))

typedef void* REModelHandle; /* c_api.h:32 */

/* c_api.h:54 -- thread-local message of the last failing call */
GPBOOST_C_EXPORT const char* LGBM_GetLastError();

/* c_api.h:1359-1391 -- 32 positional inputs + out handle; coords column-major fp64 */
GPBOOST_C_EXPORT int GPB_CreateREModel(int32_t num_data,
    const int32_t* cluster_ids_data,
    const char* re_group_data,
    int32_t num_re_group,
    const double* re_group_rand_coef_data,
    const int32_t* ind_effect_group_rand_coef,
    int32_t num_re_group_rand_coef,
    const int* drop_intercept_group_rand_effect,
    int32_t num_gp,
    const double* gp_coords_data,
    const int dim_gp_coords,
    const double* gp_rand_coef_data,
    int32_t num_gp_rand_coef,
    const char* cov_fct,
    double cov_fct_shape,
    const char* gp_approx,
    double cov_fct_taper_range,
    double cov_fct_taper_shape,
    int num_neighbors,
    const char* vecchia_ordering,
    int num_ind_points,
    double cover_tree_radius,
    const char* ind_points_selection,
    const char* likelihood,
    double likelihood_additional_param,
    const char* matrix_inversion_method,
    int seed,
    int num_parallel_threads,
    bool GPU_use,
    bool has_weights,
    const double* weights,
    double likelihood_learning_rate,
    REModelHandle* out);

/* c_api.h:1398 */
GPBOOST_C_EXPORT int GPB_REModelFree(REModelHandle handle);

/* c_api.h:1437-1467 -- used: init_cov_pars, lr, acc_rate_cov, max_iter, delta_rel_conv, use_nesterov_acc,
 * nesterov_schedule_version, trace, optimizer ("lbfgs" = default | "gradient_descent" | "nelder_mead"), momentum_offset, convergence_criterion,
 * m_lbfgs (GPB_OptimCovPar below); for the Laplace path cg_max_num_it, cg_max_num_it_tridiag, cg_delta_conv, num_rand_vec_trace,
 * seed_rand_vec_trace, delta_conv_mode_finding and cg_preconditioner_type (only "vadu").  -999 / "" / "default" keep the
 * reference's defaults.  num_covariates > 0, estimate_aux_pars and estimate_cov_par_index[0] >= 0 return -1. */
GPBOOST_C_EXPORT int GPB_SetOptimConfig(REModelHandle handle,
    double* init_cov_pars,
    double lr,
    double acc_rate_cov,
    int max_iter,
    double delta_rel_conv,
    bool use_nesterov_acc,
    int nesterov_schedule_version,
    bool trace,
    const char* optimizer,
    int momentum_offset,
    const char* convergence_criterion,
    int num_covariates,
    double* init_coef,
    double lr_coef,
    double acc_rate_coef,
    const char* optimizer_coef,
    int cg_max_num_it,
    int cg_max_num_it_tridiag,
    double cg_delta_conv,
    int num_rand_vec_trace,
    bool reuse_rand_vec_trace,
    const char* cg_preconditioner_type,
    int seed_rand_vec_trace,
    int piv_chol_rank,
    double* init_aux_pars,
    bool estimate_aux_pars,
    bool init_coef_aux_pars_from_iid_model,
    const int* estimate_cov_par_index,
    int m_lbfgs,
    double delta_conv_mode_finding);

/* c_api.h:1505-1509 -- THE metric's unit of work: cov_pars on the original scale
 * (sigma2, sigma1_2, rho); y in data order; fixed_effects optional (subtracted from y).
 * likelihood "bernoulli_logit": cov_pars = (sigma1_2, rho), y in {0,1}, the posterior mode restarts at 0 on every call
 * (re_model_template.h:3191-3212) and the value is minus the Laplace-approximated marginal log-likelihood. */
GPBOOST_C_EXPORT int GPB_EvalNegLogLikelihood(REModelHandle handle,
    const double* y_data,
    double* cov_pars,
    const double* fixed_effects,
    double* negll);

/* c_api.h:1517 */
GPBOOST_C_EXPORT int GPB_GetCurrentNegLogLikelihood(REModelHandle handle, double* negll);

/* c_api.h:1476-1478 -- maximum-likelihood estimation of (sigma2, sigma1_2, rho): the direct caller of the hot path
 * (REModel::OptimCovPar, re_model.cpp:483-546 -> REModelTemplate::OptimLinRegrCoefCovPar, re_model_template.h:972-1802).
 * Gaussian likelihood + gp_approx "vecchia"; optimizer_cov "lbfgs" (default; LBFGSpp with the backtracking Armijo line search,
 * nugget profiled out) or "gradient_descent" (Nesterov acceleration, Armijo step halving, nugget profiled out) or "nelder_mead" (the reference's OptimLib
 * simplex search: likelihood evaluations only).  y is uploaded
 * ONCE; every likelihood / gradient evaluation of the fit returns 3 or 7 doubles from the device.  Without init_cov_pars the
 * initial values are the reference's (FindInitCovPar, re_model_template.h:4849-4968, cov_fcts.h:1422-1683). */
GPBOOST_C_EXPORT int GPB_OptimCovPar(REModelHandle handle, const double* y_data, const double* fixed_effects);
/* c_api.h:1534-1536 -- (sigma2, sigma1_2, rho) on the original scale; calc_std_dev = true (Fisher information) returns -1 */
GPBOOST_C_EXPORT int GPB_GetCovPar(REModelHandle handle, double* optim_cov_pars, bool calc_std_dev);
/* c_api.h:1545-1546 -- -1 in every entry while no initial values exist */
GPBOOST_C_EXPORT int GPB_GetInitCovPar(REModelHandle handle, double* init_cov_pars);
/* c_api.h:1567-1568 */
GPBOOST_C_EXPORT int GPB_GetNumIt(REModelHandle handle, int* num_it);

/* c_api.h:1588-1610 -- kept: gp_coords_data_pred (column-major), vecchia_pred_type, num_neighbors_pred (default 2 x num_neighbors,
 * re_model_template.h:299); cluster ids / grouped effects / random coefficients / covariates return -1 */
GPBOOST_C_EXPORT int GPB_SetPredictionData(REModelHandle handle,
    int32_t num_data_pred,
    const int32_t* cluster_ids_data_pred,
    const char* re_group_data_pred,
    const double* re_group_rand_coef_data_pred,
    double* gp_coords_data_pred,
    const double* gp_rand_coef_data_pred,
    const double* covariate_data_pred,
    const char* vecchia_pred_type,
    int num_neighbors_pred,
    double cg_delta_conv_pred,
    int nsim_var_pred,
    int rank_pred_approx_matrix_lanczos);

/* c_api.h:1640-1660 -- out_predict: num_data_pred means, then the variances (predict_var) or the num_data_pred^2 covariance matrix
 * (predict_cov_mat).  Gaussian Vecchia model: the five vecchia_pred_type values of the reference ("order_obs_first_cond_obs_only" is its default for
 * a Gaussian likelihood); exact GP; full-scale Vecchia ("order_obs_first_cond_obs_only"); non-Gaussian Vecchia models: the latent process
 * ("latent_order_obs_first_cond_obs_only" / "latent_order_obs_first_cond_all": mean, variances, covariance matrix) and, predict_response, the response mean / variance
 * (PredictLaplaceApproxVecchia + PredictResponse, likelihoods.h:8563-8824, :9626-9672).  cov_pars NULL = the estimated / stored parameters, y_data
 * NULL = the response of the last call; posterior / prior samples return -1. */
GPBOOST_C_EXPORT int GPB_PredictREModel(REModelHandle handle,
    const double* y_data,
    int32_t num_data_pred,
    double* out_predict,
    bool predict_cov_mat,
    bool predict_var,
    bool predict_response,
    bool sample_posterior,
    bool sample_prior,
    int num_post_samples,
    int num_prior_samples,
    const int32_t* cluster_ids_data_pred,
    const char* re_group_data_pred,
    const double* re_group_rand_coef_data_pred,
    double* gp_coords_data_pred,
    const double* gp_rand_coef_data_pred,
    const double* cov_pars,
    const double* covariate_data_pred,
    bool use_saved_data,
    const double* fixed_effects,
    const double* fixed_effects_pred);

/* c_api.h:1686-1688 */
GPBOOST_C_EXPORT int GPB_GetLikelihoodName(REModelHandle handle, char* out_str, int* num_char);

/* ---- the rest of the reference's GPB_* surface (all 32 functions of c_api.h:1359-1824 are exported) and the log hook its Python
 *      package registers at import (python-package/gpboost/basic.py:117-129).  Getters / setters answer from the model's state;
 *      what is off the hot path (auxiliary parameters of likelihoods that have none here, grouped effects) returns -1 with a message. ---- */
typedef void* BoosterHandle; /* c_api.h:31; the reference declares four REModel getters with this handle type */

/* c_api.h:61 */
GPBOOST_C_EXPORT int LGBM_RegisterLogCallback(void (*callback)(const char*));
/* c_api.h:1490-1494 -- num_covariates = 0 / covariate_data = NULL: GPB_OptimCovPar; covariates: -1 */
GPBOOST_C_EXPORT int GPB_OptimLinRegrCoefCovPar(REModelHandle handle,
    const double* y_data,
    const double* covariate_data,
    int num_covariates,
    const double* fixed_effects);
/* c_api.h:1520-1521, 1526-1527 */
GPBOOST_C_EXPORT int GPB_CanCalculateStandardErrorsCovPars(REModelHandle handle,
    int* out);
GPBOOST_C_EXPORT int GPB_CanCalculateStandardErrorsAuxPars(REModelHandle handle,
    int* out);
/* c_api.h:1556-1558 */
GPBOOST_C_EXPORT int GPB_GetCoef(REModelHandle handle,
    double* optim_coef,
    bool calc_std_dev);
/* c_api.h:1579 */
GPBOOST_C_EXPORT int GPB_HasStdCylBesselK(int* has_bessel);
/* c_api.h:1672-1677 -- posterior mean of the latent GP at the training locations (Gaussian Vecchia model): (y - F) - Psi^-1 (y - F) */
GPBOOST_C_EXPORT int GPB_PredictREModelTrainingDataRandomEffects(REModelHandle handle,
    const double* cov_pars_pred,
    const double* y_obs,
    double* out_predict,
    const double* fixed_effects,
    bool calc_var);
/* c_api.h:1697-1699, 1708-1710, 1719-1721 */
GPBOOST_C_EXPORT int GPB_GetOptimizerCovPars(REModelHandle handle,
    char* out_str,
    int* num_char);
GPBOOST_C_EXPORT int GPB_GetOptimizerCoef(REModelHandle handle,
    char* out_str,
    int* num_char);
GPBOOST_C_EXPORT int GPB_GetCGPreconditionerType(REModelHandle handle,
    char* out_str,
    int* num_char);
/* c_api.h:1729-1730, 1738-1739, 1747-1748 */
GPBOOST_C_EXPORT int GPB_GetNumCGSteps(BoosterHandle handle,
    int* num_cg_steps);
GPBOOST_C_EXPORT int GPB_GetNumCGStepsTridiag(BoosterHandle handle,
    int* num_cg_steps);
GPBOOST_C_EXPORT int GPB_GetNumModeFindingSteps(BoosterHandle handle,
    int* num_cg_steps);
/* c_api.h:1756-1757 */
GPBOOST_C_EXPORT int GPB_SetLikelihood(REModelHandle handle,
    const char* likelihood);
/* c_api.h:1765-1766, 1774-1775, 1783-1784, 1792-1793 */
GPBOOST_C_EXPORT int GPB_GetResponseData(REModelHandle handle,
    double* response_data);
GPBOOST_C_EXPORT int GPB_GetCovariateData(REModelHandle handle,
    double* covariate_data);
GPBOOST_C_EXPORT int GPB_GetOffsetData(REModelHandle handle,
    double* fixed_effects);
GPBOOST_C_EXPORT int GPB_SetOffsetData(REModelHandle handle,
    const double* fixed_effects);
/* c_api.h:1804-1807, 1815-1816, 1824-1825 */
GPBOOST_C_EXPORT int GPB_GetAuxPars(REModelHandle handle,
    double* aux_pars,
    char* out_str,
    bool calc_std_dev);
GPBOOST_C_EXPORT int GPB_GetNumAuxPars(BoosterHandle handle,
    int* num_aux_pars);
GPBOOST_C_EXPORT int GPB_GetInitAuxPars(REModelHandle handle,
    double* aux_pars);

/* ---- additions (not in the reference ABI; used by the host mirror, tests and bench) ---- */

/* K likelihood evaluations at K parameter sets (row-major K x 3: sigma2, sigma1_2, rho) on the resident response: one synchronisation and
 * -- on a sharded handle -- ONE RCCL all-reduce for the whole batch (gpb_hip_vecchia_nll_terms_batch).  Not in the reference ABI. */
GPBOOST_C_EXPORT int GPB_HIP_EvalNegLogLikelihoodBatch(REModelHandle handle, int32_t K, const double* cov_pars_K3, double* negll_K);

/* Gradient of the nll wrt log(sigma2), log(sigma1_2/sigma2), log(a) -- the vector CalcGradPars hands
 * to the reference's optimisers (include/GPBoost/re_model_template.h:1988-2011,
 * include/GPBoost/optim_utils.h:322-338); the reference has no C entry point for it. */
GPBOOST_C_EXPORT int GPB_HIP_EvalNegLogLikelihoodAndGrad(REModelHandle handle, const double* y_data,
    double* cov_pars, const double* fixed_effects, double* negll, double* grad3);
/* y_aux = Psi^-1 y in data order at cov_pars (CalcGradientF / GetYAux, re_model_template.h:3298-3321,6430) */
GPBOOST_C_EXPORT int GPB_HIP_CalcYAux(REModelHandle handle, const double* y_data, double* cov_pars, double* y_aux);
/* Newton update of the leaf values (REModel::NewtonUpdateLeafValues, re_model.cpp:1298-1310 ->
 * re_model_template.h:4982-5063; the reference calls it from the objective, it has no C entry point of its own):
 * y_data = F - y in data order (what the objective hands to CalcGradientF), data_leaf_index = leaf of every data point;
 * factor + y_aux + H^T Psi^-1 H + solve in one call.  y_data = cov_pars = NULL: the reference's own contract (:4989,
 * CHECK(y_aux_has_been_calculated_)) -- reuse the factor and y_aux that the preceding GPB_HIP_CalcYAux left on the device. */
GPBOOST_C_EXPORT int GPB_HIP_NewtonUpdateLeafValues(REModelHandle handle, const double* y_data, double* cov_pars,
    const int32_t* data_leaf_index, int32_t num_leaves, double* leaf_values);
/* Predictive mean and variance at new locations, vecchia_pred_type "order_obs_first_cond_obs_only" (the slice of
 * GPB_PredictREModel, c_api.h:1640-1668, that runs CalcPredVecchiaObservedFirstOrder(CondObsOnly = true)): gp_coords_data_pred
 * column-major num_data_pred x d; out_var (may be NULL) includes the error variance iff predict_response. */
GPBOOST_C_EXPORT int GPB_HIP_PredictVecchiaObsOnly(REModelHandle handle, const double* y_data, double* cov_pars, int32_t num_data_pred,
    const double* gp_coords_data_pred, int32_t num_neighbors_pred, bool predict_response, double* out_mean, double* out_var);
/* Vecchia ordering (perm[k] = data index of the k-th point) and neighbour table (n x m, -1 padded) */
GPBOOST_C_EXPORT int GPB_HIP_GetVecchiaStructure(REModelHandle handle, int32_t* perm, int32_t* nn, int32_t* m_out);
/* Diagnostics of the last Laplace evaluation (likelihood != "gaussian"): the nine values documented at
 * gpb_hip_vecchia_laplace_logit (include/gpb_hip.h) -- iteration counts, log-determinant, phase times */
GPBOOST_C_EXPORT int GPB_HIP_GetLaplaceInfo(REModelHandle handle, double* out9);
/* Launch counts of the last GPB_OptimCovPar (likelihood-only launches, launches with gradient sums) and the final learning rate */
GPBOOST_C_EXPORT int GPB_HIP_GetOptimInfo(REModelHandle handle, int* num_ll_evals, int* num_grad_evals, double* lr_cov_final);
/* Test seam: the initial values GPB_OptimCovPar uses when none are given (FindInitCovPar, re_model_template.h:4849-4968 ->
 * cov_fcts.h:1422-1683), computed from host data alone: coords0_colmajor = the first cluster's coordinates in Vecchia order (n0 x dim);
 * the generator starts at `seed` and is advanced by one std::shuffle of shuffle_len elements when shuffle_len > 0 (the ordering step of a
 * one-cluster model with vecchia_ordering = "random").  theta3 = (sigma2, sigma1_2 / sigma2, a), transformed scale. */
GPBOOST_C_EXPORT int GPB_HIP_FindInitCovParHost(int32_t num_data, const double* y_data, const double* fixed_effects, int32_t n0, int32_t dim,
    const double* coords0_colmajor, int cov_type, int seed, int32_t shuffle_len, double* theta3);
/* Test seam: derivative factors dA (n x m) / dD (n) of the residual process of a full-scale Vecchia (VIF) model at cov_pars, parameter p (0 variance, 1 range) */
GPBOOST_C_EXPORT int GPB_HIP_VifGradFactor(REModelHandle handle, double* cov_pars, int p, double* dA, double* dD);

/* Test seam: host half of the full-scale Vecchia (VIF) likelihood + analytic gradient (gpb_c_api.cpp: vif_terms_core) with the two device passes
 * (gpb_hip_vecchia_vif_factor, gpb_hip_vecchia_vif_grad_sums) supplied by the caller.  t7 = {quad, logdet, bad, g1_var, g2_var, g1_range, g2_range}. */
GPBOOST_C_EXPORT int GPB_HIP_VifTermsWithCallback(int32_t k, int32_t d, const double* ip_colmajor, int cov_type, double ratio, double a, int with_grad,
                                                  int (*factor)(void*, const double*, int, double*, double*),
                                                  int (*gsums)(void*, const double*, const double*, const double*, const double*, const double*, double*),
                                                  void* ctx, double* t7);

/* Test seam: the host optimiser of GPB_OptimCovPar with a caller-supplied evaluation callback
 * terms(ctx, sigma1_2 / sigma2, a, with_grad, t7) -> 0 | -1 that fills the seven shard sums of gpb_hip_vecchia_grad_terms
 * (t7[0..1] only when with_grad == 0).  init_theta / theta_out = (sigma2, sigma1_2 / sigma2, a): the reference's transformed
 * scale (re_model.cpp:301-318).  range_const = sqrt(2 nu).  num_evals2 = {likelihood-only calls, calls with gradient}. */
GPBOOST_C_EXPORT int GPB_HIP_OptimizeGaussianWithCallback(int32_t num_data, const double* init_theta, const char* optimizer,
    double lr_cov, double acc_rate_cov, int max_iter, double delta_rel_conv, bool use_nesterov_acc, int nesterov_schedule_version,
    int momentum_offset, const char* convergence_criterion, int m_lbfgs, double range_const,
    int (*terms)(void*, double, double, int, double*), void* ctx, double* theta_out, int* num_it, double* negll, int* num_evals2,
    const int* estimate_cov_par_index /* NULL or [0] < 0: all three estimated (c_api.h:1437-1467) */);
/* Host half of the Vecchia prediction 'order_obs_first_cond_all' (CalcPredVecchiaObservedFirstOrder, CondObsOnly = false,
 * src/GPBoost/Vecchia_utils.cpp:2061-2090) and its test seam: factor rows of the appended prediction points in, mean = Bp^-1 (-Bpo y) and
 * sigma2 Bp^-1 Dp Bp^-T out (var_out / cov_out may be NULL).  GPB_PredictREModel uses it for 'order_obs_first_cond_all'. */
GPBOOST_C_EXPORT int GPB_HIP_PredictCondAllHost(int32_t n_obs, int32_t n_pred, int32_t m, const int32_t* nn_pred, const double* A_pred,
    const double* D_pred, const double* y_obs, double sigma2, bool predict_response, double* mean_out, double* var_out, double* cov_out);
/* Test seam for the host half of the unique-location mapping of one non-Gaussian GP (DetermineUniqueDuplicateCoordsFast, src/GPBoost/GP_utils.cpp:472-548,
 * as RECompGP applies it with use_Z_for_duplicates, include/GPBoost/re_comp.h:863-885): positions of the first appearances (ascending, *num_unique of
 * them) and, per point, the index of its location among them.  No device needed. */
GPBOOST_C_EXPORT int GPB_HIP_UniqueLocationsHost(int32_t n, int32_t d, const double* coords_colmajor, int32_t* num_unique, int32_t* uniques_out,
    int32_t* unique_idx_out);
/* Host half of the response-scale predictions of the non-Gaussian likelihoods and its test seams (no device needed): Likelihood::PredictResponse
 * (include/GPBoost/likelihoods.h:9626-9672) in place on the latent predictive (mean, var) -- probit: Phi(m / sqrt(1 + v)); logit: adaptive Gauss-Hermite
 * quadrature around the integrand's mode (RespMeanAdaptiveGHQuadrature, :10128-10160; delta_conv_mode_finding <= 0: the default 1e-8); Poisson:
 * exp(m + v / 2) -- var is written only if predict_var.  GPB_HIP_GaussHermiteHost: the rule's nodes and ADAPTIVE weights w_j exp(x_j^2) (the
 * reference tabulates them for order 30, :17472-17576; here they are computed). */
GPBOOST_C_EXPORT int GPB_HIP_PredictResponseHost(const char* likelihood, int32_t n, double* mean_inout, double* var_inout, bool predict_var,
    double delta_conv_mode_finding);
GPBOOST_C_EXPORT int GPB_HIP_GaussHermiteHost(int32_t order, double* nodes_out, double* adaptive_weights_out);
/* Test seam and host half of parameter estimation for non-Gaussian likelihoods (GPB_OptimCovPar drives it with the device gradient of
 * the Laplace approximation): the reference's lbfgs / gradient descent on theta = (sigma1_2, a) with a stateful evaluation
 * callback eval(ctx, op, sigma1_2, a, out3): op 0 / 1 = find the mode (warm start) and return the negative approximate marginal
 * log-likelihood (op 1: + its gradient wrt (log sigma1_2, log a) in out3[1..2]); op 2 = gradient of the current state only; op 3 =
 * reset the mode to its previous value (Likelihood::ResetModeToPreviousValue); + 16 = first gradient-descent update (CG caps / 3). */
GPBOOST_C_EXPORT int GPB_HIP_OptimizeLaplaceWithCallback(const double* init_theta2, const char* optimizer, double lr_cov,
    double acc_rate_cov, int max_iter, double delta_rel_conv, bool use_nesterov_acc, int nesterov_schedule_version, int momentum_offset,
    const char* convergence_criterion, int m_lbfgs, int (*eval)(void*, int, double, double, double*), void* ctx, double* theta_out2,
    int* num_it, double* negll, int* num_evals);
/* Test seam and host half of GPB_OptimCovPar for likelihoods with auxiliary parameters estimated jointly with the covariance parameters (round 5: gamma,
 * negative_binomial -- the shape): lbfgs on (log sigma1_2, log a, log aux_1 .. log aux_naux) as EvalLLforLBFGSpp lays the vector out with EstimateAuxPars()
 * (include/GPBoost/optim_utils.h:256-283, 345-348, 498-535).  eval(ctx, op, sigma1_2, a, aux, naux, out): op 0 / 1 = find the mode (warm start) at the
 * given parameters, out[0] = negative approximate marginal log-likelihood (op 1: + gradient wrt (log sigma1_2, log a) in out[1..2] and wrt log aux in
 * out[3 ..]); op 2 = gradient of the current state; op 3 = reset the mode to its previous value; op 4 = forget the mode. */
GPBOOST_C_EXPORT int GPB_HIP_OptimizeLaplaceAuxWithCallback(const double* init_theta2, const double* init_aux, int naux, double lr_cov, int max_iter,
    double delta_rel_conv, int m_lbfgs, int (*eval)(void*, int, double, double, const double*, int, double*), void* ctx, double* theta_out2,
    double* aux_out, int* num_it, double* negll, int* num_evals);
/* Host half of the same fit: Likelihood::FindInitialAuxPars (include/GPBoost/likelihoods.h:1851-1947) -- the shape's start value when none is given:
 * gamma: approximate MLE ignoring the effects; negative_binomial: method of moments (fixed_effects may be NULL; data order). */
GPBOOST_C_EXPORT int GPB_HIP_FindInitialAuxParsHost(const char* likelihood, int32_t n, const double* y, const double* fixed_effects, double* aux_out);
/* Test seam and host half of GPB_GetCovPar(calc_std_dev = true) for non-Gaussian models: CalcStdDevCovParAuxParsNonGaussian
 * (include/GPBoost/re_model_template.h:11029-11117) -- Hessian of the negative approximate marginal log-likelihood as the numerical Jacobian (central
 * differences on the log scale, step 1e-4 max(|log theta_i|, 1)) of its analytic gradient, Cholesky inverse, delta method -- on theta = (sigma1_2, a)
 * with the stateful evaluation callback of GPB_HIP_OptimizeLaplaceWithCallback (ops 1 and 0).  se_out2: standard errors of (sigma1_2, rho), NaN if the
 * Hessian is not positive definite. */
GPBOOST_C_EXPORT int GPB_HIP_LaplaceStdErrorsWithCallback(const double* theta2, double range_const, int (*eval)(void*, int, double, double, double*),
    void* ctx, double* se_out2);
/* Test seam and host half of GPB_OptimLinRegrCoefCovPar for NON-GAUSSIAN models (the regression coefficients are part of the lbfgs vector, the
 * reference's default: OptimExternal / EvalLLforLBFGSpp with estimate_coef_using_bfgs, include/GPBoost/optim_utils.h:283-420, 575-711; covariates
 * scaled and the intercept started at FindInitialIntercept, re_model_template.h:1112-1300).  The evaluation callback sees the linear predictor as
 * fixed effects of the location parameter and returns the boosting gradient: eval(ctx, op, sigma1_2, a, fixed_effects[n], out3, grad_F[n]) with
 * op 0 / 1 = mode finding (warm start) + value (op 1: + gradient wrt (log sigma1_2, log a) in out3[1..2] and grad_F), op 2 = gradient and grad_F
 * of the current state, op 3 = reset the mode to its previous value, op 4 = forget the mode.  Outputs: (sigma1_2, a), the coefficients on the
 * ORIGINAL scale of the covariates, iterations, negative approximate marginal log-likelihood.  init_coef NULL and init_coef_from_iid_model: the
 * initial coefficients come from the "iid model" (REModel::InitCoefAuxParsFromIidModel, src/GPBoost/re_model.cpp:380-470: the likelihood without the
 * Gaussian process, a plain GLM fitted on the host with the same lbfgs over the coefficients). */
GPBOOST_C_EXPORT int GPB_HIP_OptimizeLaplaceCoefWithCallback(const char* likelihood, int32_t n, int32_t p, const double* X_colmajor, const double* y,
    const double* fixed_effects, const double* init_theta2, const double* init_coef, bool init_coef_from_iid_model, double lr_cov, int max_iter,
    double delta_rel_conv, int m_lbfgs, int (*eval)(void*, int, double, double, const double*, double*, double*), void* ctx, double* theta_out2,
    double* coef_out, int* num_it, double* negll, double* init_coef_out /* optional: the initial coefficients used, original scale */);
/* Test seam and host half of GPB_GetCoef(calc_std_dev = true) for non-Gaussian models: CalcStdDevCoefNonGaussian (re_model_template.h:10851-10897) --
 * numerical Jacobian of X' grad_F (central differences, step coef_i eps^(1/3)), Cholesky inverse, sqrt of the diagonal; NaN if not positive definite. */
GPBOOST_C_EXPORT int GPB_HIP_LaplaceCoefStdErrorsWithCallback(int32_t n, int32_t p, const double* X_colmajor, const double* fixed_effects,
    const double* theta2, const double* coef, int (*eval)(void*, int, double, double, const double*, double*, double*), void* ctx, double* se_out);
/* The underlying gpb_hip_vecchia_t* (include/gpb_hip.h) for resident / sharded use */
GPBOOST_C_EXPORT void* GPB_HIP_GetVecchiaHandle(REModelHandle handle);

#ifdef __cplusplus
}
#endif
#endif /* GPBOOST_C_API_SUBSET_H_ */
