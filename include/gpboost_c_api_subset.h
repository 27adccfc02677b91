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
 * Supported model slice (anything else returns -1 with a message, never a silent fallback):
 *   one GP (any number of clusters = independent realisations through cluster_ids_data, Gaussian likelihood), no grouped effects /
 *   random coefficients / weights, d <= 3,
 *   cov_fct "exponential" or "matern" with shape 0.5 / 1.5 / 2.5, likelihood "gaussian", and either
 *   gp_approx "vecchia" (num_neighbors <= 62, vecchia_ordering "none" | "random") or gp_approx "none"
 *   (exact GP, dense Cholesky; likelihood and y_aux only);
 *   likelihood "bernoulli_logit" with gp_approx "vecchia" and matrix_inversion_method "default" | "iterative"
 *   (Vecchia-Laplace approximation, "vadu"-preconditioned CG + stochastic Lanczos quadrature: the reference's
 *   defaults for that model; likelihood evaluation only, cov_pars = (sigma1_2, rho), y in {0, 1}).
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

/* c_api.h:1437-1467 -- the optimiser itself is host code outside this library; what IS used on the path: trace, and for
 * the Laplace path cg_max_num_it, cg_max_num_it_tridiag, cg_delta_conv, num_rand_vec_trace, seed_rand_vec_trace,
 * delta_conv_mode_finding (-999 keeps the reference's default) and cg_preconditioner_type (only "vadu") */
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

/* c_api.h:1686-1688 */
GPBOOST_C_EXPORT int GPB_GetLikelihoodName(REModelHandle handle, char* out_str, int* num_char);

/* ---- additions (not in the reference ABI; used by the host mirror, tests and bench) ---- */

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
 * factor + y_aux + H^T Psi^-1 H + solve in one call. */
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
/* The underlying gpb_hip_vecchia_t* (include/gpb_hip.h) for resident / sharded use */
GPBOOST_C_EXPORT void* GPB_HIP_GetVecchiaHandle(REModelHandle handle);

#ifdef __cplusplus
}
#endif
#endif /* GPBOOST_C_API_SUBSET_H_ */
