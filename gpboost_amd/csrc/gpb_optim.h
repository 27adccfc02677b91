// gpboost_amd/csrc/gpb_optim.h -- the caller of the hot path: covariance-parameter estimation for one Gaussian GP
// (SURVEY.md section 8f rank 1).  Host control flow only; every likelihood / gradient evaluation goes through a
// callback that returns the seven shard sums of the device kernels (include/gpb_hip.h: gpb_hip_vecchia_grad_terms),
// so y, the coordinates and the neighbour table stay resident in HBM for the whole fit and 3 or 7 doubles cross
// PCIe per evaluation.
//
// Restates, for gauss_likelihood_ && !has_covariates_ && one GP component:
//   REModelTemplate::OptimLinRegrCoefCovPar   include/GPBoost/re_model_template.h:972-1802   (internal "gradient_descent" loop)
//   UpdateCovAuxPars / ApplyMomentumStep      :8691-8846, :5077-5100, NesterovSchedule :6143-6158
//   AvoidTooLargeLearningRatesCovAuxPars      :8353-8375, MaximalLearningRateCovAuxPars :5413-5421
//   CalcDirDerivArmijo...CovAuxPars           :8425-8476
//   ProfileOutSigma2                          :2640-2650, EvalNegLogLikelihoodOnlyUpdateNuggetVariance :3140-3143
//   CheckOptimizerHasConverged                :1928-1949
//   ApplyGaussianNuggetLowerBound             :7849-7874
//   OptimExternal / EvalLLforLBFGSpp          include/GPBoost/optim_utils.h:244-420, :575-711  ("lbfgs", the default)
//   LBFGSSolver::minimize                     external_libs/LBFGSpp/include/LBFGS.h:86-300 (GPBoost's modified copy)
//   LineSearchBacktracking::LineSearch        external_libs/LBFGSpp/include/LBFGSpp/LineSearchBacktracking.h:44-143
//   BFGSMat::add_correction / apply_Hv        external_libs/LBFGSpp/include/LBFGSpp/BFGSMat.h:89-186
// The parameter vector is the reference's TRANSFORMED one (re_model.cpp:301-318): theta = (sigma2, sigma1_2 / sigma2, a).
#ifndef GPB_OPTIM_H_
#define GPB_OPTIM_H_

#include <string>

// t7 = { y' Psi^-1 y, log det Psi, (unused), dB-term and dD-term of the variance parameter, the same two for the range }
// (the layout of gpb_hip_vecchia_grad_terms); with_grad == 0 only needs t7[0..1].  Returns 0 on success.
typedef int (*gpb_terms_fn)(void* ctx, double ratio, double a, int with_grad, double* t7);

struct GpbOptimConfig {                      // defaults: re_model_template.h:5689-5851
  std::string optimizer = "lbfgs";           // InitializeOptimSettings :8277-8280
  double lr_cov_init = -999.;                // 0.1 for gradient_descent, 1 otherwise (SetInitialValueLRCov :8318-8335)
  double acc_rate_cov = 0.5;
  double delta_rel_conv_init = -999.;        // 1e-6 (SetInitialValueDeltaRelConv :8338-8347)
  int max_iter = 1000;
  bool use_nesterov_acc = true;
  int nesterov_schedule_version = 0;
  int momentum_offset = 2;
  std::string convergence_criterion = "relative_change_in_log_likelihood";
  int m_lbfgs = 6;
  double range_const = 1.;                   // sqrt(2 nu): only used by the nugget-bound round trip (TransformBack -> Transform)
  int estimate_cov_par_index[3] = {1, 1, 1}; // <= 0: (error variance, GP variance, range) held at the initial value (Gaussian models;
                                             // re_model_template.h:930-936, ProfileOutSigma2 :2640-2650, MaybeKeepVarianceConstant :7881-7904,
                                             // zero gradient entries :1993-2011)
  bool trace = false;
  // Variables the EVALUATOR profiles out besides the error variance (the regression coefficients of a fit with covariates): lbfgs remembers them at
  // every accepted iterate and goes back to the remembered values when a line search fails (SetLag1ProfiledOutVariables /
  // ResetProfiledOutVariablesToLag1, optim_utils.h:383-390).  op 0 = remember, 1 = go back.  May be null.
  void (*profiled_lag)(void* ctx, int op) = nullptr;
  void* profiled_lag_ctx = nullptr;
  // "gradient_descent" with regression coefficients that the evaluator finds by generalised least squares: at the start of every iteration the
  // coefficients are updated at the CURRENT factor (ProfileOutCoef + EvalNegLogLikelihoodOnlyUpdateFixedEffects, re_model_template.h:1478-1481) and
  // stay fixed during that iteration's step-size search.  Returns the seven sums of the current factor for the new residual.  May be null.
  int (*coef_update)(void* ctx, double ratio, double a, double* t7) = nullptr;
  void* coef_update_ctx = nullptr;
};

struct GpbOptimResult {
  double theta[3];          // transformed scale
  int num_it = 0;
  double negll = 0.;
  int num_ll_evals = 0;     // device launches without / with gradient terms
  int num_grad_evals = 0;
  double lr_cov_final = 0.;
};

// 0 = ok, -1 = error (message in err, at most errlen bytes)
int gpb_optimize_gaussian_cov_pars(const GpbOptimConfig& cfg, int num_data, gpb_terms_fn fn, void* ctx, const double theta_init[3],
                                   GpbOptimResult* out, char* err, int errlen);

// ---- non-Gaussian likelihoods (Laplace approximation): theta = (sigma1_2, a), no nugget, nothing profiled out ----
// The approximate marginal likelihood is evaluated by mode finding that is WARM-STARTED from the previous evaluation's mode
// (likelihoods.h:3790-3797), so the evaluator is stateful:
//   op 0 / 1  find the mode at (var, a) starting from the current mode, return the negative approximate marginal log-likelihood in
//             out3[0]; op 1 also returns its gradient wrt (log var, log a) in out3[1..2]
//   op 2      gradient of the CURRENT state only (parameters and mode of the last op 0 / 1; no new mode finding)
//   op 3      reset the mode to its value before the last mode finding (Likelihood::ResetModeToPreviousValue, likelihoods.h:997-1004)
//   op 4      forget the mode: the next mode finding starts from zero (InitializeModeAvec; the restart with 'nelder_mead' after NaN / Inf,
//             re_model_template.h:1722-1726)
//   + 16      first_update: the reference divides cg_max_num_it(_tridiag) by 3 in the first gradient-descent update (likelihoods.h:3833-3836)
typedef int (*gpb_laplace_fn)(void* ctx, int op, double var, double a, double* out3);

struct GpbLaplaceOptimResult {
  double theta[2];
  int num_it = 0;
  double negll = 0.;
  int num_evals = 0;
};

int gpb_optimize_laplace_cov_pars(const GpbOptimConfig& cfg, gpb_laplace_fn fn, void* ctx, const double theta_init[2],
                                  GpbLaplaceOptimResult* out, char* err, int errlen);

// Standard errors of (sigma1_2, rho) of a non-Gaussian (Laplace) model at theta = (sigma1_2, a): CalcStdDevCovParAuxParsNonGaussian
// (include/GPBoost/re_model_template.h:11029-11117) -- the Hessian of the negative approximate marginal log-likelihood as the numerical Jacobian of
// its analytic gradient (CalcHessianCovParAuxPars, :10915-10968: central differences on the log scale, step max(|log theta_i| h, h), h = 1e-4, every
// evaluation a warm-started mode finding; symmetrised), its Cholesky inverse, and the delta method back to the original scale
// (|d sigma1_2 / d log sigma1_2| = sigma1_2, |d rho / d log a| = rho).  Five evaluations (four perturbed, one to restore the state at theta).
// se_out = NaN where the Hessian is not positive definite (the reference warns and returns NaN).  0 = ok, -1 = evaluator failed.
// estimated2 (optional): <= 0 marks a parameter held fixed (estimate_cov_par_index): NaN for it, the Hessian of the others alone is inverted.
int gpb_laplace_std_errors(gpb_laplace_fn fn, void* ctx, const double theta[2], double range_const, double se_out[2], char* err, int errlen,
                           const int* estimated2 = nullptr);

// ---- non-Gaussian likelihoods WITH a linear predictor: the regression coefficients are part of the lbfgs vector ----
// (OptimExternal / EvalLLforLBFGSpp with estimate_coef_using_bfgs, optim_utils.h:283-420, 575-711; the reference's default for these models).
// The parameter vector is (log sigma1_2, log a, beta_1 .. beta_p) with beta on the SCALED covariates (re_model_template.h:1218-1242).  The
// evaluator sees the linear predictor only as FIXED EFFECTS of the location parameter and returns the boosting gradient d(-mll)/dF:
//   op 0 / 1  mode finding at (var, a) with the location parameter offset fixed_effects (n values, data order), warm-started; out3[0] = negative
//             approximate marginal log-likelihood; op 1 also out3[1..2] = gradient wrt (log var, log a) and grad_F (n values, data order)
//   op 2      gradient and grad_F of the CURRENT state; op 3 reset the mode to its previous value; op 4 forget the mode
typedef int (*gpb_laplace_fe_fn)(void* ctx, int op, double var, double a, const double* fixed_effects, double* out3, double* grad_F);

struct GpbLaplaceCoefResult {
  double theta[2];
  int num_it = 0;
  double negll = 0.;
  int num_evals = 0;
};

// X_scaled: column-major n x p (already centred / scaled where the reference does so); offset: n values added to X beta (NULL: none);
// beta: in = initial values, out = estimates (both on the scaled covariates); C_mu, C_sigma2: FindConstantsCapTooLargeLearningRateCoef.
// learn_cov = false: the covariance parameters stay at theta_init and only beta is in the lbfgs vector (learn_cov_aux_pars = false: the fit of
// the "iid model" that supplies initial coefficients, re_model.cpp:380-470).
int gpb_optimize_laplace_coef_cov_pars(const GpbOptimConfig& cfg, gpb_laplace_fe_fn fn, void* ctx, int n, int p, const double* X_scaled,
                                       const double* offset, double C_mu, double C_sigma2, const double theta_init[2], double* beta,
                                       GpbLaplaceCoefResult* out, char* err, int errlen, bool learn_cov = true);

// ---- non-Gaussian likelihoods WITH auxiliary parameters (gamma / negative_binomial: the shape), estimated jointly with the covariance parameters ----
// (EvalLLforLBFGSpp with EstimateAuxPars(), optim_utils.h:256-283, 345-348: the lbfgs vector is (log sigma1_2, log a, log aux_1 .. log aux_naux);
// GetMaximalLearningRate -> MaximalLearningRateCovAuxPars over covariance AND auxiliary entries, :498-535.)  The evaluator is gpb_laplace_fn with
// the auxiliary parameters in and their gradient out:
//   op 0 / 1  mode finding at (var, a, aux), warm-started; out[0] = negative approximate marginal log-likelihood; op 1 also out[1..2] = gradient wrt
//             (log var, log a) and out[3 .. 3 + naux) = gradient wrt log aux
//   op 2      gradient of the CURRENT state; op 3 reset the mode to its previous value; op 4 forget the mode
typedef int (*gpb_laplace_aux_fn)(void* ctx, int op, double var, double a, const double* aux, int naux, double* out);

struct GpbLaplaceAuxResult {
  double theta[2];
  int num_it = 0;
  double negll = 0.;
  int num_evals = 0;
};

// aux: in = initial values, out = estimates (original scale).  Only optimizer_cov = "lbfgs" (the reference's default).
int gpb_optimize_laplace_cov_aux_pars(const GpbOptimConfig& cfg, gpb_laplace_aux_fn fn, void* ctx, int naux, const double theta_init[2], double* aux,
                                      GpbLaplaceAuxResult* out, char* err, int errlen);

// Standard errors of the covariance and the auxiliary parameters of such a model from the JOINT numerical Hessian (CalcStdDevCovParAuxParsNonGaussian,
// re_model_template.h:11029-11117): 2 (2 + naux) evaluations with gradient + one to restore the state; NaN where none exists (parameter not estimated, Hessian not positive definite).
int gpb_laplace_aux_std_errors(gpb_laplace_aux_fn fn, void* ctx, const double theta[2], const double* aux, int naux, double range_const, double se_cov[2],
                               double* se_aux, char* err, int errlen, const int* estimated2 = nullptr);

// Standard errors of the regression coefficients of a non-Gaussian model: CalcStdDevCoefNonGaussian (include/GPBoost/re_model_template.h:10851-10897) --
// Hessian wrt beta as the numerical Jacobian of X' grad_F (central differences, step beta_i eps^(1/3), at least eps^(1/3)), symmetrised, Cholesky
// inverse, sqrt of its diagonal ("(very) approximate", as the reference says).  X: ORIGINAL covariates (column-major n x p), beta on that scale.
// 2 p evaluations with gradient; se_out = NaN if the Hessian is not positive definite.
int gpb_laplace_coef_std_errors(gpb_laplace_fe_fn fn, void* ctx, int n, int p, const double* X, const double* offset, const double theta[2],
                                const double* beta, double* se_out, char* err, int errlen);

#endif  // GPB_OPTIM_H_
