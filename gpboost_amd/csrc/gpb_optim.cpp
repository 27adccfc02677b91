// gpboost_amd/csrc/gpb_optim.cpp -- see gpb_optim.h for the reference functions each block follows.
#include "gpb_optim.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <limits>
#include <vector>

namespace {
constexpr int kNaOrInf = 2;      // return code of the gradient-based optimisers: NaN / Inf in the parameters or the objective

constexpr double kMinNuggetVarRatio = 1e-10;          // re_model_template.h:5668
constexpr double kLrShrinkageFactor = 0.5;            // :5750
constexpr int kMaxNumberLrShrinkageSteps = 30;        // :5746
constexpr double kCArmijo = 1e-4, kCArmijoMom = 1e-4; // :5809-5811
const double kMaxGradientUpdateLogScale = std::log(100.);   // :5780-5782

constexpr int kCheckFailed = -3;     // a CHECK of the reference that ends the fit (distinct from an evaluator error, which a simplex search may survive)

struct Fail {
  char* err; int errlen;
  int operator()(const char* fmt, ...) const {
    va_list ap; va_start(ap, fmt); vsnprintf(err, errlen, fmt, ap); va_end(ap);
    return -1;
  }
};

// The model state the reference's optimiser works on (members of REModelTemplate), reduced to one Gaussian GP.
struct State {
  const GpbOptimConfig& cfg;
  int n; gpb_terms_fn fn; void* ctx;
  double cur_r = 0., cur_a = 0.;     // parameters of the factor that is "current" (SetCovParsComps + CalcCovFactor)
  bool have_factor = false, have_grad = false;
  double yPy = 0., logdet = 0.;      // yTPsiInvy_, log_det_Psi_
  double g[4] = {0., 0., 0., 0.};    // gradient shard sums of the current factor
  double sigma2 = 0., sigma2_lag1 = 0.;
  double negll = 0.;                 // neg_log_likelihood_
  int n_ll = 0, n_grad = 0;
  double th_first[3] = {0., 0., 0.}; // cov_pars_set_first_time_ (:1201)
  const Fail* fail_ = nullptr;

  // MaybeKeepVarianceConstant (:7881-7904): the nugget is estimated, the marginal variance is not -> the RATIO follows the nugget so
  // that sigma1_2 = ratio * sigma2 stays at its initial value
  void keep_variance_constant(double th[3]) const {
    if (cfg.estimate_cov_par_index[0] > 0 && cfg.estimate_cov_par_index[1] <= 0) th[1] = th_first[1] * th_first[0] / th[0];
  }

  // ApplyGaussianNuggetLowerBound on the transformed scale (:7849-7874): TransformBack, bound, Transform again
  void nugget_bound(double th[3]) const {
    const double sigma2_o = th[0], var_o = th[1] * th[0], rho_o = cfg.range_const / th[2];
    if (!std::isfinite(var_o) || var_o <= 0.) return;
    const double nugget_min = kMinNuggetVarRatio / (1. - kMinNuggetVarRatio) * var_o;
    if (std::isfinite(nugget_min) && sigma2_o < nugget_min) {
      th[0] = nugget_min;
      th[1] = var_o / nugget_min;
      th[2] = cfg.range_const / rho_o;
    }
  }
  double negll_at(double s2) const {   // :3132 / :3142
    return yPy / 2. / s2 + logdet / 2. + n / 2. * (std::log(s2) + std::log(2 * M_PI));
  }
  // CalcCovFactorOrModeAndNegLL (:2832-2852).  with_grad: also fetch the gradient sums in the same launch.
  int calc(const double th_in[3], bool with_grad) {
    double th[3] = {th_in[0], th_in[1], th_in[2]};
    keep_variance_constant(th);
    nugget_bound(th);
    // RECompGP::SetCovPars -> CovFunction::CheckPars (re_comp.h:1211-1217, cov_fcts.h:426-429): a parameter that has run to zero (exp underflow of an
    // unbounded simplex / line search) ends the fit with an error, it is not evaluated
    if (th[1] <= 0. || th[2] <= 0.) {
      if (fail_) (*fail_)("Check failed: pars[i] > 0. (a covariance parameter has reached zero during the optimisation: %g, %g on the transformed scale) ", th[1], th[2]);
      return kCheckFailed;
    }
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    if (fn(ctx, th[1], th[2], with_grad ? 1 : 0, t)) return -1;
    if (with_grad) ++n_grad; else ++n_ll;
    cur_r = th[1]; cur_a = th[2]; have_factor = true;
    yPy = t[0]; logdet = t[1];
    have_grad = with_grad;
    if (with_grad) { g[0] = t[3]; g[1] = t[4]; g[2] = t[5]; g[3] = t[6]; }
    negll = negll_at(th[0]);
    return 0;
  }
  // CalcGradPars(cov_pars, ., true, false, ., ., include_error_var = false) (:1988-2011): gradient of the CURRENT factor wrt
  // (log ratio, log a), with cov_pars[0] = s2 in the divisions
  int grad(double s2, double out[2]) {
    if (!have_grad) {
      double t[7] = {0, 0, 0, 0, 0, 0, 0};
      if (fn(ctx, cur_r, cur_a, 1, t)) return -1;
      ++n_grad;
      g[0] = t[3]; g[1] = t[4]; g[2] = t[5]; g[3] = t[6];
      have_grad = true;
    }
    out[0] = cfg.estimate_cov_par_index[1] > 0 ? g[0] / s2 + g[1] : 0.;     // parameters that are not estimated: no gradient entry (:2004)
    out[1] = cfg.estimate_cov_par_index[2] > 0 ? g[2] / s2 + g[3] : 0.;
    return 0;
  }
  void profile_out_sigma2(double th[3]) {   // :2640-2650
    if (cfg.estimate_cov_par_index[0] > 0) sigma2 = yPy / n;
    th[0] = sigma2;
    nugget_bound(th);
    sigma2 = th[0];
  }
};

double nesterov_schedule(int iter, int version, double acc_rate, int momentum_offset) {   // :6143-6158
  if (iter < momentum_offset) return 0.;
  if (version == 0) return acc_rate;
  return 1. - (3. / (6. + iter));
}

bool finite3(const double* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// ---------------------------------------------------------------------------------------------------------------------
// internal optimiser "gradient_descent" (+ Nesterov acceleration): OptimLinRegrCoefCovPar :1436-1704
// ---------------------------------------------------------------------------------------------------------------------
int run_gradient_descent(State& st, const GpbOptimConfig& cfg, double th[3], GpbOptimResult* out, const Fail& fail) {
  double lr_cov = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 0.1;
  const double delta_rel_conv = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const bool nesterov = cfg.use_nesterov_acc;
  if (nesterov && cfg.nesterov_schedule_version == 1)   // :8701-8703 (armijo_condition_ is always true)
    return fail("Armijo condition backtracking is not implemented when nesterov_schedule_version = 1 ");
  st.sigma2 = th[0];
  st.sigma2_lag1 = st.sigma2;
  if (st.calc(th, true)) return -1;
  if (!std::isfinite(st.negll))
    return fail("%s occurred in initial negative log-likelihood. Possible solutions: try other initial values ('init_cov_pars')",
                std::isnan(st.negll) ? "NaN" : "Inf");
  double aux[3] = {th[0], th[1], th[2]}, aux_lag1[3] = {th[0], th[1], th[2]};   // cov_pars_after_grad_aux(_lag1)
  double th_lag1[3];
  int num_it = cfg.max_iter;
  for (int it = 0; it < cfg.max_iter; ++it) {
    const double negll_lag1 = st.negll;
    std::copy(th, th + 3, th_lag1);
    double negll_after_lin_coef_update = negll_lag1;                 // no covariates (:1511)
    if (cfg.coef_update) {                                           // coefficients by generalised least squares at the current factor (:1478-1481)
      double t[7] = {0, 0, 0, 0, 0, 0, 0};
      if (cfg.coef_update(cfg.coef_update_ctx, st.cur_r, st.cur_a, t)) return -1;
      ++st.n_grad;
      st.yPy = t[0]; st.logdet = t[1];
      st.g[0] = t[3]; st.g[1] = t[4]; st.g[2] = t[5]; st.g[3] = t[6]; st.have_grad = true;
      negll_after_lin_coef_update = st.negll_at(th[0]);
    }
    st.profile_out_sigma2(th);                                       // :1518-1520
    double grad[2];
    if (st.grad(th[0], grad)) return -1;                             // :1524
    // AvoidTooLargeLearningRatesCovAuxPars
    const double max_lr = kMaxGradientUpdateLogScale / std::max(std::fabs(grad[0]), std::fabs(grad[1]));
    if (lr_cov > max_lr) lr_cov = max_lr;
    // CalcDirDerivArmijoAndLearningRateConstChangeCovAuxPars (armijo_condition_ = true)
    const double dir_deriv = -(grad[0] * grad[0] + grad[1] * grad[1]);
    double mom_dir_deriv = 0.;
    if (nesterov) {
      const double d1 = std::log(th[1]) - std::log(aux[1]), d2 = std::log(th[2]) - std::log(aux[2]);
      mom_dir_deriv = grad[0] * d1 + grad[1] * d2;
    }
    // UpdateCovAuxPars
    double th_new[3] = {th[0], 0., 0.};
    double lr = lr_cov, acc = cfg.acc_rate_cov;
    bool decrease_found = false, halving_done = false;
    for (int ih = 0; ih < kMaxNumberLrShrinkageSteps; ++ih) {
      th_new[0] = th[0];
      th_new[1] = std::exp(std::log(th[1]) - lr * grad[0]);          // update on the log-scale
      th_new[2] = std::exp(std::log(th[2]) - lr * grad[1]);
      st.nugget_bound(th_new);
      if (nesterov) {
        std::copy(th_new, th_new + 3, aux);
        const double mu = nesterov_schedule(it, cfg.nesterov_schedule_version, acc, cfg.momentum_offset);   // ApplyMomentumStep
        th_new[0] = aux[0];
        th_new[1] = std::exp((mu + 1.) * std::log(aux[1]) - mu * std::log(aux_lag1[1]));
        th_new[2] = std::exp((mu + 1.) * std::log(aux[2]) - mu * std::log(aux_lag1[2]));
        st.nugget_bound(th_new);
      }
      if (st.calc(th_new, ih == 0)) return -1;                       // first trial: gradient sums in the same launch
      const double mu = nesterov ? nesterov_schedule(it, cfg.nesterov_schedule_version, acc, cfg.momentum_offset) : 0.;
      if (st.negll <= negll_after_lin_coef_update + kCArmijo * lr * dir_deriv + kCArmijoMom * mu * mom_dir_deriv) decrease_found = true;
      if (decrease_found) break;
      halving_done = true;
      lr *= kLrShrinkageFactor;
      acc *= 0.5;
    }
    if (halving_done) lr_cov = lr;                                   // permanent for gradient descent (:8824)
    if (nesterov) std::copy(aux, aux + 3, aux_lag1);
    std::copy(th_new, th_new + 3, th);
    if (cfg.trace)
      fprintf(stderr, "[gpboost_amd] it %d: cov pars (transformed) %.10g %.10g %.10g negll %.10g lr %g\n", it + 1, th[0], th[1], th[2],
              st.negll, lr_cov);
    if (!std::isfinite(st.negll) || !finite3(th))
      return kNaOrInf;                                             // the caller starts again with 'nelder_mead' (:1706-1731)
    // CheckOptimizerHasConverged
    bool terminate = false;
    if (cfg.convergence_criterion == "relative_change_in_parameters") {
      double d = 0., r = 0.;
      for (int i = 0; i < 3; ++i) { d += (th[i] - th_lag1[i]) * (th[i] - th_lag1[i]); r += th_lag1[i] * th_lag1[i]; }
      terminate = std::sqrt(d) <= delta_rel_conv * std::sqrt(r);
    } else {
      terminate = (negll_lag1 - st.negll) <= delta_rel_conv * std::max(std::fabs(negll_lag1), 1.);
    }
    if (terminate) { num_it = it + 1; break; }
  }
  out->num_it = num_it;
  out->lr_cov_final = lr_cov;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// "lbfgs" (the default): LBFGSSolver<double, LineSearchBacktracking> on x = log(theta[1:]) with the nugget profiled out
// ---------------------------------------------------------------------------------------------------------------------
struct BfgsMat {     // BFGSMat.h:69-186 for n = 2
  int m; double theta = 1.; int ncorr = 0, ptr;
  std::vector<double> s, y, ys, alpha;   // s, y: column j at [2 j, 2 j + 1]
  explicit BfgsMat(int m_) : m(m_), ptr(m_), s(2 * m_), y(2 * m_), ys(m_), alpha(m_) {}
  void add_correction(const double* sv, const double* yv) {
    const int loc = ptr % m;
    s[2 * loc] = sv[0]; s[2 * loc + 1] = sv[1];
    y[2 * loc] = yv[0]; y[2 * loc + 1] = yv[1];
    const double d = sv[0] * yv[0] + sv[1] * yv[1];
    ys[loc] = d;
    theta = (yv[0] * yv[0] + yv[1] * yv[1]) / d;
    if (ncorr < m) ++ncorr;
    ptr = loc + 1;
  }
  void apply_Hv(const double* v, double a, double* res) {
    res[0] = a * v[0]; res[1] = a * v[1];
    int j = ptr % m;
    for (int i = 0; i < ncorr; ++i) {
      j = (j + m - 1) % m;
      alpha[j] = (s[2 * j] * res[0] + s[2 * j + 1] * res[1]) / ys[j];
      res[0] -= alpha[j] * y[2 * j]; res[1] -= alpha[j] * y[2 * j + 1];
    }
    res[0] /= theta; res[1] /= theta;
    for (int i = 0; i < ncorr; ++i) {
      const double beta = (y[2 * j] * res[0] + y[2 * j + 1] * res[1]) / ys[j];
      res[0] += (alpha[j] - beta) * s[2 * j]; res[1] += (alpha[j] - beta) * s[2 * j + 1];
      j = (j + 1) % m;
    }
  }
};

// EvalLLforLBFGSpp::operator() (optim_utils.h:283-420) for learn_cov_aux_pars, profile_out_error_variance, no covariates
int lbfgs_objective(State& st, const double x[2], bool eval_likelihood, bool calc_gradient, bool grad_in_same_launch, double* fx,
                    double grad[2]) {
  double th[3] = {st.sigma2, std::exp(x[0]), std::exp(x[1])};
  if (eval_likelihood) {
    if (const int rc = st.calc(th, grad_in_same_launch)) return rc;
    st.profile_out_sigma2(th);
    *fx = st.negll_at(th[0]);                         // EvalNegLogLikelihoodOnlyUpdateNuggetVariance
  }
  if (calc_gradient && st.grad(th[0], grad)) return -1;
  return 0;
}

int run_lbfgs(State& st, const GpbOptimConfig& cfg, double th[3], GpbOptimResult* out, const Fail& fail) {
  const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const double initial_step_factor = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 1.;
  const double epsilon = 1e-20, epsilon_rel = 1e-20, ftol = 1e-4;     // optim_utils.h:655-661, Param.h:188
  const int max_linesearch = 20;
  constexpr double eps = std::numeric_limits<double>::epsilon();
  st.sigma2 = th[0];
  st.sigma2_lag1 = st.sigma2;                                           // SetLag1ProfiledOutVariables (:1352)
  if (cfg.profiled_lag) cfg.profiled_lag(cfg.profiled_lag_ctx, 0);
  BfgsMat bfgs(cfg.m_lbfgs);
  double x[2] = {std::log(th[1]), std::log(th[2])}, xp[2], grad[2], gradp[2], drt[2];
  double fx = 1e99;
  if (lbfgs_objective(st, x, true, true, true, &fx, grad)) return -1;
  if (!std::isfinite(fx))
    return fail("%s occurred in initial negative log-likelihood. Possible solutions: try other initial values ('init_cov_pars')",
                std::isnan(fx) ? "NaN" : "Inf");
  double gnorm = std::sqrt(grad[0] * grad[0] + grad[1] * grad[1]);
  double fx_past = fx;                                                  // m_fx[0] (past = 1)
  int k = 1;
  bool done = gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(x[0] * x[0] + x[1] * x[1]);
  if (!done) {
    drt[0] = -grad[0]; drt[1] = -grad[1];
    double step = initial_step_factor / std::sqrt(drt[0] * drt[0] + drt[1] * drt[1]);
    for (;;) {
      xp[0] = x[0]; xp[1] = x[1]; gradp[0] = grad[0]; gradp[1] = grad[1];
      const double max_lr = kMaxGradientUpdateLogScale / std::max(std::fabs(drt[0]), std::fabs(drt[1]));   // GetMaximalLearningRate
      if (max_lr < step) step = max_lr;
      // LineSearchBacktracking (Armijo)
      bool line_search_failed = false;
      {
        if (step <= 0.) return fail("GPModel lbfgs: 'step' must be positive");
        const double fx_init = fx, dg_init = grad[0] * drt[0] + grad[1] * drt[1];
        if (dg_init > 0.) return fail("GPModel lbfgs: the moving direction increases the objective function value");
        const double test_decr = ftol * dg_init;
        int iter;
        for (iter = 0; iter < max_linesearch; ++iter) {
          x[0] = xp[0] + step * drt[0]; x[1] = xp[1] + step * drt[1];
          if (lbfgs_objective(st, x, true, false, iter == 0, &fx, grad)) return -1;
          double width;
          if (fx > fx_init + step * test_decr || (fx != fx)) {
            width = ((fx - fx_init) > 2. * std::max(std::fabs(fx_init), 1.)) ? 0.5 / 16. : 0.5;
          } else {
            break;                                                      // Armijo condition is met
          }
          step *= width;
        }
        if (iter >= max_linesearch) {
          x[0] = xp[0]; x[1] = xp[1];
          st.sigma2 = st.sigma2_lag1;                                   // ResetProfiledOutVariablesToLag1
          line_search_failed = true;
          fx = fx_init;
          step = 0.;
        }
      }
      double dummy;
      if (lbfgs_objective(st, x, false, true, false, &dummy, grad)) return -1;   // gradient of the CURRENT factor
      // (an evaluator that profiles variables out does so again when it is asked for this gradient: going back to the remembered values comes after)
      if (line_search_failed && cfg.profiled_lag) cfg.profiled_lag(cfg.profiled_lag_ctx, 1);
      gnorm = std::sqrt(grad[0] * grad[0] + grad[1] * grad[1]);
      bool has_converged = gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(x[0] * x[0] + x[1] * x[1]);
      if ((fx_past - fx) <= delta * std::max(std::fabs(fx_past), 1.)) has_converged = true;
      if (cfg.max_iter != 0 && k >= cfg.max_iter) has_converged = true;
      st.sigma2_lag1 = st.sigma2;                                       // SetLag1ProfiledOutVariables
      if (cfg.profiled_lag) cfg.profiled_lag(cfg.profiled_lag_ctx, 0);
      if (cfg.trace)
        fprintf(stderr, "[gpboost_amd] lbfgs it %d: sigma2 %.10g ratio %.10g a %.10g negll %.10g step %g\n", k, st.sigma2, std::exp(x[0]),
                std::exp(x[1]), fx, step);
      if (has_converged) break;
      const double sv[2] = {x[0] - xp[0], x[1] - xp[1]}, yv[2] = {grad[0] - gradp[0], grad[1] - gradp[1]};
      if (sv[0] * yv[0] + sv[1] * yv[1] > eps * (yv[0] * yv[0] + yv[1] * yv[1])) bfgs.add_correction(sv, yv);
      step = 1.;
      bfgs.apply_Hv(grad, -1., drt);
      fx_past = fx;
      ++k;
    }
  }
  if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(fx))
    return kNaOrInf;                                               // the caller starts again with 'nelder_mead' (:1706-1731)
  th[0] = st.sigma2; th[1] = std::exp(x[0]); th[2] = std::exp(x[1]);   // OptimExternal :690-693
  st.negll = fx;
  out->num_it = k;
  out->lr_cov_final = initial_step_factor;
  return 0;
}

// "nelder_mead": OptimLib's simplex search as GPBoost ships it (external_libs/OptimLib/unconstrained/nm.hpp:95-372, settings of
// OptimExternal, optim_utils.h:624-645, :677-684) for two parameters x = log(theta[1:]); the objective is EvalLLforOptimLib
// (optim_utils.h:61-213) -- likelihood evaluations only, which makes this optimiser a pure consumer of the hot path.  f(x, &fx) returns
// non-zero on an evaluator error.  adaptive_pars (the default) gives the classic coefficients for n = 2: reflection 1, contraction
// 0.75 - 1 / (2 n) = 0.5, expansion 1 + 2 / n = 2, shrinkage 1 - 1 / n = 0.5.  The vertex order after std::sort, the centroid of the n best
// vertices and the two relative-change measures (largest change of the SORTED value / point arrays against the arrays of the previous
// iteration) are the reference's.  Returns the iteration count in *num_it and the best vertex in x (its value in *fbest, re-evaluated as
// error_reporting does, error_reporting.ipp:25-75).
template <class F>
int nelder_mead_2d(F&& f, double x[2], const GpbOptimConfig& cfg, double delta, int* num_it, double* fbest) {
  constexpr int n = 2;
  const double par_alpha = 1.0, par_beta = 0.75 - 1.0 / (2.0 * n), par_gamma = 1.0 + 2.0 / n, par_delta = 1.0 - 1.0 / n;
  double tol_f, tol_x;
  if (cfg.convergence_criterion == "relative_change_in_parameters") { tol_x = delta; tol_f = 1e-20; }
  else { tol_f = delta; tol_x = 1e-20; }
  const size_t iter_max = (size_t)cfg.max_iter;
  double fv[n + 1], fv_old[n + 1], pt[n + 1][n], pt_old[n + 1][n];
  if (f(x, &fv[0])) return -1;
  pt[0][0] = x[0]; pt[0][1] = x[1];
  for (int i = 1; i < n + 1; ++i) {
    for (int c = 0; c < n; ++c) pt[i][c] = x[c] + (x[i - 1] != 0.0 ? 0.05 * x[i - 1] : 0.00025) * (c == i - 1 ? 1.0 : 0.0);
    if (f(pt[i], &fv[i])) return -1;
  }
  size_t iter = 0;
  double rel_f = 2 * std::fabs(tol_f), rel_x = 2 * std::fabs(tol_x);
  std::copy(fv, fv + n + 1, fv_old);
  std::copy(&pt[0][0], &pt[0][0] + (n + 1) * n, &pt_old[0][0]);
  bool has_converged = false;
  while (!has_converged) {
    ++iter;
    bool next_iter = false;
    size_t idx[n + 1] = {0, 1, 2};
    std::sort(idx, idx + n + 1, [&](size_t a, size_t b) { return fv[a] < fv[b]; });
    {
      double fs[n + 1], ps[n + 1][n];
      for (int i = 0; i < n + 1; ++i) { fs[i] = fv[idx[i]]; ps[i][0] = pt[idx[i]][0]; ps[i][1] = pt[idx[i]][1]; }
      std::copy(fs, fs + n + 1, fv);
      std::copy(&ps[0][0], &ps[0][0] + (n + 1) * n, &pt[0][0]);
    }
    double cen[n], xr[n], xt[n], ft;
    for (int c = 0; c < n; ++c) cen[c] = (pt[0][c] + pt[1][c]) / (double)n;
    for (int c = 0; c < n; ++c) xr[c] = cen[c] + par_alpha * (cen[c] - pt[n][c]);
    double fr;
    if (f(xr, &fr)) return -1;
    if (fr >= fv[0] && fr < fv[n - 1]) { pt[n][0] = xr[0]; pt[n][1] = xr[1]; fv[n] = fr; next_iter = true; }
    if (!next_iter && fr < fv[0]) {                                   // expansion
      for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_gamma * (xr[c] - cen[c]);
      if (f(xt, &ft)) return -1;
      if (ft < fr) { pt[n][0] = xt[0]; pt[n][1] = xt[1]; fv[n] = ft; } else { pt[n][0] = xr[0]; pt[n][1] = xr[1]; fv[n] = fr; }
      next_iter = true;
    }
    if (!next_iter && fr >= fv[n - 1]) {
      if (fr < fv[n]) {                                               // outside contraction
        for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_beta * (xr[c] - cen[c]);
        if (f(xt, &ft)) return -1;
        if (ft <= fr) { pt[n][0] = xt[0]; pt[n][1] = xt[1]; fv[n] = ft; next_iter = true; }
      } else {                                                        // inside contraction
        for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_beta * (pt[n][c] - cen[c]);
        if (f(xt, &ft)) return -1;
        if (ft < fv[n]) { pt[n][0] = xt[0]; pt[n][1] = xt[1]; fv[n] = ft; next_iter = true; }
      }
    }
    if (!next_iter) {                                                 // shrink towards the best vertex
      for (int i = 1; i < n + 1; ++i) for (int c = 0; c < n; ++c) pt[i][c] = pt[0][c] + par_delta * (pt[i][c] - pt[0][c]);
      for (int i = 1; i < n + 1; ++i) if (f(pt[i], &fv[i])) return -1;
    }
    double num = 0., den = 0.;
    for (int i = 0; i < n + 1; ++i) { num = std::max(num, std::fabs(fv[i] - fv_old[i])); den = std::max(den, std::fabs(fv_old[i])); }
    rel_f = num / (1.0e-08 + den);
    std::copy(fv, fv + n + 1, fv_old);
    if (tol_x >= 0.0) {
      num = 0.; den = 0.;
      for (int i = 0; i < n + 1; ++i) for (int c = 0; c < n; ++c) { num = std::max(num, std::fabs(pt[i][c] - pt_old[i][c])); den = std::max(den, std::fabs(pt_old[i][c])); }
      rel_x = num / (1.0e-08 + den);
      std::copy(&pt[0][0], &pt[0][0] + (n + 1) * n, &pt_old[0][0]);
    }
    has_converged = !(rel_f > tol_f && rel_x > tol_x && iter < iter_max);
  }
  int best = 0;
  for (int i = 1; i < n + 1; ++i) if (fv[i] < fv[best]) best = i;
  x[0] = pt[best][0]; x[1] = pt[best][1];
  if (f(x, fbest)) return -1;                                         // settings->opt_fn_value = opt_objfn(x_p) (error_reporting)
  *num_it = (int)iter;
  return 0;
}

// configured = false: the restart after NaN / Inf in a gradient-based fit -- the reference makes the 'estimate_cov_par_index' check for the
// CONFIGURED optimiser only (:1075-1080) and then searches over both parameters of the simplex
int run_nelder_mead(State& st, const GpbOptimConfig& cfg, double th[3], GpbOptimResult* out, const Fail& fail, bool configured = true) {
  for (int k = 0; k < 3 && configured; ++k)
    if (cfg.estimate_cov_par_index[k] <= 0)
      return fail("Holding fix some covariance parameters (via 'estimate_cov_par_index') when using optimizer_cov = 'nelder_mead' as optimizer is currently not supported ");   // :1075-1080
  const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-8;      // SetInitialValueDeltaRelConv :8338-8347
  st.sigma2 = th[0];
  double x[2] = {std::log(th[1]), std::log(th[2])}, fx = 1e99, g[2];
  // An evaluator error at a TRIAL vertex (parameters outside the kernels' range: a > 1e20, variance >= 1e100) is an objective of +Inf there --
  // the reference's simplex sees Inf / NaN at such a vertex and contracts away from it; only the first evaluation (the initial point: no
  // device, no response) aborts the search
  int n_ok = 0;
  auto f = [&](const double* xv, double* fv) -> int {
    const double xx[2] = {xv[0], xv[1]};
    const int rc = lbfgs_objective(st, xx, true, false, false, fv, g);
    if (rc == 0) { ++n_ok; return 0; }
    if (n_ok == 0 || rc == kCheckFailed) return rc;
    *fv = INFINITY;
    return 0;
  };
  int num_it = 0;
  if (nelder_mead_2d(f, x, cfg, delta, &num_it, &fx)) return -1;
  if (f(x, &fx)) return -1;                                           // OptimExternal re-evaluates at the solution (optim_utils.h:677-684)
  if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(fx))
    return fail("NaN or Inf occurred in covariance parameter optimization using 'nelder_mead'; try other initial values");
  th[0] = st.sigma2; th[1] = std::exp(x[0]); th[2] = std::exp(x[1]);
  st.negll = fx;
  out->num_it = num_it;
  out->lr_cov_final = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 1.;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// non-Gaussian likelihoods: the same two optimisers on theta = (sigma1_2, a); the objective is the Laplace approximation with a
// warm-started mode (EvalLLforLBFGSpp, optim_utils.h:340-344, :395-414; OptimLinRegrCoefCovPar non-Gaussian branches :1380-1410,
// :1514-1550; UpdateCovAuxPars :8737-8742, :8806-8812)
// ---------------------------------------------------------------------------------------------------------------------
struct LapState {
  gpb_laplace_fn fn; void* ctx;
  int n_evals = 0;
  double negll = 0.;
  const Fail* fail_ = nullptr;
  int eval(const double th[2], bool with_grad, bool first_update, double* grad) {
    double o[3] = {0, 0, 0};
    if (th[0] <= 0. || th[1] <= 0.)      // CovFunction::CheckPars (cov_fcts.h:426-429), as for the Gaussian models above
    {                                    // ends the fit (kCheckFailed), also inside a simplex search that would survive an evaluator error
      if (fail_) (*fail_)("Check failed: pars[i] > 0. (a covariance parameter has reached zero during the optimisation: %g, %g) ", th[0], th[1]);
      return kCheckFailed;
    }
    if (fn(ctx, (with_grad ? 1 : 0) | (first_update ? 16 : 0), th[0], th[1], o)) return -1;
    ++n_evals;
    negll = o[0];
    if (with_grad && grad) { grad[0] = o[1]; grad[1] = o[2]; }
    return 0;
  }
  int grad_current(const double th[2], double* grad) {
    double o[3] = {0, 0, 0};
    if (fn(ctx, 2, th[0], th[1], o)) return -1;
    grad[0] = o[1]; grad[1] = o[2];
    return 0;
  }
  int reset_mode() { double o[3]; return fn(ctx, 3, 0., 0., o); }
};

int run_gradient_descent_laplace(LapState& st, const GpbOptimConfig& cfg, double th[2], GpbLaplaceOptimResult* out, const Fail& fail) {
  double lr_cov = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 0.1;
  const double delta_rel_conv = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const bool nesterov = cfg.use_nesterov_acc;
  if (nesterov && cfg.nesterov_schedule_version == 1)
    return fail("Armijo condition backtracking is not implemented when nesterov_schedule_version = 1 ");
  double grad[2];
  if (st.eval(th, true, false, grad)) return -1;
  if (!std::isfinite(st.negll))
    return fail("%s occurred in initial approximate negative marginal log-likelihood. Possible solutions: try other initial values ('init_cov_pars')",
                std::isnan(st.negll) ? "NaN" : "Inf");
  double aux[2] = {th[0], th[1]}, aux_lag1[2] = {th[0], th[1]}, th_lag1[2];
  int num_it = cfg.max_iter;
  bool have_grad = true;
  for (int it = 0; it < cfg.max_iter; ++it) {
    const double negll_lag1 = st.negll;
    th_lag1[0] = th[0]; th_lag1[1] = th[1];
    if (!have_grad && st.grad_current(th, grad)) return -1;
    const double max_lr = kMaxGradientUpdateLogScale / std::max(std::fabs(grad[0]), std::fabs(grad[1]));
    if (lr_cov > max_lr) lr_cov = max_lr;
    const double dir_deriv = -(grad[0] * grad[0] + grad[1] * grad[1]);
    double mom_dir_deriv = 0.;
    if (nesterov) mom_dir_deriv = grad[0] * (std::log(th[0]) - std::log(aux[0])) + grad[1] * (std::log(th[1]) - std::log(aux[1]));
    double th_new[2], gnew[2];
    double lr = lr_cov, acc = cfg.acc_rate_cov;
    bool decrease_found = false, halving_done = false, new_grad = false;
    for (int ih = 0; ih < kMaxNumberLrShrinkageSteps; ++ih) {
      th_new[0] = std::exp(std::log(th[0]) - lr * grad[0]);
      th_new[1] = std::exp(std::log(th[1]) - lr * grad[1]);
      if (nesterov) {
        aux[0] = th_new[0]; aux[1] = th_new[1];
        const double mu = nesterov_schedule(it, cfg.nesterov_schedule_version, acc, cfg.momentum_offset);
        th_new[0] = std::exp((mu + 1.) * std::log(aux[0]) - mu * std::log(aux_lag1[0]));       // momentum on the log-scale, all parameters
        th_new[1] = std::exp((mu + 1.) * std::log(aux[1]) - mu * std::log(aux_lag1[1]));
      }
      new_grad = ih == 0;
      if (st.eval(th_new, new_grad, it == 0, gnew)) return -1;
      const double mu = nesterov ? nesterov_schedule(it, cfg.nesterov_schedule_version, acc, cfg.momentum_offset) : 0.;
      if (st.negll <= negll_lag1 + kCArmijo * lr * dir_deriv + kCArmijoMom * mu * mom_dir_deriv) decrease_found = true;
      if (decrease_found) break;
      halving_done = true;
      lr *= kLrShrinkageFactor;
      acc *= 0.5;
      if (st.reset_mode()) return -1;                                // the parameters are discarded, so is the mode found for them
    }
    if (halving_done) lr_cov = lr;
    if (nesterov) { aux_lag1[0] = aux[0]; aux_lag1[1] = aux[1]; }
    th[0] = th_new[0]; th[1] = th_new[1];
    have_grad = decrease_found && new_grad;
    if (have_grad) { grad[0] = gnew[0]; grad[1] = gnew[1]; }
    if (cfg.trace)
      fprintf(stderr, "[gpboost_amd] it %d: cov pars (transformed) %.10g %.10g negll %.10g lr %g\n", it + 1, th[0], th[1], st.negll, lr_cov);
    if (!std::isfinite(st.negll) || !std::isfinite(th[0]) || !std::isfinite(th[1]))
      return kNaOrInf;                                             // the caller starts again with 'nelder_mead' (:1706-1731)
    bool terminate = false;
    if (cfg.convergence_criterion == "relative_change_in_parameters") {
      const double d = std::sqrt((th[0] - th_lag1[0]) * (th[0] - th_lag1[0]) + (th[1] - th_lag1[1]) * (th[1] - th_lag1[1]));
      terminate = d <= delta_rel_conv * std::sqrt(th_lag1[0] * th_lag1[0] + th_lag1[1] * th_lag1[1]);
    } else {
      terminate = (negll_lag1 - st.negll) <= delta_rel_conv * std::max(std::fabs(negll_lag1), 1.);
    }
    if (terminate) { num_it = it + 1; break; }
  }
  out->num_it = num_it;
  return 0;
}

int run_lbfgs_laplace(LapState& st, const GpbOptimConfig& cfg, double th[2], GpbLaplaceOptimResult* out, const Fail& fail) {
  const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const double initial_step_factor = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 1.;
  const double epsilon = 1e-20, epsilon_rel = 1e-20, ftol = 1e-4;
  const int max_linesearch = 20;
  constexpr double eps = std::numeric_limits<double>::epsilon();
  BfgsMat bfgs(cfg.m_lbfgs);
  double x[2] = {std::log(th[0]), std::log(th[1])}, xp[2], grad[2], gradp[2], drt[2], thx[2];
  auto at = [&](const double* xx) { thx[0] = std::exp(xx[0]); thx[1] = std::exp(xx[1]); };
  // parameters that are not estimated have no gradient entry (estimate_cov_par_index, likelihoods.h:6622): lbfgs then never moves them
  const bool est0 = cfg.estimate_cov_par_index[0] > 0, est1 = cfg.estimate_cov_par_index[1] > 0;
  auto mask = [&](double* g) { if (!est0) g[0] = 0.; if (!est1) g[1] = 0.; };
  at(x);
  if (st.eval(thx, true, false, grad)) return -1;
  mask(grad);
  double fx = st.negll;
  if (!std::isfinite(fx))
    return fail("%s occurred in initial approximate negative marginal log-likelihood. Possible solutions: try other initial values ('init_cov_pars')",
                std::isnan(fx) ? "NaN" : "Inf");
  double gnorm = std::sqrt(grad[0] * grad[0] + grad[1] * grad[1]);
  double fx_past = fx;
  int k = 1;
  if (!(gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(x[0] * x[0] + x[1] * x[1]))) {
    drt[0] = -grad[0]; drt[1] = -grad[1];
    double step = initial_step_factor / std::sqrt(drt[0] * drt[0] + drt[1] * drt[1]);
    for (;;) {
      xp[0] = x[0]; xp[1] = x[1]; gradp[0] = grad[0]; gradp[1] = grad[1];
      const double max_lr = kMaxGradientUpdateLogScale / std::max(std::fabs(drt[0]), std::fabs(drt[1]));
      if (max_lr < step) step = max_lr;
      bool grad_is_current = false;
      {
        if (step <= 0.) return fail("GPModel lbfgs: 'step' must be positive");
        const double fx_init = fx, dg_init = grad[0] * drt[0] + grad[1] * drt[1];
        if (dg_init > 0.) return fail("GPModel lbfgs: the moving direction increases the objective function value");
        const double test_decr = ftol * dg_init;
        double gtrial[2];
        int iter;
        for (iter = 0; iter < max_linesearch; ++iter) {
          x[0] = xp[0] + step * drt[0]; x[1] = xp[1] + step * drt[1];
          at(x);
          if (st.eval(thx, iter == 0, false, gtrial)) return -1;    // first trial: gradient in the same evaluation (accepted most of the time)
          fx = st.negll;
          if (fx > fx_init + step * test_decr || (fx != fx)) {
            if (fx != fx && st.reset_mode()) return -1;             // NaN: ResetLaplaceApproxModeToPreviousValue (optim_utils.h:405-407)
            step *= ((fx - fx_init) > 2. * std::max(std::fabs(fx_init), 1.)) ? 0.5 / 16. : 0.5;
          } else {
            if (iter == 0) { grad[0] = gtrial[0]; grad[1] = gtrial[1]; grad_is_current = true; }
            break;
          }
        }
        if (iter >= max_linesearch) { x[0] = xp[0]; x[1] = xp[1]; fx = fx_init; step = 0.; }
      }
      at(x);
      if (!grad_is_current && st.grad_current(thx, grad)) return -1;          // gradient of the CURRENT state (last trial's mode)
      mask(grad);
      gnorm = std::sqrt(grad[0] * grad[0] + grad[1] * grad[1]);
      bool has_converged = gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(x[0] * x[0] + x[1] * x[1]);
      if ((fx_past - fx) <= delta * std::max(std::fabs(fx_past), 1.)) has_converged = true;
      if (cfg.max_iter != 0 && k >= cfg.max_iter) has_converged = true;
      if (cfg.trace)
        fprintf(stderr, "[gpboost_amd] lbfgs it %d: var %.10g a %.10g negll %.10g step %g\n", k, std::exp(x[0]), std::exp(x[1]), fx, step);
      if (has_converged) break;
      const double sv[2] = {x[0] - xp[0], x[1] - xp[1]}, yv[2] = {grad[0] - gradp[0], grad[1] - gradp[1]};
      if (sv[0] * yv[0] + sv[1] * yv[1] > eps * (yv[0] * yv[0] + yv[1] * yv[1])) bfgs.add_correction(sv, yv);
      step = 1.;
      bfgs.apply_Hv(grad, -1., drt);
      fx_past = fx;
      ++k;
    }
  }
  if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(fx))
    return kNaOrInf;                                               // the caller starts again with 'nelder_mead' (:1706-1731)
  th[0] = std::exp(x[0]); th[1] = std::exp(x[1]);
  st.negll = fx;
  out->num_it = k;
  return 0;
}

// the same simplex search on theta = (sigma1_2, a) of a non-Gaussian model: the objective is the Laplace approximation with its warm-started
// mode; a NaN / Inf value resets the mode to the one before the evaluation (EvalLLforOptimLib, optim_utils.h:196-200)
int run_nelder_mead_laplace(LapState& st, const GpbOptimConfig& cfg, double th[2], GpbLaplaceOptimResult* out, const Fail& fail) {
  const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-8;
  double x[2] = {std::log(th[0]), std::log(th[1])}, fx = 1e99;
  int n_ok = 0;
  auto f = [&](const double* xv, double* fv) -> int {
    const double thx[2] = {std::exp(xv[0]), std::exp(xv[1])};
    if (const int rc = st.eval(thx, false, false, nullptr)) {       // evaluator error at a trial vertex: +Inf there, the mode goes back (see run_nelder_mead)
      if (n_ok == 0 || rc == kCheckFailed) return rc;
      *fv = INFINITY;
      return st.reset_mode() ? -1 : 0;
    }
    ++n_ok;
    *fv = st.negll;
    if (!std::isfinite(*fv) && st.reset_mode()) return -1;
    return 0;
  };
  int num_it = 0;
  if (nelder_mead_2d(f, x, cfg, delta, &num_it, &fx)) return -1;
  if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(fx))
    return fail("NaN or Inf occurred in covariance parameter optimization using 'nelder_mead'; try other initial values");
  th[0] = std::exp(x[0]); th[1] = std::exp(x[1]);
  st.negll = fx;
  out->num_it = num_it;
  return 0;
}

}  // namespace

int gpb_optimize_gaussian_cov_pars(const GpbOptimConfig& cfg, int num_data, gpb_terms_fn fn, void* ctx, const double theta_init[3],
                                   GpbOptimResult* out, char* err, int errlen) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !out || !theta_init) return fail("gpb_optimize_gaussian_cov_pars: null argument");
  if (!(theta_init[0] > 0.) || !(theta_init[1] > 0.) || !(theta_init[2] > 0.))
    return fail("Initial covariance parameters need to be positive (found %g, %g, %g on the transformed scale)", theta_init[0], theta_init[1],
                theta_init[2]);
  State st{cfg, num_data, fn, ctx};
  st.fail_ = &fail;
  double th[3] = {theta_init[0], theta_init[1], theta_init[2]};
  std::copy(th, th + 3, st.th_first);
  st.sigma2 = th[0];
  *out = GpbOptimResult();
  int rc = 0;
  if (cfg.max_iter > 0) {
    if (cfg.optimizer == "gradient_descent") rc = run_gradient_descent(st, cfg, th, out, fail);
    else if (cfg.optimizer == "lbfgs") rc = run_lbfgs(st, cfg, th, out, fail);
    else if (cfg.optimizer == "nelder_mead") rc = run_nelder_mead(st, cfg, th, out, fail);
    else
      return fail("optimizer_cov = '%s' is not on the MI355X path of this library (supported: 'lbfgs', 'gradient_descent', 'nelder_mead')", cfg.optimizer.c_str());
    if (rc == kNaOrInf) {
      // "redo optimization with nelder_mead in case NA or Inf occurred" (re_model_template.h:1706-1731): from the initial values, with the
      // convergence tolerance the first optimiser had (delta_rel_conv_init_: 1e-6 unless given)
      fprintf(stderr, "[gpboost_amd] Warning: NaN or Inf occurred in covariance parameter optimization using '%s'. The optimization will be started a "
                      "second time using 'nelder_mead'. If you want to avoid this, try directly using a different optimizer. If you have used "
                      "'gradient_descent', you can also consider using a smaller learning rate \n", cfg.optimizer.c_str());
      GpbOptimConfig c2 = cfg;
      c2.optimizer = "nelder_mead";
      if (!(cfg.delta_rel_conv_init > 0.)) c2.delta_rel_conv_init = 1e-6;
      std::copy(theta_init, theta_init + 3, th);
      st.sigma2 = th[0];
      rc = run_nelder_mead(st, c2, th, out, fail, false);
    }
    if (rc) {
      if (!err[0]) snprintf(err, errlen, "likelihood evaluation failed during the optimisation");
      return -1;
    }
  }
  if (cfg.max_iter > 0) st.keep_variance_constant(th);                   // :1750-1755
  std::copy(th, th + 3, out->theta);
  out->negll = st.negll;
  out->num_ll_evals = st.n_ll;
  out->num_grad_evals = st.n_grad;
  return 0;
}

int gpb_optimize_laplace_cov_pars(const GpbOptimConfig& cfg, gpb_laplace_fn fn, void* ctx, const double theta_init[2],
                                  GpbLaplaceOptimResult* out, char* err, int errlen) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !out || !theta_init) return fail("gpb_optimize_laplace_cov_pars: null argument");
  if (!(theta_init[0] > 0.) || !(theta_init[1] > 0.))
    return fail("Initial covariance parameters need to be positive (found %g, %g on the transformed scale)", theta_init[0], theta_init[1]);
  LapState st{fn, ctx};
  st.fail_ = &fail;
  double th[2] = {theta_init[0], theta_init[1]};
  *out = GpbLaplaceOptimResult();
  if ((cfg.estimate_cov_par_index[0] <= 0 || cfg.estimate_cov_par_index[1] <= 0) && cfg.optimizer != "lbfgs" && cfg.max_iter > 0)
    return fail("holding covariance parameters fixed (estimate_cov_par_index) with optimizer_cov = '%s' for a non-Gaussian model is not on the MI355X path of this library (supported: 'lbfgs')", cfg.optimizer.c_str());
  if (cfg.max_iter > 0) {
    int rc;
    if (cfg.optimizer == "gradient_descent") rc = run_gradient_descent_laplace(st, cfg, th, out, fail);
    else if (cfg.optimizer == "lbfgs") rc = run_lbfgs_laplace(st, cfg, th, out, fail);
    else if (cfg.optimizer == "nelder_mead") rc = run_nelder_mead_laplace(st, cfg, th, out, fail);
    else return fail("optimizer_cov = '%s' is not on the MI355X path of this library (supported: 'lbfgs', 'gradient_descent', 'nelder_mead')", cfg.optimizer.c_str());
    if (rc == kNaOrInf) {                    // as above; the mode starts from zero again (InitializeModeAvec, :1722-1726)
      fprintf(stderr, "[gpboost_amd] Warning: NaN or Inf occurred in covariance parameter optimization using '%s'. The optimization will be started a "
                      "second time using 'nelder_mead'. \n", cfg.optimizer.c_str());
      GpbOptimConfig c2 = cfg;
      c2.optimizer = "nelder_mead";
      if (!(cfg.delta_rel_conv_init > 0.)) c2.delta_rel_conv_init = 1e-6;
      th[0] = theta_init[0]; th[1] = theta_init[1];
      double o3[3];
      if (fn(ctx, 4, 0., 0., o3)) return fail("the evaluator could not reset the mode");
      rc = run_nelder_mead_laplace(st, c2, th, out, fail);
    }
    if (rc) { if (!err[0]) snprintf(err, errlen, "likelihood evaluation failed during the optimisation"); return -1; }
  }
  out->theta[0] = th[0]; out->theta[1] = th[1];
  out->negll = st.negll;
  out->num_evals = st.n_evals;
  return 0;
}

int gpb_laplace_std_errors(gpb_laplace_fn fn, void* ctx, const double theta[2], double range_const, double se_out[2], char* err, int errlen,
                           const int* estimated2) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !theta || !se_out) return fail("gpb_laplace_std_errors: null argument");
  if (!(theta[0] > 0.) || !(theta[1] > 0.)) return fail("Covariance parameters need to be positive (found %g, %g on the transformed scale)", theta[0], theta[1]);
  const double h_eps = 1e-4;                                                  // :11047
  double delta[2], H[2][2];
  for (int i = 0; i < 2; ++i) { delta[i] = std::fabs(std::log(theta[i])) * h_eps; if (delta[i] < h_eps) delta[i] = h_eps; }   // :10939-10944
  for (int i = 0; i < 2; ++i) {
    double g[2][2];
    for (int s = 0; s < 2; ++s) {                                             // theta_i e^{+delta}, then theta_i e^{-delta} (:10949-10963)
      double t[2] = {theta[0], theta[1]}, o3[3];
      t[i] *= std::exp(s == 0 ? delta[i] : -delta[i]);
      if (fn(ctx, 1, t[0], t[1], o3)) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed while calculating standard deviations"); return -1; }
      g[s][0] = o3[1]; g[s][1] = o3[2];
    }
    for (int j = 0; j < 2; ++j) H[i][j] = (g[0][j] - g[1][j]) / (2. * delta[i]);
  }
  { double o3[3]; if (fn(ctx, 0, theta[0], theta[1], o3)) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed while calculating standard deviations"); return -1; } }   // :11051
  const double h01 = 0.5 * (H[0][1] + H[1][0]);
  const double det = H[0][0] * H[1][1] - h01 * h01;
  const double nan_value = std::numeric_limits<double>::quiet_NaN();
  se_out[0] = se_out[1] = nan_value;
  const bool e0 = !estimated2 || estimated2[0] > 0, e1 = !estimated2 || estimated2[1] > 0;
  if (!e0 || !e1) {            // parameters held fixed have no standard error (NaN); the Hessian of the estimated ones alone is inverted (:11052-11079)
    if (e0 && H[0][0] > 0.) se_out[0] = theta[0] * std::sqrt(1. / H[0][0]);
    if (e1 && H[1][1] > 0.) se_out[1] = (range_const / theta[1]) * std::sqrt(1. / H[1][1]);
    return 0;
  }
  if (H[0][0] > 0. && det > 0. && std::isfinite(det)) {                       // LLT succeeds iff positive definite
    const double inv00 = H[1][1] / det, inv11 = H[0][0] / det;
    se_out[0] = theta[0] * std::sqrt(inv00);
    se_out[1] = (range_const / theta[1]) * std::sqrt(inv11);
  } else {
    fprintf(stderr, "[gpboost_amd] Warning: Cannot calculate standard deviations for covariance / auxiliary parameters since the approximated Hessian is not positive definite \n");
  }
  return 0;
}

// The same simplex search for n >= 2 parameters (round 6: covariance + auxiliary parameters of a non-Gaussian likelihood; nm.hpp:95-372 with adaptive_pars: reflection 1,
// contraction 0.75 - 1 / (2 n), expansion 1 + 2 / n, shrinkage 1 - 1 / n; start simplex x + 0.05 x_i e_i, 0.00025 where x_i = 0).  Same order of evaluations, sorting,
// centroid of the n best vertices and relative-change measures as nelder_mead_2d.
template <class F>
int nelder_mead_nd(F&& f, int n, double* x, const GpbOptimConfig& cfg, double delta, int* num_it, double* fbest) {
  const double par_alpha = 1.0, par_beta = 0.75 - 1.0 / (2.0 * n), par_gamma = 1.0 + 2.0 / n, par_delta = 1.0 - 1.0 / n;
  double tol_f, tol_x;
  if (cfg.convergence_criterion == "relative_change_in_parameters") { tol_x = delta; tol_f = 1e-20; }
  else { tol_f = delta; tol_x = 1e-20; }
  const size_t iter_max = (size_t)cfg.max_iter;
  std::vector<double> fv(n + 1), fv_old(n + 1), pt((size_t)(n + 1) * n), pt_old, cen(n), xr(n), xt(n);
  auto P = [&](int i) { return pt.data() + (size_t)i * n; };
  if (f(x, &fv[0])) return -1;
  std::copy(x, x + n, P(0));
  for (int i = 1; i < n + 1; ++i) {
    for (int c = 0; c < n; ++c) P(i)[c] = x[c] + (x[i - 1] != 0.0 ? 0.05 * x[i - 1] : 0.00025) * (c == i - 1 ? 1.0 : 0.0);
    if (f(P(i), &fv[i])) return -1;
  }
  size_t iter = 0;
  double rel_f = 2 * std::fabs(tol_f), rel_x = 2 * std::fabs(tol_x);
  fv_old = fv; pt_old = pt;
  bool has_converged = false;
  std::vector<size_t> idx(n + 1);
  while (!has_converged) {
    ++iter;
    bool next_iter = false;
    for (int i = 0; i < n + 1; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return fv[a] < fv[b]; });
    {
      std::vector<double> fs(n + 1), ps((size_t)(n + 1) * n);
      for (int i = 0; i < n + 1; ++i) { fs[i] = fv[idx[i]]; std::copy(P((int)idx[i]), P((int)idx[i]) + n, ps.data() + (size_t)i * n); }
      fv = fs; pt = ps;
    }
    double ft, fr;
    for (int c = 0; c < n; ++c) { double acc = 0.; for (int i = 0; i < n; ++i) acc += P(i)[c]; cen[c] = acc / (double)n; }
    for (int c = 0; c < n; ++c) xr[c] = cen[c] + par_alpha * (cen[c] - P(n)[c]);
    if (f(xr.data(), &fr)) return -1;
    if (fr >= fv[0] && fr < fv[n - 1]) { std::copy(xr.begin(), xr.end(), P(n)); fv[n] = fr; next_iter = true; }
    if (!next_iter && fr < fv[0]) {                                   // expansion
      for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_gamma * (xr[c] - cen[c]);
      if (f(xt.data(), &ft)) return -1;
      if (ft < fr) { std::copy(xt.begin(), xt.end(), P(n)); fv[n] = ft; } else { std::copy(xr.begin(), xr.end(), P(n)); fv[n] = fr; }
      next_iter = true;
    }
    if (!next_iter && fr >= fv[n - 1]) {
      if (fr < fv[n]) {                                               // outside contraction
        for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_beta * (xr[c] - cen[c]);
        if (f(xt.data(), &ft)) return -1;
        if (ft <= fr) { std::copy(xt.begin(), xt.end(), P(n)); fv[n] = ft; next_iter = true; }
      } else {                                                        // inside contraction
        for (int c = 0; c < n; ++c) xt[c] = cen[c] + par_beta * (P(n)[c] - cen[c]);
        if (f(xt.data(), &ft)) return -1;
        if (ft < fv[n]) { std::copy(xt.begin(), xt.end(), P(n)); fv[n] = ft; next_iter = true; }
      }
    }
    if (!next_iter) {                                                 // shrink towards the best vertex
      for (int i = 1; i < n + 1; ++i) for (int c = 0; c < n; ++c) P(i)[c] = P(0)[c] + par_delta * (P(i)[c] - P(0)[c]);
      for (int i = 1; i < n + 1; ++i) if (f(P(i), &fv[i])) return -1;
    }
    double num = 0., den = 0.;
    for (int i = 0; i < n + 1; ++i) { num = std::max(num, std::fabs(fv[i] - fv_old[i])); den = std::max(den, std::fabs(fv_old[i])); }
    rel_f = num / (1.0e-08 + den);
    fv_old = fv;
    if (tol_x >= 0.0) {
      num = 0.; den = 0.;
      for (size_t e = 0; e < pt.size(); ++e) { num = std::max(num, std::fabs(pt[e] - pt_old[e])); den = std::max(den, std::fabs(pt_old[e])); }
      rel_x = num / (1.0e-08 + den);
      pt_old = pt;
    }
    has_converged = !(rel_f > tol_f && rel_x > tol_x && iter < iter_max);
  }
  int best = 0;
  for (int i = 1; i < n + 1; ++i) if (fv[i] < fv[best]) best = i;
  std::copy(P(best), P(best) + n, x);
  if (f(x, fbest)) return -1;                                         // settings->opt_fn_value = opt_objfn(x_p) (error_reporting)
  *num_it = (int)iter;
  return 0;
}

// Standard errors of covariance AND auxiliary parameters of a non-Gaussian model whose auxiliary parameters are estimated (round 6):
// CalcStdDevCovParAuxParsNonGaussian (include/GPBoost/re_model_template.h:11029-11117) over the vector (sigma1_2, a, aux_1 .. aux_naux) -- the Hessian wrt the logs as the
// numerical Jacobian of the gradient (CalcHessianCovParAuxPars :10915-10968: central differences, step |log p| 1e-4, at least 1e-4; symmetrised), the sub-matrix of the
// ESTIMATED parameters (covariance parameters by estimate_cov_par_index, an auxiliary parameter iff its diagonal entry is > 0: the df of "t_fix_df" has a zero
// gradient, :11059-11067) inverted by Cholesky, delta method back to the original scale with the reference's finite-difference derivative (:10977-11027: p sinh(h) / h,
// h = eps^(1/3)).  The evaluator's state is restored at (theta, aux) afterwards (:11048-11051).  NaN where no standard error exists.
int gpb_laplace_aux_std_errors(gpb_laplace_aux_fn fn, void* ctx, const double theta[2], const double* aux, int naux, double range_const, double se_cov[2],
                               double* se_aux, char* err, int errlen, const int* estimated2) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !theta || !aux || !se_cov || !se_aux || naux < 1 || naux > 6) return fail("gpb_laplace_aux_std_errors: invalid argument");
  const int N = 2 + naux;
  std::vector<double> p0(N), H((size_t)N * N, 0.), delta(N), out(3 + naux), g1(N), g2(N);
  p0[0] = theta[0]; p0[1] = theta[1];
  for (int j = 0; j < naux; ++j) p0[2 + j] = aux[j];
  for (int i = 0; i < N; ++i) if (!(p0[i] > 0.)) return fail("Parameters need to be positive for their standard deviations (found %g)", p0[i]);
  const double h_eps = 1e-4;                                                  // :11047
  for (int i = 0; i < N; ++i) { delta[i] = std::fabs(std::log(p0[i])) * h_eps; if (delta[i] < h_eps) delta[i] = h_eps; }
  auto eval_grad = [&](const std::vector<double>& p, std::vector<double>& g) -> int {
    if (fn(ctx, 1, p[0], p[1], p.data() + 2, naux, out.data())) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed while calculating standard deviations"); return -1; }
    for (int j = 0; j < N; ++j) g[j] = out[1 + j];
    return 0;
  };
  for (int i = 0; i < N; ++i) {
    std::vector<double> pa = p0, pb = p0;
    pa[i] *= std::exp(delta[i]); pb[i] *= std::exp(-delta[i]);
    if (eval_grad(pa, g1) || eval_grad(pb, g2)) return -1;
    for (int j = 0; j < N; ++j) H[(size_t)i * N + j] = (g1[j] - g2[j]) / (2. * delta[i]);
  }
  if (fn(ctx, 0, p0[0], p0[1], p0.data() + 2, naux, out.data())) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed while calculating standard deviations"); return -1; }
  for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) { const double v = 0.5 * (H[(size_t)i * N + j] + H[(size_t)j * N + i]); H[(size_t)i * N + j] = H[(size_t)j * N + i] = v; }
  std::vector<int> est;
  for (int i = 0; i < 2; ++i) if (!estimated2 || estimated2[i] > 0) est.push_back(i);
  for (int j = 0; j < naux; ++j) if (H[(size_t)(2 + j) * N + 2 + j] > 0.) est.push_back(2 + j);
  const double nan_value = std::numeric_limits<double>::quiet_NaN();
  std::vector<double> inv_diag(N, nan_value);
  const int E = (int)est.size();
  if (E > 0) {
    std::vector<double> L((size_t)E * E, 0.);
    bool pd = true;
    for (int i = 0; i < E && pd; ++i)
      for (int j = 0; j <= i; ++j) {
        double acc = H[(size_t)est[i] * N + est[j]];
        for (int k = 0; k < j; ++k) acc -= L[(size_t)i * E + k] * L[(size_t)j * E + k];
        if (i == j) { if (!(acc > 0.) || !std::isfinite(acc)) { pd = false; break; } L[(size_t)i * E + i] = std::sqrt(acc); }
        else L[(size_t)i * E + j] = acc / L[(size_t)j * E + j];
      }
    if (pd) {
      std::vector<double> col(E);
      for (int c = 0; c < E; ++c) {            // (H^-1)_cc = || L^-1 e_c ||^2
        double ss = 0.;
        for (int i = 0; i < E; ++i) {
          double acc = i == c ? 1. : 0.;
          for (int k = 0; k < i; ++k) acc -= L[(size_t)i * E + k] * col[k];
          col[i] = i < c ? 0. : acc / L[(size_t)i * E + i];
          ss += col[i] * col[i];
        }
        inv_diag[est[c]] = ss;
      }
    } else {
      fprintf(stderr, "[gpboost_amd] Warning: Cannot calculate standard deviations for covariance / auxiliary parameters since the approximated Hessian is not positive definite \n");
    }
  }
  const double h = std::pow(std::numeric_limits<double>::epsilon(), 1.0 / 3.0);
  const double fd = (std::exp(h) - std::exp(-h)) / (2. * h);       // d p / d log p by the reference's central difference
  const double orig[2] = {theta[0], range_const / theta[1]};       // sigma1_2 itself; rho = c / a: |d rho / d log a| = rho
  for (int i = 0; i < 2; ++i) se_cov[i] = (std::isfinite(inv_diag[i]) && inv_diag[i] >= 0.) ? std::fabs(orig[i] * fd) * std::sqrt(inv_diag[i]) : nan_value;
  for (int j = 0; j < naux; ++j) se_aux[j] = (std::isfinite(inv_diag[2 + j]) && inv_diag[2 + j] >= 0.) ? std::fabs(aux[j] * fd) * std::sqrt(inv_diag[2 + j]) : nan_value;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Non-Gaussian likelihoods with a linear predictor: lbfgs on (log sigma1_2, log a, beta).  The same solver as run_lbfgs_laplace (LBFGSSolver::minimize,
// LineSearchBacktracking, BFGSMat of GPBoost's LBFGSpp copy) for a vector of length 2 + p; GetMaximalLearningRate (optim_utils.h:498-535) caps the step
// with MaximalLearningRateCovAuxPars on the covariance part and MaximalLearningRateCoef (re_model_template.h:5428-5464) on the coefficients.
namespace {
struct BfgsMatN {     // BFGSMat.h:69-186
  int n, m; double theta = 1.; int ncorr = 0, ptr;
  std::vector<double> s, y, ys, alpha;   // s, y: column j at [n j, n (j + 1))
  BfgsMatN(int n_, int m_) : n(n_), m(m_), ptr(m_), s((size_t)n_ * m_), y((size_t)n_ * m_), ys(m_), alpha(m_) {}
  void add_correction(const double* sv, const double* yv) {
    const int loc = ptr % m;
    double d = 0., yy = 0.;
    for (int i = 0; i < n; ++i) { s[(size_t)n * loc + i] = sv[i]; y[(size_t)n * loc + i] = yv[i]; d += sv[i] * yv[i]; yy += yv[i] * yv[i]; }
    ys[loc] = d;
    theta = yy / d;
    if (ncorr < m) ++ncorr;
    ptr = loc + 1;
  }
  void apply_Hv(const double* v, double a, double* res) {
    for (int i = 0; i < n; ++i) res[i] = a * v[i];
    int j = ptr % m;
    for (int c = 0; c < ncorr; ++c) {
      j = (j + m - 1) % m;
      double sr = 0.;
      for (int i = 0; i < n; ++i) sr += s[(size_t)n * j + i] * res[i];
      alpha[j] = sr / ys[j];
      for (int i = 0; i < n; ++i) res[i] -= alpha[j] * y[(size_t)n * j + i];
    }
    for (int i = 0; i < n; ++i) res[i] /= theta;
    for (int c = 0; c < ncorr; ++c) {
      double yr = 0.;
      for (int i = 0; i < n; ++i) yr += y[(size_t)n * j + i] * res[i];
      const double beta = yr / ys[j];
      for (int i = 0; i < n; ++i) res[i] += (alpha[j] - beta) * s[(size_t)n * j + i];
      j = (j + 1) % m;
    }
  }
};

struct LapCoefState {
  gpb_laplace_fe_fn fn; void* ctx;
  int n, p; const double* X; const double* offset; double C_mu, C_sigma2;
  std::vector<double> fe, gF;
  int n_evals = 0;
  double negll = 0.;
  int nc = 2;                       // covariance entries at the head of the lbfgs vector: 2 = (log var, log a), 0 = held at th_fixed
  double th_fixed[2] = {1., 1.};
  // likelihoods with auxiliary parameters (no covariates: p = 0): naux entries log(aux) at the TAIL of the vector (optim_utils.h:256-283), evaluator afn
  gpb_laplace_aux_fn afn = nullptr; int naux = 0;
  int eval_aux(const double* x, int op, double* grad) {
    double auxv[8], o[3 + 8] = {0};
    for (int j = 0; j < naux; ++j) auxv[j] = std::exp(x[nc + p + j]);
    if (afn(ctx, op, var_of(x), a_of(x), auxv, naux, o)) return -1;
    op &= 15;                                                        // (bit 4: first update of a gradient-descent fit, see device_laplace)
    if (op != 2) { ++n_evals; negll = o[0]; }
    if (op >= 1 && grad) { if (nc) { grad[0] = o[1]; grad[1] = o[2]; } for (int j = 0; j < naux; ++j) grad[nc + p + j] = o[3 + j]; }
    return 0;
  }
  double var_of(const double* x) const { return nc ? std::exp(x[0]) : th_fixed[0]; }
  double a_of(const double* x) const { return nc ? std::exp(x[1]) : th_fixed[1]; }
  void linear_predictor(const double* beta) {                        // UpdateFixedEffects: fixed_effects + X beta (non-Gaussian)
    for (int i = 0; i < n; ++i) fe[i] = offset ? offset[i] : 0.;
    for (int j = 0; j < p; ++j) { const double b = beta[j]; const double* col = X + (size_t)j * n; for (int i = 0; i < n; ++i) fe[i] += col[i] * b; }
  }
  void grad_beta(double* g) const {                                  // X' grad_F (re_model_template.h:2157-2160)
    for (int j = 0; j < p; ++j) { const double* col = X + (size_t)j * n; double acc = 0.; for (int i = 0; i < n; ++i) acc += col[i] * gF[i]; g[j] = acc; }
  }
  // x = (log var, log a, beta) [nc = 2] or (beta) [nc = 0]; grad (nc + p) filled if with_grad
  int eval(const double* x, bool with_grad, double* grad) {
    if (afn) return eval_aux(x, with_grad ? 1 : 0, grad);
    linear_predictor(x + nc);
    double o[3] = {0, 0, 0};
    if (fn(ctx, with_grad ? 1 : 0, var_of(x), a_of(x), fe.data(), o, gF.data())) return -1;
    ++n_evals;
    negll = o[0];
    if (with_grad) { if (nc) { grad[0] = o[1]; grad[1] = o[2]; } grad_beta(grad + nc); }
    return 0;
  }
  int grad_current(const double* x, double* grad) {
    if (afn) return eval_aux(x, 2, grad);
    double o[3] = {0, 0, 0};
    if (fn(ctx, 2, var_of(x), a_of(x), fe.data(), o, gF.data())) return -1;
    if (nc) { grad[0] = o[1]; grad[1] = o[2]; }
    grad_beta(grad + nc);
    return 0;
  }
  int reset_mode() { double o[3 + 8]; if (afn) return afn(ctx, 3, 0., 0., nullptr, naux, o); return fn(ctx, 3, 0., 0., nullptr, o, nullptr); }
  // MaximalLearningRateCoef (re_model_template.h:5428-5464): the step along neg_step_dir may move the mean of the linear predictor by at most
  // C_mu C_MAX_CHANGE_COEF and its variance by at most C_sigma2 C_MAX_CHANGE_COEF
  double max_lr_coef(const double* beta, const double* dir) const {
    if (p == 0) return 1e99;                                         // no coefficients in the vector
    const double C_MAX_CHANGE_COEF = 10.;                            // :5795
    double m_ch = 0., m_l1 = 0., v_ch = 0., c_ch = 0.;
    for (int i = 0; i < n; ++i) {
      double ch = 0., l1 = 0.;
      for (int j = 0; j < p; ++j) { const double x = X[(size_t)j * n + i]; ch += x * dir[j]; l1 += x * beta[j]; }
      m_ch += ch; m_l1 += l1; v_ch += ch * ch; c_ch += ch * l1;
    }
    m_ch /= n; m_l1 /= n; v_ch /= n; c_ch /= n;
    v_ch -= m_ch * m_ch;
    c_ch -= m_ch * m_l1;
    const double max_lr_mu = C_mu * C_MAX_CHANGE_COEF / std::fabs(m_ch);
    const double max_lr_var = (std::fabs(c_ch) + std::sqrt(c_ch * c_ch + 4. * v_ch * C_sigma2 * C_MAX_CHANGE_COEF)) / 2. / v_ch;
    return std::min(max_lr_mu, max_lr_var);
  }
};

int run_lbfgs_laplace_coef(LapCoefState& st, const GpbOptimConfig& cfg, std::vector<double>& x, int* num_it, const Fail& fail) {
  const int N = (int)x.size();
  const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const double initial_step_factor = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 1.;
  const double epsilon = 1e-20, epsilon_rel = 1e-20, ftol = 1e-4;
  const int max_linesearch = 20;
  constexpr double eps = std::numeric_limits<double>::epsilon();
  BfgsMatN bfgs(N, cfg.m_lbfgs);
  std::vector<double> xp(N), grad(N), gradp(N), drt(N), gtrial(N), sv(N), yv(N), ndir(N);
  auto norm = [N](const std::vector<double>& v) { double a = 0.; for (int i = 0; i < N; ++i) a += v[i] * v[i]; return std::sqrt(a); };
  auto dot = [N](const std::vector<double>& a, const std::vector<double>& b) { double r = 0.; for (int i = 0; i < N; ++i) r += a[i] * b[i]; return r; };
  const bool est0 = cfg.estimate_cov_par_index[0] > 0, est1 = cfg.estimate_cov_par_index[1] > 0;
  auto mask = [&](std::vector<double>& g) { if (st.nc) { if (!est0) g[0] = 0.; if (!est1) g[1] = 0.; } };      // estimate_cov_par_index (likelihoods.h:6622)
  if (st.eval(x.data(), true, grad.data())) return -1;
  mask(grad);
  double fx = st.negll;
  if (!std::isfinite(fx))
    return fail("%s occurred in initial approximate negative marginal log-likelihood. Possible solutions: try other initial values ('init_cov_pars' and 'init_coef')",
                std::isnan(fx) ? "NaN" : "Inf");
  double gnorm = norm(grad);
  double fx_past = fx;
  int k = 1;
  if (!(gnorm <= epsilon || gnorm <= epsilon_rel * norm(x))) {
    for (int i = 0; i < N; ++i) drt[i] = -grad[i];
    double step = initial_step_factor / norm(drt);
    for (;;) {
      xp = x; gradp = grad;
      // GetMaximalLearningRate (optim_utils.h:498-535)
      const int nc = st.nc;
      double max_lr = 1e99;
      {       // MaximalLearningRateCovAuxPars over the covariance AND the auxiliary entries (optim_utils.h:515-523)
        double mx = 0.;
        if (nc) mx = std::max(std::fabs(drt[0]), std::fabs(drt[1]));
        for (int j = 0; j < st.naux; ++j) mx = std::max(mx, std::fabs(drt[nc + st.p + j]));
        if (nc || st.naux) max_lr = kMaxGradientUpdateLogScale / mx;
      }
      for (int i = 0; i < N; ++i) ndir[i] = -drt[i];
      const double max_lr_beta = st.max_lr_coef(x.data() + nc, ndir.data() + nc);
      if (max_lr_beta < max_lr) max_lr = max_lr_beta;
      if (max_lr < step) step = max_lr;
      bool grad_is_current = false;
      {
        if (step <= 0.) return fail("GPModel lbfgs: 'step' must be positive");
        const double fx_init = fx, dg_init = dot(grad, drt);
        if (dg_init > 0.) return fail("GPModel lbfgs: the moving direction increases the objective function value");
        const double test_decr = ftol * dg_init;
        int iter;
        for (iter = 0; iter < max_linesearch; ++iter) {
          for (int i = 0; i < N; ++i) x[i] = xp[i] + step * drt[i];
          if (st.eval(x.data(), iter == 0, gtrial.data())) return -1;
          fx = st.negll;
          if (fx > fx_init + step * test_decr || (fx != fx)) {
            if (fx != fx && st.reset_mode()) return -1;
            step *= ((fx - fx_init) > 2. * std::max(std::fabs(fx_init), 1.)) ? 0.5 / 16. : 0.5;
          } else {
            if (iter == 0) { grad = gtrial; grad_is_current = true; }
            break;
          }
        }
        if (iter >= max_linesearch) { x = xp; fx = fx_init; step = 0.; if (!st.afn) st.linear_predictor(x.data() + st.nc); }
      }
      if (!grad_is_current && st.grad_current(x.data(), grad.data())) return -1;
      mask(grad);
      gnorm = norm(grad);
      bool has_converged = gnorm <= epsilon || gnorm <= epsilon_rel * norm(x);
      if ((fx_past - fx) <= delta * std::max(std::fabs(fx_past), 1.)) has_converged = true;
      if (cfg.max_iter != 0 && k >= cfg.max_iter) has_converged = true;
      if (cfg.trace) {
        fprintf(stderr, "[gpboost_amd] lbfgs it %d: var %.10g a %.10g coef", k, st.var_of(x.data()), st.a_of(x.data()));
        for (int i = st.nc; i < N; ++i) fprintf(stderr, " %.10g", x[i]);
        fprintf(stderr, " negll %.10g step %g\n", fx, step);
      }
      if (has_converged) break;
      for (int i = 0; i < N; ++i) { sv[i] = x[i] - xp[i]; yv[i] = grad[i] - gradp[i]; }
      if (dot(sv, yv) > eps * dot(yv, yv)) bfgs.add_correction(sv.data(), yv.data());
      step = 1.;
      bfgs.apply_Hv(grad.data(), -1., drt.data());
      fx_past = fx;
      ++k;
    }
  }
  for (int i = 0; i < N; ++i) if (!std::isfinite(x[i])) return kNaOrInf;
  if (!std::isfinite(fx)) return kNaOrInf;
  st.negll = fx;
  *num_it = k;
  return 0;
}
}  // namespace

int gpb_optimize_laplace_coef_cov_pars(const GpbOptimConfig& cfg, gpb_laplace_fe_fn fn, void* ctx, int n, int p, const double* X_scaled,
                                       const double* offset, double C_mu, double C_sigma2, const double theta_init[2], double* beta,
                                       GpbLaplaceCoefResult* out, char* err, int errlen, bool learn_cov) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !out || !theta_init || !beta || !X_scaled || n < 1 || p < 1) return fail("gpb_optimize_laplace_coef_cov_pars: invalid argument");
  if (!(theta_init[0] > 0.) || !(theta_init[1] > 0.))
    return fail("Initial covariance parameters need to be positive (found %g, %g on the transformed scale)", theta_init[0], theta_init[1]);
  if (cfg.optimizer != "lbfgs")
    return fail("optimizer_cov = '%s' for a non-Gaussian model with a linear predictor is not on the MI355X path of this library (supported: 'lbfgs', the reference's default)", cfg.optimizer.c_str());
  LapCoefState st{fn, ctx, n, p, X_scaled, offset, C_mu, C_sigma2, std::vector<double>(n), std::vector<double>(n)};
  st.nc = learn_cov ? 2 : 0;
  st.th_fixed[0] = theta_init[0]; st.th_fixed[1] = theta_init[1];
  const int nc = st.nc;
  std::vector<double> x(nc + p);
  if (nc) { x[0] = std::log(theta_init[0]); x[1] = std::log(theta_init[1]); }
  for (int j = 0; j < p; ++j) x[nc + j] = beta[j];
  *out = GpbLaplaceCoefResult();
  if (cfg.max_iter > 0) {
    const int rc = run_lbfgs_laplace_coef(st, cfg, x, &out->num_it, fail);
    if (rc == kNaOrInf) return fail("NaN or Inf occurred in the parameter optimisation of a non-Gaussian model with a linear predictor (the reference restarts with 'nelder_mead', which is not on this path for such models)");
    if (rc) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed during the optimisation"); return -1; }
  }
  out->theta[0] = st.var_of(x.data()); out->theta[1] = st.a_of(x.data());
  for (int j = 0; j < p; ++j) beta[j] = x[nc + j];
  out->negll = st.negll;
  out->num_evals = st.n_evals;
  return 0;
}

// optimizer_cov = "gradient_descent" with estimated auxiliary parameters (round 6): the reference's internal loop (OptimLinRegrCoefCovPar, re_model_template.h:1514-1660) on
// x = (log sigma1_2, log a, log aux_1 ..) with TWO learning rates -- lr_cov_ for the covariance block, lr_aux_pars_ (same initial value, SetInitialValueLRCov :8316-8333) for
// the auxiliary block: each capped by MAX_GRADIENT_UPDATE_LOG_SCALE_ / max |gradient of its block| (AvoidTooLargeLearningRatesCovAuxPars :8354-8375), each with its own
// directional derivative -|g_block|^2 and momentum term (CalcDirDerivArmijoAndLearningRateConstChangeCovAuxPars :8420-8470); a trial step is accepted when BOTH Armijo
// conditions hold (UpdateCovAuxPars :8768-8776), otherwise both rates are halved and the mode goes back (:8793-8809); the halved rates stay (:8819-8824).  A rate that fell
// below 1e-4 of its value after the first iteration while its block stopped moving is set back once the OTHER block has moved by more than 1 % (:1564-1612), with a mode
// finding at the current parameters (RecalculateModeLaplaceApprox).
int run_gradient_descent_laplace_aux(LapCoefState& st, const GpbOptimConfig& cfg, std::vector<double>& x, int* num_it_out, const Fail& fail) {
  const int N = (int)x.size(), NC = 2, NA = N - 2;
  double lr_cov = cfg.lr_cov_init > 0. ? cfg.lr_cov_init : 0.1, lr_aux = lr_cov;
  const double delta_rel_conv = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-6;
  const bool nesterov = cfg.use_nesterov_acc;
  if (nesterov && cfg.nesterov_schedule_version == 1)
    return fail("Armijo condition backtracking is not implemented when nesterov_schedule_version = 1 ");
  const double kLrIsSmallRelChange = 1e-4, kMinRelChangeOther = 1e-2;     // LR_IS_SMALL_REL_CHANGE_IN_PARS_THRESHOLD_, MIN_REL_CHANGE_IN_OTHER_PARS_FOR_RESETTING_LR_ (re_model_template.h:5808-5812)
  std::vector<double> grad(N), gnew(N), xa(x), xa_lag1(x), x_lag1(N), x_new(N);
  if (st.eval_aux(x.data(), 1, grad.data())) return -1;
  if (!std::isfinite(st.negll))
    return fail("%s occurred in initial approximate negative marginal log-likelihood. Possible solutions: try other initial values ('init_cov_pars')",
                std::isnan(st.negll) ? "NaN" : "Inf");
  auto blk_norm = [&](const std::vector<double>& a, const std::vector<double>* b, int i0, int i1) {      // norms on the ORIGINAL scale, as the reference's cov_aux_pars
    double s2 = 0.; for (int i = i0; i < i1; ++i) { const double d = std::exp(a[i]) - (b ? std::exp((*b)[i]) : 0.); s2 += d * d; } return std::sqrt(s2);
  };
  int num_it = cfg.max_iter;
  bool have_grad = true;
  double lr_cov_after_first = lr_cov, lr_aux_after_first = lr_aux, thr_cov = 0., thr_aux = 0.;
  bool lr_cov_is_small = false, lr_aux_is_small = false;
  std::vector<double> aux_before_lr_cov_small, cov_before_lr_aux_small;
  for (int it = 0; it < cfg.max_iter; ++it) {
    const double negll_lag1 = st.negll;
    x_lag1 = x;
    if (!have_grad && st.eval_aux(x.data(), 2, grad.data())) return -1;
    double gmax_c = 0., gmax_a = 0., dd_c = 0., dd_a = 0.;
    for (int i = 0; i < NC; ++i) { gmax_c = std::max(gmax_c, std::fabs(grad[i])); dd_c -= grad[i] * grad[i]; }
    for (int i = NC; i < N; ++i) { gmax_a = std::max(gmax_a, std::fabs(grad[i])); dd_a -= grad[i] * grad[i]; }
    if (lr_cov > kMaxGradientUpdateLogScale / gmax_c) lr_cov = kMaxGradientUpdateLogScale / gmax_c;
    if (lr_aux > kMaxGradientUpdateLogScale / gmax_a) lr_aux = kMaxGradientUpdateLogScale / gmax_a;
    double mom_c = 0., mom_a = 0.;
    if (nesterov) {
      for (int i = 0; i < NC; ++i) mom_c += grad[i] * (x[i] - xa[i]);
      for (int i = NC; i < N; ++i) mom_a += grad[i] * (x[i] - xa[i]);
    }
    double lc = lr_cov, la = lr_aux, acc = cfg.acc_rate_cov;
    bool decrease_found = false, halving_done = false, new_grad = false;
    for (int ih = 0; ih < kMaxNumberLrShrinkageSteps; ++ih) {
      for (int i = 0; i < N; ++i) x_new[i] = x[i] - (i < NC ? lc : la) * grad[i];
      double mu = 0.;
      if (nesterov) {
        xa = x_new;
        mu = nesterov_schedule(it, cfg.nesterov_schedule_version, acc, cfg.momentum_offset);
        for (int i = 0; i < N; ++i) x_new[i] = (mu + 1.) * xa[i] - mu * xa_lag1[i];
      }
      new_grad = ih == 0;
      if (st.eval_aux(x_new.data(), (new_grad ? 1 : 0) | (it == 0 ? 16 : 0), gnew.data())) return -1;
      if (st.negll <= negll_lag1 + kCArmijo * lc * dd_c + kCArmijoMom * mu * mom_c &&
          st.negll <= negll_lag1 + kCArmijo * la * dd_a + kCArmijoMom * mu * mom_a) decrease_found = true;
      if (decrease_found) break;
      halving_done = true;
      lc *= kLrShrinkageFactor; la *= kLrShrinkageFactor;
      acc *= 0.5;
      if (st.reset_mode()) return -1;
    }
    if (halving_done) { lr_cov = lc; lr_aux = la; }
    if (nesterov) xa_lag1 = xa;
    x = x_new;
    have_grad = decrease_found && new_grad;
    if (have_grad) grad = gnew;
    if (it == 0) { lr_cov_after_first = lr_cov; thr_cov = lr_cov / 1e4; lr_aux_after_first = lr_aux; thr_aux = lr_aux / 1e4; }
    // a very small rate whose block has stopped moving is remembered ... (:1564-1590)
    if (lr_cov < thr_cov && !lr_cov_is_small && blk_norm(x, &x_lag1, 0, NC) < kLrIsSmallRelChange * blk_norm(x_lag1, nullptr, 0, NC)) {
      lr_cov_is_small = true; aux_before_lr_cov_small.assign(x.begin() + NC, x.end());
    }
    if (lr_aux < thr_aux && !lr_cov_is_small && !lr_aux_is_small && blk_norm(x, &x_lag1, NC, N) < kLrIsSmallRelChange * blk_norm(x_lag1, nullptr, NC, N)) {
      lr_aux_is_small = true; cov_before_lr_aux_small.assign(x.begin(), x.begin() + NC);
    }
    // ... and set back when the other block has moved on (:1601-1622); the mode is found again at the current parameters
    bool recalculated = false;
    auto moved = [&](const std::vector<double>& ref, int i0) {
      double d2 = 0., r2 = 0.;
      for (size_t q = 0; q < ref.size(); ++q) { const double a0 = std::exp(ref[q]), a1 = std::exp(x[i0 + q]); d2 += (a1 - a0) * (a1 - a0); r2 += a0 * a0; }
      return std::sqrt(d2) > kMinRelChangeOther * std::sqrt(r2);
    };
    if (lr_aux_is_small && moved(cov_before_lr_aux_small, 0)) {
      lr_aux = lr_aux_after_first; lr_aux_is_small = false;
      if (st.eval_aux(x.data(), 0, nullptr)) return -1;
      recalculated = true; have_grad = false;
    }
    if (lr_cov_is_small && moved(aux_before_lr_cov_small, NC)) {
      lr_cov = lr_cov_after_first; lr_cov_is_small = false;
      if (!recalculated) { if (st.eval_aux(x.data(), 0, nullptr)) return -1; have_grad = false; }
    }
    if (cfg.trace) {
      fprintf(stderr, "[gpboost_amd] it %d: pars (log)", it + 1);
      for (int i = 0; i < N; ++i) fprintf(stderr, " %.10g", x[i]);
      fprintf(stderr, " negll %.10g lr_cov %g lr_aux %g\n", st.negll, lr_cov, lr_aux);
    }
    bool finite = std::isfinite(st.negll);
    for (int i = 0; i < N; ++i) finite = finite && std::isfinite(std::exp(x[i]));
    if (!finite) return kNaOrInf;
    bool terminate = false;
    if (cfg.convergence_criterion == "relative_change_in_parameters") {
      terminate = blk_norm(x, &x_lag1, 0, N) <= delta_rel_conv * blk_norm(x_lag1, nullptr, 0, N);
    } else {
      terminate = (negll_lag1 - st.negll) <= delta_rel_conv * std::max(std::fabs(negll_lag1), 1.);
    }
    if (terminate) { num_it = it + 1; break; }
  }
  (void)NA;
  *num_it_out = num_it;
  return 0;
}

int gpb_optimize_laplace_cov_aux_pars(const GpbOptimConfig& cfg, gpb_laplace_aux_fn fn, void* ctx, int naux, const double theta_init[2], double* aux,
                                      GpbLaplaceAuxResult* out, char* err, int errlen) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !out || !theta_init || !aux || naux < 1 || naux > 8) return fail("gpb_optimize_laplace_cov_aux_pars: invalid argument");
  if (!(theta_init[0] > 0.) || !(theta_init[1] > 0.))
    return fail("Initial covariance parameters need to be positive (found %g, %g on the transformed scale)", theta_init[0], theta_init[1]);
  for (int j = 0; j < naux; ++j) if (!(aux[j] > 0.)) return fail("Initial auxiliary parameters need to be positive (found %g)", aux[j]);
  if (cfg.optimizer != "lbfgs" && cfg.optimizer != "nelder_mead" && cfg.optimizer != "gradient_descent")
    return fail("optimizer_cov = '%s' for a likelihood with estimated auxiliary parameters is not on the MI355X path of this library (supported: 'lbfgs', the reference's default, 'gradient_descent' and 'nelder_mead')", cfg.optimizer.c_str());
  if (cfg.optimizer == "nelder_mead") {
    // round 6: OptimLib's simplex search over (log sigma1_2, log a, log aux_1 ..) -- EvalLLforOptimLib with EstimateAuxPars() (optim_utils.h:61-213: SetAuxPars(exp(tail)) at
    // every evaluation), likelihood evaluations only; an evaluator error at a trial vertex counts as +Inf there and the mode goes back (as run_nelder_mead_laplace)
    *out = GpbLaplaceAuxResult();
    std::vector<double> xn(2 + naux), o(3 + naux), av(naux);
    xn[0] = std::log(theta_init[0]); xn[1] = std::log(theta_init[1]);
    for (int j = 0; j < naux; ++j) xn[2 + j] = std::log(aux[j]);
    int n_ok = 0, n_evals = 0;
    auto f = [&](const double* xv, double* fv) -> int {
      for (int j = 0; j < naux; ++j) av[j] = std::exp(xv[2 + j]);
      ++n_evals;
      if (fn(ctx, 0, std::exp(xv[0]), std::exp(xv[1]), av.data(), naux, o.data())) {
        if (n_ok == 0) return -1;
        *fv = INFINITY;
        return fn(ctx, 3, 0., 0., av.data(), naux, o.data()) ? -1 : 0;
      }
      ++n_ok;
      *fv = o[0];
      if (!std::isfinite(*fv) && fn(ctx, 3, 0., 0., av.data(), naux, o.data())) return -1;
      return 0;
    };
    double fx = 1e99;
    if (cfg.max_iter > 0) {
      const double delta = cfg.delta_rel_conv_init > 0. ? cfg.delta_rel_conv_init : 1e-8;
      if (nelder_mead_nd(f, 2 + naux, xn.data(), cfg, delta, &out->num_it, &fx)) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed during the optimisation"); return -1; }
      for (double v : xn) if (!std::isfinite(v)) return fail("NaN or Inf occurred in covariance parameter optimization using 'nelder_mead'; try other initial values");
      if (!std::isfinite(fx)) return fail("NaN or Inf occurred in covariance parameter optimization using 'nelder_mead'; try other initial values");
    }
    out->theta[0] = std::exp(xn[0]); out->theta[1] = std::exp(xn[1]);
    for (int j = 0; j < naux; ++j) aux[j] = std::exp(xn[2 + j]);
    out->negll = fx; out->num_evals = n_evals;
    return 0;
  }
  LapCoefState st{nullptr, ctx, 0, 0, nullptr, nullptr, 1., 1., std::vector<double>(), std::vector<double>()};
  st.afn = fn; st.naux = naux; st.nc = 2;
  std::vector<double> x(2 + naux);
  x[0] = std::log(theta_init[0]); x[1] = std::log(theta_init[1]);
  for (int j = 0; j < naux; ++j) x[2 + j] = std::log(aux[j]);
  *out = GpbLaplaceAuxResult();
  if (cfg.max_iter > 0) {
    const int rc = cfg.optimizer == "gradient_descent" ? run_gradient_descent_laplace_aux(st, cfg, x, &out->num_it, fail) : run_lbfgs_laplace_coef(st, cfg, x, &out->num_it, fail);
    if (rc == kNaOrInf) return fail("NaN or Inf occurred in the parameter optimisation of a likelihood with auxiliary parameters (the reference restarts with 'nelder_mead', which is not on this path for such models)");
    if (rc) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed during the optimisation"); return -1; }
  }
  out->theta[0] = std::exp(x[0]); out->theta[1] = std::exp(x[1]);
  for (int j = 0; j < naux; ++j) aux[j] = std::exp(x[2 + j]);
  out->negll = st.negll;
  out->num_evals = st.n_evals;
  return 0;
}

int gpb_laplace_coef_std_errors(gpb_laplace_fe_fn fn, void* ctx, int n, int p, const double* X, const double* offset, const double theta[2],
                                const double* beta, double* se_out, char* err, int errlen) {
  const Fail fail{err, errlen};
  if (err && errlen > 0) err[0] = 0;
  if (!fn || !X || !theta || !beta || !se_out || n < 1 || p < 1) return fail("gpb_laplace_coef_std_errors: invalid argument");
  LapCoefState st{fn, ctx, n, p, X, offset, 1., 1., std::vector<double>(n), std::vector<double>(n)};
  const double h = std::pow(std::numeric_limits<double>::epsilon(), 1.0 / 3.0);
  std::vector<double> x(2 + p), H((size_t)p * p), g1(2 + p), g2(2 + p);
  x[0] = std::log(theta[0]); x[1] = std::log(theta[1]);
  for (int i = 0; i < p; ++i) {
    double delta = beta[i] * h;
    if (std::fabs(delta) < h) delta = h;                                        // :10858-10863
    for (int sgn = 0; sgn < 2; ++sgn) {
      for (int j = 0; j < p; ++j) x[2 + j] = beta[j];
      x[2 + i] += sgn == 0 ? delta : -delta;
      if (st.eval(x.data(), true, sgn == 0 ? g1.data() : g2.data())) { if (err && !err[0]) snprintf(err, errlen, "likelihood evaluation failed while calculating standard deviations"); return -1; }
    }
    for (int j = 0; j < p; ++j) H[(size_t)i * p + j] = (g1[2 + j] - g2[2 + j]) / (2. * delta);
  }
  for (int i = 0; i < p; ++i) for (int j = 0; j < i; ++j) { const double v = 0.5 * (H[(size_t)i * p + j] + H[(size_t)j * p + i]); H[(size_t)i * p + j] = H[(size_t)j * p + i] = v; }
  const double nan_value = std::numeric_limits<double>::quiet_NaN();
  for (int j = 0; j < p; ++j) se_out[j] = nan_value;
  // Cholesky H = L L' (lower, in place); (H^-1)_jj = || L^-1 e_j ||^2
  std::vector<double> L = H;
  bool pd = true;
  for (int i = 0; i < p && pd; ++i)
    for (int j = 0; j <= i; ++j) {
      double acc = L[(size_t)i * p + j];
      for (int k = 0; k < j; ++k) acc -= L[(size_t)i * p + k] * L[(size_t)j * p + k];
      if (i == j) { if (!(acc > 0.) || !std::isfinite(acc)) { pd = false; break; } L[(size_t)i * p + i] = std::sqrt(acc); }
      else L[(size_t)i * p + j] = acc / L[(size_t)j * p + j];
    }
  if (!pd) {
    fprintf(stderr, "[gpboost_amd] Warning: Cannot calculate standard deviations for regression coefficients since the approximated Hessian is not positive definite \n");
    return 0;
  }
  std::vector<double> col(p);
  for (int j = 0; j < p; ++j) {
    double ss = 0.;
    for (int i = j; i < p; ++i) {
      double v = (i == j) ? 1. : 0.;
      for (int k = j; k < i; ++k) v -= L[(size_t)i * p + k] * col[k];
      col[i] = v / L[(size_t)i * p + i];
      ss += col[i] * col[i];
    }
    se_out[j] = std::sqrt(ss);
  }
  return 0;
}
