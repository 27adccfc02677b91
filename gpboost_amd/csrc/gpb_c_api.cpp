// gpboost_amd/csrc/gpb_c_api.cpp -- the hot-path slice of the reference C API
// (include/gpboost_c_api_subset.h) on top of the gfx950 shim (include/gpb_hip.h).
//
// What lives here is exactly the host-side pre/post-processing the reference does around its
// per-point loop, re-stated for one Gaussian Vecchia GP:
//   ordering          src/GPBoost/Vecchia_utils.cpp:1129-1138  (std::shuffle(std::mt19937(seed)))
//   TransformCovPars  src/GPBoost/re_model.cpp:768-778 -> include/GPBoost/cov_fcts.h:485-516
//   nugget bound      include/GPBoost/re_model_template.h:7849-7874
//   SetY permutation  include/GPBoost/re_model_template.h:6185-6222
//   negll formula     include/GPBoost/re_model_template.h:3132
//   gradient assembly include/GPBoost/re_model_template.h:1988-2011
// and, for likelihood = "bernoulli_logit" (Vecchia-Laplace, iterative methods; BASELINE config 4):
//   label check       include/GPBoost/likelihoods.h:1321-1329 (CheckY)
//   mode reset + sign include/GPBoost/re_model_template.h:3191-3212 (EvalLaplaceApproxNegLogLikelihood)
//   CG / SLQ settings include/GPBoost/re_model_template.h:863-891, :943-948, defaults :5860-5876, :5507
// All per-point arithmetic happens on the device.
#include "../../include/gpboost_c_api_subset.h"
#include "../../include/gpb_hip.h"
#include "gpb_optim.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <limits>
#include <vector>

namespace {

thread_local char g_last_error[512] = "Everything is fine";   // c_api.h:1837-1849

thread_local void (*g_log_callback)(const char*) = nullptr;   // Log::ResetCallBack (log.h:260-264): thread-local like the reference's
void log_info(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g_log_callback) g_log_callback(buf);
  else { std::fputs(buf, stdout); std::fflush(stdout); }
}

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return -1;
}
int shim_error() { return set_error("%s", gpb_hip_get_last_error()); }

constexpr double kMinNuggetVarRatio = 1e-10;   // re_model_template.h:5668

template <class F>
struct ScopeExitApi { F f; ~ScopeExitApi() { f(); } };
template <class F>
ScopeExitApi<F> scope_exit_api(F f) { return ScopeExitApi<F>{ f }; }

struct REModelHip {
  int n = 0, d = 0, m = 0;
  int num_neighbors = 0;        // num_neighbors_ as given (re_model_template.h:288-299); num_neighbors_pred_ defaults to twice this
  int cov_type = 0;
  std::vector<int> perm;        // data_indices_per_cluster_: Vecchia position -> data index
  gpb_hip_vecchia_t* vh = nullptr;          // cluster 0 (the only one unless cluster_ids distinguishes independent realisations)
  std::vector<gpb_hip_vecchia_t*> vhs;      // one Vecchia state per cluster, in order of first appearance (re_model_template.h:6820-6852)
  std::vector<int> cl_off;                  // offsets of the clusters in perm / ybuf (size #clusters + 1)
  gpb_hip_exact_t* eh = nullptr;   // gp_approx == "none": dense path, data order (no Vecchia ordering)
  // gp_approx == "full_scale_vecchia" ("vif"): predictive process on num_ind_points inducing points + Vecchia approximation of the residual process
  bool has_weights = false;        // sample weights (Gaussian Vecchia model): observation-specific nuggets live in the device handles
  std::vector<double> lik_weights; bool lik_weights_pushed = false;   // sample weights of a non-Gaussian model (data order): factors of the per-datum likelihood terms (round 5)
  std::vector<double> nug_v;       // ... and, Vecchia order, here: 1 / w_i (the 'latent_*' prediction types need R^-1 = diag(w) on the host)
  bool vif = false;
  int num_ind_points = 0;
  std::vector<double> ip;          // inducing points, column-major num_ind_points x d (kmeans++ from the model's generator)
  double* ybuf = nullptr;       // y in Vecchia order, page-locked (uploaded on every host-pointer call)
  size_t ybuf_cap = 0;
  double cur_negll = 0.;
  bool negll_valid = false;
  bool has_duplicates = false;
  std::vector<int32_t> cluster_id_values;   // unique_clusters_: the id of every cluster in vhs order (empty: no cluster_ids_data was given = one cluster with id 0, re_model_template.h:6836)
  bool trace = false;
  std::string likelihood = "gaussian";
  // iterative-method settings of the Laplace path (re_model_template.h:5860-5876, :5507)
  int cg_max_num_it = 1000, cg_max_num_it_tridiag = 1000, num_rand_vec_trace = 50, seed_rand_vec_trace = 1;
  double cg_delta_conv = 1e-2, delta_conv_mode_finding = 1e-8;
  std::vector<int> labels;      // y in {0,1}, Vecchia order
  // likelihoods with an auxiliary parameter (round 5: gamma, negative_binomial -- the shape; likelihoods.h:298-322): num_aux_pars_, aux_pars_ (original
  // scale), AuxParsHaveBeenSet(), init_aux_pars_ / init_aux_pars_given_ (re_model.cpp:327-344), estimate_aux_pars_ (re_model_template.h:909-912)
  int num_aux = 0; double aux_pars[2] = {1., 2.}; bool aux_set = false; double init_aux[2] = {-1., -1.}; bool init_aux_given = false; bool estimate_aux_pars = true;      // (t: scale, df -- internal default df 2, likelihoods.h:392)
  // "t_fix_df" (ParseLikelihoodAliasEstimateAdditionalPars, likelihoods.h:10466-10471): estimate_df_t_ = false -> num_aux_pars_estim_ = 1 of num_aux_pars_ = 2 (:402-407).
  // The df stay IN the optimiser's vector with a zero gradient (SetGradAuxParsNotEstimated :16179-16183) and SetAuxPars copies only the first num_aux_pars_estim_ values (:2780-2789)
  bool estimate_df_t = true;
  int num_aux_estim() const { return (likelihood == "t" && !estimate_df_t) ? 1 : num_aux; }
  std::vector<double> resp_real;   // gamma: the real-valued response, Vecchia order
  // repeated locations of a non-Gaussian model (the reference's unique-location mapping, Vecchia_utils.cpp:1156-1168, re_comp.h:863-885): the
  // Vecchia handle lives on the n_re unique locations; datum at shuffled position k belongs to random effect re_of[k]; dorder lists the
  // shuffled positions grouped by random effect (stable), re_ptr is the CSR of that grouping.  n_re == 0: no repeated locations.
  int n_re = 0; std::vector<int> re_of, dorder, re_ptr;
  double lap_info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool lap_fit_first_eval = true;
  GpbLaplaceOptimResult last_fit_lap;
  // parameter estimation (REModel members of re_model.h: cov_pars_, init_cov_pars_, num_it_; all on the TRANSFORMED scale)
  std::mt19937 rng;             // rng_ of the reference: orderings first, then the sub-sample of FindInitCovPar
  std::vector<double> coords0;  // coordinates of the first cluster in Vecchia order, column-major (for FindInitCovPar)
  int n0 = 0;
  GpbOptimConfig optim;
  bool optimizer_unsupported_alias = false;
  double cov_pars_tr[3] = {0, 0, 0}, init_cov_pars_tr[3] = {0, 0, 0};
  bool cov_pars_initialized = false, init_cov_pars_provided = false;
  bool yaux_valid = false;      // y_aux_has_been_calculated_: factor + y_aux of the last GPB_HIP_CalcYAux are still on the device
  bool y_set = false;           // y_has_been_set_: ybuf / the device copy hold the response of the last call that passed one
  bool dev_y_is_host = false;   // the device response is exactly y_host (no offset subtracted, not a boosting seam's working response): see prediction_response
  // GPB_SetPredictionData (re_model_template.h:3337-3400)
  std::vector<double> coords_pred; int num_data_pred = 0;
  std::string vecchia_pred_type = "order_obs_first_cond_obs_only";   // default for the Gaussian likelihood (:7112-7115)
  int num_neighbors_pred = 0;   // 0 = default (2 * num_neighbors_, :299)
  int num_it = 0;
  GpbOptimResult last_fit;
  // linear-regression covariates (GPB_OptimLinRegrCoefCovPar; Gaussian likelihood, coefficients profiled out by GLS = the reference's default "wls")
  int p_cov = 0;
  bool coef_by_iteration = false;         // ... except with optimizer_cov 'gradient_descent': one least-squares update per ITERATION (re_model_template.h:1478-1481)
  bool fitting_with_covariates = false;   // inside GPB_OptimLinRegrCoefCovPar only: every evaluation of the objective profiles the coefficients out (optim_utils.h:296-302)
  std::vector<double> X;        // data order, column-major n x p (X_)
  std::vector<double> beta;     // beta_
  std::vector<double> beta_lag1;   // beta_lag1_: the coefficients at the last accepted lbfgs iterate (SetLag1ProfiledOutVariables, re_model_template.h)
  std::vector<double> chol_XtPsiInvX;   // lower Cholesky factor (p x p, row-major) of X' Psi^-1 X (Psi on the error-variance-free scale) at the last GLS step
  bool coef_estimated = false;
  std::string optimizer_coef = "";   // as given to GPB_SetOptimConfig ("" = default: "wls" for the Gaussian likelihood)
  std::vector<double> init_coef;     // init_coef of GPB_SetOptimConfig (non-Gaussian models with covariates: start of the lbfgs vector)
  bool init_coef_from_iid_model = true;   // init_coef_aux_pars_from_iid_model (re_model.cpp:345; the packages' default)
  std::string cg_preconditioner_type = "vadu";   // ParsePreconditionerAlias default for a non-Gaussian Vecchia model (re_model_template.h:7137)
  int piv_chol_rank = 50;                         // fitc_piv_chol_preconditioner_rank_ for "pivoted_cholesky" (default_piv_chol_preconditioner_rank_, re_model_template.h:5922) / "fitc" (200, :5921)
  std::vector<double> pc_ip; bool pc_ip_pushed = false;   // inducing points of the "fitc" preconditioner (k x d column-major; ind_points_determined_for_preconditioner_) and whether the device has them
  std::vector<double> offset;                     // GPB_SetOffsetData (fixed_effects_, has_fixed_effects_; re_model_template.h:6318-6321)
  bool has_offset = false;
  std::vector<double> y_host;                     // the response as last passed in (original order, no offset subtracted): y_vec_ of the Gaussian model
  bool model_has_been_estimated = false;
  ~REModelHip() { for (auto* v : vhs) gpb_hip_vecchia_free(v); if (eh) gpb_hip_exact_free(eh); if (ybuf) gpb_hip_pinned_free(ybuf); }
};

// The host-pointer entry points permute n-vectors between data order and Vecchia order (gathers / scatters over 8 MB at n = 1e6:
// latency-bound on one core, ~1 ms); a few threads bring that to a fraction of the H2D copy that follows.
template <class F>
void parallel_for(int n, F&& body) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int nt = n < 200000 ? 1 : std::max(1, std::min(8, hw > 0 ? hw : 1));
  if (nt == 1) { body(0, n); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  const int per = (n + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { body(std::min(n, t * per), std::min(n, (t + 1) * per)); });
  body(0, std::min(n, per));
  for (auto& x : th) x.join();
}

bool near(double a, double b) { return std::fabs(a - b) < 1e-10 * std::max({1.0, std::fabs(a), std::fabs(b)}); }  // utils.h:55

// (sigma2, sigma1_2, rho) -> (sigma2, sigma1_2 / sigma2, sqrt(2 nu) / rho), with the nugget lower bound
int transform_cov_pars(const REModelHip* mdl, const double* cov_pars, double* tr) {
  double sigma2 = cov_pars[0];
  const double sigma1_2 = cov_pars[1], rho = cov_pars[2];
  if (!(sigma2 > 0.) || !(sigma1_2 > 0.) || !(rho > 0.))
    return set_error("Covariance parameters need to be positive (found %g, %g, %g)", sigma2, sigma1_2, rho);
  const double nugget_min = kMinNuggetVarRatio / (1. - kMinNuggetVarRatio) * sigma1_2;   // :7866
  if (sigma2 < nugget_min) sigma2 = nugget_min;
  tr[0] = sigma2;
  tr[1] = sigma1_2 / sigma2;
  const double c = mdl->cov_type == 0 ? 1. : (mdl->cov_type == 1 ? std::sqrt(3.) : std::sqrt(5.));
  tr[2] = c / rho;
  return 0;
}

// Round 5: proportions under the logit / probit links -- binomial_logit / binomial_probit (y = successes / trials, the trials are the sample weights) and
// quasi_bernoulli_logit / _probit (any y in [0, 1]): the Bernoulli terms with a real-valued y (LogLikBernoulliLogit<double>, LogLikBinomialProbit, likelihoods.h:11394-11404).
bool is_proportion_likelihood(const std::string& lik) {
  return lik == "binomial_logit" || lik == "binomial_probit" || lik == "quasi_bernoulli_logit" || lik == "quasi_bernoulli_probit";
}
bool is_logit_link(const std::string& lik) { return lik == "bernoulli_logit" || lik == "binomial_logit" || lik == "quasi_bernoulli_logit"; }
bool is_probit_link(const std::string& lik) { return lik == "bernoulli_probit" || lik == "binomial_probit" || lik == "quasi_bernoulli_probit"; }
int laplace_link_id(const std::string& lik) {
  return is_probit_link(lik) ? 1 : (lik == "poisson" ? 2 : (lik == "gamma" ? 3 : (lik == "negative_binomial" ? 4 : (lik == "beta" ? 5 : (lik == "t" ? 6 : (lik == "lognormal" ? 7 : (lik == "gaussian_latent" ? 8 : 0)))))));
}
bool supported_non_gaussian(const std::string& lik) {
  return lik == "bernoulli_logit" || lik == "bernoulli_probit" || lik == "poisson" || lik == "gamma" || lik == "negative_binomial" || lik == "beta" || lik == "t" || lik == "lognormal" || lik == "gaussian_latent" || is_proportion_likelihood(lik);
}
// Likelihood::ParseLikelihoodAlias (likelihoods.h:10254-10275)
// "<likelihood>_fix_df" -> "<likelihood>", *fix_df = true (ParseLikelihoodAliasEstimateAdditionalPars, likelihoods.h:10466-10471: the suffix is stripped from any name)
std::string strip_fix_df(const std::string& lik, bool* fix_df) {
  *fix_df = lik.size() > 7 && lik.compare(lik.size() - 7, 7, "_fix_df") == 0;
  return *fix_df ? lik.substr(0, lik.size() - 7) : lik;
}
std::string parse_likelihood_alias(const std::string& lik) {
  if (lik == "binary_probit") return "bernoulli_probit";
  if (lik == "binary" || lik == "binary_logit") return "bernoulli_logit";
  if (lik == "binomial") return "binomial_logit";
  if (lik == "quasi_binary_probit") return "quasi_bernoulli_probit";
  if (lik == "quasi_binary" || lik == "quasi_binary_logit") return "quasi_bernoulli_logit";
  return lik;
}
int num_aux_of(const std::string& lik) { return lik == "t" ? 2 : ((lik == "gamma" || lik == "negative_binomial" || lik == "beta" || lik == "lognormal" || lik == "gaussian_latent") ? 1 : 0); }      // t: scale, df (likelihoods.h:398-407)
// the model's auxiliary parameters to the device (Likelihood::SetAuxPars); a no-op for likelihoods without any
// cg_preconditioner_type of the iterative methods (SetPropertiesLikelihood, re_model_template.h:7516-7524)
int laplace_push_preconditioner(REModelHip* mdl) {
  if (!mdl->vh || mdl->likelihood == "gaussian" || mdl->eh) return 0;
  const int type = mdl->cg_preconditioner_type == "pivoted_cholesky" ? 1 : (mdl->cg_preconditioner_type == "fitc" ? 2 : (mdl->cg_preconditioner_type == "vecchia_response" ? 3 :
                   (mdl->cg_preconditioner_type == "vifdu" ? 4 : (mdl->cg_preconditioner_type == "none" ? 5 : 0))));
  if (gpb_hip_vecchia_laplace_set_preconditioner(mdl->vh, type, mdl->piv_chol_rank)) return shim_error();
  return 0;
}
int kmeans_plusplus(const std::vector<double>& x, int n, int d, int k, std::mt19937& gen, int max_it, std::vector<double>* means_out);
// "fitc" preconditioner: its inducing points are determined ONCE, at the first covariance factor of the model (Calc_FITC_Preconditioner_Vecchia, re_model_template.h:9502-9593:
// kmeans++ on the Vecchia-ordered (unique) coordinates from the model's generator, whatever that generator has drawn before -- the ordering shuffle, FindInitCovPar's
// sub-sample), so this runs right before an evaluation, not in GPB_SetOptimConfig
int laplace_prepare_preconditioner(REModelHip* mdl) {
  if (mdl->cg_preconditioner_type != "fitc" || mdl->pc_ip_pushed || !mdl->vh) return 0;
  const int k = mdl->piv_chol_rank;
  if (mdl->n <= k) return set_error("Need to have less inducing points (currently fitc_piv_chol_preconditioner_rank = %d) than data points (%d) for cg_preconditioner_type = '%s' ", k, mdl->n, "fitc");
  if (mdl->n0 < k) return set_error("Cannot have more inducing points than unique coordinates for cg_preconditioner_type = '%s' ", "fitc");
  if (mdl->pc_ip.empty() && kmeans_plusplus(mdl->coords0, mdl->n0, mdl->d, k, mdl->rng, 1000, &mdl->pc_ip)) return -1;
  if (gpb_hip_vecchia_laplace_set_inducing_points(mdl->vh, k, mdl->pc_ip.data())) return shim_error();
  mdl->pc_ip_pushed = true;
  return 0;
}
int laplace_push_aux(REModelHip* mdl) {
  if (mdl->num_aux < 1) return 0;
  if (gpb_hip_vecchia_laplace_set_aux_pars(mdl->vh, mdl->aux_pars, mdl->num_aux)) return shim_error();
  return 0;
}
// Likelihood::FindInitialAuxPars (likelihoods.h:1851-1947) for gamma (approximate MLE of the shape ignoring the effects) and negative_binomial
// (method of moments); y, fixed_effects in data order
// (wts: sample weights in the order of y, or NULL -- weighted moments with sum of weights in place of n, likelihoods.h:1856-1910)
double initial_aux_par(const std::string& lik, int n, const double* y, const double* fe, const double* wts = nullptr) {
  if (lik == "t") {         // MAD as a robust start of the scale, the inter-quartile range if it is zero (likelihoods.h:1973-2010); the df keep their current value
    std::vector<double> v(n);
    for (int i = 0; i < n; ++i) v[i] = fe ? y[i] - fe[i] : y[i];
    auto median = [](std::vector<double>& a) {      // CalculateMedianPartiallySortInput (utils.h): nth_element, mean of the two middle values for an even count
      const size_t nn = a.size(), pos = nn / 2;
      std::nth_element(a.begin(), a.begin() + pos, a.end());
      double med = a[pos];
      if (nn % 2 == 0) { std::nth_element(a.begin(), a.begin() + pos - 1, a.end()); med = (a[pos - 1] + med) / 2.; }
      return med;
    };
    const double med = median(v);
    for (int i = 0; i < n; ++i) v[i] = std::fabs(v[i] - med);
    double sc = 1.4826 * median(v);
    if (sc <= 1e-10) {
      for (int i = 0; i < n; ++i) v[i] = fe ? y[i] - fe[i] : y[i];
      int pos = (int)(n * 0.25);
      std::nth_element(v.begin(), v.begin() + pos, v.end());
      const double q25 = v[pos];
      pos = (int)(n * 0.75);
      std::nth_element(v.begin(), v.begin() + pos, v.end());
      sc = (v[pos] - q25) / 1.349;
    }
    return sc;
  }
  if (lik == "beta") {      // method of moments for the precision, phi = mu (1 - mu) / var - 1, clipped to [0.1, 100] (likelihoods.h:1952-1972; the fixed effects are not used there)
    double avg = 0., sum_sq = 0., sw = 0.;
    for (int i = 0; i < n; ++i) { const double w = wts ? wts[i] : 1.0; avg += w * y[i]; sum_sq += w * y[i] * y[i]; sw += w; }
    avg /= sw;
    const double sample_var = std::max((sum_sq - sw * avg * avg) / (sw - 1), 1e-6);
    double phi = avg * (1.0 - avg) / sample_var - 1.0;
    if (std::isnan(phi) || phi <= 0.0) phi = 1.0;
    return std::min(std::max(phi, 0.1), 100.0);
  }
  if (lik == "lognormal") {      // moment-based: the (weighted) variance of log y - offset, at least 1e-6 (likelihoods.h:2015-2030)
    double mean_log = 0., mean_log_sq = 0., sw = 0.;
    for (int i = 0; i < n; ++i) {
      const double w = wts ? wts[i] : 1.0, z = fe ? std::log(y[i]) - fe[i] : std::log(y[i]);
      mean_log += w * z; mean_log_sq += w * z * z; sw += w;
    }
    mean_log /= sw; mean_log_sq /= sw;
    return std::max(mean_log_sq - mean_log * mean_log, 1e-6);
  }
  if (lik == "gamma") {
    double log_avg = 0., avg_log = 0., sw = 0.;
    for (int i = 0; i < n; ++i) {
      const double w = wts ? wts[i] : 1.0;
      if (fe) { log_avg += w * y[i] / std::exp(fe[i]); avg_log += w * (std::log(y[i]) - fe[i]); }
      else { log_avg += w * y[i]; avg_log += w * std::log(y[i]); }
      sw += w;
    }
    log_avg = std::log(log_avg / sw); avg_log /= sw;
    const double s = std::max(log_avg - avg_log, 1e-8);
    return (3. - s + std::sqrt((s - 3.) * (s - 3.) + 24. * s)) / (12. * s);
  }
  if (lik == "negative_binomial" || lik == "gaussian_latent") {      // (gaussian_latent shares the moment pass of the counts, y / exp(fixed effect) included: likelihoods.h:1857-1882; its start is half the sample variance, :2012-2014)
    double avg = 0., sum_sq = 0., sw = 0.;
    for (int i = 0; i < n; ++i) { const double w = wts ? wts[i] : 1.0; const double v = fe ? y[i] / std::exp(fe[i]) : y[i]; avg += w * v; sum_sq += w * v * v; sw += w; }
    avg /= sw;
    const double avg_sq = avg * avg;
    const double sample_var = std::max((sum_sq - sw * avg_sq) / (sw - 1), 1e-6);
    if (lik == "gaussian_latent") return sample_var / 2.;
    return sample_var <= avg ? 100 * avg_sq : avg_sq / (sample_var - avg);
  }
  return 1.;
}
void find_initial_aux_pars(REModelHip* mdl, const double* y, const double* fe) {
  // (sic) the reference hands FindInitialAuxPars the caller's y in DATA order (re_model_template.h:1343) while Likelihood::weights_ is in the order of the
  // cluster's data, i.e. Vecchia order (:425-431): datum i is paired with the weight of datum perm[i].  Reproduced, because the start value decides where a
  // flat likelihood's fit ends (verified against the reference: with this pairing its default fit is reproduced, with the "right" one it is not).
  std::vector<double> wv;
  if (!mdl->lik_weights.empty()) { wv.resize(mdl->n); for (int i = 0; i < mdl->n; ++i) wv[i] = mdl->lik_weights[mdl->perm[i]]; }
  mdl->aux_pars[0] = initial_aux_par(mdl->likelihood, mdl->n, y, fe, wv.empty() ? nullptr : wv.data());
  mdl->aux_set = true;
}

// sample weights of a non-Gaussian model -> the device state, in the order it keeps the data (Vecchia order, grouped by random effect for repeated locations)
int laplace_push_weights(REModelHip* mdl) {
  if (mdl->lik_weights.empty() || mdl->lik_weights_pushed) return 0;
  std::vector<double> w(mdl->n);
  if (mdl->n_re > 0) for (int g = 0; g < mdl->n; ++g) w[g] = mdl->lik_weights[mdl->perm[mdl->dorder[g]]];
  else for (int k = 0; k < mdl->n; ++k) w[k] = mdl->lik_weights[mdl->perm[k]];
  if (gpb_hip_vecchia_laplace_set_weights(mdl->vh, w.data())) return shim_error();
  mdl->lik_weights_pushed = true;
  return 0;
}

// location parameter = mode + fixed effects (likelihoods.h:3861-3870), Vecchia order; NULL clears the offset
int laplace_upload_fixed_effects(REModelHip* mdl, const double* fixed_effects) {
  if (fixed_effects) {
    std::vector<double> fe(mdl->n);
    if (mdl->n_re > 0) for (int g = 0; g < mdl->n; ++g) fe[g] = fixed_effects[mdl->perm[mdl->dorder[g]]];       // grouped by random effect
    else for (int k = 0; k < mdl->n; ++k) fe[k] = fixed_effects[mdl->perm[k]];
    if (gpb_hip_vecchia_laplace_set_fixed_effects(mdl->vh, fe.data())) return shim_error();
  } else if (gpb_hip_vecchia_laplace_set_fixed_effects(mdl->vh, nullptr)) return shim_error();
  return 0;
}

// response (validated against the likelihood) and fixed effects of the Vecchia-Laplace path, Vecchia order
int laplace_upload_data(REModelHip* mdl, const double* y_data, const double* fixed_effects) {
  if (!y_data) return set_error("y_data is NULL: the HIP hot path evaluates the likelihood at the response passed in");
  mdl->labels.resize(mdl->n);
  const bool poisson = mdl->likelihood == "poisson" || mdl->likelihood == "negative_binomial";     // integer-valued responses >= 0
  if (mdl->likelihood == "gamma" || mdl->likelihood == "beta" || mdl->likelihood == "t" || mdl->likelihood == "lognormal" || mdl->likelihood == "gaussian_latent") {      // likelihoods.h:1365-1373 (gamma, lognormal): strictly positive, real-valued; beta: :1403-1409, strictly inside (0, 1); t: any real value
    const bool is_beta = mdl->likelihood == "beta", is_t = mdl->likelihood == "t" || mdl->likelihood == "gaussian_latent";      // (is_t: any finite real value)
    mdl->resp_real.resize(mdl->n);
    for (int k = 0; k < mdl->n; ++k) {
      const double yk = y_data[mdl->perm[k]];
      if (is_beta) { if (!(yk > 0. && yk < 1.)) return set_error(" Must have 0 < y < 1 for the response variable ('y') for likelihood = '%s', found %g ", mdl->likelihood.c_str(), yk); }
      else if (is_t) { if (!std::isfinite(yk)) return set_error("NaN or Inf in the response variable ('y') for likelihood = '%s' ", mdl->likelihood.c_str()); }
      else if (!(yk > 0.)) return set_error(" Must have y > 0 for the response variable ('y') for likelihood = '%s', found %g ", mdl->likelihood.c_str(), yk);
      mdl->resp_real[k] = yk; mdl->labels[k] = 0;
    }
    if (gpb_hip_vecchia_laplace_set_likelihood(mdl->vh, laplace_link_id(mdl->likelihood))) return shim_error();
    if (mdl->n_re > 0) {
      std::vector<double> grouped(mdl->n);
      for (int g = 0; g < mdl->n; ++g) grouped[g] = mdl->resp_real[mdl->dorder[g]];
      if (gpb_hip_vecchia_laplace_set_response_real(mdl->vh, grouped.data())) return shim_error();
    } else if (gpb_hip_vecchia_laplace_set_response_real(mdl->vh, mdl->resp_real.data())) return shim_error();
    if (laplace_push_aux(mdl) || laplace_push_weights(mdl)) return -1;
    mdl->y_set = true;
    return laplace_upload_fixed_effects(mdl, fixed_effects);
  }
  if (is_proportion_likelihood(mdl->likelihood)) {        // likelihoods.h:1330-1337: proportions in [0, 1]; :669-674: the binomial likelihoods need the trials as weights
    const bool binom = mdl->likelihood == "binomial_logit" || mdl->likelihood == "binomial_probit";
    if (binom && mdl->lik_weights.empty())
      return set_error("'weights' are missing. For the likelihood '%s', 'weights' should contain the number of trials n_i (and 'y' the ratios of successes / trials). ", mdl->likelihood.c_str());
    mdl->resp_real.resize(mdl->n);
    for (int k = 0; k < mdl->n; ++k) {
      const double yk = y_data[mdl->perm[k]];
      if (yk < 0. || yk > 1. || !(yk == yk)) return set_error(" Must have 0 <= y <= 1 for the response variable ('y') for likelihood = '%s', found %g. Note that the response variable should be the proportion of successes / trials ", mdl->likelihood.c_str(), yk);
      mdl->resp_real[k] = yk; mdl->labels[k] = 0;
    }
    if (gpb_hip_vecchia_laplace_set_likelihood(mdl->vh, laplace_link_id(mdl->likelihood))) return shim_error();
    if (gpb_hip_vecchia_laplace_set_binomial(mdl->vh, binom ? 1 : 0)) return shim_error();
    if (mdl->n_re > 0) {
      std::vector<double> grouped(mdl->n);
      for (int g = 0; g < mdl->n; ++g) grouped[g] = mdl->resp_real[mdl->dorder[g]];
      if (gpb_hip_vecchia_laplace_set_response_real(mdl->vh, grouped.data())) return shim_error();
    } else if (gpb_hip_vecchia_laplace_set_response_real(mdl->vh, mdl->resp_real.data())) return shim_error();
    if (laplace_push_weights(mdl)) return -1;
    mdl->y_set = true;
    return laplace_upload_fixed_effects(mdl, fixed_effects);
  }
  mdl->resp_real.clear();      // integer / binary responses live in `labels` (GPB_GetResponseData tells the two apart by this)
  for (int k = 0; k < mdl->n; ++k) {
    const double yk = y_data[mdl->perm[k]];
    if (poisson) {                                        // likelihoods.h:1338-1350
      double intpart;
      if (yk < 0.) return set_error(" Must have y >= 0 for the response variable ('y') for likelihood = '%s', found %g ", mdl->likelihood.c_str(), yk);
      if (std::modf(yk, &intpart) != 0.0) return set_error("Found non-integer response variable ('y'). Response variable can only be integer valued for likelihood = '%s' ", mdl->likelihood.c_str());
      if (yk > 2147483647.) return set_error("response %g is too large for likelihood = '%s'", yk, mdl->likelihood.c_str());
      mdl->labels[k] = (int)yk;
      continue;
    }
    if (std::fabs(yk) >= 1e-10 && !near(yk, 1.))       // likelihoods.h:1321-1329
      return set_error("The response variable ('y') needs to be 0 or 1 for likelihood = '%s' ", mdl->likelihood.c_str());
    mdl->labels[k] = std::fabs(yk) < 1e-10 ? 0 : 1;
  }
  if (gpb_hip_vecchia_laplace_set_likelihood(mdl->vh, laplace_link_id(mdl->likelihood))) return shim_error();
  if (gpb_hip_vecchia_laplace_set_binomial(mdl->vh, 0)) return shim_error();
  if (mdl->n_re > 0) {
    std::vector<int> grouped(mdl->n);
    for (int g = 0; g < mdl->n; ++g) grouped[g] = mdl->labels[mdl->dorder[g]];
    if (gpb_hip_vecchia_laplace_set_labels(mdl->vh, grouped.data())) return shim_error();
  } else if (gpb_hip_vecchia_laplace_set_labels(mdl->vh, mdl->labels.data())) return shim_error();
  if (laplace_push_aux(mdl) || laplace_push_weights(mdl)) return -1;
  mdl->y_set = true;
  return laplace_upload_fixed_effects(mdl, fixed_effects);
}

// gpb_laplace_fn (gpb_optim.h) on the device: the stateful evaluator behind GPB_OptimCovPar for non-Gaussian likelihoods.  The mode stays
// in HBM between evaluations (warm start, likelihoods.h:3790-3797); 3 doubles cross PCIe per evaluation.
int device_laplace(void* ctx, int op_in, double var, double a, double* out3) {
  auto* mdl = static_cast<REModelHip*>(ctx);
  const bool first_update = (op_in & 16) != 0;
  const int op = op_in & 15;
  if (op == 3) return gpb_hip_vecchia_laplace_reset_mode_to_previous(mdl->vh) ? -1 : 0;
  if (op == 4) { mdl->lap_fit_first_eval = true; return 0; }      // the next evaluation starts its mode finding from zero
  // first gradient-descent update: cg_max_num_it(_tridiag) / 3 (likelihoods.h:3833-3836)
  const int cg = first_update ? (int)std::round(mdl->cg_max_num_it / 3.) : mdl->cg_max_num_it;
  const int cgt = first_update ? (int)std::round(mdl->cg_max_num_it_tridiag / 3.) : mdl->cg_max_num_it_tridiag;
  if (op == 0 || op == 1) {
    const int reset = mdl->lap_fit_first_eval ? 1 : 0;        // the fit starts from mode 0 (InitializeModeAvec), then warm-starts
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_eval(mdl->vh, mdl->cov_type, var, a, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, std::max(cg, 1),
                                     std::max(cgt, 1), mdl->cg_delta_conv, mdl->delta_conv_mode_finding, reset, 1, mdl->lap_info, nullptr)) return -1;
    mdl->lap_fit_first_eval = false;
    out3[0] = -mdl->lap_info[0];
    if (op == 0) return 0;
  }
  double g2[2];
  if (gpb_hip_vecchia_laplace_grad_current(mdl->vh, std::max(cg, 1), mdl->cg_delta_conv, g2, nullptr, nullptr)) return -1;
  out3[1] = g2[0]; out3[2] = g2[1];
  return 0;
}

// gpb_laplace_aux_fn (gpb_optim.h) on the device: likelihoods whose auxiliary parameters are estimated with the covariance parameters
int device_laplace_aux(void* ctx, int op_in, double var, double a, const double* aux, int naux, double* out) {
  auto* mdl = static_cast<REModelHip*>(ctx);
  const bool first_update = (op_in & 16) != 0;       // first gradient-descent update: cg_max_num_it(_tridiag) / 3 (likelihoods.h:3833-3836), as device_laplace
  const int op = op_in & 15;
  const int cg_it = first_update ? (int)std::round(mdl->cg_max_num_it / 3.) : mdl->cg_max_num_it;
  const int cgt_it = first_update ? (int)std::round(mdl->cg_max_num_it_tridiag / 3.) : mdl->cg_max_num_it_tridiag;
  if (op == 3) return gpb_hip_vecchia_laplace_reset_mode_to_previous(mdl->vh) ? -1 : 0;
  if (op == 4) { mdl->lap_fit_first_eval = true; return 0; }
  if (op == 0 || op == 1) {
    for (int j = 0; j < naux && j < mdl->num_aux_estim(); ++j) mdl->aux_pars[j] = aux[j];      // SetAuxPars at every evaluation (optim_utils.h:279-282): the first num_aux_pars_estim_ values
    mdl->aux_set = true;
    if (laplace_push_aux(mdl)) return -1;
    const int reset = mdl->lap_fit_first_eval ? 1 : 0;
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_eval(mdl->vh, mdl->cov_type, var, a, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, std::max(cg_it, 1),
                                     std::max(cgt_it, 1), mdl->cg_delta_conv, mdl->delta_conv_mode_finding, reset, 1, mdl->lap_info, nullptr)) return -1;
    mdl->lap_fit_first_eval = false;
    out[0] = -mdl->lap_info[0];
    if (op == 0) return 0;
  }
  double g2[2], g4[8];                               // 4 doubles per auxiliary parameter
  if (gpb_hip_vecchia_laplace_grad_current(mdl->vh, std::max(cg_it, 1), mdl->cg_delta_conv, g2, nullptr, nullptr)) return -1;
  if (gpb_hip_vecchia_laplace_grad_aux_current(mdl->vh, g4)) return -1;
  out[1] = g2[0]; out[2] = g2[1];
  for (int j = 0; j < naux && j < 2; ++j) out[3 + j] = j < mdl->num_aux_estim() ? g4[4 * j] : 0.;      // SetGradAuxParsNotEstimated (likelihoods.h:16179-16183)
  return 0;
}

// gpb_laplace_fe_fn (gpb_optim.h) on the device: the evaluator of the fits WITH a linear predictor.  The linear predictor arrives as the fixed
// effects of the location parameter (data order); the boosting gradient d(-mll)/dF goes back in data order (X' grad_F is the host's part).
int device_laplace_fe(void* ctx, int op, double var, double a, const double* fixed_effects, double* out3, double* grad_F) {
  auto* mdl = static_cast<REModelHip*>(ctx);
  if (op == 3) return gpb_hip_vecchia_laplace_reset_mode_to_previous(mdl->vh) ? -1 : 0;
  if (op == 4) { mdl->lap_fit_first_eval = true; return 0; }
  if (op == 0 || op == 1) {
    if (laplace_upload_fixed_effects(mdl, fixed_effects)) return -1;
    const int reset = mdl->lap_fit_first_eval ? 1 : 0;
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_eval(mdl->vh, mdl->cov_type, var, a, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, std::max(mdl->cg_max_num_it, 1),
                                     std::max(mdl->cg_max_num_it_tridiag, 1), mdl->cg_delta_conv, mdl->delta_conv_mode_finding, reset, 1, mdl->lap_info, nullptr)) return -1;
    mdl->lap_fit_first_eval = false;
    out3[0] = -mdl->lap_info[0];
    if (op == 0) return 0;
  }
  double g2[2];
  if (gpb_hip_vecchia_laplace_grad_current(mdl->vh, std::max(mdl->cg_max_num_it, 1), mdl->cg_delta_conv, g2, nullptr, nullptr)) return -1;
  out3[1] = g2[0]; out3[2] = g2[1];
  std::vector<double> gF(mdl->n);                       // Vecchia order, or -- repeated locations -- per datum grouped by random effect
  if (gpb_hip_vecchia_laplace_grad_F_current(mdl->vh, gF.data())) return -1;
  if (mdl->n_re > 0) for (int g = 0; g < mdl->n; ++g) grad_F[mdl->perm[mdl->dorder[g]]] = gF[g];
  else for (int k = 0; k < mdl->n; ++k) grad_F[mdl->perm[k]] = gF[k];
  return 0;
}

// SetInitialValueLRCov / SetInitialValueDeltaRelConv (re_model_template.h:8318-8347): the defaults that depend on the optimiser are resolved at the
// FIRST fit and then kept by the model -- a later fit with another optimiser inherits them (lr_cov 0.1 after a first gradient-descent fit becomes the
// initial step factor of a later lbfgs fit; delta_rel_conv 1e-6 after a first lbfgs fit is what a later simplex search stops at)
void resolve_optimizer_defaults_once(REModelHip* mdl) {
  if (!(mdl->optim.lr_cov_init > 0.)) mdl->optim.lr_cov_init = mdl->optim.optimizer == "gradient_descent" ? 0.1 : 1.;
  if (!(mdl->optim.delta_rel_conv_init > 0.)) mdl->optim.delta_rel_conv_init = mdl->optim.optimizer == "nelder_mead" ? 1e-8 : 1e-6;
}

// remember = false: the boosting seams (GPB_HIP_CalcYAux, ...), whose y is a working response of one iteration, not the model's y_vec_
int upload_y(REModelHip* mdl, const double* y_data, const double* fixed_effects, bool remember = true) {
  if (!y_data) return set_error("y_data is NULL: the HIP hot path evaluates the likelihood at the response passed in");
  const int n = mdl->n;
  if (mdl->ybuf_cap < (size_t)n) {
    if (mdl->ybuf) { gpb_hip_pinned_free(mdl->ybuf); mdl->ybuf = nullptr; mdl->ybuf_cap = 0; }
    void* p = nullptr;
    if (gpb_hip_pinned_alloc(sizeof(double) * (size_t)n, &p)) return shim_error();
    mdl->ybuf = static_cast<double*>(p); mdl->ybuf_cap = (size_t)n;
  }
  if (fixed_effects) {
    parallel_for(n, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) { const int id = mdl->perm[k]; mdl->ybuf[k] = y_data[id] - fixed_effects[id]; } });   // :2909-2915
  } else {
    parallel_for(n, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) mdl->ybuf[k] = y_data[mdl->perm[k]]; });
  }
  mdl->yaux_valid = false;
  mdl->dev_y_is_host = remember && !fixed_effects;
  if (remember && y_data != mdl->y_host.data()) mdl->y_host.assign(y_data, y_data + n);   // later calls with an offset but without y start from THIS, not from y - offset
  if (mdl->eh) { if (gpb_hip_exact_set_y(mdl->eh, mdl->ybuf)) return shim_error(); mdl->y_set = true; return 0; }
  for (size_t k = 0; k < mdl->vhs.size(); ++k)
    if (gpb_hip_vecchia_set_y(mdl->vhs[k], mdl->ybuf + mdl->cl_off[k])) return shim_error();
  mdl->y_set = true;
  return 0;
}

// The response a prediction conditions on (SetYCalcCovCalcYAuxForPred, re_model_template.h:11141-11166): y_obs if given, else y_vec_, minus this call's
// offset.  Without y_data and offset the device must hold y_vec_ itself: an earlier call may have left y - F there (an evaluation or prediction with
// fixed_effects) or a boosting seam's working response -- then y_host goes up again.
int prediction_response(REModelHip* mdl, const double* y_data, const double* fe) {
  if (fe) return upload_y(mdl, y_data ? y_data : mdl->y_host.data(), fe);
  if (y_data) return upload_y(mdl, y_data, nullptr);
  if (!mdl->dev_y_is_host && !mdl->y_host.empty()) return upload_y(mdl, mdl->y_host.data(), nullptr);
  return 0;
}

double range_const(const REModelHip* mdl) { return mdl->cov_type == 0 ? 1. : (mdl->cov_type == 1 ? std::sqrt(3.) : std::sqrt(5.)); }

// REModelTemplate::FindInitCovPar (re_model_template.h:4849-4968) -> RECompGP::FindInitCovPar (re_comp.h:1249-1267) ->
// CovFunction::FindInitCovPar (cov_fcts.h:1422-1683) for one Gaussian Matern GP.  The result is used as cov_pars_ directly, i.e. it
// is on the TRANSFORMED scale: (var(y) / 2, 1, a) with a such that the correlation is ~0.05 at half the median distance.
int find_init_cov_par_core(int n, const double* y_data, const double* fixed_effects, int nd, int d, const double* c /* column-major nd x d */,
                           int cov_type, std::mt19937& rng, double* theta) {
  double mean = 0., var = 0.;
  for (int i = 0; i < n; ++i) mean += fixed_effects ? y_data[i] - fixed_effects[i] : y_data[i];
  mean /= n;
  for (int i = 0; i < n; ++i) { const double r = (fixed_effects ? y_data[i] - fixed_effects[i] : y_data[i]) - mean; var += r * r; }
  var /= (n - 1);
  theta[0] = var / 2.;
  theta[1] = 1.;                                   // init_marg_var = 1 for the Gaussian likelihood (:4865, :4912)
  const int MAX_POINTS_INIT_RANGE = 1000;          // cov_fcts.h:1444
  const int ns = nd > MAX_POINTS_INIT_RANGE ? MAX_POINTS_INIT_RANGE : nd;
  std::vector<int> sample_ind;
  if (ns < nd) {
    std::uniform_int_distribution<> dis(0, nd - 1);
    sample_ind.resize(ns);
    for (int i = 0; i < ns; ++i) sample_ind[i] = dis(rng);
  }
  std::vector<double> distances((size_t)(ns * (ns - 1) / 2.));
  for (int i = 0; i < ns - 1; ++i) {
    const int ii = sample_ind.empty() ? i : sample_ind[i];
    for (int j = i + 1; j < ns; ++j) {
      const int jj = sample_ind.empty() ? j : sample_ind[j];
      double s2 = 0.;
      for (int k = 0; k < d; ++k) { const double df = c[(size_t)k * nd + ii] - c[(size_t)k * nd + jj]; s2 += df * df; }
      distances[(size_t)i * (2 * ns - i - 1) / 2 + j - (i + 1)] = std::sqrt(s2);
    }
  }
  // CalculateMedianPartiallySortInput (utils.h:191-204)
  const size_t num_el = distances.size(), pos_med = num_el / 2;
  std::nth_element(distances.begin(), distances.begin() + pos_med, distances.end());
  double med = distances[pos_med];
  if (num_el % 2 == 0) {
    std::nth_element(distances.begin(), distances.begin() + pos_med - 1, distances.end());
    med = (med + distances[pos_med - 1]) / 2.;
  }
  if (med < 1e-10) {                               // EPSILON_NUMBERS: fall back to the mean distance
    double sum = 0.;
    for (double v : distances) sum += v;
    med = sum / num_el;
  }
  if (med < 1e-10)
    return set_error("Cannot find an initial value for the range parameter since both the median and the average distances among coordinates are zero %s",
                     sample_ind.empty() ? "" : "on a random sub-sample of size 1000 ");
  theta[2] = cov_type == 0 ? 2. * 3. / med : (cov_type == 1 ? 2. * 4.7 / med : 2. * 5.9 / med);   // cov_fcts.h:1601-1611
  return 0;
}

int find_init_cov_par(REModelHip* mdl, const double* y_data, const double* fixed_effects, double* theta) {
  return find_init_cov_par_core(mdl->n, y_data, fixed_effects, mdl->n0, mdl->d, mdl->coords0.data(), mdl->cov_type, mdl->rng, theta);
}

// REModel::InitializeCovParsIfNotDefined (re_model.cpp:1312-1334)
int initialize_cov_pars_if_not_defined(REModelHip* mdl, const double* y_data, const double* fixed_effects) {
  if (mdl->cov_pars_initialized) return 0;
  if (mdl->init_cov_pars_provided) {
    std::copy(mdl->init_cov_pars_tr, mdl->init_cov_pars_tr + 3, mdl->cov_pars_tr);
  } else {
    if (!y_data) return set_error("y_data is NULL: initial covariance parameters cannot be determined");
    if (mdl->likelihood != "gaussian") {     // (sigma1_2, a): marginal variance 1 (re_model_template.h:4865, :4904-4913), the same range heuristic
      double th3[3];
      if (find_init_cov_par(mdl, y_data, fixed_effects, th3)) return -1;
      mdl->cov_pars_tr[0] = mdl->optim.optimizer == "nelder_mead" ? 0.1 : 1.;       // init_marg_var (re_model_template.h:4904-4909)
      if (mdl->likelihood == "gaussian_latent") {      // IsGaussianLikelihood(): half the sample variance of y - fixed effects, whatever the optimiser (:4866-4896, :4903-4906)
        double mean = 0., var = 0.;
        for (int i = 0; i < mdl->n; ++i) mean += fixed_effects ? y_data[i] - fixed_effects[i] : y_data[i];
        mean /= mdl->n;
        for (int i = 0; i < mdl->n; ++i) { const double r = (fixed_effects ? y_data[i] - fixed_effects[i] : y_data[i]) - mean; var += r * r; }
        mdl->cov_pars_tr[0] = var / (mdl->n - 1) / 2.;
      }
      mdl->cov_pars_tr[1] = th3[2]; mdl->cov_pars_tr[2] = 0.;
    } else if (find_init_cov_par(mdl, y_data, fixed_effects, mdl->cov_pars_tr)) return -1;
    std::copy(mdl->cov_pars_tr, mdl->cov_pars_tr + 3, mdl->init_cov_pars_tr);
  }
  mdl->cov_pars_initialized = true;
  return 0;
}

// ProfileOutCoef (re_model_template.h:2665-2683) with UpdateCoefGLS (:10012-10019): beta = (X' Psi^-1 X)^-1 X' Psi^-1 y0 at the covariance
// parameters of this evaluation -- the Gram matrix of B [X, y0] with weights 1 / D comes from the device in one pass -- then the response on the
// device becomes the residual y0 - X beta (UpdateFixedEffects, :2859-2871).
int profile_out_coef(REModelHip* mdl, double ratio, double a) {
  const int p = mdl->p_cov, q = p + 1;
  if (gpb_hip_vecchia_factor(mdl->vh, mdl->cov_type, ratio, a, 1)) return shim_error();
  std::vector<double> G((size_t)q * q);
  if (gpb_hip_vecchia_gram(mdl->vh, G.data())) return shim_error();
  // Cholesky of X' Psi^-1 X (Eigen's llt().solve)
  std::vector<double> L((size_t)p * p, 0.), rhs(p);
  for (int i = 0; i < p; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sacc = G[(size_t)i * q + j];
      for (int k = 0; k < j; ++k) sacc -= L[(size_t)i * p + k] * L[(size_t)j * p + k];
      if (i == j) {
        if (!(sacc > 0.)) return set_error("The matrix X' Psi^-1 X of the linear regression covariates is not positive definite (collinear covariates?)");
        L[(size_t)i * p + i] = std::sqrt(sacc);
      } else L[(size_t)i * p + j] = sacc / L[(size_t)j * p + j];
    }
    rhs[i] = G[(size_t)i * q + p];
  }
  for (int i = 0; i < p; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * p + k] * rhs[k]; rhs[i] = v / L[(size_t)i * p + i]; }
  for (int i = p - 1; i >= 0; --i) { double v = rhs[i]; for (int k = i + 1; k < p; ++k) v -= L[(size_t)k * p + i] * rhs[k]; rhs[i] = v / L[(size_t)i * p + i]; }
  mdl->beta = rhs;
  mdl->chol_XtPsiInvX = L;
  if (gpb_hip_vecchia_set_resid(mdl->vh, mdl->beta.data())) return shim_error();
  mdl->yaux_valid = false;
  return 0;
}

// Inducing points of the full-scale Vecchia approximation: kmeans++ as the reference runs it (src/GPBoost/GP_utils.cpp: random_plusplus
// :208-235 -- every new mean is a draw from std::discrete_distribution weighted by the PLAIN distance to the closest mean chosen so far --,
// calculate_means :237-280, kmeans_plusplus :282-308: Lloyd iterations until the means repeat, at most max_it) on the model's ONE
// generator, which has already shuffled the ordering (re_model_template.h:351-355).  x: column-major n x d (Vecchia order); means: column-major k x d.
int kmeans_plusplus(const std::vector<double>& x, int n, int d, int k, std::mt19937& gen, int max_it, std::vector<double>* means_out) {
  auto dist = [&](int r, const double* mean) { double s2 = 0.; for (int c = 0; c < d; ++c) { const double t = x[(size_t)c * n + r] - mean[c]; s2 += t * t; } return std::sqrt(s2); };
  std::vector<double> means((size_t)k * d, 0.), w(n, 1.0);           // means row-major here
  for (int i = 0; i < k; ++i) {
    if (i == 1) for (auto& v : w) v *= -1.;
    if (i > 0) for (int r = 0; r < n; ++r) { const double dd = dist(r, &means[(size_t)(i - 1) * d]); if (w[r] > dd || w[r] < 0) w[r] = dd; }
    double sum = 0.; for (double v : w) sum += v;
    int v;
    if (sum > 0.) v = std::discrete_distribution<>(w.data(), w.data() + n)(gen);
    else v = std::uniform_int_distribution<>(0, n - 1)(gen);
    for (int c = 0; c < d; ++c) means[(size_t)i * d + c] = x[(size_t)c * n + v];
  }
  // Lloyd iterations (calculate_means, :237-280): the n x k distances of the assignment step on the device, the ordered mean update on the host
  // (gpb_hip_kmeans_lloyd) -- 10 s of model creation at n = 1e5, k = 200 were this loop on the host's cores
  int its = 0;
  if (gpb_hip_kmeans_lloyd(n, d, x.data(), k, means.data(), max_it, &its)) return shim_error();
  means_out->assign((size_t)k * d, 0.);
  for (int j = 0; j < k; ++j) for (int c = 0; c < d; ++c) (*means_out)[(size_t)c * k + j] = means[(size_t)j * d + c];
  return 0;
}

// lower Cholesky factor of the k x k row-major matrix M (in place, upper part zeroed); false if not positive definite
bool cholesky_lower(std::vector<double>& M, int k) {
  for (int i = 0; i < k; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sacc = M[(size_t)i * k + j];
      for (int q = 0; q < j; ++q) sacc -= M[(size_t)i * k + q] * M[(size_t)j * k + q];
      if (i == j) { if (!(sacc > 0.)) return false; M[(size_t)i * k + i] = std::sqrt(sacc); }
      else M[(size_t)i * k + j] = sacc / M[(size_t)j * k + j];
    }
    for (int j = i + 1; j < k; ++j) M[(size_t)i * k + j] = 0.;
  }
  return true;
}

// Full-scale Vecchia, Gaussian likelihood: y' Psi^-1 y and log|Psi| at (ratio, a) by the Woodbury identity with the residual-process
// Vecchia factor on the device (CalcSigmaComps re_model_template.h:8151-8200, CalcCovFactorFITC_FSA :9646-9745, CalcYAux :9785-9806,
// the log-determinant :2950-2966) and -- with_grad -- their derivatives wrt (log ratio, log a): the analytic gradient of
// CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i (:2205-2330, 2447-2452) in the form of DESIGN.md 4.12.  The k x k work -- Sigma_m, its Cholesky
// factor and inverse, the Woodbury matrix and its inverse, the traces with dSigma_m -- is host work (k <= 256); everything with an n in it is the
// device's: `factor` = gpb_hip_vecchia_vif_factor, `gsums` = gpb_hip_vecchia_vif_grad_sums (function pointers so that the CPU suite can drive this
// routine with a numpy restatement of the two, GPB_HIP_VifTermsWithCallback).
struct VifSolve { std::vector<double> Linv, Lw, v; };      // for the prediction: L_m^-1, the Cholesky factor of the Woodbury matrix, W^-1 (B C)' D^-1 B y
typedef int (*vif_factor_fn)(void* ctx, const double* Linv, int with_grad, double* out3, double* G);
typedef int (*vif_gsums_fn)(void* ctx, const double* Winv, const double* Si, const double* N0, const double* negMp1, const double* w, double* sums12);

// C = A B for k x k row-major matrices
void matmul_kk(const std::vector<double>& A, const std::vector<double>& B, int k, std::vector<double>* C) {
  C->assign((size_t)k * k, 0.);
  parallel_for(k, [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i)
      for (int q = 0; q < k; ++q) {
        const double aiq = A[(size_t)i * k + q];
        if (aiq == 0.) continue;
        for (int j = 0; j < k; ++j) (*C)[(size_t)i * k + j] += aiq * B[(size_t)q * k + j];
      }
  });
}
// inverse of a lower-triangular k x k row-major matrix (row-major lower-triangular result), column by column
void tri_inverse_lower(const std::vector<double>& L, int k, std::vector<double>* Linv) {
  Linv->assign((size_t)k * k, 0.);
  for (int c = 0; c < k; ++c) {
    (*Linv)[(size_t)c * k + c] = 1. / L[(size_t)c * k + c];
    for (int i = c + 1; i < k; ++i) {
      double sacc = 0.;
      for (int q = c; q < i; ++q) sacc -= L[(size_t)i * k + q] * (*Linv)[(size_t)q * k + c];
      (*Linv)[(size_t)i * k + c] = sacc / L[(size_t)i * k + i];
    }
  }
}
// (L L')^-1 = Linv' Linv from the lower-triangular inverse
void spd_inverse_from_tri(const std::vector<double>& Linv, int k, std::vector<double>* inv) {
  inv->assign((size_t)k * k, 0.);
  parallel_for(k, [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i)
      for (int j = 0; j <= i; ++j) {
        double sacc = 0.;
        for (int q = i; q < k; ++q) sacc += Linv[(size_t)q * k + i] * Linv[(size_t)q * k + j];      // rows q >= max(i, j) = i
        (*inv)[(size_t)i * k + j] = sacc;
      }
  });
  for (int i = 0; i < k; ++i) for (int j = i + 1; j < k; ++j) (*inv)[(size_t)i * k + j] = (*inv)[(size_t)j * k + i];
}

int vif_terms_core(int k, int d, const double* ip_colmajor, int cov_type, double ratio, double a, int with_grad, vif_factor_fn factor, vif_gsums_fn gsums,
                   void* ctx, double* t7, VifSolve* keep) {
  std::vector<double> Sm0((size_t)k * k), dSm1;
  if (with_grad) dSm1.assign((size_t)k * k, 0.);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= i; ++j) {
      double s2 = 0.;
      for (int c = 0; c < d; ++c) { const double t = ip_colmajor[(size_t)c * k + i] - ip_colmajor[(size_t)c * k + j]; s2 += t * t; }
      const double r = a * std::sqrt(s2), e = ratio * std::exp(-r);
      const double kv = cov_type == 0 ? e : (cov_type == 1 ? e * (1. + r) : e * (1. + r + r * r / 3.));
      Sm0[(size_t)i * k + j] = Sm0[(size_t)j * k + i] = kv;
      if (with_grad) {      // d/d log a (GradientRangeMaternShape0_5 / 1_5 / 2_5 with transf_scale, cov_fcts.h:2535-2554)
        const double dk = cov_type == 0 ? -r * e : (cov_type == 1 ? -r * r * e : -r * r * (1. + r) * e / 3.);
        dSm1[(size_t)i * k + j] = dSm1[(size_t)j * k + i] = dk;
      }
    }
  std::vector<double> Sm = Sm0;
  for (int i = 0; i < k; ++i) Sm[(size_t)i * k + i] *= 1. + 1e-6;                   // JITTER_MULT_IP_FITC_FSA (utils.h:41): only what is FACTORISED carries it
  std::vector<double> L = Sm;
  if (!cholesky_lower(L, k)) return set_error("The covariance matrix of the inducing points is not positive definite");
  std::vector<double> Linv;
  tri_inverse_lower(L, k, &Linv);
  double o3[3];
  const int q = k + 1;
  std::vector<double> G((size_t)q * q);
  if (factor(ctx, Linv.data(), with_grad, o3, G.data())) return -1;                // device: (B [C_nm, y])' D^-1 (B [C_nm, y]) and the factor's sums
  std::vector<double> W((size_t)k * k), r(k);
  for (int i = 0; i < k; ++i) { for (int j = 0; j < k; ++j) W[(size_t)i * k + j] = Sm[(size_t)i * k + j] + G[(size_t)i * q + j]; r[i] = G[(size_t)i * q + k]; }
  if (!cholesky_lower(W, k)) return set_error("The Woodbury matrix of the full-scale Vecchia approximation is not positive definite");
  double ldm = 0., ldw = 0.;
  for (int i = 0; i < k; ++i) { ldm += std::log(L[(size_t)i * k + i]); ldw += std::log(W[(size_t)i * k + i]); }
  for (int i = 0; i < k; ++i) { double v = r[i]; for (int j = 0; j < i; ++j) v -= W[(size_t)i * k + j] * r[j]; r[i] = v / W[(size_t)i * k + i]; }   // L_W^-1 r
  double rr = 0.; for (int i = 0; i < k; ++i) rr += r[i] * r[i];
  std::vector<double> w = r;                                                       // w = L_W^-T (L_W^-1 r) = W^-1 (B C)' D^-1 B y
  for (int i = k - 1; i >= 0; --i) { double v = w[i]; for (int j = i + 1; j < k; ++j) v -= W[(size_t)j * k + i] * w[j]; w[i] = v / W[(size_t)i * k + i]; }
  if (keep) { keep->Linv = Linv; keep->Lw = W; keep->v = w; }
  t7[0] = o3[0] - rr;
  t7[1] = o3[1] - 2. * ldm + 2. * ldw;
  t7[2] = o3[2];
  if (!with_grad) return 0;
  // ---- gradient -------------------------------------------------------------------------------------------------------------------------
  std::vector<double> LwInv, Winv, Si, T, Mp0, Mp1;
  tri_inverse_lower(W, k, &LwInv);
  spd_inverse_from_tri(LwInv, k, &Winv);
  spd_inverse_from_tri(Linv, k, &Si);
  matmul_kk(Si, Sm0, k, &T); matmul_kk(T, Si, k, &Mp0);                            // Si dSm^var Si   (dSm^var = the UN-jittered Sigma_m, as the reference's GetZSigmaZtGrad)
  matmul_kk(Si, dSm1, k, &T); matmul_kk(T, Si, k, &Mp1);                           // Si dSm^range Si
  std::vector<double> N0((size_t)k * k), negMp1((size_t)k * k);
  for (size_t e = 0; e < N0.size(); ++e) { N0[e] = 2. * Si[e] - Mp0[e]; negMp1[e] = -Mp1[e]; }
  double S[12];
  if (gsums(ctx, Winv.data(), Si.data(), N0.data(), negMp1.data(), w.data(), S)) return -1;
  for (int p = 0; p < 2; ++p) {
    const std::vector<double>& dSm = p == 0 ? Sm0 : dSm1;
    double wdw = 0., trSi = 0., trW = 0.;
    for (int i = 0; i < k; ++i) {
      double acc = 0.;
      for (int j = 0; j < k; ++j) { const double v = dSm[(size_t)i * k + j]; acc += v * w[j]; trSi += Si[(size_t)i * k + j] * v; trW += Winv[(size_t)i * k + j] * v; }
      wdw += w[i] * acc;
    }
    const double S1 = S[0 + p], S2 = S[2 + p], S3 = S[4 + p], S4 = S[6 + p], S5 = S[8 + p], S6 = S[10 + p];
    const double dquad = S2 - 2. * S6 + wdw;                                        // d(y' Psi^-1 y) / d log theta_p
    const double dlogdet = S1 - trSi + trW + 2. * S5 + 2. * S3 - S4;                // d log|Psi| / d log theta_p
    t7[3 + 2 * p] = 0.5 * dquad;
    t7[4 + 2 * p] = 0.5 * dlogdet;
  }
  return 0;
}

int vif_terms(REModelHip* mdl, double ratio, double a, double* t7, VifSolve* keep = nullptr, int with_grad = 0, int keep_grad_factor = 0) {
  struct Ctx { REModelHip* mdl; double ratio, a; int keep; } c{mdl, ratio, a, keep_grad_factor};
  const vif_factor_fn factor = [](void* ctx, const double* Linv, int wg, double* out3, double* G) -> int {
    auto* x = reinterpret_cast<Ctx*>(ctx);
    return gpb_hip_vecchia_vif_factor(x->mdl->vh, x->mdl->cov_type, x->ratio, x->a, Linv, wg, out3, G) ? shim_error() : 0;
  };
  const vif_gsums_fn gsums = [](void* ctx, const double* Winv, const double* Si, const double* N0, const double* negMp1, const double* w, double* sums12) -> int {
    auto* x = reinterpret_cast<Ctx*>(ctx);
    return gpb_hip_vecchia_vif_grad_sums(x->mdl->vh, x->mdl->cov_type, x->ratio, x->a, Winv, Si, N0, negMp1, w, x->keep, sums12) ? shim_error() : 0;
  };
  double t[7] = {0, 0, 0, 0, 0, 0, 0};
  const int rc = vif_terms_core(mdl->num_ind_points, mdl->d, mdl->ip.data(), mdl->cov_type, ratio, a, with_grad, factor, gsums, &c, t, keep);
  if (rc) return -1;
  for (int q = 0; q < (with_grad ? 7 : 3); ++q) t7[q] = t[q];
  mdl->yaux_valid = false;
  return 0;
}

// the optimiser's window on the device: the shard sums of all clusters at (ratio, a); y is already resident
int device_terms(void* ctx, double ratio, double a, int with_grad, double* t7) {
  auto* mdl = reinterpret_cast<REModelHip*>(ctx);
  for (int q = 0; q < 7; ++q) t7[q] = 0.;
  if (mdl->vif) {
    return vif_terms(mdl, ratio, a, t7, nullptr, with_grad);      // with_grad: the analytic gradient, two more passes over the n x k matrices
  }
  // covariate fit: response := y0 - X beta_GLS(ratio, a) before the terms are evaluated.  Only there -- GPB_EvalNegLogLikelihood and a later
  // GPB_OptimCovPar stay plain evaluations of y - fixed_effects (re_model.cpp:755-790), whatever was fitted before
  if (mdl->fitting_with_covariates && mdl->p_cov > 0 && !mdl->coef_by_iteration && profile_out_coef(mdl, ratio, a)) return -1;
  if (mdl->eh) {      // exact GP (gp_approx "none"): dense Cholesky; the gradient through one partial factorisation of [[Psi, .], [I, 0]]
    if (with_grad) { if (gpb_hip_exact_grad_terms(mdl->eh, mdl->cov_type, ratio, a, t7)) return shim_error(); }
    else if (gpb_hip_exact_nll_terms(mdl->eh, mdl->cov_type, ratio, a, t7, nullptr, nullptr)) return shim_error();
    return 0;
  }
  for (auto* v : mdl->vhs) {
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    int world = 0;
    if (gpb_hip_vecchia_comm_info(v, nullptr, &world)) return shim_error();
    int rc;
    if (world > 0)      // sharded handle: job-wide sums on every rank (one ncclAllReduce of 3 / 7 doubles per evaluation)
      rc = with_grad ? gpb_hip_vecchia_grad_terms_allreduce(v, mdl->cov_type, ratio, a, t) : gpb_hip_vecchia_nll_terms_allreduce(v, mdl->cov_type, ratio, a, 1, t);
    else
      rc = with_grad ? gpb_hip_vecchia_grad_terms(v, mdl->cov_type, ratio, a, t) : gpb_hip_vecchia_nll_terms(v, mdl->cov_type, ratio, a, 1, t);
    if (rc) return shim_error();
    for (int q = 0; q < (with_grad ? 7 : 3); ++q) t7[q] += t[q];
  }
  return 0;
}

void transform_back(const REModelHip* mdl, const double* tr, double* orig) {   // TransformBackCovPars (cov_fcts.h:560-600)
  orig[0] = tr[0];
  orig[1] = tr[1] * tr[0];
  orig[2] = range_const(mdl) / tr[2];
}

const char* kDuplicatesNonGaussianMessage =
    "Duplicates found in the coordinates for the Gaussian process. This is currently not supported for the Vecchia approximation for non-Gaussian likelihoods ";   // Vecchia_utils.cpp:1211-1214

// CanCalculateStandardErrorsCovPars (re_model_template.h:1804-1807) restricted to what GPB_GetCovPar(calc_std_dev = true) does on the device
// what gpb_hip_vecchia_fisher_std_errors covers (its per-point derivative kernel: m <= 62, d <= 3; an unsharded handle): the capability
// query must not promise more than GPB_GetCovPar(calc_std_dev) delivers
bool can_calc_std_dev(const REModelHip* mdl) {
  if (mdl->likelihood == "gaussian" && mdl->eh) return mdl->n <= 24000;      // exact GP: the dense Fisher information (gpb_hip_exact_fisher_std_errors)
  if (mdl->likelihood != "gaussian")      // non-Gaussian: numerical Jacobian of the gradient of the Laplace approximation (its derivative kernel's limits)
    return !mdl->eh && mdl->vhs.size() == 1 && (!mdl->vif || mdl->cg_preconditioner_type == "fitc") && std::min(mdl->m, (mdl->n_re > 0 ? mdl->n_re : mdl->n) - 1) <= 126 && mdl->d <= 3;      // (full-scale Vecchia: its gradient is built for "fitc")
  if (mdl->eh || mdl->vhs.size() != 1 || mdl->vif || mdl->has_weights) return false;
  int world = 0;
  if (gpb_hip_vecchia_comm_info(mdl->vhs[0], nullptr, &world) || world > 1) return false;
  return std::min(mdl->m, mdl->n - 1) <= 126 && mdl->d <= 3;
}

// DetermineUniqueDuplicateCoordsFast (src/GPBoost/GP_utils.cpp:472-548) as RECompGP uses it for one non-Gaussian GP (re_comp.h:863-885):
// uniques = positions of the FIRST appearance of every distinct location, ascending (two locations are the same if their squared distance
// is below EPSILON_NUMBERS^2 = 1e-20); unique_idx[i] = index into uniques of point i.  Candidates share their coordinate sum.
// coords: column-major n x d.
// ---- response-scale predictions of the non-Gaussian likelihoods (Likelihood::PredictResponse, include/GPBoost/likelihoods.h:9626-9672) ----
// Gauss-Hermite rule of the given order (weight exp(-x^2)): nodes x_j ascending and the ADAPTIVE weights w_j exp(x_j^2) the reference tabulates as
// GH_nodes_ / adaptive_GH_weights_ (:17472-17576, order_GH_ = 30).  Computed: the nodes are the eigenvalues of the Jacobi matrix (zero diagonal,
// off-diagonals sqrt(k / 2)) -- bisection on its Sturm sequence, polished by Newton steps on the orthonormal Hermite recurrence; the weights
// are the Christoffel numbers 1 / sum_{k < n} p_k(x_j)^2.
void gauss_hermite_adaptive(int order, std::vector<double>* nodes, std::vector<double>* adaptive_weights) {
  const int n = order;
  nodes->assign(n, 0.0); adaptive_weights->assign(n, 0.0);
  auto count_below = [n](double x) {          // eigenvalues of the Jacobi matrix that are < x
    int cnt = 0;
    double dk = -x;
    if (dk < 0.) ++cnt;
    for (int k = 1; k < n; ++k) {
      if (dk == 0.) dk = 1e-300;
      dk = -x - (0.5 * k) / dk;
      if (dk < 0.) ++cnt;
    }
    return cnt;
  };
  const double pi_m14 = std::pow(M_PI, -0.25);
  auto poly = [n, pi_m14](double x, double* pn1, double* sumsq) {      // p_n(x); p_{n-1}(x); sum_{k<n} p_k^2
    double pm = 0.0, pc = pi_m14, acc = 0.0;
    for (int k = 0; k < n; ++k) {
      acc += pc * pc;
      const double pnext = x * std::sqrt(2.0 / (k + 1)) * pc - std::sqrt((double)k / (k + 1)) * pm;
      pm = pc; pc = pnext;
    }
    *pn1 = pm; *sumsq = acc;
    return pc;
  };
  const double bound = std::sqrt(2.0 * n + 1.0) + 1.0;
  for (int j = 0; j < n; ++j) {
    double lo = -bound, hi = bound;                // the (j + 1)-th smallest eigenvalue
    for (int it = 0; it < 80; ++it) {
      const double mid = 0.5 * (lo + hi);
      if (count_below(mid) > j) hi = mid; else lo = mid;
    }
    double x = 0.5 * (lo + hi), pn1 = 0., ss = 0.;
    for (int it = 0; it < 4; ++it) {               // p_n' = sqrt(2 n) p_{n-1}
      const double pn = poly(x, &pn1, &ss);
      const double dp = std::sqrt(2.0 * n) * pn1;
      if (dp == 0.) break;
      x -= pn / dp;
    }
    poly(x, &pn1, &ss);
    (*nodes)[j] = x;
    (*adaptive_weights)[j] = std::exp(x * x) / ss;
  }
  for (int j = 0; j < n / 2; ++j) {                // exact symmetry
    const double xs = 0.5 * ((*nodes)[n - 1 - j] - (*nodes)[j]), ws = 0.5 * ((*adaptive_weights)[j] + (*adaptive_weights)[n - 1 - j]);
    (*nodes)[j] = -xs; (*nodes)[n - 1 - j] = xs; (*adaptive_weights)[j] = ws; (*adaptive_weights)[n - 1 - j] = ws;
  }
  if (n % 2) (*nodes)[n / 2] = 0.0;
}

inline double normal_pdf(double x) { return std::exp(-0.5 * x * x) / std::sqrt(2.0 * M_PI); }
inline double normal_cdf(double x) { return 0.5 * std::erfc(-x * M_SQRT1_2); }

// E[sigmoid(b)], b ~ N(latent_mean, latent_var): RespMeanAdaptiveGHQuadrature (likelihoods.h:10128-10160) for the Bernoulli-logit likelihood --
// Newton from 0 to the mode of sigmoid(x) N(x; m, v) (at most 100 steps, stop at |update| / |previous value| < delta), then the adaptive rule
double resp_mean_logit(double latent_mean, double latent_var, double delta, const std::vector<double>& xs, const std::vector<double>& aw) {
  const double s2i = 1.0 / latent_var, ss = std::sqrt(s2i);
  double mode = 0.0;
  for (int it = 0; it < 100; ++it) {
    const double last = mode;
    const double p = 1.0 / (1.0 + std::exp(-mode));
    const double upd = ((1.0 - p) - s2i * (mode - latent_mean)) / (-p * (1.0 - p) - s2i);   // d log sigmoid = 1 - p, d2 = -p (1 - p)
    mode -= upd;
    if (std::fabs(upd) / std::fabs(last) < delta) break;
  }
  const double p = 1.0 / (1.0 + std::exp(-mode));
  const double sh = M_SQRT2 / std::sqrt(p * (1.0 - p) + s2i);
  double acc = 0.0;
  for (size_t j = 0; j < xs.size(); ++j) {
    const double x = sh * xs[j] + mode;
    acc += aw[j] * (1.0 / (1.0 + std::exp(-x))) * normal_pdf(ss * (x - latent_mean));
  }
  return acc * sh * ss;
}

// in place: latent (mean, var) -> response (mean, var if predict_var); false = likelihood not on the path
// beta: E[Var(y | b)] with Var(y | b) = mu (1 - mu) / (1 + precision), mu = sigmoid(b): ExpectedValueCondRespVarAdaptiveGHQuadrature (likelihoods.h:10167-10193) with
// CondVarLikelihood / its log-derivatives (:15633-15680); second: E[sigmoid(b)^2], RespMeanAdaptiveGHQuadrature(second_moment = true) (:10128-10160)
double resp_gh_beta(double latent_mean, double latent_var, double delta, const std::vector<double>& xs, const std::vector<double>& aw, int what, double aux) {
  const double s2i = 1.0 / latent_var, ss = std::sqrt(s2i);
  double mode = 0.0;
  auto d1 = [&](double x) { return what == 2 ? (-1. + 2. / (1. + std::exp(x))) : 2.0 * (1.0 / (1.0 + std::exp(x))); };         // what 1: c_mult = 2 times d log sigmoid = sigmoid(-x)
  auto d2 = [&](double x) { const double e = std::exp(x); return what == 2 ? -2 * e / ((1. + e) * (1. + e)) : 2.0 * -(1.0 / (1.0 + std::exp(-x))) * (1.0 - 1.0 / (1.0 + std::exp(-x))); };
  for (int it = 0; it < 100; ++it) {
    const double last = mode;
    const double upd = (d1(mode) - s2i * (mode - latent_mean)) / (d2(mode) - s2i);
    mode -= upd;
    if (std::fabs(upd) / std::fabs(last) < delta) break;
  }
  const double sh = M_SQRT2 / std::sqrt(-d2(mode) + s2i);
  double acc = 0.0;
  for (size_t j = 0; j < xs.size(); ++j) {
    const double x = sh * xs[j] + mode;
    double f;
    if (what == 2) { const double em = std::exp(-x); f = em / ((1. + em) * (1. + em)) / (1. + aux); }
    else { const double p = 1.0 / (1.0 + std::exp(-x)); f = p * p; }
    acc += aw[j] * f * normal_pdf(ss * (x - latent_mean));
  }
  return acc * sh * ss;
}

bool predict_response_host(const std::string& lik, int n, double* mean, double* var, bool predict_var, double delta, double aux = 1.0) {
  if (lik == "t") {                         // likelihoods.h:9824-9832: the mean is the latent mean; the squared scale is added to the latent variances
    if (predict_var) for (int i = 0; i < n; ++i) var[i] += aux * aux;
    return true;
  }
  if (lik == "beta") {                      // likelihoods.h:9805-9824: mean E[sigmoid(b)]; variance Var(E[y | b]) + E[Var(y | b)]
    std::vector<double> xs, aw;
    gauss_hermite_adaptive(30, &xs, &aw);
    for (int i = 0; i < n; ++i) {
      const double rm = resp_mean_logit(mean[i], var[i], delta, xs, aw);
      if (predict_var) {
        const double var_E = resp_gh_beta(mean[i], var[i], delta, xs, aw, 1, aux) - rm * rm;
        const double E_var = resp_gh_beta(mean[i], var[i], delta, xs, aw, 2, aux);
        var[i] = var_E + E_var;
      }
      mean[i] = rm;
    }
    return true;
  }
  if (lik == "gaussian_latent") {           // likelihoods.h:9853-9857: the latent mean; variance + the error variance
    if (predict_var) for (int i = 0; i < n; ++i) var[i] += aux;
    return true;
  }
  if (lik == "lognormal") {                 // likelihoods.h:9868-9888: mean exp(m + v / 2); variance Var(E[y | b]) + E[Var(y | b)] with aux = variance of log y
    const double exp_s2_m1 = std::expm1(aux);
    for (int i = 0; i < n; ++i) {
      const double pm = std::exp(mean[i] + 0.5 * var[i]);
      if (predict_var) { const double exp_v_m1 = std::expm1(var[i]), pm2 = pm * pm; var[i] = exp_v_m1 * pm2 + exp_s2_m1 * pm2 * (exp_v_m1 + 1.); }
      mean[i] = pm;
    }
    return true;
  }
  if (lik == "gamma") {                     // likelihoods.h:9715-9728
    for (int i = 0; i < n; ++i) {
      const double pm = std::exp(mean[i] + 0.5 * var[i]);
      if (predict_var) var[i] = (std::exp(var[i]) - 1.) * pm * pm + std::exp(2 * mean[i] + 2 * var[i]) / aux;
      mean[i] = pm;
    }
    return true;
  }
  if (lik == "negative_binomial") {         // likelihoods.h:9783-9793
    for (int i = 0; i < n; ++i) {
      const double pm = std::exp(mean[i] + 0.5 * var[i]);
      if (predict_var) var[i] = std::exp(2 * (mean[i] + var[i])) * (1 + 1 / aux) + pm * (1 - pm);
      mean[i] = pm;
    }
    return true;
  }
  if (is_probit_link(lik)) {             // (binomial_probit / quasi_bernoulli_probit alike, likelihoods.h:9631-9643)
    for (int i = 0; i < n; ++i) { mean[i] = normal_cdf(mean[i] / std::sqrt(1.0 + var[i])); if (predict_var) var[i] = mean[i] * (1.0 - mean[i]); }
    return true;
  }
  if (is_logit_link(lik)) {              // (binomial_logit / quasi_bernoulli_logit alike, :9645-9659)
    std::vector<double> xs, aw;
    gauss_hermite_adaptive(30, &xs, &aw);
    for (int i = 0; i < n; ++i) { mean[i] = resp_mean_logit(mean[i], var[i], delta, xs, aw); if (predict_var) var[i] = mean[i] * (1.0 - mean[i]); }
    return true;
  }
  if (lik == "poisson") {
    for (int i = 0; i < n; ++i) {
      const double pm = std::exp(mean[i] + 0.5 * var[i]);
      if (predict_var) var[i] = pm * ((std::exp(var[i]) - 1.0) * pm + 1.0);
      mean[i] = pm;
    }
    return true;
  }
  return false;
}

// ---- non-Gaussian models with a linear predictor: host preparation of the fit (REModelTemplate::OptimLinRegrCoefCovPar, re_model_template.h:1112-1300) ----
// Phi^-1(p): the reference uses Wichura's AS 241 (DF_utils.h:256); here Newton steps on the erfc-based normal_cdf from a rational start (the two
// agree to the last digits of a double; the value only seeds the intercept of a probit model)
double normal_quantile(double p) {
  if (!(p > 0.)) return -std::numeric_limits<double>::infinity();
  if (!(p < 1.)) return std::numeric_limits<double>::infinity();
  const double t = std::sqrt(-2. * std::log(p < 0.5 ? p : 1. - p));
  double x = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1. + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);   // Abramowitz & Stegun 26.2.23
  if (p < 0.5) x = -x;
  for (int it = 0; it < 8; ++it) {
    const double f = normal_cdf(x) - p, d = normal_pdf(x);
    if (!(d > 0.)) break;
    const double step = f / d;
    x -= step / (1. + 0.5 * x * step);          // Halley
    if (std::fabs(step) < 1e-16 * std::max(1., std::fabs(x))) break;
  }
  return x;
}

struct LaplaceCoefSetup {
  int n = 0, p = 0;
  bool has_intercept = false, scale = false;
  int intercept_col = -1;
  std::vector<double> Xs, loc, scl;     // covariates as the optimiser sees them (column-major n x p, data order) and their transformation
  std::vector<double> beta;             // initial coefficients on that scale
  double C_mu = 1., C_sigma2 = 1.;
};

// intercept detection (:1114-1133), scaling of the covariates (:1218-1242), initial coefficients (:1243-1275: zeros, or init_coef transformed; without
// init_coef the intercept starts at Likelihood::FindInitialIntercept, likelihoods.h:1455-1540) and the constants of the step cap
// (FindConstantsCapTooLargeLearningRateCoef, likelihoods.h:2618-2664).  init_var = total variance of the random effects at the initial parameters.
// w (round 6): the sample weights in DATA order (GetWeightsAllClusters, re_model_template.h:8385-8399), or nullptr: they weight the start of the intercept
// (FindInitialIntercept, likelihoods.h:1455-1560) and the constants of the step cap (FindConstantsCapTooLargeLearningRateCoef, :2618-2660); the scaling of the covariates does not use them
int laplace_coef_setup(const std::string& lik, int n, int p, const double* X, const double* y, const double* fixed_effects, double init_var,
                       const double* init_coef, LaplaceCoefSetup* s, const double* w = nullptr) {
  s->n = n; s->p = p;
  s->Xs.assign(X, X + (size_t)n * p);
  for (int j = 0; j < p && !s->has_intercept; ++j) {
    const double* col = X + (size_t)j * n;
    bool constant = true;
    for (int i = 1; i < n && constant; ++i)
      constant = std::fabs(col[i] - col[0]) < 1e-10 * std::max({1.0, std::fabs(col[i]), std::fabs(col[0])});      // TwoNumbersAreEqual (utils.h:54-56)
    if (constant) { s->has_intercept = true; s->intercept_col = j; }
  }
  s->scale = !(s->has_intercept && p == 1);       // optimizer_cov 'lbfgs' with the coefficients in its vector (:1220-1222)
  s->loc.assign(p, 0.); s->scl.assign(p, 1.);
  if (s->scale)
    for (int j = 0; j < p; ++j) {
      if (s->has_intercept && j == s->intercept_col) continue;
      double* col = s->Xs.data() + (size_t)j * n;
      double mean = 0.;
      for (int i = 0; i < n; ++i) mean += col[i];
      mean /= n;
      double ss = 0.;
      for (int i = 0; i < n; ++i) { col[i] -= mean; ss += col[i] * col[i]; }
      const double sd = std::sqrt(ss / n);
      if (!(sd > 0.)) return set_error("GPB_OptimLinRegrCoefCovPar: covariate %d is constant (a second intercept)", j + 1);
      for (int i = 0; i < n; ++i) col[i] /= sd;
      s->loc[j] = mean; s->scl[j] = sd;
    }
  s->beta.assign(p, 0.);
  if (init_coef) {
    std::copy(init_coef, init_coef + p, s->beta.begin());
    if (s->scale) {                                // TransformCoef (:8083-8103)
      for (int j = 0; j < p; ++j) {
        if (s->has_intercept && j == s->intercept_col) continue;
        if (s->has_intercept) s->beta[s->intercept_col] += s->beta[j] * s->loc[j];
        s->beta[j] *= s->scl[j];
      }
    }
  } else if (s->has_intercept) {
    if (!(init_var > 0.)) return set_error("GPB_OptimLinRegrCoefCovPar: the initial marginal variance must be positive");
    double b0;
    if (lik == "bernoulli_logit" || lik == "bernoulli_probit") {
      double sy = 0., sw = 0.;
      for (int i = 0; i < n; ++i) { const double wi = w ? w[i] : 1.0; sy += wi * y[i]; sw += wi; }
      double pavg = (sy > 0. && sw > 0.) ? sy / sw : 0.5;
      pavg = std::min(std::max(pavg, 1e-12), 1. - 1e-12);
      b0 = lik == "bernoulli_logit" ? std::log(pavg) - std::log1p(-pavg) : normal_quantile(pavg);
      b0 = std::min(std::max(b0, -3.0), 3.0);
    } else if (lik == "poisson") {
      double avg = 0., sw = 0.;
      for (int i = 0; i < n; ++i) { const double wi = w ? w[i] : 1.0; avg += wi * (fixed_effects ? y[i] / std::exp(fixed_effects[i]) : y[i]); sw += wi; }
      avg = std::max(avg / sw, 1e-12);
      b0 = std::log(avg) - 0.5 * init_var;
    } else return set_error("GPB_OptimLinRegrCoefCovPar: likelihood '%s' is not on the MI355X hot path of this library", lik.c_str());
    s->beta[s->intercept_col] = b0;
  }
  if (lik == "poisson") {
    double mean = 0., sec = 0., sw = 0.;
    for (int i = 0; i < n; ++i) { const double wi = w ? w[i] : 1.0; mean += wi * y[i]; sec += wi * y[i] * y[i]; sw += wi; }
    mean /= sw; sec /= sw;
    const double var = sec - mean * mean;
    s->C_mu = std::fabs(mean > 0. ? std::log(mean) : -std::numeric_limits<double>::infinity());
    s->C_sigma2 = std::fabs(var > 0. ? std::log(var) : -std::numeric_limits<double>::infinity());
  } else { s->C_mu = 1.; s->C_sigma2 = 1.; }
  if (s->C_mu < 1.) s->C_mu = 1.;                       // likelihoods.h:2741-2743
  return 0;
}

// ---- initial coefficients from the "iid model" (REModel::InitCoefAuxParsFromIidModel, src/GPBoost/re_model.cpp:380-470; the packages' default
// init_coef_aux_pars_from_iid_model = true): the same likelihood WITHOUT the Gaussian process -- one grouped random effect whose variance is set to
// 1e-20 and whose mode stays at zero (iid_model_, re_model_template.h:451-456, 986-995; likelihoods.h:3281-3293), i.e. a plain GLM: objective
// -LogLikelihood(F + X beta), gradient X' (-d log p / d loc) -- fitted with the same lbfgs over the (scaled) coefficients only, at least 1000
// iterations allowed.  O(n p) host work per evaluation, once per fit, no device involved (there is no GP in it).
double log1p_exp_neg_abs_plus_max(double x) { return std::log1p(std::exp(-std::fabs(x))) + std::max(x, 0.0); }      // GPBoost::softplus (DF_utils.h:57-60)
double normal_log_cdf(double x) {                                             // GPBoost::normalLogCDF (DF_utils.h:74-92)
  if (x < 0.0) {
    const double e = std::erfc(-x * M_SQRT1_2);
    if (e > 0.0) return std::log(0.5) + std::log(e);
    const double u = -x, u2 = u * u;
    return -0.5 * u2 - std::log(u) - 0.5 * std::log(2 * M_PI) + std::log(1.0 - 1.0 / u2 + 3.0 / (u2 * u2));
  }
  const double Q = 0.5 * std::erfc(x * M_SQRT1_2);
  return Q == 0.0 ? 0.0 : std::log1p(-Q);
}
struct GlmCtx { int lik; int n; const double* y; double log_norm_const; const double* w = nullptr; };     // lik: 0 logit, 1 probit, 2 Poisson; w: sample weights (data order) or nullptr
int glm_eval(void* ctx, int op, double, double, const double* fe, double* out3, double* grad_F) {
  const auto* c = static_cast<const GlmCtx*>(ctx);
  if (op == 3 || op == 4) return 0;
  if (op == 0 || op == 1) {
    double ll = 0.;
    for (int i = 0; i < c->n; ++i) {
      const double x = fe[i], y = c->y[i];
      const double wi = c->w ? c->w[i] : 1.0;                                                // Likelihood::weights_ multiplies every per-datum term (likelihoods.h:666-668)
      if (c->lik == 0) ll += wi * (y * x - log1p_exp_neg_abs_plus_max(x));                   // LogLikBernoulliLogit (likelihoods.h:11401-11403)
      else if (c->lik == 1) ll += wi * (y > 0.5 ? normal_log_cdf(x) : normal_log_cdf(-x));   // LogLikBernoulliProbit (:11385-11392)
      else ll += wi * (y * x - std::exp(x));                                                 // LogLikPoisson (:11407-11415)
    }
    out3[0] = -(ll + c->log_norm_const);
    if (op == 0) return 0;
  }
  out3[1] = 0.; out3[2] = 0.;
  for (int i = 0; i < c->n; ++i) {
    const double x = fe[i], y = c->y[i];
    double first;
    if (c->lik == 0) first = y - (x >= 0. ? 1. / (1. + std::exp(-x)) : std::exp(x) / (1. + std::exp(x)));
    else if (c->lik == 1) {
      const double z = y > 0.5 ? x : -x;
      const double r = std::exp(-0.5 * z * z - 0.5 * std::log(2 * M_PI) - normal_log_cdf(z));   // InvMillsRatio
      first = y > 0.5 ? r : -r;
    } else first = y - std::exp(x);
    grad_F[i] = -(c->w ? c->w[i] : 1.0) * first;
  }
  return 0;
}

void transform_back_coef(const LaplaceCoefSetup& s, std::vector<double>& beta);
int iid_model_init_coef(const std::string& lik, int n, int p, const double* X, const double* y, const double* fixed_effects, const GpbOptimConfig& cfg,
                        std::vector<double>* coef_out, int* num_it_out, const double* w = nullptr) {
  LaplaceCoefSetup su;
  if (laplace_coef_setup(lik, n, p, X, y, fixed_effects, 1e-20, nullptr, &su, w)) return -1;
  GlmCtx g{ lik == "bernoulli_logit" ? 0 : (lik == "bernoulli_probit" ? 1 : 2), n, y, 0., w };      // the iid model is created with the model's weights (re_model.cpp:401-409)
  if (g.lik == 2) for (int i = 0; i < n; ++i) g.log_norm_const -= (w ? w[i] : 1.0) * std::lgamma(y[i] + 1.);     // log_normalizing_constant_ (likelihoods.h:10750-10757)
  GpbOptimConfig c2 = cfg;
  c2.optimizer = "lbfgs";
  c2.max_iter = std::max(cfg.max_iter, 1000);
  c2.trace = false;
  const double th[2] = {1e-20, 1.};
  char err[512] = "";
  GpbLaplaceCoefResult res;
  std::vector<double> beta = su.beta;
  if (gpb_optimize_laplace_coef_cov_pars(c2, glm_eval, &g, n, p, su.Xs.data(), fixed_effects, su.C_mu, su.C_sigma2, th, beta.data(), &res, err, (int)sizeof(err), false))
    return set_error("initial coefficients from the model without the Gaussian process: %s", err[0] ? err : "the fit failed");
  transform_back_coef(su, beta);
  *coef_out = beta;
  if (num_it_out) *num_it_out = res.num_it;
  return 0;
}

void transform_back_coef(const LaplaceCoefSetup& s, std::vector<double>& beta) {     // TransformBackCoef (:8105-8125)
  if (!s.scale) return;
  for (int j = 0; j < s.p; ++j) {
    if (s.has_intercept && j == s.intercept_col) continue;
    beta[j] /= s.scl[j];
    if (s.has_intercept) beta[s.intercept_col] -= beta[j] * s.loc[j];
  }
}

// residual norm at which the block CG of the predictive variances of the non-Gaussian models stops: the quadratic forms are then exact to ~1e-7
// relative (gpb_hip_vecchia_laplace_predict, include/gpb_hip.h)
constexpr double kPredVarCgTol = 1e-8;

void unique_locations(const std::vector<double>& coords, int n, int d, std::vector<int>* uniques, std::vector<int>* unique_idx) {
  std::vector<double> csum(n);
  for (int i = 0; i < n; ++i) { double sacc = coords[i]; for (int c = 1; c < d; ++c) sacc += coords[(size_t)c * n + i]; csum[i] = sacc; }
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return csum[a] < csum[b]; });
  std::vector<int> rep(n);
  std::iota(rep.begin(), rep.end(), 0);
  auto smaller = [](double a, double b) { return b - a > 1e-10 * std::max({std::fabs(a), std::fabs(b), 1.0}); };   // NumberIsSmallerThan (utils.h)
  for (int i = 0; i < n; ) {
    int j = i + 1;
    while (j < n && !smaller(csum[order[i]], csum[order[j]])) ++j;
    if (j - i > 1) {
      std::vector<int> grp(order.begin() + i, order.begin() + j), reps;
      std::sort(grp.begin(), grp.end());                                     // ascending position: the first appearance represents its location
      for (int p : grp) {
        bool dup = false;
        for (int r : reps) {
          double s2 = 0.;
          for (int c = 0; c < d; ++c) { const double dd = coords[(size_t)c * n + p] - coords[(size_t)c * n + r]; s2 += dd * dd; }
          if (s2 < 1e-20) { rep[p] = r; dup = true; break; }
        }
        if (!dup) reps.push_back(p);
      }
    }
    i = j;
  }
  uniques->clear();
  std::vector<int> pos(n, -1);
  for (int i = 0; i < n; ++i) if (rep[i] == i) { pos[i] = (int)uniques->size(); uniques->push_back(i); }
  unique_idx->resize(n);
  for (int i = 0; i < n; ++i) (*unique_idx)[i] = pos[rep[i]];
}

double negll_from_terms(int n, double yPy, double logdet, double sigma2) {
  return yPy / 2. / sigma2 + logdet / 2. + n / 2. * (std::log(sigma2) + std::log(2 * M_PI));   // :3132
}

}  // namespace

#define C_API_BEGIN() try {
#define C_API_END()                                                         \
  }                                                                         \
  catch (const std::exception& ex) { return set_error("%s", ex.what()); }   \
  catch (...) { return set_error("unknown exception"); }                    \
  return 0;

extern "C" {

const char* LGBM_GetLastError() { return g_last_error; }

/* c_api.cpp:1005-1009: the reference's Python package registers its logger at import (basic.py:117-129) */
int LGBM_RegisterLogCallback(void (*callback)(const char*)) {
  g_log_callback = callback;
  return 0;
}

int GPB_CreateREModel(int32_t num_data, const int32_t* cluster_ids_data, const char* /*re_group_data*/,
                      int32_t num_re_group, const double* /*re_group_rand_coef_data*/,
                      const int32_t* /*ind_effect_group_rand_coef*/, int32_t num_re_group_rand_coef,
                      const int* /*drop_intercept_group_rand_effect*/, int32_t num_gp, const double* gp_coords_data,
                      const int dim_gp_coords, const double* /*gp_rand_coef_data*/, int32_t num_gp_rand_coef,
                      const char* cov_fct, double cov_fct_shape, const char* gp_approx, double /*cov_fct_taper_range*/,
                      double /*cov_fct_taper_shape*/, int num_neighbors, const char* vecchia_ordering,
                      int num_ind_points, double /*cover_tree_radius*/, const char* ind_points_selection,
                      const char* likelihood, double likelihood_additional_param,
                      const char* matrix_inversion_method, int seed, int /*num_parallel_threads*/, bool /*GPU_use*/,
                      bool has_weights, const double* weights, double /*likelihood_learning_rate*/,
                      REModelHandle* out) {
  C_API_BEGIN();
  if (!out) return set_error("GPB_CreateREModel: out is NULL");
  *out = nullptr;
  const char* scope = "is not on the MI355X hot path of this library (one Gaussian Vecchia GP; see include/gpboost_c_api_subset.h)";
  if (num_re_group > 0 || num_re_group_rand_coef > 0) return set_error("GPB_CreateREModel: grouped random effects %s", scope);
  if (num_gp != 1 || num_gp_rand_coef > 0) return set_error("GPB_CreateREModel: num_gp = %d / num_gp_rand_coef = %d %s", num_gp, num_gp_rand_coef, scope);
  if (has_weights && !weights) return set_error("GPB_CreateREModel: has_weights is set but weights is NULL");
  if (!gp_coords_data) return set_error("GPB_CreateREModel: gp_coords_data is NULL");
  const std::string cov = cov_fct ? cov_fct : "";
  const std::string approx = gp_approx ? gp_approx : "";
  const std::string ordering = vecchia_ordering ? vecchia_ordering : "";
  const std::string lik = likelihood ? likelihood : "";
  int cov_type = -1;
  if (cov == "exponential") cov_type = 0;
  else if (cov == "matern") {
    if (near(cov_fct_shape, 0.5)) cov_type = 0;
    else if (near(cov_fct_shape, 1.5)) cov_type = 1;
    else if (near(cov_fct_shape, 2.5)) cov_type = 2;
  }
  if (cov_type < 0) return set_error("GPB_CreateREModel: cov_fct '%s' (shape %g) %s", cov.c_str(), cov_fct_shape, scope);
  // "vif" / "VIF" = "full_scale_vecchia" with Euclidean ("nearest") neighbours (re_model_template.h:207-209); the *_correlation_based forms
  // select neighbours by residual correlation with a cover tree (:201-206) and are not on this path
  const bool vif = approx == "full_scale_vecchia" || approx == "vif" || approx == "VIF";
  if (approx != "vecchia" && approx != "none" && !vif) return set_error("GPB_CreateREModel: gp_approx '%s' %s", approx.c_str(), scope);
  if (vif) {
    const std::string sel = ind_points_selection ? ind_points_selection : "";
    if (sel != "" && sel != "kmeans++") return set_error("GPB_CreateREModel: ind_points_selection '%s' with gp_approx '%s' %s", sel.c_str(), approx.c_str(), scope);
    // (round 6: non-Gaussian likelihoods with gp_approx 'full_scale_vecchia' -- FindModePostRandEffCalcMLLFSVA, likelihoods.h:3379-3750 -- iterative methods with the 'fitc' preconditioner)
    if (dim_gp_coords > 3) return set_error("GPB_CreateREModel: %d coordinate dimensions with gp_approx '%s' %s", dim_gp_coords, approx.c_str(), scope);
    if (num_ind_points <= 0) num_ind_points = 200;                                 // re_model_template.h:319-330
    if (num_ind_points > 256) return set_error("GPB_CreateREModel: num_ind_points = %d (at most 256 on this path) %s", num_ind_points, scope);
    if (num_neighbors <= 0) num_neighbors = 30;                                    // :296
  }
  bool fix_df = false;
  const std::string lik_name = parse_likelihood_alias(strip_fix_df(lik, &fix_df));
  if (lik_name != "gaussian" && !supported_non_gaussian(lik_name)) return set_error("GPB_CreateREModel: likelihood '%s' %s", lik.c_str(), scope);
  // likelihood_additional_param (likelihoods.h:223, 391-399): -999 = "not given"; 't' reads its degrees of freedom from it (aux_pars_ = {1, df}); none of the
  // other likelihoods on this path takes one ('tweedie_fixed_p', 'asymmetric_laplace' do in the reference and are not built) -- a value is refused, never dropped
  const bool add_par_given = !near(likelihood_additional_param, -999.);
  if (add_par_given && lik_name == "t" && likelihood_additional_param < 0.)
    return set_error("The 'likelihood_additional_param' (df) is not > 0, found = %g ", likelihood_additional_param);      // likelihoods.h:394-396
  if (add_par_given && lik_name != "t")
    return set_error("GPB_CreateREModel: likelihood_additional_param = %g with likelihood '%s' %s", likelihood_additional_param, lik.c_str(), scope);
  if (lik_name != "gaussian") {
    const std::string inv = matrix_inversion_method ? matrix_inversion_method : "default";
    if (approx != "vecchia" && !vif) return set_error("GPB_CreateREModel: likelihood '%s' with gp_approx '%s' %s", lik_name.c_str(), approx.c_str(), scope);
    // "default" resolves to "iterative" for a non-Gaussian Vecchia model (re_model_template.h:5722-5735)
    if (inv != "default" && inv != "iterative") return set_error("GPB_CreateREModel: matrix_inversion_method '%s' for likelihood '%s' %s", inv.c_str(), lik_name.c_str(), scope);
  }
  if (ordering != "none" && ordering != "random") return set_error("GPB_CreateREModel: vecchia_ordering '%s' %s", ordering.c_str(), scope);
  if (has_weights) {       // re_model_template.h:403-431
    if (approx != "vecchia" && !(vif && lik_name != "gaussian")) return set_error("GPB_CreateREModel: sample weights with likelihood '%s' / gp_approx '%s' %s", lik.c_str(), approx.c_str(), scope);      // (round 6: + full-scale Vecchia with a non-Gaussian likelihood)
    double sum_w = 0.;
    for (int i = 0; i < num_data; ++i) {
      if (weights[i] < 0.) return set_error(" Found negative values in 'weights' ");
      if (lik_name == "gaussian" && weights[i] == 0.) return set_error("Found zero values in 'weights'. For likelihood = 'gaussian', all weights must be strictly positive ");
      // (non-Gaussian: the reference admits zeros and then estimates diag((Sigma^-1 + W)^-1) stochastically where d information / d loc vanishes,
      //  likelihoods.h:6754-6768; this library keeps the closed form and asks for strictly positive weights)
      if (weights[i] == 0.) return set_error("GPB_CreateREModel: a sample weight that is exactly zero (datum %d) with likelihood '%s' %s -- drop the datum instead", i, lik.c_str(), scope);
      if (!std::isfinite(weights[i])) return set_error("NaN or Inf in 'weights' ");
      sum_w += weights[i];
    }
    if (sum_w == 0.) return set_error("The total sum of the 'weights' is zero ");
  }
  if (num_data < 2) return set_error("GPB_CreateREModel: num_data = %d", num_data);
  if (num_neighbors <= 0) num_neighbors = 20;   // re_model_template.h:288-294

  auto mdl = std::unique_ptr<REModelHip>(new REModelHip());
  mdl->n = num_data; mdl->d = dim_gp_coords; mdl->cov_type = cov_type; mdl->likelihood = lik_name; mdl->num_aux = num_aux_of(lik_name); if (lik_name == "lognormal") mdl->aux_pars[0] = 0.5;  /* likelihoods.h:506 */ mdl->num_neighbors = num_neighbors;
  if (lik_name == "t" && add_par_given) mdl->aux_pars[1] = likelihood_additional_param;      // aux_pars_ = {1, additional_param} (likelihoods.h:397-399)
  mdl->estimate_df_t = !fix_df;
  if (vif && lik_name != "gaussian") { mdl->cg_preconditioner_type = "fitc"; mdl->piv_chol_rank = 200; }      // re_model_template.h:7137-7150 (default of a non-Gaussian full_scale_vecchia model)
  if (has_weights && lik_name != "gaussian") mdl->lik_weights.assign(weights, weights + num_data);     // factors of the per-datum likelihood terms (likelihoods.h:666-668)
  mdl->perm.resize(num_data);
  std::iota(mdl->perm.begin(), mdl->perm.end(), 0);
  if (approx == "none") {   // exact GP: dense Cholesky (re_model_template.h:8151, :9273-9287, :6491-6494); no ordering
    if (cluster_ids_data)
      for (int i = 1; i < num_data; ++i)
        if (cluster_ids_data[i] != cluster_ids_data[0]) return set_error("GPB_CreateREModel: more than one cluster with gp_approx 'none' %s", scope);
    if (gpb_hip_exact_create(num_data, dim_gp_coords, gp_coords_data, &mdl->eh)) return shim_error();
    mdl->m = 0;
    mdl->rng = std::mt19937(seed);                                   // rng_ (re_model_template.h:161): FindInitCovPar draws its sub-sample from it
    mdl->coords0.assign(gp_coords_data, gp_coords_data + (size_t)num_data * dim_gp_coords); mdl->n0 = num_data;
    mdl->cl_off = {0, num_data};
    *out = mdl.release();
    return 0;
  }
  // clusters = independent realisations of the GP, in order of first appearance (SetUpClusterIds, re_model_template.h:6820-6852)
  std::vector<std::vector<int>> clusters;
  if (cluster_ids_data) {
    std::vector<int32_t> ids;
    for (int i = 0; i < num_data; ++i) {
      size_t k = 0;
      while (k < ids.size() && ids[k] != cluster_ids_data[i]) ++k;
      if (k == ids.size()) { ids.push_back(cluster_ids_data[i]); clusters.emplace_back(); }
      if (ids.size() > 4096) return set_error("GPB_CreateREModel: more than 4096 clusters %s", scope);
      clusters[k].push_back(i);
    }
    mdl->cluster_id_values = ids;
  } else {
    clusters.emplace_back(mdl->perm);
  }
  if (clusters.size() > 1 && lik_name != "gaussian") return set_error("GPB_CreateREModel: several clusters with likelihood '%s' %s", lik.c_str(), scope);
  if (clusters.size() > 1 && vif) return set_error("GPB_CreateREModel: several clusters with gp_approx '%s' %s", approx.c_str(), scope);
  mdl->vif = vif; mdl->num_ind_points = vif ? num_ind_points : 0;
  mdl->rng = std::mt19937(seed);                                   // ONE generator for all clusters (re_model_template.h:161, type_defs.h:52)
  std::mt19937& rng = mdl->rng;
  mdl->perm.clear(); mdl->cl_off.assign(1, 0);
  mdl->m = 0;
  for (auto& idx : clusters) {
    const int nc = (int)idx.size();
    if (nc < 2) return set_error("GPB_CreateREModel: a cluster with %d data point(s) %s", nc, scope);
    if (ordering == "random") std::shuffle(idx.begin(), idx.end(), rng);   // Vecchia_utils.cpp:1129-1131
    std::vector<double> coords((size_t)nc * dim_gp_coords);
    for (int j = 0; j < dim_gp_coords; ++j)                          // Vecchia_utils.cpp:1132-1138
      for (int k = 0; k < nc; ++k) coords[(size_t)j * nc + k] = gp_coords_data[(size_t)j * num_data + idx[k]];
    if (vif) {     // CreateREComponentsFITC_FSA (re_model_template.h:7639-7720) runs between the shuffle and the neighbour search, on the same generator
      if (nc <= num_ind_points) return set_error("Need to have less inducing points (currently num_ind_points = %d) than data points (%d) if gp_approx = 'full_scale_vecchia' ", num_ind_points, nc);
      if (kmeans_plusplus(coords, nc, dim_gp_coords, num_ind_points, rng, 1000, &mdl->ip)) return -1;
    }
    gpb_hip_vecchia_t* vh = nullptr;
    int n_pts = nc;                                                    // points of the Vecchia approximation
    if (lik_name != "gaussian") {
      // one non-Gaussian GP: repeated locations share ONE random effect (use_Z_for_duplicates, Vecchia_utils.cpp:1156-1168); the approximation
      // is built on the unique locations in the order of their first appearance in the (shuffled) data
      std::vector<int> uniques, uidx;
      unique_locations(coords, nc, dim_gp_coords, &uniques, &uidx);
      if ((int)uniques.size() < nc) {
        if (vif) return set_error("GPB_CreateREModel: repeated locations with likelihood '%s' and gp_approx '%s' %s", lik.c_str(), approx.c_str(), scope);
        const int nu = (int)uniques.size();
        if (nu < 2) return set_error("GPB_CreateREModel: %d unique location(s) %s", nu, scope);
        std::vector<double> cu((size_t)nu * dim_gp_coords);
        for (int j = 0; j < dim_gp_coords; ++j) for (int u = 0; u < nu; ++u) cu[(size_t)j * nu + u] = coords[(size_t)j * nc + uniques[u]];
        coords.swap(cu);
        n_pts = nu;
        mdl->n_re = nu; mdl->re_of = uidx;
        mdl->dorder.resize(nc);
        std::iota(mdl->dorder.begin(), mdl->dorder.end(), 0);
        std::stable_sort(mdl->dorder.begin(), mdl->dorder.end(), [&](int a, int b) { return uidx[a] < uidx[b]; });
        mdl->re_ptr.assign(nu + 1, 0);
        for (int k = 0; k < nc; ++k) mdl->re_ptr[uidx[k] + 1]++;
        for (int u = 0; u < nu; ++u) mdl->re_ptr[u + 1] += mdl->re_ptr[u];
      }
    }
    if (gpb_hip_vecchia_create(n_pts, dim_gp_coords, num_neighbors, coords.data(), &vh)) return shim_error();
    if (mdl->vhs.empty()) { mdl->coords0 = coords; mdl->n0 = n_pts; }
    mdl->vhs.push_back(vh);
    if (mdl->n_re > 0 && gpb_hip_vecchia_laplace_set_data_map(vh, mdl->re_ptr.data())) return shim_error();
    if (has_weights && lik_name == "gaussian") {     // nugget 1 / w_i of every observation, Vecchia order (GetGaussianNuggetDiagFromWeights, :6393-6417)
      std::vector<double> nug((size_t)nc);
      for (int k = 0; k < nc; ++k) nug[k] = 1. / weights[idx[k]];
      if (gpb_hip_vecchia_set_nugget_diag(vh, nug.data())) return shim_error();
      mdl->has_weights = true;
      mdl->nug_v.insert(mdl->nug_v.end(), nug.begin(), nug.end());
    }
    int dup = 0;
    if (gpb_hip_vecchia_find_neighbors(vh, &dup)) return shim_error();
    mdl->has_duplicates = mdl->has_duplicates || dup != 0;
    mdl->m = std::max(mdl->m, std::min(num_neighbors, n_pts - 1));
    mdl->perm.insert(mdl->perm.end(), idx.begin(), idx.end());
    mdl->cl_off.push_back((int)mdl->perm.size());
  }
  mdl->vh = mdl->vhs[0];
  if (vif) {
    if (mdl->has_duplicates) return set_error("GPB_CreateREModel: duplicate coordinates with gp_approx '%s' %s", approx.c_str(), scope);
    if (gpb_hip_vecchia_vif_set_inducing_points(mdl->vh, num_ind_points, mdl->ip.data())) return shim_error();
    if (lik_name != "gaussian" && laplace_push_preconditioner(mdl.get())) return -1;      // "fitc", rank 200: the default of these models reaches the device without a GPB_SetOptimConfig call
  }
  // the reference maps repeated locations to unique random effects for one non-Gaussian GP and stops if duplicates remain
  // (Vecchia_utils.cpp:1156-1158, 1208-1214); the unique-location mapping is not on this path, so duplicates are an error here
  if (lik_name != "gaussian" && mdl->has_duplicates) return set_error("%s", kDuplicatesNonGaussianMessage);
  *out = mdl.release();
  C_API_END();
}

int GPB_REModelFree(REModelHandle handle) {
  C_API_BEGIN();
  delete reinterpret_cast<REModelHip*>(handle);
  C_API_END();
}

int GPB_SetOptimConfig(REModelHandle handle, double* init_cov_pars, double lr, double acc_rate_cov, int max_iter, double delta_rel_conv,
                       bool use_nesterov_acc, int nesterov_schedule_version, bool trace, const char* optimizer, int momentum_offset,
                       const char* convergence_criterion, int num_covariates, double* init_coef, double /*lr_coef*/,
                       double /*acc_rate_coef*/, const char* optimizer_coef, int cg_max_num_it, int cg_max_num_it_tridiag,
                       double cg_delta_conv, int num_rand_vec_trace, bool /*reuse_rand_vec_trace*/, const char* cg_preconditioner_type,
                       int seed_rand_vec_trace, int piv_chol_rank, double* init_aux_pars, bool estimate_aux_pars,
                       bool init_coef_aux_pars_from_iid_model, const int* estimate_cov_par_index, int m_lbfgs,
                       double delta_conv_mode_finding) {
  C_API_BEGIN();
  if (!handle) return set_error("GPB_SetOptimConfig: null handle");
  // covariates arrive with GPB_OptimLinRegrCoefCovPar; init_coef is irrelevant when the coefficients are profiled out ("wls", Gaussian models) and is
  // the start of the lbfgs vector for non-Gaussian models
  auto* mdl0 = reinterpret_cast<REModelHip*>(handle);
  if (init_coef && num_covariates > 0) mdl0->init_coef.assign(init_coef, init_coef + num_covariates); else mdl0->init_coef.clear();
  mdl0->init_coef_from_iid_model = init_coef_aux_pars_from_iid_model;
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  // auxiliary parameters (gamma / negative_binomial; re_model.cpp:327-344, re_model_template.h:909-912): initial values are applied at once (SetAuxPars)
  mdl->estimate_aux_pars = estimate_aux_pars;
  if (init_aux_pars && mdl->num_aux > 0) {
    if (mdl->num_aux_estim() < mdl->num_aux && !mdl->aux_set && !near(init_aux_pars[1], mdl->aux_pars[1]))       // likelihoods.h:2763-2768
      fprintf(stderr, "[gpboost_amd] Warning: The 'df' parameter provided in 'init_aux_pars' (= %g) and 'likelihood_additional_param' (= %g) are not equal. Will use the value provided in 'likelihood_additional_param' \n", init_aux_pars[1], mdl->aux_pars[1]);
    for (int j = 0; j < mdl->num_aux; ++j) {
      mdl->init_aux[j] = init_aux_pars[j];                                     // init_aux_pars_ keeps what the caller passed (re_model.cpp:327-344)
      if (j >= mdl->num_aux_estim()) continue;                                 // SetAuxPars copies the first num_aux_pars_estim_ values only (likelihoods.h:2780-2789)
      if (!(init_aux_pars[j] > 0.)) return set_error("The %s parameter is not > 0 (found %g)", mdl->likelihood == "t" ? (j == 0 ? "scale" : "df") : (mdl->likelihood == "lognormal" ? "log_variance" : "shape"), init_aux_pars[j]);
      mdl->aux_pars[j] = init_aux_pars[j];
    }
    mdl->init_aux_given = true; mdl->aux_set = true;
  } else mdl->init_aux_given = false;
  if (estimate_cov_par_index && estimate_cov_par_index[0] >= 0) {          // re_model_template.h:930-936
    if (mdl->likelihood != "gaussian") {      // two covariance parameters (sigma1_2, rho): lbfgs leaves the ones marked 0 at their initial values
      mdl->optim.estimate_cov_par_index[0] = estimate_cov_par_index[0]; mdl->optim.estimate_cov_par_index[1] = estimate_cov_par_index[1];
      mdl->optim.estimate_cov_par_index[2] = 1;
    } else
    std::copy(estimate_cov_par_index, estimate_cov_par_index + 3, mdl->optim.estimate_cov_par_index);
  }
  mdl->trace = trace;
  if (optimizer_coef && optimizer_coef[0]) mdl->optimizer_coef = optimizer_coef;
  // REModel::SetOptimConfig (re_model.cpp:301-318): initial values are kept on the transformed scale
  if (init_cov_pars) {
    if (mdl->likelihood != "gaussian") {        // (sigma1_2, rho) -> (sigma1_2, a): no error variance (re_model.cpp:301-318 with gauss_likelihood_ = false)
      if (!(init_cov_pars[0] > 0.) || !(init_cov_pars[1] > 0.)) return set_error("Covariance parameters need to be positive (found %g, %g)", init_cov_pars[0], init_cov_pars[1]);
      mdl->init_cov_pars_tr[0] = init_cov_pars[0]; mdl->init_cov_pars_tr[1] = range_const(mdl) / init_cov_pars[1]; mdl->init_cov_pars_tr[2] = 0.;
      std::copy(mdl->init_cov_pars_tr, mdl->init_cov_pars_tr + 3, mdl->cov_pars_tr);
      mdl->cov_pars_initialized = true;
      mdl->init_cov_pars_provided = true;
    } else {
    double tr[3];
    if (transform_cov_pars(mdl, init_cov_pars, tr)) return -1;
    std::copy(tr, tr + 3, mdl->init_cov_pars_tr);
    std::copy(tr, tr + 3, mdl->cov_pars_tr);
    mdl->cov_pars_initialized = true;
    mdl->init_cov_pars_provided = true;
    }
  }
  // REModelTemplate::SetOptimConfig (re_model_template.h:743-960); -999 = keep the default
  GpbOptimConfig& oc = mdl->optim;
  oc.trace = trace;
  if (acc_rate_cov > 0.) oc.acc_rate_cov = acc_rate_cov;
  else if (!near(acc_rate_cov, -999.)) return set_error("acc_rate_cov is not > 0, found = %g ", acc_rate_cov);
  if (max_iter >= 0) oc.max_iter = max_iter;
  else if (max_iter != -999) return set_error("max_iter is not >= 0, found = %d ", max_iter);
  oc.use_nesterov_acc = use_nesterov_acc;
  if (nesterov_schedule_version == 0 || nesterov_schedule_version == 1) oc.nesterov_schedule_version = nesterov_schedule_version;
  else if (nesterov_schedule_version != -999) return set_error("nesterov_schedule_version is not 0 or 1, found = %d ", nesterov_schedule_version);
  if (optimizer && std::string(optimizer) != "") {
    std::string o = optimizer;
    mdl->optimizer_unsupported_alias = (o == "gradient_descent_constant_change" || o == "gradient_descent_reset_lr");   // change the step rule (:788-818)
    if (o == "gradient_descent_constant_change" || o == "gradient_descent_increase_lr" || o == "gradient_descent_reset_lr") o = "gradient_descent";
    if (o == "lbfgs_not_profile_out_nugget") mdl->optimizer_unsupported_alias = true;
    oc.optimizer = o;
  }
  if (momentum_offset >= 0) oc.momentum_offset = momentum_offset;
  else if (momentum_offset != -999) return set_error("momentum_offset is not >= 0, found = %d ", momentum_offset);
  if (convergence_criterion && std::string(convergence_criterion) != "default") {
    const std::string cc = convergence_criterion;
    if (cc != "relative_change_in_log_likelihood" && cc != "relative_change_in_parameters")
      return set_error("Convergence criterion '%s' is not supported.", cc.c_str());
    oc.convergence_criterion = cc;
  }
  if (delta_rel_conv > 0.) oc.delta_rel_conv_init = delta_rel_conv;
  else if (!near(delta_rel_conv, -999.)) return set_error("delta_rel_conv is not > 0, found = %g ", delta_rel_conv);
  if (lr > 0.) oc.lr_cov_init = lr;
  else if (!near(lr, -999.)) return set_error("lr_cov is not > 0, found = %g ", lr);
  if (m_lbfgs > 0) oc.m_lbfgs = m_lbfgs;
  else if (m_lbfgs != -999) return set_error("m_lbfgs is not > 0, found = %d ", m_lbfgs);
  if (num_rand_vec_trace > 0) mdl->num_rand_vec_trace = num_rand_vec_trace;
  else if (num_rand_vec_trace != -999) return set_error("num_rand_vec_trace is not > 0, found = %d ", num_rand_vec_trace);
  mdl->seed_rand_vec_trace = seed_rand_vec_trace;
  if (cg_max_num_it > 0) mdl->cg_max_num_it = cg_max_num_it;
  else if (cg_max_num_it != -999) return set_error("cg_max_num_it is not > 0, found = %d ", cg_max_num_it);
  if (cg_max_num_it_tridiag > 0) mdl->cg_max_num_it_tridiag = cg_max_num_it_tridiag;
  else if (cg_max_num_it_tridiag != -999) return set_error("cg_max_num_it_tridiag is not > 0, found = %d ", cg_max_num_it_tridiag);
  if (cg_delta_conv > 0.) mdl->cg_delta_conv = cg_delta_conv;
  else if (!near(cg_delta_conv, -999.)) return set_error("cg_delta_conv is not > 0, found = %g ", cg_delta_conv);
  if (delta_conv_mode_finding > 0.) mdl->delta_conv_mode_finding = delta_conv_mode_finding;
  else if (!near(delta_conv_mode_finding, -999.)) return set_error("delta_conv_mode_finding is not > 0, found = %g ", delta_conv_mode_finding);
  if (mdl->likelihood != "gaussian") {
    const std::string pc = cg_preconditioner_type ? cg_preconditioner_type : "";      // (NULL = not given, as the reference's packages pass it: the rank below is read all the same, re_model_template.h:895-914)
    if (pc != "" && mdl->cg_preconditioner_type != pc && mdl->model_has_been_estimated)      // re_model_template.h:891-895 (the comparison is with the string as given, before the alias is resolved)
      return set_error("Cannot change 'cg_preconditioner_type' after a model has been fitted ");
    // ParsePreconditionerAlias (re_model_template.h:7482-7513); SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_ (:5906): "vadu", "pivoted_cholesky", "fitc" (round 5) and "vecchia_response" (round 6) are built; "incomplete_cholesky" is not
    if (mdl->vif) {      // SUPPORTED_PRECONDITIONERS_NONGAUSS_VIF_ (:5910): "fitc" (the default), "vifdu", "none"
      if (pc == "fitc" || pc == "FITC" || pc == "predictive_process_plus_diagonal" || pc == "") { if (pc != "") mdl->cg_preconditioner_type = "fitc"; }
      else if (pc == "vifdu" || pc == "VIFDU" || pc == "Bt_Sigma_inv_plus_W_B") mdl->cg_preconditioner_type = "vifdu";      // evaluation + Nelder-Mead fits: the gradient is built for "fitc" only
      else if (pc == "none") mdl->cg_preconditioner_type = "none";
      else return set_error("Preconditioner type '%s' is not supported for gp_approx = '%s' and likelihood = '%s'", pc.c_str(), "full_scale_vecchia", mdl->likelihood.c_str());
    }
    else if (pc == "" || pc == "vadu" || pc == "VADU" || pc == "vecchia_approximation_with_diagonal_update" || pc == "Sigma_inv_plus_BtWB") { if (pc != "") mdl->cg_preconditioner_type = "vadu"; }
    else if (pc == "pivoted_cholesky" || pc == "piv_chol" || pc == "piv_chol_on_Sigma") mdl->cg_preconditioner_type = "pivoted_cholesky";
    else if (pc == "fitc" || pc == "FITC" || pc == "predictive_process_plus_diagonal") mdl->cg_preconditioner_type = "fitc";
    else if (pc == "vecchia_response" || pc == "vecchia_observable" || pc == "vecchia") mdl->cg_preconditioner_type = "vecchia_response";     // evaluation only: the reference refuses gradients with it (likelihoods.h:6570-6572)
    else return set_error("GPB_SetOptimConfig: cg_preconditioner_type '%s' is not on the MI355X hot path of this library ('vadu', 'pivoted_cholesky', 'fitc' and 'vecchia_response' are)", pc.c_str());
    const int rank_before = mdl->piv_chol_rank;
    if (piv_chol_rank > 0) mdl->piv_chol_rank = piv_chol_rank;                      // re_model_template.h:900-914
    else if (piv_chol_rank != -999) return set_error("fitc_piv_chol_preconditioner_rank is not > 0, found = %d ", piv_chol_rank);
    else if (mdl->cg_preconditioner_type == "fitc") mdl->piv_chol_rank = 200;          // (a call without a rank puts the type's default back, whether or not it named the type: :907-914)
    else if (mdl->cg_preconditioner_type == "pivoted_cholesky") mdl->piv_chol_rank = 50;
    if (mdl->cg_preconditioner_type == "fitc" && mdl->piv_chol_rank != rank_before && !mdl->pc_ip.empty() && (int)mdl->pc_ip.size() != mdl->piv_chol_rank * mdl->d)
      return set_error("GPB_SetOptimConfig: the inducing points of the fitc preconditioner have been determined for rank %d; the rank cannot change afterwards on this path", (int)mdl->pc_ip.size() / mdl->d);
    mdl->pc_ip_pushed = false;
    if (mdl->cg_preconditioner_type == "pivoted_cholesky" && !mdl->vif && !mdl->eh && mdl->piv_chol_rank > (mdl->n_re > 0 ? mdl->n_re : mdl->n))
      return set_error("'fitc_piv_chol_preconditioner_rank' cannot be larger than the dimension of the mode (= number of unique locations) ");     // likelihoods.h:936-938
    if (laplace_push_preconditioner(mdl)) return -1;
  }
  C_API_END();
}

int GPB_EvalNegLogLikelihood(REModelHandle handle, const double* y_data, double* cov_pars, const double* fixed_effects,
                             double* negll) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !negll) return set_error("GPB_EvalNegLogLikelihood: null argument");
  if (mdl->likelihood != "gaussian") {   // cov_pars = (sigma1_2, rho): no error variance (re_model_template.h:3191-3212)
    // y_data == NULL: the response of the last call that passed one (CHECK(y_has_been_set_), re_model_template.h:3183-3190)
    if (!y_data && !mdl->y_set) return set_error("Check failed: y_has_been_set_ (GPB_EvalNegLogLikelihood: y_data is NULL and no response has been set)");
    double stored[2] = {0., 0.};
    if (!cov_pars) {                        // re_model.cpp:759-766: the stored (initial or estimated) parameters
      if (y_data && initialize_cov_pars_if_not_defined(mdl, y_data, fixed_effects)) return -1;
      if (!mdl->cov_pars_initialized) return set_error("Check failed: cov_pars_initialized_ (GPB_EvalNegLogLikelihood: cov_pars is NULL and no parameters are stored)");
      stored[0] = mdl->cov_pars_tr[0]; stored[1] = range_const(mdl) / mdl->cov_pars_tr[1];
      cov_pars = stored;
    }
    const double sigma1_2 = cov_pars[0], rho = cov_pars[1];
    if (!(sigma1_2 > 0.) || !(rho > 0.)) return set_error("Covariance parameters need to be positive (found %g, %g)", sigma1_2, rho);
    if (y_data) { if (laplace_upload_data(mdl, y_data, fixed_effects)) return -1; }
    else { if (laplace_upload_fixed_effects(mdl, fixed_effects)) return -1; if (laplace_push_aux(mdl)) return -1; }   // labels stay resident; the offset and the auxiliary parameters are this call's
    const double cc = mdl->cov_type == 0 ? 1. : (mdl->cov_type == 1 ? std::sqrt(3.) : std::sqrt(5.));
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_logit(mdl->vh, mdl->cov_type, sigma1_2, cc / rho, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace,
                                      mdl->cg_max_num_it, mdl->cg_max_num_it_tridiag, mdl->cg_delta_conv, mdl->delta_conv_mode_finding,
                                      1 /* mode reset to 0, :3199-3201 */, mdl->lap_info, nullptr)) return shim_error();
    mdl->cur_negll = -mdl->lap_info[0];
    mdl->negll_valid = true;
    *negll = mdl->cur_negll;
    return 0;
  }
  double tr[3];
  if (cov_pars) { if (transform_cov_pars(mdl, cov_pars, tr)) return -1; }
  else {                                            // re_model.cpp:759-766: the stored (initial or estimated) parameters
    if (y_data && initialize_cov_pars_if_not_defined(mdl, y_data, fixed_effects)) return -1;
    if (!mdl->cov_pars_initialized) return set_error("Check failed: cov_pars_initialized_ (GPB_EvalNegLogLikelihood: cov_pars is NULL and no parameters are stored)");
    std::copy(mdl->cov_pars_tr, mdl->cov_pars_tr + 3, tr);
  }
  if (y_data) { if (upload_y(mdl, y_data, fixed_effects)) return -1; }
  else {
    // y_data == NULL: the response already resident in HBM is used -- nothing crosses PCIe but the parameters and the value
    // (EvalNegLogLikelihoodGauss calls SetY only for a non-NULL y_data, re_model_template.h:2905-2921; this is what the
    //  reference's own optimiser and BASELINE.json's metric do between SetY calls)
    if (fixed_effects) return set_error("EvalNegLogLikelihoodGauss: 'y_data' cannot nullptr when 'fixed_effects' is provided ");   // :2907-2909
    if (!mdl->y_set) return set_error("GPB_EvalNegLogLikelihood: y_data is NULL and no response has been set (pass y_data once, or call GPB_OptimCovPar / GPB_SetY first)");
  }
  double t3[3] = {0., 0., 0.};
  if (mdl->eh) { if (gpb_hip_exact_nll_terms(mdl->eh, mdl->cov_type, tr[1], tr[2], t3, nullptr, nullptr)) return shim_error(); }
  else {
    // block-diagonal Psi: the quadratic forms and log-determinants of the clusters add up; a sharded handle (one process per GPU,
    // gpb_hip_vecchia_comm_init) returns the job-wide sums on every rank (one ncclAllReduce of 3 doubles per evaluation)
    double t7[7];
    if (device_terms(mdl, tr[1], tr[2], 0, t7)) return -1;
    t3[0] = t7[0]; t3[1] = t7[1]; t3[2] = t7[2];
  }
  mdl->cur_negll = negll_from_terms(mdl->n, t3[0], t3[1], tr[0]);
  mdl->negll_valid = true;
  *negll = mdl->cur_negll;
  C_API_END();
}

/* ---- parameter estimation: the direct caller of the hot path ---- */
int GPB_OptimCovPar(REModelHandle handle, const double* y_data, const double* fixed_effects) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_OptimCovPar: null handle");
  // a fit without a linear regression term forgets the covariates of an earlier one (re_model.cpp:483-546: OptimCovPar is
  // OptimLinRegrCoefCovPar without covariate data, has_covariates_ = false)
  if (!mdl->fitting_with_covariates && mdl->p_cov > 0) {
    mdl->p_cov = 0; mdl->coef_estimated = false; mdl->X.clear(); mdl->beta.clear();
    if (mdl->vh) (void)gpb_hip_vecchia_set_covariates(mdl->vh, 0, nullptr);
  }
  // a fit with fixed effects keeps them as the model's offset for later predictions (re_model_template.h:1185-1188)
  if (fixed_effects) { mdl->offset.assign(fixed_effects, fixed_effects + mdl->n); mdl->has_offset = true; }
  resolve_optimizer_defaults_once(mdl);
  const char* scope = "is not on the MI355X path of this library yet (parameter estimation: Gaussian likelihood, gp_approx 'vecchia')";
  if (mdl->likelihood != "gaussian") {     // theta = (sigma1_2, a), Laplace approximation + its gradient on the device (gpb_optim.h: gpb_laplace_fn)
    if (mdl->optimizer_unsupported_alias) return set_error("GPB_OptimCovPar: this variant of optimizer_cov %s", scope);
    if (!y_data) return set_error("GPB_OptimCovPar: y_data is NULL");
    if (initialize_cov_pars_if_not_defined(mdl, y_data, fixed_effects)) return -1;                  // re_model.cpp:487-491
    // initial values of the auxiliary parameters if none were given (re_model_template.h:1332-1349)
    if (mdl->num_aux > 0 && mdl->estimate_aux_pars && !mdl->aux_set) find_initial_aux_pars(mdl, y_data, fixed_effects);
    if (laplace_upload_data(mdl, y_data, fixed_effects)) return -1;
    mdl->lap_fit_first_eval = true;
    GpbOptimConfig cfg = mdl->optim;
    cfg.range_const = range_const(mdl);
    char err[512] = "";
    GpbLaplaceOptimResult res;
    if (mdl->num_aux > 0 && mdl->estimate_aux_pars) {      // the shape is part of the lbfgs vector (optim_utils.h:256-283)
      GpbLaplaceAuxResult ra;
      double aux[2] = {mdl->aux_pars[0], mdl->aux_pars[1]};
      if (gpb_optimize_laplace_cov_aux_pars(cfg, device_laplace_aux, mdl, mdl->num_aux, mdl->cov_pars_tr, aux, &ra, err, (int)sizeof(err))) {
        const char* why = gpb_hip_get_last_error();
        if (err[0] && why && why[0]) return set_error("%s: %s", err, why);
        return err[0] ? set_error("%s", err) : shim_error();
      }
      if (cfg.max_iter > 0) {
        mdl->cov_pars_tr[0] = ra.theta[0]; mdl->cov_pars_tr[1] = ra.theta[1];
        for (int j = 0; j < mdl->num_aux_estim(); ++j) mdl->aux_pars[j] = aux[j];
        if (laplace_push_aux(mdl)) return -1;
        mdl->cur_negll = ra.negll;
        mdl->negll_valid = true;
      }
      mdl->num_it = ra.num_it;
      res.theta[0] = ra.theta[0]; res.theta[1] = ra.theta[1]; res.num_it = ra.num_it; res.negll = ra.negll; res.num_evals = ra.num_evals;
      mdl->last_fit_lap = res;
      mdl->model_has_been_estimated = true;
      return 0;
    }
    if (gpb_optimize_laplace_cov_pars(cfg, device_laplace, mdl, mdl->cov_pars_tr, &res, err, (int)sizeof(err))) {
      const char* why = gpb_hip_get_last_error();           // what the device path said, not only "evaluation failed"
      if (err[0] && why && why[0]) return set_error("%s: %s", err, why);
      return err[0] ? set_error("%s", err) : shim_error();
    }
    if (cfg.max_iter > 0) {
      mdl->cov_pars_tr[0] = res.theta[0]; mdl->cov_pars_tr[1] = res.theta[1];
      mdl->cur_negll = res.negll;
      mdl->negll_valid = true;
    }
    mdl->num_it = res.num_it;
    mdl->last_fit_lap = res;
    mdl->model_has_been_estimated = true;
    return 0;
  }
  if (mdl->optimizer_unsupported_alias) return set_error("GPB_OptimCovPar: this variant of optimizer_cov %s", scope);
  if (!y_data) return set_error("GPB_OptimCovPar: y_data is NULL");
  for (int i = 0; i < mdl->n; ++i)
    if (!std::isfinite(y_data[i])) return set_error("NaN or Inf in response variable / label ");   // re_model_template.h:1090-1094
  if (initialize_cov_pars_if_not_defined(mdl, y_data, fixed_effects)) return -1;                    // re_model.cpp:487-491
  if (upload_y(mdl, y_data, fixed_effects)) return -1;   // ONE H2D of y for the whole fit (SetY, re_model_template.h:1204-1206, :1324-1331)
  GpbOptimConfig cfg = mdl->optim;
  cfg.range_const = range_const(mdl);
  if (mdl->fitting_with_covariates && mdl->p_cov > 0 && mdl->coef_by_iteration) {
    // gradient descent: the response starts as y - fixed effects - X beta_init; one least-squares update of the coefficients per iteration
    if (gpb_hip_vecchia_set_resid(mdl->vh, mdl->beta.data())) return shim_error();
    cfg.coef_update_ctx = mdl;
    cfg.coef_update = [](void* c, double ratio, double a, double* t7) {
      auto* m = static_cast<REModelHip*>(c);
      if (profile_out_coef(m, ratio, a)) return -1;
      return device_terms(m, ratio, a, 1, t7);
    };
  } else if (mdl->fitting_with_covariates && mdl->p_cov > 0) {   // the coefficients are profiled out by the evaluator (device_terms): lbfgs keeps / restores them with the error variance
    cfg.profiled_lag_ctx = mdl;
    cfg.profiled_lag = [](void* c, int op) { auto* m = static_cast<REModelHip*>(c); if (op == 0) m->beta_lag1 = m->beta; else m->beta = m->beta_lag1; };
  }
  // (gp_approx 'full_scale_vecchia': lbfgs / gradient_descent run on fourth-order central differences of the device likelihood, device_terms)
  if (mdl->vif && mdl->p_cov > 0 && mdl->fitting_with_covariates) return set_error("GPB_OptimLinRegrCoefCovPar: covariates with gp_approx 'full_scale_vecchia' %s", scope);
  char err[512] = "";
  GpbOptimResult res;
  if (gpb_optimize_gaussian_cov_pars(cfg, mdl->n, device_terms, mdl, mdl->cov_pars_tr, &res, err, (int)sizeof(err))) {
    const char* why = gpb_hip_get_last_error();
    if (err[0] && why && why[0] && !std::strstr(err, why)) return set_error("%s: %s", err, why);
    return err[0] ? set_error("%s", err) : -1;
  }
  if (cfg.max_iter > 0) {
    std::copy(res.theta, res.theta + 3, mdl->cov_pars_tr);
    mdl->cur_negll = res.negll;
    mdl->negll_valid = true;
  }
  mdl->num_it = res.num_it;
  mdl->last_fit = res;
  mdl->model_has_been_estimated = true;
  C_API_END();
}

int GPB_GetCovPar(REModelHandle handle, double* optim_cov_pars, bool calc_std_dev) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !optim_cov_pars) return set_error("GPB_GetCovPar: null argument");
  if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or correctly set ");   // re_model.cpp:922-924
  if (mdl->likelihood != "gaussian") {     // two parameters: (sigma1_2, rho)
    optim_cov_pars[0] = mdl->cov_pars_tr[0]; optim_cov_pars[1] = range_const(mdl) / mdl->cov_pars_tr[1];
    if (calc_std_dev) {     // CalcStdDevCovParAuxParsNonGaussian (re_model_template.h:11029-11117): numerical Jacobian of the device gradient, delta method
      if (!can_calc_std_dev(mdl)) return set_error("GPB_GetCovPar: standard deviations of a non-Gaussian model need its gradient on the MI355X path (one cluster, at most 126 neighbours, coordinate dimensions 1..3; GPB_CanCalculateStandardErrorsCovPars answers 0 otherwise)");
      if (!mdl->y_set) return set_error("GPB_GetCovPar: standard deviations need the response of a fit or an evaluation (none has been set)");
      const double th[2] = {mdl->cov_pars_tr[0], mdl->cov_pars_tr[1]};
      double se[2];
      char err[512] = "";
      {     // the location parameter of the fit: offset + X beta (whatever an evaluation in between has left on the device)
        const double* fel = mdl->has_offset ? mdl->offset.data() : nullptr;
        std::vector<double> fe_lin;
        if (mdl->p_cov > 0 && mdl->coef_estimated) {
          fe_lin.assign(mdl->n, 0.);
          for (int i = 0; i < mdl->n; ++i) { double v = fel ? fel[i] : 0.; for (int j = 0; j < mdl->p_cov; ++j) v += mdl->X[(size_t)j * mdl->n + i] * mdl->beta[j]; fe_lin[i] = v; }
          fel = fe_lin.data();
        }
        if (laplace_upload_fixed_effects(mdl, fel)) return -1;
      }
      int rc_se;
      if (mdl->num_aux > 0 && mdl->estimate_aux_pars) {      // the JOINT Hessian over covariance and auxiliary parameters (re_model_template.h:11034-11047)
        double se_aux[2];
        rc_se = gpb_laplace_aux_std_errors(device_laplace_aux, mdl, th, mdl->aux_pars, mdl->num_aux, range_const(mdl), se, se_aux, err, (int)sizeof(err), mdl->optim.estimate_cov_par_index);
      } else rc_se = gpb_laplace_std_errors(device_laplace, mdl, th, range_const(mdl), se, err, (int)sizeof(err), mdl->optim.estimate_cov_par_index);
      if (rc_se) {
        const char* why = gpb_hip_get_last_error();
        return (why && why[0]) ? set_error("%s: %s", err[0] ? err : "GPB_GetCovPar", why) : set_error("%s", err[0] ? err : "evaluation failed");
      }
      optim_cov_pars[2] = se[0]; optim_cov_pars[3] = se[1];
    }
    return 0;
  }
  transform_back(mdl, mdl->cov_pars_tr, optim_cov_pars);
  if (calc_std_dev) {      // CalculateStandardErrorsCovPars -> CalcFisherInformation_Vecchia (stochastic trace, re_model_template.h:10137-10230)
    if (!can_calc_std_dev(mdl)) return set_error("GPB_GetCovPar: standard deviations are on the MI355X path of this library for a one-cluster, unsharded Gaussian Vecchia model with at most 126 neighbours and coordinate dimensions 1..3, and for the exact GP up to n = 24000, only (GPB_CanCalculateStandardErrorsCovPars answers 0 otherwise)");
    double se[3];
    if (mdl->eh) {             // CalcFisherInformation, dense branch (re_model_template.h:10066-10127)
      if (gpb_hip_exact_fisher_std_errors(mdl->eh, mdl->cov_type, optim_cov_pars[0], mdl->cov_pars_tr[1], mdl->cov_pars_tr[2], optim_cov_pars[2], se)) return shim_error();
    } else if (gpb_hip_vecchia_fisher_std_errors(mdl->vh, mdl->cov_type, optim_cov_pars[0], optim_cov_pars[1], optim_cov_pars[2], mdl->cov_pars_tr[1], mdl->cov_pars_tr[2],
                                          mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, se)) return shim_error();
    for (int j = 0; j < 3; ++j) optim_cov_pars[3 + j] = se[j];     // re_model.cpp:961-963
    mdl->yaux_valid = false;                                        // the factor on the device was recomputed
  }
  C_API_END();
}

int GPB_GetInitCovPar(REModelHandle handle, double* init_cov_pars) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !init_cov_pars) return set_error("GPB_GetInitCovPar: null argument");
  if (mdl->likelihood != "gaussian") {
    if (!(mdl->cov_pars_initialized || mdl->init_cov_pars_provided)) { init_cov_pars[0] = -1.; init_cov_pars[1] = -1.; return 0; }
    init_cov_pars[0] = mdl->init_cov_pars_tr[0]; init_cov_pars[1] = range_const(mdl) / mdl->init_cov_pars_tr[1];
    return 0;
  }
  if (!(mdl->cov_pars_initialized || mdl->init_cov_pars_provided)) {
    for (int j = 0; j < 3; ++j) init_cov_pars[j] = -1.;   // re_model.cpp GetInitCovPar: -1 if not available
    return 0;
  }
  transform_back(mdl, mdl->init_cov_pars_tr, init_cov_pars);
  C_API_END();
}

int GPB_GetNumIt(REModelHandle handle, int* num_it) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !num_it) return set_error("GPB_GetNumIt: null argument");
  *num_it = mdl->num_it;
  C_API_END();
}

int GPB_HIP_GetOptimInfo(REModelHandle handle, int* num_ll_evals, int* num_grad_evals, double* lr_cov_final) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_HIP_GetOptimInfo: null handle");
  if (num_ll_evals) *num_ll_evals = mdl->last_fit.num_ll_evals;
  if (num_grad_evals) *num_grad_evals = mdl->last_fit.num_grad_evals;
  if (lr_cov_final) *lr_cov_final = mdl->last_fit.lr_cov_final;
  C_API_END();
}

/* Test seam: the host optimiser for non-Gaussian likelihoods -- theta = (sigma1_2, a) -- with a caller-supplied stateful evaluator of the
   Laplace approximation (ops documented at gpb_laplace_fn in gpb_optim.h).  GPB_OptimCovPar drives the same optimiser with the device
   evaluator (device_laplace); the CPU tests drive it with the oracle. */
int GPB_HIP_OptimizeLaplaceWithCallback(const double* init_theta2, const char* optimizer, double lr_cov, double acc_rate_cov, int max_iter,
                                        double delta_rel_conv, bool use_nesterov_acc, int nesterov_schedule_version, int momentum_offset,
                                        const char* convergence_criterion, int m_lbfgs, int (*eval)(void*, int, double, double, double*),
                                        void* ctx, double* theta_out2, int* num_it, double* negll, int* num_evals) {
  C_API_BEGIN();
  if (!init_theta2 || !eval || !theta_out2) return set_error("GPB_HIP_OptimizeLaplaceWithCallback: null argument");
  GpbOptimConfig cfg;
  if (optimizer && optimizer[0]) cfg.optimizer = optimizer;
  if (lr_cov > 0.) cfg.lr_cov_init = lr_cov;
  if (acc_rate_cov > 0.) cfg.acc_rate_cov = acc_rate_cov;
  if (max_iter >= 0) cfg.max_iter = max_iter;
  if (delta_rel_conv > 0.) cfg.delta_rel_conv_init = delta_rel_conv;
  cfg.use_nesterov_acc = use_nesterov_acc;
  if (nesterov_schedule_version >= 0) cfg.nesterov_schedule_version = nesterov_schedule_version;
  if (momentum_offset >= 0) cfg.momentum_offset = momentum_offset;
  if (convergence_criterion && convergence_criterion[0] && std::string(convergence_criterion) != "default") cfg.convergence_criterion = convergence_criterion;
  if (m_lbfgs > 0) cfg.m_lbfgs = m_lbfgs;
  char err[512] = "";
  GpbLaplaceOptimResult res;
  if (gpb_optimize_laplace_cov_pars(cfg, eval, ctx, init_theta2, &res, err, (int)sizeof(err)))
    return set_error("%s", err[0] ? err : "evaluation callback failed");
  theta_out2[0] = res.theta[0]; theta_out2[1] = res.theta[1];
  if (num_it) *num_it = res.num_it;
  if (negll) *negll = res.negll;
  if (num_evals) *num_evals = res.num_evals;
  C_API_END();
}

/* Test seam and host half of GPB_OptimCovPar for likelihoods whose auxiliary parameters are estimated with the covariance parameters (gamma,
   negative_binomial): lbfgs on (log sigma1_2, log a, log aux) with the evaluation callback of gpb_laplace_aux_fn (gpb_optim.h). */
int GPB_HIP_OptimizeLaplaceAuxWithCallback(const double* init_theta2, const double* init_aux, int naux, double lr_cov, int max_iter, double delta_rel_conv,
                                           int m_lbfgs, int (*eval)(void*, int, double, double, const double*, int, double*), void* ctx,
                                           double* theta_out2, double* aux_out, int* num_it, double* negll, int* num_evals) {
  C_API_BEGIN();
  if (!init_theta2 || !init_aux || !eval || !theta_out2 || !aux_out || naux < 1 || naux > 8) return set_error("GPB_HIP_OptimizeLaplaceAuxWithCallback: invalid argument");
  GpbOptimConfig cfg;
  if (lr_cov > 0.) cfg.lr_cov_init = lr_cov;
  if (max_iter >= 0) cfg.max_iter = max_iter;
  if (delta_rel_conv > 0.) cfg.delta_rel_conv_init = delta_rel_conv;
  if (m_lbfgs > 0) cfg.m_lbfgs = m_lbfgs;
  char err[512] = "";
  GpbLaplaceAuxResult res;
  std::vector<double> aux(init_aux, init_aux + naux);
  if (gpb_optimize_laplace_cov_aux_pars(cfg, eval, ctx, naux, init_theta2, aux.data(), &res, err, (int)sizeof(err)))
    return set_error("%s", err[0] ? err : "evaluation callback failed");
  theta_out2[0] = res.theta[0]; theta_out2[1] = res.theta[1];
  std::copy(aux.begin(), aux.end(), aux_out);
  if (num_it) *num_it = res.num_it;
  if (negll) *negll = res.negll;
  if (num_evals) *num_evals = res.num_evals;
  C_API_END();
}

int GPB_HIP_FindInitialAuxParsHost(const char* likelihood, int32_t n, const double* y, const double* fixed_effects, double* aux_out) {
  C_API_BEGIN();
  if (!likelihood || !y || !aux_out || n < 2) return set_error("GPB_HIP_FindInitialAuxParsHost: invalid argument");
  if (num_aux_of(likelihood) < 1) return set_error("GPB_HIP_FindInitialAuxParsHost: likelihood '%s' has no auxiliary parameters on this path", likelihood);
  aux_out[0] = initial_aux_par(likelihood, n, y, fixed_effects);
  C_API_END();
}

/* Test seam and host half of the standard errors of a non-Gaussian model's covariance parameters (GPB_GetCovPar(calc_std_dev = true) drives it with the
   device evaluator): CalcStdDevCovParAuxParsNonGaussian (re_model_template.h:11029-11117) on theta = (sigma1_2, a) with the evaluation callback of
   GPB_HIP_OptimizeLaplaceWithCallback.  se_out2 = standard errors of (sigma1_2, rho). */
int GPB_HIP_LaplaceStdErrorsWithCallback(const double* theta2, double range_const_, int (*eval)(void*, int, double, double, double*), void* ctx,
                                         double* se_out2) {
  C_API_BEGIN();
  if (!theta2 || !eval || !se_out2) return set_error("GPB_HIP_LaplaceStdErrorsWithCallback: null argument");
  char err[512] = "";
  if (gpb_laplace_std_errors(eval, ctx, theta2, range_const_, se_out2, err, (int)sizeof(err))) return set_error("%s", err[0] ? err : "evaluation callback failed");
  C_API_END();
}

/* Test seam and host half of the fits of non-Gaussian models WITH a linear predictor (GPB_OptimLinRegrCoefCovPar drives it with the device
   evaluator): intercept detection, scaling of the covariates, initial coefficients and step-cap constants (laplace_coef_setup), lbfgs on
   (log sigma1_2, log a, beta) (gpb_optimize_laplace_coef_cov_pars), coefficients back on the original scale.  eval: gpb_laplace_fe_fn (gpb_optim.h). */
int GPB_HIP_OptimizeLaplaceCoefWithCallback(const char* likelihood, int32_t n, int32_t p, const double* X_colmajor, const double* y,
                                            const double* fixed_effects, const double* init_theta2, const double* init_coef, bool init_coef_from_iid_model,
                                            double lr_cov, int max_iter, double delta_rel_conv, int m_lbfgs,
                                            int (*eval)(void*, int, double, double, const double*, double*, double*),
                                            void* ctx, double* theta_out2, double* coef_out, int* num_it, double* negll, double* init_coef_out) {
  C_API_BEGIN();
  if (!likelihood || n < 1 || p < 1 || !X_colmajor || !y || !init_theta2 || !eval || !theta_out2 || !coef_out) return set_error("GPB_HIP_OptimizeLaplaceCoefWithCallback: invalid argument");
  GpbOptimConfig cfg;
  if (lr_cov > 0.) cfg.lr_cov_init = lr_cov;
  if (max_iter >= 0) cfg.max_iter = max_iter;
  if (delta_rel_conv > 0.) cfg.delta_rel_conv_init = delta_rel_conv;
  if (m_lbfgs > 0) cfg.m_lbfgs = m_lbfgs;
  if (const char* e = std::getenv("GPB_OPTIM_TRACE")) cfg.trace = std::atoi(e) != 0;
  std::vector<double> ic;
  if (init_coef) ic.assign(init_coef, init_coef + p);
  else if (init_coef_from_iid_model && iid_model_init_coef(std::string(likelihood), n, p, X_colmajor, y, fixed_effects, cfg, &ic, nullptr)) return -1;
  if (init_coef_out && !ic.empty()) std::copy(ic.begin(), ic.end(), init_coef_out);
  LaplaceCoefSetup su;
  if (laplace_coef_setup(std::string(likelihood), n, p, X_colmajor, y, fixed_effects, init_theta2[0], ic.empty() ? nullptr : ic.data(), &su)) return -1;
  char err[512] = "";
  GpbLaplaceCoefResult res;
  std::vector<double> beta = su.beta;
  if (gpb_optimize_laplace_coef_cov_pars(cfg, eval, ctx, n, p, su.Xs.data(), fixed_effects, su.C_mu, su.C_sigma2, init_theta2, beta.data(), &res, err, (int)sizeof(err)))
    return set_error("%s", err[0] ? err : "evaluation callback failed");
  transform_back_coef(su, beta);
  theta_out2[0] = res.theta[0]; theta_out2[1] = res.theta[1];
  std::copy(beta.begin(), beta.end(), coef_out);
  if (num_it) *num_it = res.num_it;
  if (negll) *negll = res.negll;
  C_API_END();
}

/* Test seam and host half of GPB_GetCoef(calc_std_dev = true) for non-Gaussian models: CalcStdDevCoefNonGaussian (re_model_template.h:10851-10897) with
   the evaluation callback of GPB_HIP_OptimizeLaplaceCoefWithCallback.  X: original covariates (column-major n x p); theta2 = (sigma1_2, a). */
int GPB_HIP_LaplaceCoefStdErrorsWithCallback(int32_t n, int32_t p, const double* X_colmajor, const double* fixed_effects, const double* theta2,
                                             const double* coef, int (*eval)(void*, int, double, double, const double*, double*, double*), void* ctx,
                                             double* se_out) {
  C_API_BEGIN();
  if (n < 1 || p < 1 || !X_colmajor || !theta2 || !coef || !eval || !se_out) return set_error("GPB_HIP_LaplaceCoefStdErrorsWithCallback: invalid argument");
  char err[512] = "";
  if (gpb_laplace_coef_std_errors(eval, ctx, n, p, X_colmajor, fixed_effects, theta2, coef, se_out, err, (int)sizeof(err))) return set_error("%s", err[0] ? err : "evaluation callback failed");
  C_API_END();
}

/* Test seam: FindInitCovPar (re_model_template.h:4849-4968, cov_fcts.h:1422-1683) on host data alone.  coords0_colmajor = the first
   cluster's coordinates in Vecchia order; the generator is seeded with `seed` and advanced by one std::shuffle of `shuffle_len`
   elements when shuffle_len > 0 (what vecchia_ordering = "random" does to a one-cluster model before the initial values are drawn).
   theta3 = (sigma2, sigma1_2 / sigma2, a) on the transformed scale. */
int GPB_HIP_FindInitCovParHost(int32_t num_data, const double* y_data, const double* fixed_effects, int32_t n0, int32_t dim,
                               const double* coords0_colmajor, int cov_type, int seed, int32_t shuffle_len, double* theta3) {
  C_API_BEGIN();
  if (!y_data || !coords0_colmajor || !theta3 || num_data < 2 || n0 < 2 || dim < 1 || cov_type < 0 || cov_type > 2)
    return set_error("GPB_HIP_FindInitCovParHost: invalid argument");
  std::mt19937 rng(seed);
  if (shuffle_len > 0) { std::vector<int> idx(shuffle_len); std::iota(idx.begin(), idx.end(), 0); std::shuffle(idx.begin(), idx.end(), rng); }
  if (find_init_cov_par_core(num_data, y_data, fixed_effects, n0, dim, coords0_colmajor, cov_type, rng, theta3)) return -1;
  C_API_END();
}

/* Test seam: the host half of the full-scale Vecchia (VIF) likelihood and its analytic gradient (vif_terms_core: Sigma_m and its factor, the Woodbury
   matrix, the k x k inverses and traces) with the two device passes supplied by the caller -- the CPU suite hands it a numpy restatement of
   gpb_hip_vecchia_vif_factor / gpb_hip_vecchia_vif_grad_sums (tests/test_vif.py).  t7 = {y' Psi^-1 y, log|Psi|, #(D <= 0), g1_var, g2_var, g1_range, g2_range}. */
int GPB_HIP_VifTermsWithCallback(int32_t k, int32_t d, const double* ip_colmajor, int cov_type, double ratio, double a, int with_grad,
                                 int (*factor)(void*, const double*, int, double*, double*),
                                 int (*gsums)(void*, const double*, const double*, const double*, const double*, const double*, double*), void* ctx, double* t7) {
  C_API_BEGIN();
  if (k < 1 || d < 1 || !ip_colmajor || !factor || (with_grad && !gsums) || !t7) return set_error("GPB_HIP_VifTermsWithCallback: invalid argument");
  double t[7] = {0, 0, 0, 0, 0, 0, 0};
  if (vif_terms_core(k, d, ip_colmajor, cov_type, ratio, a, with_grad, factor, gsums, ctx, t, nullptr)) return -1;
  std::copy(t, t + 7, t7);
  C_API_END();
}

/* Test seam: the same host optimiser with a caller-supplied evaluation callback (tests/ drives it with the CPU oracle, so the
   control flow is checked against the reference's trajectories without a GPU).  init_theta / theta_out: TRANSFORMED scale. */
int GPB_HIP_OptimizeGaussianWithCallback(int32_t num_data, const double* init_theta, const char* optimizer, double lr_cov, double acc_rate_cov,
                                         int max_iter, double delta_rel_conv, bool use_nesterov_acc, int nesterov_schedule_version,
                                         int momentum_offset, const char* convergence_criterion, int m_lbfgs, double range_const_,
                                         int (*terms)(void*, double, double, int, double*), void* ctx, double* theta_out, int* num_it,
                                         double* negll, int* num_evals2, const int* estimate_cov_par_index) {
  C_API_BEGIN();
  if (!init_theta || !terms || !theta_out) return set_error("GPB_HIP_OptimizeGaussianWithCallback: null argument");
  GpbOptimConfig cfg;
  if (estimate_cov_par_index && estimate_cov_par_index[0] >= 0) std::copy(estimate_cov_par_index, estimate_cov_par_index + 3, cfg.estimate_cov_par_index);
  if (optimizer && optimizer[0]) cfg.optimizer = optimizer;
  if (lr_cov > 0.) cfg.lr_cov_init = lr_cov;
  if (acc_rate_cov > 0.) cfg.acc_rate_cov = acc_rate_cov;
  if (max_iter >= 0) cfg.max_iter = max_iter;
  if (delta_rel_conv > 0.) cfg.delta_rel_conv_init = delta_rel_conv;
  cfg.use_nesterov_acc = use_nesterov_acc;
  if (nesterov_schedule_version >= 0) cfg.nesterov_schedule_version = nesterov_schedule_version;
  if (momentum_offset >= 0) cfg.momentum_offset = momentum_offset;
  if (convergence_criterion && convergence_criterion[0] && std::string(convergence_criterion) != "default") cfg.convergence_criterion = convergence_criterion;
  if (m_lbfgs > 0) cfg.m_lbfgs = m_lbfgs;
  cfg.range_const = range_const_;
  char err[512] = "";
  GpbOptimResult res;
  if (gpb_optimize_gaussian_cov_pars(cfg, num_data, terms, ctx, init_theta, &res, err, (int)sizeof(err)))
    return set_error("%s", err[0] ? err : "evaluation callback failed");
  std::copy(res.theta, res.theta + 3, theta_out);
  if (num_it) *num_it = res.num_it;
  if (negll) *negll = res.negll;
  if (num_evals2) { num_evals2[0] = res.num_ll_evals; num_evals2[1] = res.num_grad_evals; }
  C_API_END();
}

/* Host half of the Vecchia prediction 'order_obs_first_cond_all' (CalcPredVecchiaObservedFirstOrder with CondObsOnly = false,
   src/GPBoost/Vecchia_utils.cpp:2061-2090), and test seam for it: given the factor rows of the APPENDED prediction points -- neighbour
   indices into (observed, prediction) points, nn < n_obs + i for row i, -1 padded; A_i = C_nn^-1 c; D_i with the nugget, transformed scale
   (what vecchia_point_kernel<MODE_FACTOR> leaves for them, neighbours searched with end_search_at = -1) --
     mean = Bp^-1 (-Bpo y)          forward substitution: Bp = I - A_pp is unit lower triangular (:2061-2064)
     cov  = sigma2 Bp^-1 Dp Bp^-T   rows of Bp^-1 by the same recursion (:2077-2090); nugget removed from the diagonal unless predict_response
   var_out (n_pred) and cov_out (n_pred x n_pred, row-major) may each be NULL.  Dense in the number of prediction points (the reference keeps
   Bp^-1 sparse): refused above 20000 points.  The device wiring into GPB_PredictREModel is not done yet. */
int GPB_HIP_PredictCondAllHost(int32_t n_obs, int32_t n_pred, int32_t m, const int32_t* nn_pred, const double* A_pred, const double* D_pred,
                               const double* y_obs, double sigma2, bool predict_response, double* mean_out, double* var_out, double* cov_out) {
  C_API_BEGIN();
  if (!nn_pred || !A_pred || !D_pred || !y_obs || !mean_out || n_obs < 1 || n_pred < 1 || m < 1)
    return set_error("GPB_HIP_PredictCondAllHost: invalid argument");
  const bool need_rows = var_out || cov_out;
  if (need_rows && n_pred > 20000) return set_error("GPB_HIP_PredictCondAllHost: predictive (co)variances for %d points (dense limit: 20000)", n_pred);
  std::vector<double> L;                       // row i of Bp^-1, columns 0..i (lower triangular, row-major full storage)
  if (need_rows) L.assign((size_t)n_pred * n_pred, 0.);
  for (int i = 0; i < n_pred; ++i) {
    double mu = 0.;
    if (need_rows) L[(size_t)i * n_pred + i] = 1.;
    for (int j = 0; j < m; ++j) {
      const int c = nn_pred[(size_t)i * m + j];
      if (c < 0) continue;
      if (c >= n_obs + i) return set_error("GPB_HIP_PredictCondAllHost: row %d has neighbour %d that does not precede it", i, c);
      const double a = A_pred[(size_t)i * m + j];
      if (c < n_obs) mu += a * y_obs[c];
      else {
        const int q = c - n_obs;
        mu += a * mean_out[q];
        if (need_rows) for (int k = 0; k <= q; ++k) L[(size_t)i * n_pred + k] += a * L[(size_t)q * n_pred + k];
      }
    }
    mean_out[i] = mu;
  }
  if (need_rows) {
    for (int i = 0; i < n_pred; ++i) {
      for (int k = cov_out ? 0 : i; k <= i; ++k) {
        double sacc = 0.;
        for (int j = 0; j <= k; ++j) sacc += L[(size_t)i * n_pred + j] * D_pred[j] * L[(size_t)k * n_pred + j];
        double v = sigma2 * sacc;
        if (i == k && !predict_response) v -= sigma2;
        if (cov_out) { cov_out[(size_t)i * n_pred + k] = v; cov_out[(size_t)k * n_pred + i] = v; }
        if (var_out && i == k) var_out[i] = v;
      }
    }
  }
  C_API_END();
}

/* Test seam for the host half of the unique-location mapping (no device needed): uniques_out (n, the first *num_unique entries are the positions of
   the first appearances, ascending) and unique_idx_out (n) of DetermineUniqueDuplicateCoordsFast (src/GPBoost/GP_utils.cpp:472-548) as
   GPB_CreateREModel applies it to the (shuffled) coordinates of one non-Gaussian GP.  coords: column-major n x d. */
int GPB_HIP_UniqueLocationsHost(int32_t n, int32_t d, const double* coords_colmajor, int32_t* num_unique, int32_t* uniques_out, int32_t* unique_idx_out) {
  C_API_BEGIN();
  if (n < 1 || d < 1 || !coords_colmajor || !num_unique || !uniques_out || !unique_idx_out) return set_error("GPB_HIP_UniqueLocationsHost: invalid argument");
  std::vector<double> c(coords_colmajor, coords_colmajor + (size_t)n * d);
  std::vector<int> uq, ui;
  unique_locations(c, n, d, &uq, &ui);
  *num_unique = (int32_t)uq.size();
  std::copy(uq.begin(), uq.end(), uniques_out);
  std::copy(ui.begin(), ui.end(), unique_idx_out);
  C_API_END();
}

/* Test seams of the host half of the response-scale predictions (no device needed): Likelihood::PredictResponse (likelihoods.h:9626-9672) in place on
   (mean, var) -- var is read always and written if predict_var -- and the Gauss-Hermite rule it integrates the logit likelihood with. */
int GPB_HIP_PredictResponseHost(const char* likelihood, int32_t n, double* mean_inout, double* var_inout, bool predict_var, double delta_conv_mode_finding) {
  C_API_BEGIN();
  if (!likelihood || n < 0 || !mean_inout || !var_inout) return set_error("GPB_HIP_PredictResponseHost: invalid argument");
  if (!predict_response_host(std::string(likelihood), n, mean_inout, var_inout, predict_var, delta_conv_mode_finding > 0. ? delta_conv_mode_finding : 1e-8))
    return set_error("GPB_HIP_PredictResponseHost: likelihood '%s' is not on the MI355X hot path of this library", likelihood);
  C_API_END();
}
int GPB_HIP_GaussHermiteHost(int32_t order, double* nodes_out, double* adaptive_weights_out) {
  C_API_BEGIN();
  if (order < 1 || order > 200 || !nodes_out || !adaptive_weights_out) return set_error("GPB_HIP_GaussHermiteHost: invalid argument");
  std::vector<double> xs, aw;
  gauss_hermite_adaptive(order, &xs, &aw);
  std::copy(xs.begin(), xs.end(), nodes_out);
  std::copy(aw.begin(), aw.end(), adaptive_weights_out);
  C_API_END();
}

/* c_api.h:1588-1610 -- only what the obs-only Vecchia prediction needs is kept: coordinates, prediction type, #neighbours */
int GPB_SetPredictionData(REModelHandle handle, int32_t num_data_pred, const int32_t* cluster_ids_data_pred, const char* re_group_data_pred,
                          const double* re_group_rand_coef_data_pred, double* gp_coords_data_pred, const double* gp_rand_coef_data_pred,
                          const double* covariate_data_pred, const char* vecchia_pred_type, int num_neighbors_pred,
                          double /*cg_delta_conv_pred*/, int /*nsim_var_pred*/, int /*rank_pred_approx_matrix_lanczos*/) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_SetPredictionData: null handle");
  const char* scope = "is not on the MI355X path of this library";
  if (cluster_ids_data_pred || re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred || covariate_data_pred)
    return set_error("GPB_SetPredictionData: cluster ids / grouped effects / random coefficients / covariates for prediction %s", scope);
  if (vecchia_pred_type && vecchia_pred_type[0]) {
    const std::string t = vecchia_pred_type;
    // SUPPORTED_VECCHIA_PRED_TYPES_GAUSS_ of the reference; only the first is implemented on the device
    if (t != "order_obs_first_cond_obs_only" && t != "order_obs_first_cond_all" && t != "order_pred_first" &&
        t != "latent_order_obs_first_cond_obs_only" && t != "latent_order_obs_first_cond_all")
      return set_error("Prediction type '%s' is not supported for the Veccia approximation ", t.c_str());
    mdl->vecchia_pred_type = t;
  }
  if (num_neighbors_pred > 0) mdl->num_neighbors_pred = num_neighbors_pred;
  if (gp_coords_data_pred) {
    if (num_data_pred <= 0) return set_error("GPB_SetPredictionData: num_data_pred = %d", num_data_pred);
    mdl->coords_pred.assign(gp_coords_data_pred, gp_coords_data_pred + (size_t)num_data_pred * mdl->d);
    mdl->num_data_pred = num_data_pred;
  }
  C_API_END();
}

/* c_api.h:1640-1660 -- predictive mean and variances (or the, here diagonal, covariance matrix) at new locations for the Gaussian
 * one-cluster Vecchia model, vecchia_pred_type "order_obs_first_cond_obs_only" (REModel::Predict, re_model.cpp:1083-1215 ->
 * CalcPredVecchiaObservedFirstOrder(CondObsOnly = true), Vecchia_utils.cpp:1701-2060) */
int GPB_PredictREModel(REModelHandle handle, const double* y_data, int32_t num_data_pred, double* out_predict, bool predict_cov_mat,
                       bool predict_var, bool predict_response, bool sample_posterior, bool sample_prior, int /*num_post_samples*/,
                       int /*num_prior_samples*/, const int32_t* cluster_ids_data_pred, const char* re_group_data_pred,
                       const double* re_group_rand_coef_data_pred, double* gp_coords_data_pred, const double* gp_rand_coef_data_pred,
                       const double* cov_pars, const double* covariate_data_pred, bool use_saved_data, const double* fixed_effects,
                       const double* fixed_effects_pred) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !out_predict) return set_error("GPB_PredictREModel: null argument");
  if (mdl && mdl->vif && mdl->likelihood == "gaussian") {
    // full-scale Vecchia, 'order_obs_first_cond_obs_only' (the reference's default for Gaussian data; CalcPredVecchiaObservedFirstOrder with the
    // full_scale_vecchia arguments, Vecchia_utils.cpp:1701-2060; re_model_template.h:4041-4056): y_p = C_p Sigma_m^-1 eta + e_p with the residual
    // e_p conditioning on the nearest observed points -> mean = A_p y_nn + (B C)_p W^-1 (B C)' D^-1 B y, var = sigma2 (D_p + (B C)_p W^-1 (B C)_p')
    const char* vscope = "is not on the MI355X path of this library (full-scale Vecchia prediction: 'order_obs_first_cond_obs_only' / 'order_obs_first_cond_all', no covariates / samples)";
    if (sample_posterior || sample_prior) return set_error("GPB_PredictREModel: samples of a full-scale Vecchia model %s", vscope);
    if (predict_cov_mat && predict_var) return set_error("Calculation of both the predictive covariance matrix and variances is not supported. Choose one option (predict_cov_mat or predict_var)");
    if (cluster_ids_data_pred || re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred || covariate_data_pred || mdl->p_cov > 0)
      return set_error("GPB_PredictREModel: cluster ids / grouped effects / random coefficients / covariates for prediction %s", vscope);
    // (the reference has exactly these two for full-scale Vecchia models: 'order_pred_first' and the latent types are fatal there, re_model_template.h:4072-4110)
    const bool v_cond_all = mdl->vecchia_pred_type == "order_obs_first_cond_all";
    if (mdl->vecchia_pred_type != "order_obs_first_cond_obs_only" && !v_cond_all) return set_error("GPB_PredictREModel: vecchia_pred_type '%s' of a full-scale Vecchia model %s", mdl->vecchia_pred_type.c_str(), vscope);
    const double* cpv = gp_coords_data_pred;
    int npv = num_data_pred;
    if (use_saved_data) { cpv = mdl->coords_pred.empty() ? nullptr : mdl->coords_pred.data(); npv = mdl->num_data_pred; }
    if (!cpv || npv <= 0) return set_error("GPB_PredictREModel: no coordinates for prediction (gp_coords_data_pred / GPB_SetPredictionData)");
    double trv[3];
    if (cov_pars) { double c3[3] = {cov_pars[0], cov_pars[1], cov_pars[2]}; if (transform_cov_pars(mdl, c3, trv)) return -1; }
    else {
      if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");
      std::copy(mdl->cov_pars_tr, mdl->cov_pars_tr + 3, trv);
    }
    if (!y_data && !mdl->y_set) return set_error("GPB_PredictREModel: y_data is NULL and no response has been set by an earlier call");
    const double* fev = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
    if (prediction_response(mdl, y_data, fev)) return -1;
    VifSolve vs;
    double t3[3];
    if (vif_terms(mdl, trv[1], trv[2], t3, &vs)) return -1;
    int nnpv = mdl->num_neighbors_pred > 0 ? mdl->num_neighbors_pred : 2 * mdl->num_neighbors;
    if (nnpv > (v_cond_all ? mdl->n + npv - 1 : mdl->n)) nnpv = v_cond_all ? mdl->n + npv - 1 : mdl->n;
    if (nnpv > 126) nnpv = 126;
    const int kv = mdl->num_ind_points;
    std::vector<double> up(npv), Dp(npv), BC((size_t)npv * kv);
    if (v_cond_all) {
      // 'order_obs_first_cond_all' (CalcPredVecchiaObservedFirstOrder, CondObsOnly = false, Vecchia_utils.cpp:1975-2046): the device returns the rows of
      // [Bpo Bp] of the appended points, -(Bpo y) and (B~ C~)_p; here the forward substitutions with Bp = I - A_pp (unit lower triangular: a neighbour
      // precedes its point) --  mean = Bp^-1 (-Bpo y + (B~ C~)_p v),  T = Bp^-1 (B~ C~)_p,  rows of Bp^-1 --  and
      // cov = sigma2 (Bp^-1 Dp Bp^-T + T W^-1 T')  [- sigma2 I for the latent process].  The reference's eight-term expression (:2028-2046) is T W^-1 T'
      // written out with W = Sigma_m + (B C)' D^-1 (B C).  Dense in the number of prediction points, as the 'cond_all' type of the plain Vecchia model here.
      const bool second2 = predict_var || predict_cov_mat;
      if (second2 && npv > 20000) return set_error("GPB_PredictREModel: predictive (co)variances of 'order_obs_first_cond_all' for %d points (dense limit: 20000)", npv);
      int mu_ = 0;
      std::vector<int32_t> nnr((size_t)npv * nnpv);
      std::vector<double> Ap((size_t)npv * nnpv);
      if (gpb_hip_vecchia_vif_predict_cond_all(mdl->vh, npv, cpv, nnpv, mdl->ip.data(), mdl->cov_type, trv[1], trv[2], vs.Linv.data(), &mu_, nnr.data(), Ap.data(),
                                               up.data(), Dp.data(), BC.data(), nullptr)) return shim_error();
      std::vector<double> Lr(second2 ? (size_t)npv * npv : 0, 0.);       // rows of Bp^-1
      std::vector<double> mean(npv);
      for (int i = 0; i < npv; ++i) {
        double* bc = BC.data() + (size_t)i * kv;
        if (second2) Lr[(size_t)i * npv + i] = 1.;
        double w = -up[i];
        for (int j = 0; j < kv; ++j) w += bc[j] * vs.v[j];
        for (int j = 0; j < mu_; ++j) {
          const int c = nnr[(size_t)i * mu_ + j];
          if (c < mdl->n) continue;                              // (-1 padding and observed neighbours: already in u and BC)
          const int q = c - mdl->n;
          if (q >= i) return set_error("GPB_PredictREModel: prediction point %d has neighbour %d that does not precede it", i, q);
          const double aij = Ap[(size_t)i * mu_ + j];
          w += aij * mean[q];
          if (second2) {
            const double* bq = BC.data() + (size_t)q * kv;       // (already T_q: rows are finished in order)
            for (int r = 0; r < kv; ++r) bc[r] += aij * bq[r];
            for (int r = 0; r <= q; ++r) Lr[(size_t)i * npv + r] += aij * Lr[(size_t)q * npv + r];
          }
        }
        mean[i] = w;
        out_predict[i] = w + (fixed_effects_pred ? fixed_effects_pred[i] : 0.);
      }
      if (second2) {
        std::vector<double> T((size_t)npv * kv);
        for (int i = 0; i < npv; ++i) {                          // T_i <- L_W^-1 T_i
          const double* bc = BC.data() + (size_t)i * kv;
          double* tmp = T.data() + (size_t)i * kv;
          for (int r = 0; r < kv; ++r) {
            double v = bc[r];
            for (int j = 0; j < r; ++j) v -= vs.Lw[(size_t)r * kv + j] * tmp[j];
            tmp[r] = v / vs.Lw[(size_t)r * kv + r];
          }
        }
        for (int i = 0; i < npv; ++i) {
          for (int j = predict_cov_mat ? 0 : i; j <= i; ++j) {
            double qq = 0.;
            for (int r = 0; r < kv; ++r) qq += T[(size_t)i * kv + r] * T[(size_t)j * kv + r];
            for (int r = 0; r <= j; ++r) qq += Lr[(size_t)i * npv + r] * Dp[r] * Lr[(size_t)j * npv + r];
            const double v = trv[0] * (qq - ((i == j && !predict_response) ? 1. : 0.));
            if (predict_cov_mat) { out_predict[npv + (size_t)i * npv + j] = v; out_predict[npv + (size_t)j * npv + i] = v; }
            else out_predict[npv + i] = v;
          }
        }
      }
      mdl->yaux_valid = false;
      return 0;
    }
    if (gpb_hip_vecchia_vif_predict_obs_only(mdl->vh, npv, cpv, nnpv, mdl->ip.data(), mdl->cov_type, trv[1], trv[2], vs.Linv.data(), up.data(), Dp.data(), BC.data(), nullptr))
      return shim_error();
    // T = L_W^-1 (B C)_p' column by column (only when second moments are asked for): var_p = D_p + ||T_p||^2, and -- the residuals of two
    // prediction points being independent given the observed ones (Bp = I) -- cov_pq = T_p . T_q off the diagonal
    const bool second = predict_var || predict_cov_mat;
    std::vector<double> T(second ? (size_t)npv * kv : 0);
    for (int i = 0; i < npv; ++i) {
      const double* bc = BC.data() + (size_t)i * kv;
      double mu = -up[i];
      for (int j = 0; j < kv; ++j) mu += bc[j] * vs.v[j];
      out_predict[i] = mu + (fixed_effects_pred ? fixed_effects_pred[i] : 0.);
      if (second) {
        double* tmp = T.data() + (size_t)i * kv;
        for (int r = 0; r < kv; ++r) {
          double v = bc[r];
          for (int j = 0; j < r; ++j) v -= vs.Lw[(size_t)r * kv + j] * tmp[j];
          tmp[r] = v / vs.Lw[(size_t)r * kv + r];
        }
      }
    }
    if (second) for (int i = 0; i < npv; ++i) {
      for (int j = predict_cov_mat ? 0 : i; j <= i; ++j) {
        double qq = 0.;
        for (int r = 0; r < kv; ++r) qq += T[(size_t)i * kv + r] * T[(size_t)j * kv + r];
        const double v = trv[0] * (qq + (i == j ? Dp[i] - (predict_response ? 0. : 1.) : 0.));
        if (predict_cov_mat) { out_predict[npv + (size_t)i * npv + j] = v; out_predict[npv + (size_t)j * npv + i] = v; }
        else out_predict[npv + i] = v;
      }
    }
    mdl->yaux_valid = false;
    return 0;
  }
  const char* scope = "is not on the MI355X path of this library (prediction: one-cluster Gaussian Vecchia model, 'order_obs_first_cond_obs_only')";
  if (mdl->likelihood != "gaussian" && !mdl->eh && mdl->vhs.size() == 1) {
    // non-Gaussian (Vecchia-Laplace) models, the reference's default prediction type for them ('latent_order_obs_first_cond_obs_only'):
    //   latent mean -Bpo mode, latent (co)variances Dp + Bpo (Sigma^-1 + W)^-1 Bpo' (PredictLaplaceApproxVecchia, likelihoods.h:8563-8824 -- the
    //   exact value of its "cholesky" branch, which its "iterative" branch estimates with nsim_var_pred random vectors), and the response-scale
    //   predictions from them (PredictResponse, :9626-9672).  Repeated prediction locations share a random effect (re_model_template.h:3976-3988).
    const char* lscope = "is not on the MI355X path of this library (non-Gaussian likelihoods: 'latent_order_obs_first_cond_obs_only' / 'latent_order_obs_first_cond_all', no samples)";
    if (sample_posterior || sample_prior) return set_error("GPB_PredictREModel: posterior / prior samples %s", lscope);
    if (predict_response && predict_cov_mat) return set_error("Calculation of the predictive covariance matrix is not supported when predicting the response variable (label) for non-Gaussian likelihoods");   // :3526-3529
    if (predict_cov_mat && predict_var) return set_error("Calculation of both the predictive covariance matrix and variances is not supported. Choose one option (predict_cov_mat or predict_var)");
    if (cluster_ids_data_pred || re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred)
      return set_error("GPB_PredictREModel: cluster ids / grouped effects / random coefficients for prediction %s", lscope);
    if (mdl->p_cov > 0 && !mdl->coef_estimated) return set_error("GPB_PredictREModel: the model has covariates but no estimated coefficients");
    if (mdl->p_cov > 0 && !covariate_data_pred) return set_error("No covariate data is provided in 'covariate_data_pred' but the model has linear regression covariates");   // re_model.cpp Predict
    if (mdl->p_cov == 0 && covariate_data_pred) return set_error("Covariate data is provided in 'covariate_data_pred' but the model has no linear regression covariates");
    if (mdl->p_cov > 0 && use_saved_data) return set_error("GPB_PredictREModel: saved prediction data together with covariates %s", lscope);
    const std::string& pt = mdl->vecchia_pred_type;
    // for non-Gaussian likelihoods the two 'order_obs_first_*' names mean their latent counterparts (re_model_template.h:7112-7124)
    const bool lat_cond_all = pt == "latent_order_obs_first_cond_all" || pt == "order_obs_first_cond_all";
    if (!pt.empty() && pt != "order_obs_first_cond_obs_only" && pt != "latent_order_obs_first_cond_obs_only" && !lat_cond_all)
      return set_error("GPB_PredictREModel: vecchia_pred_type '%s' %s", pt.c_str(), lscope);
    // full-scale Vecchia (round 6; PredictLaplaceApproxFSVA, likelihoods.h:7999-8535): means, variances and covariance matrices of both latent prediction types;
    // with covariates the mode is found at the location parameter fixed effects + X beta and X_pred beta joins the latent mean, as for the Vecchia models
    const double* cpl = gp_coords_data_pred;
    int npl = num_data_pred;
    if (use_saved_data) { cpl = mdl->coords_pred.empty() ? nullptr : mdl->coords_pred.data(); npl = mdl->num_data_pred; }
    if (!cpl || npl <= 0) return set_error("GPB_PredictREModel: no coordinates for prediction (gp_coords_data_pred / GPB_SetPredictionData)");
    if (!y_data && !mdl->y_set) return set_error("GPB_PredictREModel: y_data is NULL and no response has been set by an earlier call");
    if (predict_cov_mat && npl > 20000) return set_error("GPB_PredictREModel: predictive covariance matrix for %d points (dense limit: 20000)", npl);
    double s12, rho;
    if (cov_pars) { s12 = cov_pars[0]; rho = cov_pars[1]; }
    else {
      if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");
      s12 = mdl->cov_pars_tr[0]; rho = range_const(mdl) / mdl->cov_pars_tr[1];
    }
    if (!(s12 > 0.) || !(rho > 0.)) return set_error("Covariance parameters need to be positive (found %g, %g)", s12, rho);
    const double* fel = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
    std::vector<double> fe_lin;                  // location parameter of the observed data = fixed effects + X beta (UpdateFixedEffects, re_model_template.h:2859-2871)
    if (mdl->p_cov > 0) {
      fe_lin.assign(mdl->n, 0.);
      for (int i = 0; i < mdl->n; ++i) { double v = fel ? fel[i] : 0.; for (int j = 0; j < mdl->p_cov; ++j) v += mdl->X[(size_t)j * mdl->n + i] * mdl->beta[j]; fe_lin[i] = v; }
      fel = fe_lin.data();
    }
    if (y_data) { if (laplace_upload_data(mdl, y_data, fel)) return -1; }
    else if (laplace_upload_fixed_effects(mdl, fel)) return -1;
    const double a_tr = range_const(mdl) / rho;
    std::vector<double> mode(mdl->n_re > 0 ? mdl->n_re : mdl->n);
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_logit(mdl->vh, mdl->cov_type, s12, a_tr, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, mdl->cg_max_num_it,
                                      mdl->cg_max_num_it_tridiag, mdl->cg_delta_conv, mdl->delta_conv_mode_finding, 1, mdl->lap_info, mode.data()))
      return shim_error();
    if (!mdl->vif && gpb_hip_vecchia_set_y(mdl->vh, mode.data())) return shim_error();          // the "response" of the prediction is the mode (Vecchia order)
    const int nnpl = mdl->num_neighbors_pred > 0 ? mdl->num_neighbors_pred : 2 * mdl->num_neighbors;     // re_model_template.h:299
    const bool need_var = predict_var || predict_response;            // every likelihood on the path needs the latent variance for its response mean
    // unique prediction locations: first appearances, in the order given (DetermineUniqueDuplicateCoordsFast on the prediction coordinates)
    const int dl = mdl->d;
    std::vector<double> cpv(cpl, cpl + (size_t)npl * dl);
    std::vector<int> uq, ui;
    unique_locations(cpv, npl, dl, &uq, &ui);
    const int nu = (int)uq.size();
    std::vector<double> cu((size_t)nu * dl);
    for (int j = 0; j < dl; ++j) for (int u = 0; u < nu; ++u) cu[(size_t)j * nu + u] = cpv[(size_t)j * npl + uq[u]];
    std::vector<double> mu_u(nu), var_u(need_var ? nu : 0), cov_u(predict_cov_mat ? (size_t)nu * nu : 0);
    int cg_it = 0;
    if (mdl->vif) {      // both latent types in one call: a negative neighbour count = 'latent_order_obs_first_cond_all'
      const int nnv = std::min(nnpl, 126);
      if (gpb_hip_vecchia_vif_laplace_predict(mdl->vh, nu, cu.data(), lat_cond_all ? -nnv : nnv, mdl->cov_type, s12, a_tr, mdl->cg_max_num_it, kPredVarCgTol, mu_u.data(),
                                              need_var ? var_u.data() : nullptr, predict_cov_mat ? cov_u.data() : nullptr, nullptr, &cg_it)) return shim_error();
    } else if (lat_cond_all) {
      // 'latent_order_obs_first_cond_all' (PredictLaplaceApproxVecchia with CondObsOnly = false, likelihoods.h:8603-8606, 8790-8821): the prediction points
      // condition on the observed AND the preceding prediction points.  Device: factor rows of the appended points (latent: no nugget); host: the
      // rows of Bp^-1 (forward substitution, as GPB_HIP_PredictCondAllHost) and of C = Bp^-1 Bpo; device: the quadratic forms C (Sigma^-1 + W)^-1 C'.
      //   mean = -C mode,   cov = Bp^-1 Dp Bp^-T + C (Sigma^-1 + W)^-1 C'
      if (nu > 5000) return set_error("GPB_PredictREModel: '%s' for %d > 5000 distinct prediction locations %s", pt.c_str(), nu, lscope);
      const int n_obs = mdl->n_re > 0 ? mdl->n_re : mdl->n;
      int mcap = std::min(nnpl, 126);
      if (mcap > n_obs + nu - 1) mcap = n_obs + nu - 1;
      int mu_ = 0;
      std::vector<int32_t> nnr((size_t)nu * mcap);
      std::vector<double> Ap((size_t)nu * mcap), Dp(nu);
      if (gpb_hip_vecchia_predict_cond_all_latent(mdl->vh, nu, cu.data(), mcap, mdl->cov_type, s12, a_tr, &mu_, nnr.data(), Ap.data(), Dp.data(), nullptr)) return shim_error();
      std::vector<double> R((size_t)nu * nu, 0.);                       // rows of Bp^-1 (unit lower triangular)
      std::vector<std::vector<std::pair<int, double>>> Crow(nu);        // rows of C = Bp^-1 Bpo as (column, value), columns ascending
      std::vector<double> acc(n_obs, 0.); std::vector<char> seen(n_obs, 0); std::vector<int> touched;
      for (int k = 0; k < nu; ++k) {
        double* Rk = R.data() + (size_t)k * nu;
        Rk[k] = 1.;
        for (int j = 0; j < mu_; ++j) {
          const int c = nnr[(size_t)k * mu_ + j];
          if (c < n_obs) continue;
          if (c >= n_obs + k) return set_error("GPB_PredictREModel: prediction point %d has neighbour %d that does not precede it", k, c - n_obs);
          const double av = Ap[(size_t)k * mu_ + j];
          const double* Rj = R.data() + (size_t)(c - n_obs) * nu;
          for (int q = 0; q <= c - n_obs; ++q) Rk[q] += av * Rj[q];       // Bp = I - A_pp  =>  row_k(Bp^-1) = e_k + sum_j A_kj row_j(Bp^-1)
        }
        touched.clear();
        for (int q = 0; q <= k; ++q) {
          const double rq = Rk[q];
          if (rq == 0.) continue;
          for (int j = 0; j < mu_; ++j) {
            const int c = nnr[(size_t)q * mu_ + j];
            if (c < 0 || c >= n_obs) continue;
            if (!seen[c]) { seen[c] = 1; touched.push_back(c); }            // every column once (the device writes one right-hand-side entry per column)
            acc[c] += rq * (-Ap[(size_t)q * mu_ + j]);                     // Bpo = -A_po
          }
        }
        std::sort(touched.begin(), touched.end());
        double m_k = 0.;
        for (int c : touched) { Crow[k].emplace_back(c, acc[c]); m_k -= acc[c] * mode[c]; acc[c] = 0.; seen[c] = 0; }
        mu_u[k] = m_k;
      }
      if (need_var || predict_cov_mat) {
        size_t mmax = 1;
        for (int k = 0; k < nu; ++k) mmax = std::max(mmax, Crow[k].size());
        std::vector<int32_t> cols((size_t)nu * mmax, -1);
        std::vector<double> vals((size_t)nu * mmax, 0.);
        for (int k = 0; k < nu; ++k) for (size_t e = 0; e < Crow[k].size(); ++e) { cols[(size_t)k * mmax + e] = Crow[k][e].first; vals[(size_t)k * mmax + e] = Crow[k][e].second; }
        std::vector<double> q(predict_cov_mat ? (size_t)nu * nu : (size_t)nu);
        if (gpb_hip_vecchia_laplace_quad_forms(mdl->vh, nu, (int)mmax, cols.data(), vals.data(), mdl->cg_max_num_it, kPredVarCgTol, predict_cov_mat ? 1 : 0, q.data(), &cg_it)) return shim_error();
        for (int r = 0; r < nu; ++r)
          for (int c2 = (predict_cov_mat ? 0 : r); c2 <= r; ++c2) {      // prior part Bp^-1 Dp Bp^-T (lower triangle; the diagonal only for variances)
            double pr = 0.;
            for (int j = 0; j <= c2; ++j) pr += R[(size_t)r * nu + j] * Dp[j] * R[(size_t)c2 * nu + j];
            if (predict_cov_mat) { cov_u[(size_t)r * nu + c2] = cov_u[(size_t)c2 * nu + r] = pr + q[(size_t)r * nu + c2]; }
            if (r == c2 && need_var) var_u[r] = pr + (predict_cov_mat ? q[(size_t)r * nu + r] : q[r]);
          }
      }
    } else if (gpb_hip_vecchia_laplace_predict(mdl->vh, nu, cu.data(), std::min(nnpl, 126), mdl->cov_type, s12, a_tr, mdl->cg_max_num_it, kPredVarCgTol, mu_u.data(),
                                        need_var ? var_u.data() : nullptr, predict_cov_mat ? cov_u.data() : nullptr, nullptr, &cg_it))
      return shim_error();
    std::vector<double> mu(npl), var(need_var ? npl : 0);
    for (int k = 0; k < npl; ++k) { mu[k] = mu_u[ui[k]]; if (need_var) var[k] = var_u[ui[k]]; }
    if (fixed_effects_pred) for (int k = 0; k < npl; ++k) mu[k] += fixed_effects_pred[k];
    if (mdl->p_cov > 0) for (int j = 0; j < mdl->p_cov; ++j) for (int k = 0; k < npl; ++k) mu[k] += covariate_data_pred[(size_t)j * npl + k] * mdl->beta[j];   // + X_pred beta (:3868-3880)
    if (predict_response) {
      if (!predict_response_host(mdl->likelihood, npl, mu.data(), var.data(), predict_var, mdl->delta_conv_mode_finding, mdl->aux_pars[0]))
        return set_error("GPB_PredictREModel: response predictions for likelihood '%s' %s", mdl->likelihood.c_str(), lscope);
    }
    std::copy(mu.begin(), mu.end(), out_predict);
    if (predict_var) std::copy(var.begin(), var.end(), out_predict + npl);
    if (predict_cov_mat)            // Zpred cov Zpred' (re_model_template.h:4306-4316), column-major like the reference's output (symmetric)
      for (int i = 0; i < npl; ++i) for (int j = 0; j < npl; ++j) out_predict[(size_t)npl + (size_t)i * npl + j] = cov_u[(size_t)ui[j] * nu + ui[i]];
    return 0;
  }
  if (mdl->likelihood == "gaussian" && mdl->eh) {
    // exact GP (gp_approx = "none"): mean = Sigma_po Psi^-1 y, covariance = Sigma_pp [+ sigma2 I] - Sigma_po Psi^-1 Sigma_op (the dense Gaussian
    // branch of REModelTemplate::Predict, re_model_template.h:4239-4330); both reductions are blocks of one Schur complement on the device
    const char* escope = "is not on the MI355X path of this library (exact GP prediction: coordinates only, no covariates / samples)";
    if (sample_posterior || sample_prior) return set_error("GPB_PredictREModel: posterior / prior samples %s", escope);
    if (predict_cov_mat && predict_var) return set_error("Calculation of both the predictive covariance matrix and variances is not supported. Choose one option (predict_cov_mat or predict_var)");
    if (cluster_ids_data_pred || re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred || covariate_data_pred || mdl->p_cov > 0)
      return set_error("GPB_PredictREModel: cluster ids / grouped effects / random coefficients / covariates for prediction %s", escope);
    const double* cpe = gp_coords_data_pred;
    int npe = num_data_pred;
    if (use_saved_data) { cpe = mdl->coords_pred.empty() ? nullptr : mdl->coords_pred.data(); npe = mdl->num_data_pred; }
    if (!cpe || npe <= 0) return set_error("GPB_PredictREModel: no coordinates for prediction (gp_coords_data_pred / GPB_SetPredictionData)");
    double tre[3];
    if (cov_pars) { double c3[3] = {cov_pars[0], cov_pars[1], cov_pars[2]}; if (transform_cov_pars(mdl, c3, tre)) return -1; }
    else {
      if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");
      std::copy(mdl->cov_pars_tr, mdl->cov_pars_tr + 3, tre);
    }
    if (!y_data && !mdl->y_set) return set_error("GPB_PredictREModel: y_data is NULL and no response has been set by an earlier call");
    const double* fee = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
    if (prediction_response(mdl, y_data, fee)) return -1;
    const bool need_cov = predict_var || predict_cov_mat;
    std::vector<double> q;
    if (need_cov) q.resize((size_t)npe * npe);
    if (gpb_hip_exact_predict(mdl->eh, mdl->cov_type, tre[1], tre[2], npe, cpe, out_predict, need_cov ? q.data() : nullptr)) return shim_error();
    if (fixed_effects_pred) for (int k = 0; k < npe; ++k) out_predict[k] += fixed_effects_pred[k];
    const double nug = predict_response ? 1. : 0.;
    if (predict_var) for (int k = 0; k < npe; ++k) out_predict[npe + k] = tre[0] * (tre[1] + nug - q[(size_t)k * npe + k]);
    if (predict_cov_mat) {
      auto kern = [&](double dist) {
        const double r = tre[2] * dist, e = tre[1] * std::exp(-r);
        return mdl->cov_type == 0 ? e : (mdl->cov_type == 1 ? e * (1. + r) : e * (1. + r + r * r / 3.));
      };
      for (int i = 0; i < npe; ++i)
        for (int j = 0; j <= i; ++j) {
          double s2 = 0.;
          for (int c = 0; c < mdl->d; ++c) { const double dd = cpe[(size_t)c * npe + i] - cpe[(size_t)c * npe + j]; s2 += dd * dd; }
          const double v = tre[0] * ((i == j ? tre[1] + nug : kern(std::sqrt(s2))) - q[(size_t)i * npe + j]);
          out_predict[npe + (size_t)i * npe + j] = v; out_predict[npe + (size_t)j * npe + i] = v;
        }
    }
    mdl->yaux_valid = false;
    return 0;
  }
  if (mdl->likelihood == "gaussian" && !mdl->eh && (mdl->vhs.size() > 1 || cluster_ids_data_pred)) {
    // Several clusters = independent realisations of the GP (cluster_ids_data of GPB_CreateREModel) and / or cluster ids for the prediction points:
    // REModelTemplate::Predict treats every prediction cluster on its own (re_model_template.h:3700-3760) -- a cluster WITH observations conditions on
    // those observations only (Case 2, :3940-4330: here the one-cluster path below on a view of that cluster's device state); a cluster WITHOUT
    // observations gets the prior (Case 1, :3750-3936: mean = fixed effects, covariance sigma1_2 k(.) [+ sigma2 for the response]); no covariance
    // between clusters.
    const char* cscope = "is not on the MI355X path of this library (prediction with cluster ids: Gaussian Vecchia model, no covariates / saved data / samples)";
    if (sample_posterior || sample_prior) return set_error("GPB_PredictREModel: posterior / prior samples %s", cscope);
    if (predict_cov_mat && predict_var) return set_error("Calculation of both the predictive covariance matrix and variances is not supported. Choose one option (predict_cov_mat or predict_var)");
    if (re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred || covariate_data_pred || mdl->p_cov > 0 || use_saved_data)
      return set_error("GPB_PredictREModel: grouped effects / random coefficients / covariates / saved prediction data %s", cscope);
    if (!cluster_ids_data_pred) return set_error("Missing cluster_id data ('cluster_ids_pred') for making predictions");   // :3522-3524
    const int np = num_data_pred;
    if (!gp_coords_data_pred || np <= 0) return set_error("GPB_PredictREModel: no coordinates for prediction (gp_coords_data_pred)");
    double c3[3], tr[3];
    if (cov_pars) std::copy(cov_pars, cov_pars + 3, c3);
    else {
      if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");
      transform_back(mdl, mdl->cov_pars_tr, c3);
    }
    if (transform_cov_pars(mdl, c3, tr)) return -1;
    if (!y_data && !mdl->y_set) return set_error("GPB_PredictREModel: y_data is NULL and no response has been set by an earlier call");
    const double* fe = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
    if (prediction_response(mdl, y_data, fe)) return -1;   // as in the one-cluster path: the residual becomes the response
    mdl->yaux_valid = false;
    // prediction points by cluster, clusters in the order of their first appearance
    std::vector<int32_t> pid; std::vector<std::vector<int>> pidx;
    for (int i = 0; i < np; ++i) {
      size_t k = 0;
      while (k < pid.size() && pid[k] != cluster_ids_data_pred[i]) ++k;
      if (k == pid.size()) { pid.push_back(cluster_ids_data_pred[i]); pidx.emplace_back(); }
      pidx[k].push_back(i);
    }
    const int d = mdl->d;
    if (predict_cov_mat) std::fill(out_predict + np, out_predict + np + (size_t)np * np, 0.);
    for (size_t pc = 0; pc < pid.size(); ++pc) {
      const std::vector<int>& ix = pidx[pc];
      const int npc = (int)ix.size();
      int ci = -1;
      if (mdl->cluster_id_values.empty()) { if (pid[pc] == 0) ci = 0; }
      else for (size_t k = 0; k < mdl->cluster_id_values.size(); ++k) if (mdl->cluster_id_values[k] == pid[pc]) { ci = (int)k; break; }
      std::vector<double> cc((size_t)npc * d), fep;
      for (int j = 0; j < d; ++j) for (int k = 0; k < npc; ++k) cc[(size_t)j * npc + k] = gp_coords_data_pred[(size_t)j * np + ix[k]];
      if (fixed_effects_pred) { fep.resize(npc); for (int k = 0; k < npc; ++k) fep[k] = fixed_effects_pred[ix[k]]; }
      std::vector<double> outc((size_t)npc * (predict_cov_mat ? 1 + (size_t)npc : (predict_var ? 2 : 1)), 0.);
      if (ci < 0) {
        // no observations for this cluster: the prior.  The reference builds a Vecchia approximation of the prior among these points
        // (:3760-3838), which is exact for up to num_neighbors + 1 of them; beyond that only the variances are on this path.
        for (int k = 0; k < npc; ++k) outc[k] = fixed_effects_pred ? fep[k] : 0.;
        const double vdiag = tr[0] * (tr[1] + (predict_response ? 1. : 0.));
        if (predict_var) for (int k = 0; k < npc; ++k) outc[npc + k] = vdiag;
        if (predict_cov_mat) {
          if (npc > mdl->num_neighbors + 1) return set_error("GPB_PredictREModel: covariance matrix of %d prediction points in a cluster without observations (more than num_neighbors + 1 = %d) %s", npc, mdl->num_neighbors + 1, cscope);
          for (int a1 = 0; a1 < npc; ++a1) for (int b1 = 0; b1 < npc; ++b1) {
            double s2 = 0.;
            for (int j = 0; j < d; ++j) { const double t = cc[(size_t)j * npc + a1] - cc[(size_t)j * npc + b1]; s2 += t * t; }
            const double r = tr[2] * std::sqrt(s2), e = std::exp(-r);
            const double kv = mdl->cov_type == 0 ? e : (mdl->cov_type == 1 ? e * (1. + r) : e * (1. + r + r * r / 3.));
            outc[npc + (size_t)a1 * npc + b1] = a1 == b1 ? vdiag : tr[0] * tr[1] * kv;
          }
        }
      } else {
        // a view of cluster ci as a one-cluster model: shares the device state and the resident response, owns nothing
        REModelHip view;
        const auto disown = scope_exit_api([&] { view.vhs.clear(); view.vh = nullptr; view.ybuf = nullptr; view.ybuf_cap = 0; });
        const int o0 = mdl->cl_off[ci], nc = mdl->cl_off[ci + 1] - o0;
        view.n = nc; view.d = d; view.m = mdl->m; view.num_neighbors = mdl->num_neighbors; view.cov_type = mdl->cov_type;
        view.perm.resize(nc); std::iota(view.perm.begin(), view.perm.end(), 0);
        view.vh = mdl->vhs[ci]; view.vhs.assign(1, view.vh); view.cl_off = {0, nc};
        view.has_weights = mdl->has_weights;
        if (mdl->has_weights) view.nug_v.assign(mdl->nug_v.begin() + o0, mdl->nug_v.begin() + o0 + nc);
        view.ybuf = mdl->ybuf + o0; view.ybuf_cap = (size_t)nc; view.y_set = true;
        view.vecchia_pred_type = mdl->vecchia_pred_type; view.num_neighbors_pred = mdl->num_neighbors_pred;
        if (GPB_PredictREModel(&view, nullptr, npc, outc.data(), predict_cov_mat, predict_var, predict_response, false, false, 0, 0, nullptr, nullptr, nullptr,
                               cc.data(), nullptr, c3, nullptr, false, nullptr, fixed_effects_pred ? fep.data() : nullptr)) return -1;
      }
      for (int k = 0; k < npc; ++k) out_predict[ix[k]] = outc[k];
      if (predict_var) for (int k = 0; k < npc; ++k) out_predict[np + ix[k]] = outc[npc + k];
      if (predict_cov_mat)
        for (int a1 = 0; a1 < npc; ++a1) for (int b1 = 0; b1 < npc; ++b1) out_predict[np + (size_t)ix[a1] * np + ix[b1]] = outc[npc + (size_t)a1 * npc + b1];
    }
    return 0;
  }
  if (mdl->likelihood != "gaussian" || mdl->eh || mdl->vhs.size() != 1) return set_error("GPB_PredictREModel: this model %s", scope);
  if (sample_posterior || sample_prior) return set_error("GPB_PredictREModel: posterior / prior samples %s", scope);
  if (predict_cov_mat && predict_var) return set_error("Calculation of both the predictive covariance matrix and variances is not supported. Choose one option (predict_cov_mat or predict_var)");   // re_model.cpp Predict
  if (cluster_ids_data_pred || re_group_data_pred || re_group_rand_coef_data_pred || gp_rand_coef_data_pred)
    return set_error("GPB_PredictREModel: cluster ids / grouped effects / random coefficients for prediction %s", scope);
  if (mdl->p_cov > 0 && !mdl->coef_estimated) return set_error("GPB_PredictREModel: the model has covariates but no estimated coefficients");
  if (mdl->p_cov > 0 && !covariate_data_pred && !use_saved_data) return set_error("No covariate data is provided in 'covariate_data_pred' but the model has linear regression covariates");   // re_model.cpp Predict
  if (mdl->p_cov == 0 && covariate_data_pred) return set_error("Covariate data is provided in 'covariate_data_pred' but the model has no linear regression covariates");
  if (mdl->p_cov > 0 && use_saved_data) return set_error("GPB_PredictREModel: saved prediction data together with covariates %s", scope);
  const bool cond_all = mdl->vecchia_pred_type == "order_obs_first_cond_all";
  const bool pred_first = mdl->vecchia_pred_type == "order_pred_first";
  const bool latent_all = mdl->vecchia_pred_type == "latent_order_obs_first_cond_all";
  const bool latent = latent_all || mdl->vecchia_pred_type == "latent_order_obs_first_cond_obs_only";
  if (mdl->vecchia_pred_type != "order_obs_first_cond_obs_only" && !cond_all && !pred_first && !latent)
    return set_error("GPB_PredictREModel: vecchia_pred_type '%s' %s", mdl->vecchia_pred_type.c_str(), scope);
  const double* cp = gp_coords_data_pred;
  int np = num_data_pred;
  if (use_saved_data) { cp = mdl->coords_pred.empty() ? nullptr : mdl->coords_pred.data(); np = mdl->num_data_pred; }
  if (!cp || np <= 0) return set_error("GPB_PredictREModel: no coordinates for prediction (gp_coords_data_pred / GPB_SetPredictionData)");
  double tr[3];
  if (cov_pars) { double c3[3] = {cov_pars[0], cov_pars[1], cov_pars[2]}; if (transform_cov_pars(mdl, c3, tr)) return -1; }
  else {
    if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");   // re_model.cpp:1119-1121
    std::copy(mdl->cov_pars_tr, mdl->cov_pars_tr + 3, tr);
  }
  // response and offset (SetYCalcCovCalcYAuxForPred, re_model_template.h:11141-11166): with an offset -- the argument, else the one saved by
  // GPB_SetOffsetData (:3601-3607) -- the residual (y_obs or the stored response) minus the offset becomes the response
  if (!y_data && !mdl->y_set) return set_error("GPB_PredictREModel: y_data is NULL and no response has been set by an earlier call");
  const double* fe = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
  if (prediction_response(mdl, y_data, fe)) return -1;   // y_obs, else y_vec_ = the response as it was passed in (NOT y - offset of the fit)
  if (mdl->p_cov > 0 && gpb_hip_vecchia_set_resid(mdl->vh, mdl->beta.data())) return shim_error();   // resid -= X beta (:11150-11152); the host copy for 'cond_all' below
  std::vector<double> resid_v;                          // response the prediction conditions on, Vecchia order
  const double* yv = mdl->ybuf;
  if (mdl->p_cov > 0) {
    resid_v.assign(mdl->ybuf, mdl->ybuf + mdl->n);
    for (int j = 0; j < mdl->p_cov; ++j) for (int k = 0; k < mdl->n; ++k) resid_v[k] -= mdl->X[(size_t)j * mdl->n + mdl->perm[k]] * mdl->beta[j];
    yv = resid_v.data();
  }
  auto add_linear_predictor = [&](double* mean) {       // mu += X_pred beta (:3868-3880) and the external fixed effects (:3862-3867)
    if (mdl->p_cov > 0) for (int j = 0; j < mdl->p_cov; ++j) for (int k = 0; k < np; ++k) mean[k] += covariate_data_pred[(size_t)j * np + k] * mdl->beta[j];
    if (fixed_effects_pred) for (int k = 0; k < np; ++k) mean[k] += fixed_effects_pred[k];
  };
  mdl->yaux_valid = false;
  // num_neighbors_pred: default 2 * num_neighbors (:299), at most the number of observed points (Vecchia_utils.cpp:755-758) and the
  // device kernels' GPB_MAX_NEIGHBORS
  int nnp = mdl->num_neighbors_pred > 0 ? mdl->num_neighbors_pred : 2 * mdl->num_neighbors;
  if (nnp > mdl->n) nnp = mdl->n;
  if (nnp > 126) {
    log_info("[GPBoost-AMD] [Warning] num_neighbors_pred = %d exceeds the %d neighbours the MI355X kernels support; %d are used\n", nnp, 126, 126);
    nnp = 126;
  }
  if (cond_all) {
    // 'order_obs_first_cond_all': search + factor of the appended rows on the device, then the forward substitution with Bp and the rows
    // of Bp^-1 on the host (Vecchia_utils.cpp:2061-2090); y of the observed points in Vecchia order is the resident response
    if (nnp > mdl->n + np - 1) nnp = mdl->n + np - 1;
    if (nnp > 126) nnp = 126;
    int mu = 0;
    std::vector<int32_t> nnp_rows((size_t)np * nnp);
    std::vector<double> Ap((size_t)np * nnp), Dp(np);
    if (gpb_hip_vecchia_predict_cond_all(mdl->vh, np, cp, nnp, mdl->cov_type, tr[1], tr[2], &mu, nnp_rows.data(), Ap.data(), Dp.data(), nullptr)) return shim_error();
    if (GPB_HIP_PredictCondAllHost(mdl->n, np, mu, nnp_rows.data(), Ap.data(), Dp.data(), yv, tr[0], predict_response, out_predict,
                                   predict_var ? out_predict + np : nullptr, predict_cov_mat ? out_predict + np : nullptr)) return -1;
    add_linear_predictor(out_predict);
    return 0;
  }
  if (pred_first || latent) {
    // 'order_pred_first' (CalcPredVecchiaPredictedFirstOrder, Vecchia_utils.cpp:2203-2444) and 'latent_order_obs_first_cond_*'
    // (CalcPredVecchiaLatentObservedFirstOrder, :2446-2666): neighbour search + factor of EVERY point of the joint ordering on the device; the
    // conditional precision is assembled here from the factor rows and solved / inverted by the dense Cholesky on the device.
    //   order_pred_first: cond_prec = Bp' Dp^-1 Bp + Bop' Do^-1 Bop over the prediction points, mean = -cond_prec^-1 Bop' Do^-1 Bo y (:2419-2423)
    //   latent_*: the reference forms Sigma = B^-1 D B^-T and conditions on y = b_obs + eps by the Woodbury identity (:2597-2650); the same
    //     posterior is M^-1 Z_o' R^-1 y and M^-1 restricted to the prediction points with M = B' D^-1 B + Z_o' R^-1 Z_o, which needs no B^-1
    const int n = mdl->n, n_all = n + np;
    const char* dense_scope = "is not on the MI355X path of this library (the conditional precision is handled densely; use 'order_obs_first_cond_obs_only' / 'order_obs_first_cond_all')";
    if (pred_first && np > 24000) return set_error("GPB_PredictREModel: 'order_pred_first' for %d > 24000 prediction points %s", np, dense_scope);
    if (latent && n_all > 24000) return set_error("GPB_PredictREModel: '%s' for %d > 24000 observed + prediction points %s", mdl->vecchia_pred_type.c_str(), n_all, dense_scope);
    const int mcap = (pred_first || latent_all) ? n_all - 1 : n;
    if (nnp > mcap) nnp = mcap;
    std::vector<int32_t> nn((size_t)n_all * nnp);
    std::vector<double> A((size_t)n_all * nnp), D(n_all), u(n_all);
    int mu = 0, dup = 0;
    if (gpb_hip_vecchia_predict_joint_factor(mdl->vh, np, cp, nnp, pred_first ? 1 : 0, (pred_first || latent_all) ? 1 : 0, latent ? 0 : 1, mdl->cov_type,
                                             tr[1], tr[2], &mu, nn.data(), A.data(), D.data(), u.data(), &dup)) return shim_error();
    if (latent && dup) return set_error("Duplicates found among training and test coordinates. This is not supported for predictions with a Vecchia approximation for the latent process ('latent_') ");   // :2563-2566
    const bool need_cov = predict_var || predict_cov_mat;
    const int q = pred_first ? np : n_all;                  // dimension of the precision matrix
    const int col_end = pred_first ? np : n_all;            // columns of B that enter it
    std::vector<double> M((size_t)q * q, 0.), rhs(q, 0.), x(q), inv;
    std::vector<int> ec; std::vector<double> ev;
    ec.reserve(mu + 1); ev.reserve(mu + 1);
    for (int i = 0; i < n_all; ++i) {
      ec.clear(); ev.clear();
      if (i < col_end) { ec.push_back(i); ev.push_back(1.); }
      for (int j = 0; j < mu; ++j) {
        const int c = nn[(size_t)i * mu + j];
        if (c >= 0 && c < col_end) { ec.push_back(c); ev.push_back(-A[(size_t)i * mu + j]); }
      }
      const double dinv = 1. / D[i];
      for (size_t a1 = 0; a1 < ec.size(); ++a1) {
        const double va = ev[a1] * dinv;
        for (size_t b1 = 0; b1 < ec.size(); ++b1) if (ec[b1] <= ec[a1]) M[(size_t)ec[a1] * q + ec[b1]] += va * ev[b1];      // lower triangle
        if (pred_first && i >= np) rhs[ec[a1]] -= va * u[i];                                                                 // -Bop' Do^-1 (Bo y)
      }
    }
    if (latent) for (int k = 0; k < n; ++k) {               // + Z_o' R^-1 Z_o and the right-hand side Z_o' R^-1 y, R^-1 = diag(w) (:2502-2506)
      const double w = mdl->has_weights ? 1. / mdl->nug_v[k] : 1.;
      M[(size_t)k * q + k] += w;
      rhs[k] = w * yv[k];
    }
    const int sub0 = pred_first ? 0 : n;
    if (need_cov) inv.resize((size_t)np * np);
    if (gpb_hip_dense_spd_solve(q, M.data(), rhs.data(), x.data(), sub0, need_cov ? inv.data() : nullptr)) return shim_error();
    for (int k = 0; k < np; ++k) out_predict[k] = x[sub0 + k];
    add_linear_predictor(out_predict);
    // latent types: the error variance is ADDED for the response (:2624-2626, 2645-2647); 'order_pred_first' has it in the factor and it is
    // REMOVED for the latent process (re_model_template.h:4134-4150)
    const double dshift = latent ? (predict_response ? 1. : 0.) : (predict_response ? 0. : -1.);
    if (predict_var) for (int k = 0; k < np; ++k) out_predict[np + k] = tr[0] * (inv[(size_t)k * np + k] + dshift);
    if (predict_cov_mat) {
      for (size_t k = 0; k < (size_t)np * np; ++k) out_predict[np + k] = tr[0] * inv[k];
      for (int k = 0; k < np; ++k) out_predict[np + (size_t)k * np + k] += tr[0] * dshift;
    }
    return 0;
  }
  std::vector<double> D(np);
  if (gpb_hip_vecchia_predict_obs_only(mdl->vh, np, cp, nnp, mdl->cov_type, tr[1], tr[2], out_predict, D.data(), nullptr)) return shim_error();
  add_linear_predictor(out_predict);
  if (predict_var)
    for (int k = 0; k < np; ++k) out_predict[np + k] = tr[0] * (predict_response ? D[k] : D[k] - 1.);
  if (predict_cov_mat) {                       // neighbours are observed points only: Bp = I, the predictive covariance is diag(Dp)
    std::fill(out_predict + np, out_predict + np + (size_t)np * np, 0.);
    for (int k = 0; k < np; ++k) out_predict[np + (size_t)k * np + k] = tr[0] * (predict_response ? D[k] : D[k] - 1.);
  }
  C_API_END();
}

int GPB_GetCurrentNegLogLikelihood(REModelHandle handle, double* negll) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !negll) return set_error("GPB_GetCurrentNegLogLikelihood: null argument");
  if (!mdl->negll_valid) return set_error("The negative log-likelihood has not been evaluated yet");
  *negll = mdl->cur_negll;
  C_API_END();
}

int GPB_GetLikelihoodName(REModelHandle handle, char* out_str, int* num_char) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !out_str || !num_char) return set_error("GPB_GetLikelihoodName: null argument");
  *num_char = (int)mdl->likelihood.size() + 1;   // c_api.cpp: size + 1, then memcpy incl. terminator
  std::memcpy(out_str, mdl->likelihood.c_str(), mdl->likelihood.size() + 1);
  C_API_END();
}

/* ---- the rest of the GPB_* surface the reference's GPModel binds (python-package/gpboost/basic.py:5206-7118): state getters / setters
   are answered from the model's host state; estimation with covariates and auxiliary parameters are off the hot path -> -1 + message ---- */

static int copy_string_out(const std::string& v, char* out_str, int* num_char, const char* who) {
  if (!out_str || !num_char) return set_error("%s: null argument", who);
  *num_char = (int)v.size() + 1;                     // c_api.cpp:3025-3034: size + 1, memcpy incl. the terminator
  std::memcpy(out_str, v.c_str(), v.size() + 1);
  return 0;
}

int GPB_OptimLinRegrCoefCovPar(REModelHandle handle, const double* y_data, const double* covariate_data, int num_covariates,
                               const double* fixed_effects) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_OptimLinRegrCoefCovPar: null handle");
  if (num_covariates <= 0 || !covariate_data) return GPB_OptimCovPar(handle, y_data, fixed_effects);   // (forgets the covariates of an earlier fit)
  C_API_BEGIN();
  const char* scope = "is not on the MI355X path of this library (covariates: one-cluster Gaussian Vecchia model, optimizer_cov 'lbfgs' or 'gradient_descent', coefficients by 'wls')";
  if (mdl->likelihood != "gaussian" && !mdl->eh && mdl->vhs.size() == 1) {
    // (full-scale Vecchia models too -- round 6: the boosting / coefficient gradient d(-mll)/dF is the same expression in the by-products of the VIF gradient, likelihoods.h:5598-5604)
    // non-Gaussian model with a linear predictor: the coefficients are part of the lbfgs vector (the reference's default for these models,
    // optim_utils.h:283-420), covariates scaled, the linear predictor enters the device as fixed effects, its gradient is X' grad_F
    const char* lscope = "is not on the MI355X path of this library (non-Gaussian models with covariates: optimizer_cov 'lbfgs' with the coefficients in its vector)";
    if (mdl->optimizer_unsupported_alias || (mdl->optim.optimizer != "" && mdl->optim.optimizer != "lbfgs")) return set_error("GPB_OptimLinRegrCoefCovPar: optimizer_cov '%s' with covariates %s", mdl->optim.optimizer.c_str(), lscope);
    if (mdl->optimizer_coef != "" && mdl->optimizer_coef != "lbfgs") return set_error("GPB_OptimLinRegrCoefCovPar: optimizer_coef '%s' %s", mdl->optimizer_coef.c_str(), lscope);
    if (num_covariates > 256) return set_error("GPB_OptimLinRegrCoefCovPar: %d covariates %s", num_covariates, lscope);
    const double* wdat = mdl->lik_weights.empty() ? nullptr : mdl->lik_weights.data();      // round 6: sample weights together with covariates (data order)
    if (is_proportion_likelihood(mdl->likelihood)) return set_error("GPB_OptimLinRegrCoefCovPar: covariates for likelihood '%s' %s", mdl->likelihood.c_str(), lscope);
    if (!y_data) return set_error("GPB_OptimLinRegrCoefCovPar: y_data is NULL");
    if (!mdl->init_coef.empty() && (int)mdl->init_coef.size() != num_covariates) return set_error("GPB_OptimLinRegrCoefCovPar: %d initial coefficients for %d covariates", (int)mdl->init_coef.size(), num_covariates);
    const int n = mdl->n, p = num_covariates;
    if (fixed_effects) { mdl->offset.assign(fixed_effects, fixed_effects + n); mdl->has_offset = true; }      // kept for later predictions (re_model_template.h:1185-1188)
    resolve_optimizer_defaults_once(mdl);
    if (initialize_cov_pars_if_not_defined(mdl, y_data, fixed_effects)) return -1;
    std::vector<double> init_coef = mdl->init_coef;
    if (init_coef.empty() && mdl->init_coef_from_iid_model) {       // re_model.cpp:556-569
      GpbOptimConfig cfg0 = mdl->optim;
      if (iid_model_init_coef(mdl->likelihood, n, p, covariate_data, y_data, fixed_effects, cfg0, &init_coef, nullptr, wdat)) return -1;
    }
    LaplaceCoefSetup su;
    if (laplace_coef_setup(mdl->likelihood, n, p, covariate_data, y_data, fixed_effects, mdl->cov_pars_tr[0], init_coef.empty() ? nullptr : init_coef.data(), &su, wdat)) return -1;
    const double* offs = fixed_effects;      // the fit sees the argument only (re_model_template.h:1184, fixed_effects_ptr = fixed_effects); a stored offset serves prediction
    if (laplace_upload_data(mdl, y_data, offs)) return -1;
    mdl->lap_fit_first_eval = true;
    GpbOptimConfig cfg = mdl->optim;
    if (cfg.optimizer.empty()) cfg.optimizer = "lbfgs";
    cfg.range_const = range_const(mdl);
    char err[512] = "";
    GpbLaplaceCoefResult res;
    std::vector<double> beta = su.beta;
    if (gpb_optimize_laplace_coef_cov_pars(cfg, device_laplace_fe, mdl, n, p, su.Xs.data(), offs, su.C_mu, su.C_sigma2, mdl->cov_pars_tr, beta.data(), &res, err, (int)sizeof(err))) {
      const char* why = gpb_hip_get_last_error();
      if (err[0] && why && why[0]) return set_error("%s: %s", err, why);
      return err[0] ? set_error("%s", err) : shim_error();
    }
    transform_back_coef(su, beta);
    mdl->p_cov = p;
    mdl->X.assign(covariate_data, covariate_data + (size_t)n * p);
    mdl->beta = beta;
    mdl->chol_XtPsiInvX.clear();
    mdl->coef_estimated = true;
    if (cfg.max_iter > 0) {
      mdl->cov_pars_tr[0] = res.theta[0]; mdl->cov_pars_tr[1] = res.theta[1];
      mdl->cur_negll = res.negll;
      mdl->negll_valid = true;
    }
    mdl->num_it = res.num_it;
    mdl->model_has_been_estimated = true;
    // leave the device with the fitted linear predictor as its fixed effects (prediction and standard errors start from this state)
    std::vector<double> fe_fit(n);
    for (int i = 0; i < n; ++i) { double v = offs ? offs[i] : 0.; for (int j = 0; j < p; ++j) v += covariate_data[(size_t)j * n + i] * beta[j]; fe_fit[i] = v; }
    if (laplace_upload_fixed_effects(mdl, fe_fit.data())) return -1;
    return 0;
  }
  if (mdl->likelihood != "gaussian" || mdl->eh || mdl->vhs.size() != 1 || mdl->vif) return set_error("GPB_OptimLinRegrCoefCovPar: this model %s", scope);
  const bool by_iteration = mdl->optim.optimizer == "gradient_descent";
  if (mdl->optim.optimizer != "" && mdl->optim.optimizer != "lbfgs" && !by_iteration)      // ('nelder_mead' would search over the coefficients too, optim_utils.h:607-609)
    return set_error("GPB_OptimLinRegrCoefCovPar: optimizer_cov '%s' with covariates %s", mdl->optim.optimizer.c_str(), scope);
  if (by_iteration && mdl->optim.convergence_criterion != "relative_change_in_log_likelihood")
    return set_error("GPB_OptimLinRegrCoefCovPar: optimizer_cov 'gradient_descent' with covariates and convergence_criterion '%s' %s", mdl->optim.convergence_criterion.c_str(), scope);
  if (mdl->optimizer_coef != "" && mdl->optimizer_coef != "wls") return set_error("GPB_OptimLinRegrCoefCovPar: optimizer_coef '%s' %s", mdl->optimizer_coef.c_str(), scope);
  if (num_covariates > 256) return set_error("GPB_OptimLinRegrCoefCovPar: %d covariates %s", num_covariates, scope);
  if (!y_data) return set_error("GPB_OptimLinRegrCoefCovPar: y_data is NULL");
  const int n = mdl->n, p = num_covariates;
  mdl->p_cov = p;
  mdl->X.assign(covariate_data, covariate_data + (size_t)n * p);             // column-major n x p, data order (X_)
  std::vector<double> Xv((size_t)n * p);                                      // Vecchia order for the device
  for (int j = 0; j < p; ++j) for (int k = 0; k < n; ++k) Xv[(size_t)j * n + k] = covariate_data[(size_t)j * n + mdl->perm[k]];
  if (gpb_hip_vecchia_set_covariates(mdl->vh, p, Xv.data())) return shim_error();
  mdl->beta.assign(p, 0.);
  if (by_iteration) {
    // initial coefficients (re_model_template.h:1245-1278): init_coef, else zero with the intercept at the mean of y - fixed effects
    // (Likelihood::FindInitialIntercept, "gaussian" branch)
    if ((int)mdl->init_coef.size() == p) mdl->beta = mdl->init_coef;
    else {
      for (int j = 0; j < p; ++j) {
        const double* col = covariate_data + (size_t)j * n;
        bool constant = true;
        for (int i = 1; i < n && constant; ++i)
          constant = std::fabs(col[i] - col[0]) < 1e-10 * std::max({1.0, std::fabs(col[i]), std::fabs(col[0])});      // TwoNumbersAreEqual (utils.h:54-56)
        if (constant) {
          double avg = 0.;
          for (int i = 0; i < n; ++i) avg += y_data[i] - (fixed_effects ? fixed_effects[i] : 0.);
          mdl->beta[j] = avg / n;
          break;
        }
      }
    }
  }
  mdl->fitting_with_covariates = true;
  mdl->coef_by_iteration = by_iteration;
  const int rc = GPB_OptimCovPar(handle, y_data, fixed_effects);             // y0 = y - fixed_effects is uploaded there; every evaluation profiles beta out (device_terms)
  mdl->fitting_with_covariates = false;
  mdl->coef_by_iteration = false;
  if (rc != 0) { mdl->p_cov = 0; (void)gpb_hip_vecchia_set_covariates(mdl->vh, 0, nullptr); return rc; }
  // The coefficients are those of the optimiser's LAST likelihood evaluation (OptimExternal does not evaluate again after lbfgs, optim_utils.h:681-688):
  // its covariance parameters are the final ones except when a parameter is held fixed on the original scale -- then the ratio of that evaluation was
  // formed with the error variance of the evaluation before (MaybeKeepVarianceConstant).  The factor of X' Psi^-1 X for the standard deviations
  // (CalcStdDevCoef) and the residual on the device are those of the final covariance parameters.
  const std::vector<double> beta_last = mdl->beta;
  if (profile_out_coef(mdl, mdl->cov_pars_tr[1], mdl->cov_pars_tr[2])) return -1;
  if ((int)beta_last.size() == p) {
    mdl->beta = beta_last;
    if (gpb_hip_vecchia_set_resid(mdl->vh, mdl->beta.data())) return shim_error();
  }
  mdl->coef_estimated = true;
  C_API_END();
}

int GPB_CanCalculateStandardErrorsCovPars(REModelHandle handle, int* out) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !out) return set_error("GPB_CanCalculateStandardErrorsCovPars: null argument");
  out[0] = can_calc_std_dev(mdl) ? 1 : 0;
  C_API_END();
}

int GPB_CanCalculateStandardErrorsAuxPars(REModelHandle handle, int* out) {
  C_API_BEGIN();
  if (!handle || !out) return set_error("GPB_CanCalculateStandardErrorsAuxPars: null argument");
  out[0] = 0;                                        // none of the supported likelihoods has auxiliary parameters
  C_API_END();
}

int GPB_GetCoef(REModelHandle handle, double* optim_coef, bool calc_std_dev) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !optim_coef) return set_error("GPB_GetCoef: null argument");
  if (mdl->p_cov < 1 || !mdl->coef_estimated) return set_error("Linear regression coefficients have not been estimated (the model has no covariates or has not been fitted with them)");
  std::copy(mdl->beta.begin(), mdl->beta.end(), optim_coef);
  if (calc_std_dev && mdl->likelihood != "gaussian") {
    // CalcStdDevCoefNonGaussian (re_model_template.h:10851-10897): numerical Jacobian of X' grad_F, 2 p evaluations on the device
    if (!mdl->y_set) return set_error("GPB_GetCoef: standard deviations need the response of the fit (none has been set)");
    const double th[2] = {mdl->cov_pars_tr[0], mdl->cov_pars_tr[1]};
    const double* offs = mdl->has_offset ? mdl->offset.data() : nullptr;
    char err[512] = "";
    if (gpb_laplace_coef_std_errors(device_laplace_fe, mdl, mdl->n, mdl->p_cov, mdl->X.data(), offs, th, mdl->beta.data(), optim_coef + mdl->p_cov, err, (int)sizeof(err))) {
      const char* why = gpb_hip_get_last_error();
      return (why && why[0]) ? set_error("%s: %s", err[0] ? err : "GPB_GetCoef", why) : set_error("%s", err[0] ? err : "evaluation failed");
    }
    return 0;
  }
  if (calc_std_dev) {
    // CalcStdDevCoef (re_model_template.h:10823-10841): sqrt(diag((X' Psi^-1 X / sigma2)^-1)); the factor is that of the final GLS step
    const int p = mdl->p_cov;
    if (p >= mdl->n) { for (int j = 0; j < p; ++j) optim_coef[p + j] = std::numeric_limits<double>::quiet_NaN(); return 0; }
    const std::vector<double>& L = mdl->chol_XtPsiInvX;
    std::vector<double> col(p);
    for (int j = 0; j < p; ++j) {          // (L L')^-1_jj = || L^-1 e_j ||^2
      double ss = 0.;
      for (int i = j; i < p; ++i) {
        double v = (i == j) ? 1. : 0.;
        for (int k = j; k < i; ++k) v -= L[(size_t)i * p + k] * col[k];
        col[i] = v / L[(size_t)i * p + i];
        ss += col[i] * col[i];
      }
      optim_coef[p + j] = std::sqrt(mdl->cov_pars_tr[0] * ss);
    }
  }
  return 0;
}

int GPB_HasStdCylBesselK(int* has_bessel) {
  if (!has_bessel) return set_error("GPB_HasStdCylBesselK: null argument");
  has_bessel[0] = 0;                                 // general-shape Matern (Bessel K) is not on the hot path: shapes 0.5 / 1.5 / 2.5 only
  return 0;
}

int GPB_GetOptimizerCovPars(REModelHandle handle, char* out_str, int* num_char) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_GetOptimizerCovPars: null handle");
  return copy_string_out(mdl->optim.optimizer.empty() ? std::string("lbfgs") : mdl->optim.optimizer, out_str, num_char, "GPB_GetOptimizerCovPars");   // re_model_template.h:8277-8280
  C_API_END();
}

int GPB_GetOptimizerCoef(REModelHandle handle, char* out_str, int* num_char) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_GetOptimizerCoef: null handle");
  if (!mdl->optimizer_coef.empty()) return copy_string_out(mdl->optimizer_coef, out_str, num_char, "GPB_GetOptimizerCoef");
  return copy_string_out(mdl->likelihood == "gaussian" ? "wls" : "lbfgs", out_str, num_char, "GPB_GetOptimizerCoef");   // defaults, :8281-8288
  C_API_END();
}

int GPB_GetCGPreconditionerType(REModelHandle handle, char* out_str, int* num_char) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_GetCGPreconditionerType: null handle");
  return copy_string_out(mdl->cg_preconditioner_type, out_str, num_char, "GPB_GetCGPreconditionerType");
  C_API_END();
}

int GPB_GetNumCGSteps(REModelHandle handle, int* num_cg_steps) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !num_cg_steps) return set_error("GPB_GetNumCGSteps: null argument");
  num_cg_steps[0] = mdl->likelihood == "gaussian" ? 0 : (int)mdl->lap_info[2];
  C_API_END();
}

int GPB_GetNumCGStepsTridiag(REModelHandle handle, int* num_cg_steps) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !num_cg_steps) return set_error("GPB_GetNumCGStepsTridiag: null argument");
  num_cg_steps[0] = mdl->likelihood == "gaussian" ? 0 : (int)mdl->lap_info[4];
  C_API_END();
}

int GPB_GetNumModeFindingSteps(REModelHandle handle, int* num_steps) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !num_steps) return set_error("GPB_GetNumModeFindingSteps: null argument");
  num_steps[0] = mdl->likelihood == "gaussian" ? 0 : (int)mdl->lap_info[1];
  C_API_END();
}

int GPB_SetLikelihood(REModelHandle handle, const char* likelihood) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !likelihood) return set_error("GPB_SetLikelihood: null argument");
  if (mdl && mdl->vif) return set_error("GPB_SetLikelihood: gp_approx 'full_scale_vecchia' -- likelihood evaluation, its gradient and fits are on the MI355X path of this library, this call is not yet");
  bool fix_df = false;
  const std::string lik = parse_likelihood_alias(strip_fix_df(likelihood ? likelihood : "", &fix_df));
  if (mdl->model_has_been_estimated && lik != mdl->likelihood) return set_error("Cannot change likelihood after a model has been estimated ");   // re_model.cpp:154-160
  if (lik == mdl->likelihood) return 0;
  if (lik != "gaussian" && !supported_non_gaussian(lik))
    return set_error("GPB_SetLikelihood: likelihood '%s' is not on the MI355X hot path of this library", likelihood);
  if (lik != "gaussian" && (mdl->eh || mdl->vhs.size() != 1)) return set_error("GPB_SetLikelihood: likelihood '%s' needs gp_approx 'vecchia' and one cluster on this path", likelihood);
  if (lik != "gaussian" && mdl->has_duplicates) return set_error(kDuplicatesNonGaussianMessage);
  if (lik == "gaussian" && mdl->n_re > 0)
    return set_error("GPB_SetLikelihood: this model was created with repeated locations under a non-Gaussian likelihood -- its Vecchia approximation lives on the %d unique locations (Vecchia_utils.cpp:1156-1168); create a new model for the Gaussian likelihood", mdl->n_re);
  mdl->likelihood = lik; mdl->estimate_df_t = !fix_df;
  mdl->num_aux = num_aux_of(lik); mdl->aux_pars[0] = lik == "lognormal" ? 0.5 : 1.; mdl->aux_pars[1] = 2.; mdl->aux_set = false; mdl->init_aux_given = false;      // a new Likelihood object (re_model_template.h SetLikelihood)
  mdl->cov_pars_initialized = false; mdl->init_cov_pars_provided = false; mdl->negll_valid = false; mdl->y_set = false; mdl->yaux_valid = false;
  C_API_END();
}

int GPB_GetResponseData(REModelHandle handle, double* response_data) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !response_data) return set_error("GPB_GetResponseData: null argument");
  if (!mdl->y_set) return set_error("Respone variable data has not been set");      // re_model_template.h:6258-6261 (sic)
  if (mdl->likelihood == "gaussian") { std::copy(mdl->y_host.begin(), mdl->y_host.end(), response_data); }   // y_vec_: the response as passed in
  else if (!mdl->resp_real.empty()) { for (int k = 0; k < mdl->n; ++k) response_data[mdl->perm[k]] = mdl->resp_real[k]; }      // real-valued responses: gamma / beta / t / lognormal and the proportions of binomial_* / quasi_bernoulli_* (laplace_upload_data)
  else { for (int k = 0; k < mdl->n; ++k) response_data[mdl->perm[k]] = (double)mdl->labels[k]; }
  C_API_END();
}

int GPB_GetCovariateData(REModelHandle handle, double* covariate_data) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !covariate_data) return set_error("GPB_GetCovariateData: null argument");
  if (mdl->p_cov < 1) return set_error("Model does not have covariates for a linear predictor");          // re_model_template.h GetCovariateData
  std::copy(mdl->X.begin(), mdl->X.end(), covariate_data);
  return 0;
}

int GPB_GetOffsetData(REModelHandle handle, double* fixed_effects) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !fixed_effects) return set_error("GPB_GetOffsetData: null argument");
  if (!mdl->has_offset) return set_error("Model does not have an offset term ");     // :6304-6307
  std::copy(mdl->offset.begin(), mdl->offset.end(), fixed_effects);
  C_API_END();
}

int GPB_SetOffsetData(REModelHandle handle, const double* fixed_effects) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !fixed_effects) return set_error("GPB_SetOffsetData: null argument");
  mdl->offset.assign(fixed_effects, fixed_effects + mdl->n);                          // :6318-6321
  mdl->has_offset = true;
  C_API_END();
}

int GPB_GetAuxPars(REModelHandle handle, double* aux_pars, char* out_str, bool calc_std_dev) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_GetAuxPars: null handle");
  if (mdl->num_aux < 1) { if (out_str) out_str[0] = 0; return 0; }      // no auxiliary parameters: empty name, nothing written (NumAuxPars = 0)
  if (aux_pars) for (int j = 0; j < mdl->num_aux; ++j) aux_pars[j] = mdl->aux_pars[j];       // REModel::GetAuxPars (re_model.cpp:1364-1405), original scale
  if (calc_std_dev) {      // round 6: CalculateStandardErrorsAuxPars (re_model_template.h:1894-1909) -> the joint Hessian of CalcStdDevCovParAuxParsNonGaussian; written behind the values
    C_API_BEGIN();
    if (!aux_pars) return set_error("GPB_GetAuxPars: aux_pars is NULL");
    if (!mdl->estimate_aux_pars) return set_error("GPB_GetAuxPars: standard deviations of auxiliary parameters that are not estimated ('estimate_aux_pars' is false)");      // CHECK(estimate_aux_pars_), :1897
    if (!can_calc_std_dev(mdl)) return set_error("GPB_GetAuxPars: standard deviations need the model's gradient on the MI355X path (one cluster, at most 126 neighbours, coordinate dimensions 1..3)");
    if (!mdl->cov_pars_initialized || !mdl->y_set) return set_error("GPB_GetAuxPars: standard deviations need the covariance parameters and the response of a fit or an evaluation");
    {     // the location parameter of the fit: offset + X beta (GetFixedEffectsPtrForStdDevCalc, :1916-1926)
      const double* fel = mdl->has_offset ? mdl->offset.data() : nullptr;
      std::vector<double> fe_lin;
      if (mdl->p_cov > 0 && mdl->coef_estimated) {
        fe_lin.assign(mdl->n, 0.);
        for (int i = 0; i < mdl->n; ++i) { double v = fel ? fel[i] : 0.; for (int j = 0; j < mdl->p_cov; ++j) v += mdl->X[(size_t)j * mdl->n + i] * mdl->beta[j]; fe_lin[i] = v; }
        fel = fe_lin.data();
      }
      if (laplace_upload_fixed_effects(mdl, fel)) return -1;
    }
    const double th[2] = {mdl->cov_pars_tr[0], mdl->cov_pars_tr[1]};
    double se_cov[2], se_aux[2] = {0., 0.};
    char err[512] = "";
    if (gpb_laplace_aux_std_errors(device_laplace_aux, mdl, th, mdl->aux_pars, mdl->num_aux, range_const(mdl), se_cov, se_aux, err, (int)sizeof(err), mdl->optim.estimate_cov_par_index)) {
      const char* why = gpb_hip_get_last_error();
      return (why && why[0]) ? set_error("%s: %s", err[0] ? err : "GPB_GetAuxPars", why) : set_error("%s", err[0] ? err : "evaluation failed");
    }
    for (int j = 0; j < mdl->num_aux; ++j) aux_pars[mdl->num_aux + j] = se_aux[j];             // re_model.cpp:1400-1402
    }   // (C_API_BEGIN's try block)
    catch (const std::exception& e) { return set_error("%s", e.what()); } catch (...) { return set_error("unknown exception"); }
  }
  if (out_str) std::strcpy(out_str, mdl->likelihood == "t" ? "scale_SEP_df" : (mdl->likelihood == "beta" ? "precision" : (mdl->likelihood == "lognormal" ? "log_variance" : (mdl->likelihood == "gaussian_latent" ? "error_variance" : "shape"))));      // lognormal: likelihoods.h:507       // GetNamesAuxPars joins with "_SEP_" (likelihoods.h:2809-2814)         // names_aux_pars_ of gamma / negative_binomial (likelihoods.h:300, :319), beta (:380)
  return 0;
}

int GPB_GetNumAuxPars(REModelHandle handle, int* num_aux_pars) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !num_aux_pars) return set_error("GPB_GetNumAuxPars: null argument");
  num_aux_pars[0] = mdl->num_aux;                     // gaussian / bernoulli_logit / bernoulli_probit / poisson have none (likelihoods.h: num_aux_pars_)
  return 0;
}

int GPB_GetInitAuxPars(REModelHandle handle, double* aux_pars) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_GetInitAuxPars: null handle");
  if (mdl->num_aux > 0 && aux_pars) for (int j = 0; j < mdl->num_aux; ++j) aux_pars[j] = mdl->init_aux_given ? mdl->init_aux[j] : -1.;      // re_model.cpp:1423-1434: -1 = found internally
  return 0;
}

/* PredictTrainingDataRandomEffects (re_model.cpp:1217-1275, re_model_template.h:4455-4514): posterior mean of the latent GP at the
   training locations, Gaussian Vecchia model: b^ = (y - F) - y_aux with y_aux = Psi^-1 (y - F) (:4502-4505); with calc_var the second n
   entries of out_predict are sigma2 (1 - diag(Psi^-1)) (:4508-4514). */
int GPB_PredictREModelTrainingDataRandomEffects(REModelHandle handle, const double* cov_pars_pred, const double* y_obs, double* out_predict,
                                                const double* fixed_effects, bool calc_var) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !out_predict) return set_error("GPB_PredictREModelTrainingDataRandomEffects: null argument");
  // re_model.cpp / re_model_template.h: the reference refuses this call for the approximation, Gaussian or not -- the same words
  if (mdl && mdl->vif) return set_error("PredictTrainingDataRandomEffects() is currently not implemented for the 'full_scale_vecchia' approximation. Call the predict() function instead ");
  if (mdl->likelihood != "gaussian") {
    // non-Gaussian Vecchia models (re_model_template.h:4683-4725): the mode of the latent process, mapped to the data by random_effects_indices_of_data_
    // if locations repeat, and -- calc_var -- diag((Sigma^-1 + W)^-1) (CalcVarLaplaceApproxVecchia): exact, one block solve per 52 random effects
    if (mdl->eh || mdl->vhs.size() != 1) return set_error("GPB_PredictREModelTrainingDataRandomEffects: this model is not on the MI355X path of this library");
    double s12, rho;
    if (cov_pars_pred) { s12 = cov_pars_pred[0]; rho = cov_pars_pred[1]; }
    else {
      if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");   // re_model.cpp:1238-1240
      s12 = mdl->cov_pars_tr[0]; rho = range_const(mdl) / mdl->cov_pars_tr[1];
    }
    if (!(s12 > 0.) || !(rho > 0.)) return set_error("Covariance parameters need to be positive (found %g, %g)", s12, rho);
    const double* fel = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);
    std::vector<double> fe_lin;                  // + X beta of a model fitted with covariates
    if (mdl->p_cov > 0 && mdl->coef_estimated) {
      fe_lin.assign(mdl->n, 0.);
      for (int i = 0; i < mdl->n; ++i) { double v = fel ? fel[i] : 0.; for (int j = 0; j < mdl->p_cov; ++j) v += mdl->X[(size_t)j * mdl->n + i] * mdl->beta[j]; fe_lin[i] = v; }
      fel = fe_lin.data();
    }
    if (y_obs) { if (laplace_upload_data(mdl, y_obs, fel)) return -1; }
    else {
      if (!mdl->y_set) return set_error("Response variable data is not provided and has not been set before");   // :4473-4477
      if (laplace_upload_fixed_effects(mdl, fel)) return -1;
    }
    const int nr = mdl->n_re > 0 ? mdl->n_re : mdl->n;
    if (calc_var && nr > 20000) return set_error("GPB_PredictREModelTrainingDataRandomEffects: variances for %d random effects of a non-Gaussian model (one block solve per 52 of them; limit 20000)", nr);
    std::vector<double> mode(nr), var(calc_var ? nr : 0);
    if (laplace_prepare_preconditioner(mdl)) return -1;
    if (gpb_hip_vecchia_laplace_logit(mdl->vh, mdl->cov_type, s12, range_const(mdl) / rho, mdl->num_rand_vec_trace, mdl->seed_rand_vec_trace, mdl->cg_max_num_it,
                                      mdl->cg_max_num_it_tridiag, mdl->cg_delta_conv, mdl->delta_conv_mode_finding, 1, mdl->lap_info, mode.data()))
      return shim_error();
    if (calc_var && gpb_hip_vecchia_laplace_mode_var(mdl->vh, mdl->cg_max_num_it, kPredVarCgTol, var.data(), nullptr)) return shim_error();
    for (int k = 0; k < mdl->n; ++k) {
      const int r = mdl->n_re > 0 ? mdl->re_of[k] : k;
      out_predict[mdl->perm[k]] = mode[r];
      if (calc_var) out_predict[(size_t)mdl->n + mdl->perm[k]] = var[r];
    }
    return 0;
  }
  double cp[3];
  if (cov_pars_pred) std::copy(cov_pars_pred, cov_pars_pred + 3, cp);
  else {
    if (!mdl->cov_pars_initialized) return set_error("Covariance parameters have not been estimated or are not given.");   // re_model.cpp:1238-1240
    transform_back(mdl, mdl->cov_pars_tr, cp);
  }
  const double* fe = fixed_effects ? fixed_effects : (mdl->has_offset ? mdl->offset.data() : nullptr);   // :4486-4492
  std::vector<double> yc(mdl->n), ya(mdl->n);
  if (y_obs) { for (int i = 0; i < mdl->n; ++i) yc[i] = y_obs[i] - (fe ? fe[i] : 0.); }
  else {
    if (!mdl->y_set) return set_error("Response variable data is not provided and has not been set before");   // :4473-4477
    for (int i = 0; i < mdl->n; ++i) yc[i] = mdl->y_host[i] - (fe ? fe[i] : 0.);   // y_vec_ - fixed effects (:11143-11160)
  }
  {
    std::vector<double> y_keep;                   // yc is a residual, not a response: y_vec_ stays what the caller passed in last
    if (!y_obs) y_keep.swap(mdl->y_host);
    const int rc = GPB_HIP_CalcYAux(handle, yc.data(), cp, ya.data());
    if (!y_obs) mdl->y_host.swap(y_keep); else mdl->y_host.assign(y_obs, y_obs + mdl->n);
    if (rc) return -1;
  }
  for (int i = 0; i < mdl->n; ++i) out_predict[i] = yc[i] - ya[i];
  if (calc_var && mdl->eh) {   // exact GP: Cov[b | y] = Sigma - Sigma Psi^-1 Sigma = sigma2 (I - Psi_t^-1) (the dense branch's M_aux products, :4515-4620)
    double trx[3];
    if (transform_cov_pars(mdl, cp, trx)) return -1;
    std::vector<double> dg(mdl->n);
    if (gpb_hip_exact_psi_inv_diag(mdl->eh, mdl->cov_type, trx[1], trx[2], dg.data())) return shim_error();
    for (int k = 0; k < mdl->n; ++k) out_predict[mdl->n + mdl->perm[k]] = cp[0] * (1. - dg[k]);
  } else if (calc_var) {      // :4508-4514: sigma2 (1 - column sums of B o (D^-1 B)) = sigma2 (1 - diag(B' D^-1 B)); the factor of CalcYAux is resident
    std::vector<double> dg(mdl->n);
    for (size_t c = 0; c < mdl->vhs.size(); ++c)
      if (gpb_hip_vecchia_psi_inv_diag(mdl->vhs[c], dg.data() + mdl->cl_off[c])) return shim_error();
    for (int k = 0; k < mdl->n; ++k) out_predict[mdl->n + mdl->perm[k]] = cp[0] * (1. - dg[k]);
    mdl->yaux_valid = false;                            // the diagonal pass used the y_aux scratch vector
  }
  C_API_END();
}

int GPB_HIP_EvalNegLogLikelihoodAndGrad(REModelHandle handle, const double* y_data, double* cov_pars,
                                        const double* fixed_effects, double* negll, double* grad3) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !negll || !grad3 || !cov_pars) return set_error("GPB_HIP_EvalNegLogLikelihoodAndGrad: null argument");
  if (mdl->likelihood != "gaussian") return set_error("GPB_HIP_EvalNegLogLikelihoodAndGrad: Gaussian likelihood only (the gradient of the Laplace approximation is behind GPB_OptimCovPar and gpb_hip_vecchia_laplace_grad_current)");
  double tr[3];
  if (transform_cov_pars(mdl, cov_pars, tr)) return -1;
  if (upload_y(mdl, y_data, fixed_effects)) return -1;
  double t7[7] = {0, 0, 0, 0, 0, 0, 0};
  if (device_terms(mdl, tr[1], tr[2], 1, t7)) return -1;      // Vecchia: fused point kernel; exact GP: dense path (gpb_hip_exact_grad_terms)
  mdl->cur_negll = negll_from_terms(mdl->n, t7[0], t7[1], tr[0]);
  mdl->negll_valid = true;
  *negll = mdl->cur_negll;
  grad3[0] = -1. * t7[0] / tr[0] / 2. + mdl->n / 2.;   // re_model_template.h:1994
  grad3[1] = t7[3] / tr[0] + t7[4];                    // :2004
  grad3[2] = t7[5] / tr[0] + t7[6];
  C_API_END();
}

/* Test seam, full-scale Vecchia (VIF) models: the derivative factors of the residual process at cov_pars (original scale) on the resident response --
   dA (n x m, Vecchia order, aligned with the neighbour table) and dD (n) of parameter p (0: variance, 1: range; wrt the log of the transformed
   parameter): the reference's -B_grad / D_grad (src/GPBoost/Vecchia_utils.cpp:1640-1656). */
int GPB_HIP_VifGradFactor(REModelHandle handle, double* cov_pars, int p, double* dA, double* dD) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !cov_pars || !dA || !dD) return set_error("GPB_HIP_VifGradFactor: null argument");
  if (!mdl->vif) return set_error("GPB_HIP_VifGradFactor: gp_approx 'full_scale_vecchia' models only");
  if (!mdl->y_set) return set_error("GPB_HIP_VifGradFactor: no response has been set (call GPB_EvalNegLogLikelihood with y_data once)");
  double tr[3];
  if (transform_cov_pars(mdl, cov_pars, tr)) return -1;
  double t7[7];
  if (vif_terms(mdl, tr[1], tr[2], t7, nullptr, 1, 1)) return -1;
  if (gpb_hip_vecchia_vif_get_grad_factor(mdl->vh, p, dA, dD)) return shim_error();
  C_API_END();
}

/* K evaluations of the Gaussian Vecchia likelihood at K parameter sets (row-major K x 3, original scale) on the response already resident
   (GPB_EvalNegLogLikelihood with y_data once, or a fit): one synchronisation and, on a sharded handle, one all-reduce for the whole batch. */
int GPB_HIP_EvalNegLogLikelihoodBatch(REModelHandle handle, int32_t K, const double* cov_pars_K3, double* negll_K) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !cov_pars_K3 || !negll_K || K < 1) return set_error("GPB_HIP_EvalNegLogLikelihoodBatch: invalid argument");
  if (mdl && mdl->vif) return set_error("GPB_HIP_EvalNegLogLikelihoodBatch: gp_approx 'full_scale_vecchia' -- likelihood evaluation, its gradient and fits are on the MI355X path of this library, this call is not yet");
  if (mdl->likelihood != "gaussian" || mdl->eh || mdl->vhs.size() != 1) return set_error("GPB_HIP_EvalNegLogLikelihoodBatch: one-cluster Gaussian Vecchia model only");
  if (!mdl->y_set) return set_error("GPB_HIP_EvalNegLogLikelihoodBatch: no response has been set (call GPB_EvalNegLogLikelihood with y_data once)");
  std::vector<double> var(K), a(K), s2(K), t3((size_t)3 * K);
  for (int k = 0; k < K; ++k) {
    double tr[3];
    if (transform_cov_pars(mdl, cov_pars_K3 + (size_t)3 * k, tr)) return -1;
    s2[k] = tr[0]; var[k] = tr[1]; a[k] = tr[2];
  }
  if (gpb_hip_vecchia_nll_terms_batch(mdl->vh, mdl->cov_type, K, var.data(), a.data(), 1, t3.data())) return shim_error();
  for (int k = 0; k < K; ++k) negll_K[k] = negll_from_terms(mdl->n, t3[(size_t)3 * k], t3[(size_t)3 * k + 1], s2[k]);
  mdl->cur_negll = negll_K[K - 1]; mdl->negll_valid = true;
  C_API_END();
}

int GPB_HIP_CalcYAux(REModelHandle handle, const double* y_data, double* cov_pars, double* y_aux) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !y_aux || !cov_pars) return set_error("GPB_HIP_CalcYAux: null argument");
  if (mdl && mdl->vif) return set_error("GPB_HIP_CalcYAux: gp_approx 'full_scale_vecchia' -- likelihood evaluation, its gradient and fits are on the MI355X path of this library, this call is not yet");
  if (mdl->likelihood != "gaussian") return set_error("GPB_HIP_CalcYAux: only defined for the Gaussian likelihood");
  double tr[3];
  if (transform_cov_pars(mdl, cov_pars, tr)) return -1;
  if (upload_y(mdl, y_data, nullptr, false)) return -1;
  if (mdl->eh) {
    double t2[2];
    if (gpb_hip_exact_nll_terms(mdl->eh, mdl->cov_type, tr[1], tr[2], t2, y_aux, nullptr)) return shim_error();
    return 0;
  }
  std::vector<double> ya(mdl->n);
  for (size_t c = 0; c < mdl->vhs.size(); ++c) {
    if (gpb_hip_vecchia_factor(mdl->vhs[c], mdl->cov_type, tr[1], tr[2], 1)) return shim_error();
    if (gpb_hip_vecchia_yaux(mdl->vhs[c], ya.data() + mdl->cl_off[c])) return shim_error();
  }
  parallel_for(mdl->n, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) y_aux[mdl->perm[k]] = ya[k]; });   // back to data order (GetYAux, :6430); perm is a permutation: disjoint writes
  mdl->yaux_valid = mdl->vhs.size() == 1;
  C_API_END();
}

int GPB_HIP_NewtonUpdateLeafValues(REModelHandle handle, const double* y_data, double* cov_pars, const int32_t* data_leaf_index,
                                   int32_t num_leaves, double* leaf_values) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !data_leaf_index || !leaf_values) return set_error("GPB_HIP_NewtonUpdateLeafValues: null argument");
  if (mdl && mdl->vif) return set_error("GPB_HIP_NewtonUpdateLeafValues: gp_approx 'full_scale_vecchia' -- likelihood evaluation, its gradient and fits are on the MI355X path of this library, this call is not yet");
  if ((y_data == nullptr) != (cov_pars == nullptr)) return set_error("GPB_HIP_NewtonUpdateLeafValues: pass both y_data and cov_pars, or neither (= reuse the state of the last GPB_HIP_CalcYAux)");
  if (mdl->likelihood != "gaussian") return set_error("Newton updates for leaf values is only supported for Gaussian data");   // re_model_template.h:4986-4988
  if (mdl->eh) return set_error("GPB_HIP_NewtonUpdateLeafValues: the exact (dense) GP is not on the MI355X hot path of this library for this call");
  if (mdl->vhs.size() != 1) return set_error("GPB_HIP_NewtonUpdateLeafValues: models with several clusters are not on the MI355X hot path of this library for this call");
  if (!cov_pars) {      // the reference's own contract (re_model_template.h:4989: CHECK(y_aux_has_been_calculated_)): the gradient call ran before
    if (!mdl->yaux_valid) return set_error("GPB_HIP_NewtonUpdateLeafValues: y_aux has not been calculated (call GPB_HIP_CalcYAux first, or pass y_data and cov_pars)");
  } else {
    double tr[3];
    if (transform_cov_pars(mdl, cov_pars, tr)) return -1;
    if (upload_y(mdl, y_data, nullptr, false)) return -1;
    if (gpb_hip_vecchia_factor(mdl->vh, mdl->cov_type, tr[1], tr[2], 1)) return shim_error();
    std::vector<double> ya(mdl->n);
    if (gpb_hip_vecchia_yaux(mdl->vh, ya.data())) return shim_error();
  }
  std::vector<int32_t> leaf(mdl->n);
  parallel_for(mdl->n, [&](int k0, int k1) { for (int k = k0; k < k1; ++k) leaf[k] = data_leaf_index[mdl->perm[k]]; });     // :4999 (data_indices_per_cluster_)
  if (gpb_hip_vecchia_newton_leaf_values(mdl->vh, leaf.data(), num_leaves, leaf_values)) return shim_error();
  C_API_END();
}

int GPB_HIP_PredictVecchiaObsOnly(REModelHandle handle, const double* y_data, double* cov_pars, int32_t num_data_pred,
                                  const double* gp_coords_data_pred, int32_t num_neighbors_pred, bool predict_response,
                                  double* out_mean, double* out_var) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !cov_pars || !gp_coords_data_pred || !out_mean) return set_error("GPB_HIP_PredictVecchiaObsOnly: null argument");
  if (mdl && mdl->vif) return set_error("GPB_HIP_PredictVecchiaObsOnly: gp_approx 'full_scale_vecchia' -- likelihood evaluation, its gradient and fits are on the MI355X path of this library, this call is not yet");
  if (mdl->likelihood != "gaussian" || mdl->eh || mdl->vhs.size() != 1) return set_error("GPB_HIP_PredictVecchiaObsOnly: only the one-cluster Gaussian Vecchia model is on the MI355X hot path of this library");
  double tr[3];
  if (transform_cov_pars(mdl, cov_pars, tr)) return -1;
  if (upload_y(mdl, y_data, nullptr, false)) return -1;
  if (num_neighbors_pred <= 0) num_neighbors_pred = 2 * mdl->num_neighbors;   // re_model_template.h:299: num_neighbors_pred_ = 2 * num_neighbors_
  std::vector<double> D(num_data_pred);
  if (gpb_hip_vecchia_predict_obs_only(mdl->vh, num_data_pred, gp_coords_data_pred, num_neighbors_pred, mdl->cov_type, tr[1], tr[2],
                                       out_mean, D.data(), nullptr)) return shim_error();
  if (out_var)
    for (int k = 0; k < num_data_pred; ++k) out_var[k] = tr[0] * (predict_response ? D[k] : D[k] - 1.);
  C_API_END();
}

int GPB_HIP_GetVecchiaStructure(REModelHandle handle, int32_t* perm, int32_t* nn, int32_t* m_out) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl) return set_error("GPB_HIP_GetVecchiaStructure: null handle");
  if (mdl->eh) return set_error("GPB_HIP_GetVecchiaStructure: the model is an exact GP (gp_approx = 'none')");
  if (mdl->vhs.size() != 1) return set_error("GPB_HIP_GetVecchiaStructure: the model has %d clusters (one neighbour table per cluster)", (int)mdl->vhs.size());
  if (perm) std::copy(mdl->perm.begin(), mdl->perm.end(), perm);
  if (m_out) *m_out = mdl->m;
  if (nn && gpb_hip_vecchia_get_neighbors(mdl->vh, nn)) return shim_error();
  C_API_END();
}

int GPB_HIP_GetLaplaceInfo(REModelHandle handle, double* out9) {
  C_API_BEGIN();
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  if (!mdl || !out9) return set_error("GPB_HIP_GetLaplaceInfo: null argument");
  if (mdl->likelihood == "gaussian" || !mdl->negll_valid) return set_error("GPB_HIP_GetLaplaceInfo: no Laplace approximation has been evaluated");
  std::copy(mdl->lap_info, mdl->lap_info + 9, out9);
  C_API_END();
}

void* GPB_HIP_GetVecchiaHandle(REModelHandle handle) {
  auto* mdl = reinterpret_cast<REModelHip*>(handle);
  return (mdl && mdl->vhs.size() == 1) ? mdl->vh : nullptr;
}

}  // extern "C"
