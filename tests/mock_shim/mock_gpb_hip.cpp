// tests/mock_shim/mock_gpb_hip.cpp -- TEST INFRASTRUCTURE ONLY.  NOT part of the product, never installed next to the package, never found by
// gpboost_amd/libpath.py unless a TEST sets GPBOOST_AMD_LIB to it.
//
// A CPU restatement of the part of the shim (include/gpb_hip.h) that the host half of the C API (gpboost_amd/csrc/gpb_c_api.cpp, gpb_optim.cpp)
// calls, written on top of the ORACLE (oracle/gpb_oracle.c, orc_stdlib.cpp).  Linked with the unmodified gpb_c_api.cpp / gpb_optim.cpp it gives
// tests/mock_shim/libgpb_c_api_on_oracle_TEST_ONLY.so, with which the CPU suite can run the C API's host orchestration -- cluster handling, covariates,
// prediction bookkeeping, the optimisers' control flow -- end to end without a device (tests/test_c_api_host_logic.py).  It says nothing about the
// HIP kernels: those are tested on the MI355X by the -m gpu tests, which load the real library.
// Everything the C API may call but this file does not restate returns -1 with a message.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

extern "C" {
// ---- the oracle's entry points used here (oracle/gpb_oracle.c, oracle/orc_stdlib.cpp) ----
void orc_coords_sum(const double* coords, int n, int d, double* coords_sum);
void orc_vecchia_neighbors_range(const double* coords, int n, int d, int m, const int* sort_sum, int start_at, int end_search_at, int* nn, double* nn_sqd);
int orc_vecchia_factor(const double* coords, int n, int d, const int* nn, int m, int cov_type, double var, double a, int gauss, double* A, double* D,
                       double* A_grad, double* D_grad);
void orc_vecchia_By(const double* A, const int* nn, int n, int m, const double* y, double* u);
void orc_vecchia_yaux(const double* A, const double* D, const int* nn, int n, int m, const double* y, double* y_aux);
int orc_newton_leaf_values(const double* A, const double* D, const int* nn, int n, int m, const double* yaux, const int* leaf, int L, double* leaf_values);
void orc_gen_rand_normal(int seed, unsigned long long run_id, int n, int t, double* out);
void orc_set_aux(double aux, const double* y_real, double* aux_grad4);
void orc_set_aux2(double aux2);
void orc_set_weights(const double* w);
void orc_set_binomial(int on);
void orc_clear_aux(void);
int orc_pivoted_cholesky(const double* coords, int n, int d, int cov_type, double var, double a, int max_it, double err_tol, double* L_out);
void orc_set_pivchol(const double* L_k, int k, const double* rand_vec2);
void orc_clear_pivchol(void);
void orc_set_vecchia_response(const double* coords, int d, int cov_type, double var, double a);
void orc_set_fitc(const double* C_nk, const double* V_nk, const double* Sm_kk, double logdet_Sm, int k, const double* rand_vec2);
int orc_vecchia_laplace_grad_map_dbg(int link, const double* A, const double* D, const double* Ag, const double* Dg, const int* nn, int n, int m,
                                     const int* dptr, const int* y_int, const double* fe, const double* rand_vec, int t, int cg_max_num_it,
                                     int cg_max_num_it_tridiag, double cg_delta_conv, double delta_conv_mode_finding, double* out6, double* grad2,
                                     double* mode_io, int use_mode_init, double* dbg);
}

namespace {
thread_local char g_err[512] = "";
int fail(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); return -1; }

double sigmoid(double x) { return x >= 0. ? 1. / (1. + std::exp(-x)) : std::exp(x) / (1. + std::exp(x)); }
double normal_log_cdf(double x) {
  if (x < 0.0) {
    const double e = std::erfc(-x * M_SQRT1_2);
    if (e > 0.0) return std::log(0.5) + std::log(e);
    const double u = -x, u2 = u * u;
    return -0.5 * u2 - std::log(u) - 0.5 * std::log(2 * M_PI) + std::log(1.0 - 1.0 / u2 + 3.0 / (u2 * u2));
  }
  const double Q = 0.5 * std::erfc(x * M_SQRT1_2);
  return Q == 0.0 ? 0.0 : std::log1p(-Q);
}
// d log p / d loc, information, d information / d loc
double g_mock_aux = 1.0;        // auxiliary parameter seen by lik_terms (links 3 / 4; set by the entry points from the handle)
double g_mock_aux2 = 2.0;       // t: degrees of freedom
// beta (link 5): polygamma functions as the reference's (src/GPBoost/DF_utils.cpp:82-201), mean = clamped sigmoid (include/GPBoost/DF_utils.h:48-55)
double mock_digamma(double x) {
  if (x <= 0.000001) return -0.57721566490153286060 - 1.0 / x + 1.6449340668482264365 * x;
  double v = 0.;
  while (x < 8.5) { v -= 1. / x; x += 1.; }
  double r = 1. / x;
  v += std::log(x) - 0.5 * r;
  r = r * r;
  return v - r * (1. / 12. - r * (1. / 120. - r * (1. / 252. - r * (1. / 240. - r * (1. / 132.)))));
}
double mock_trigamma(double x) {
  if (x <= 0.0001) return 1.0 / x / x;
  double value = 0.0, z = x;
  while (z < 5.0) { value = value + 1.0 / z / z; z = z + 1.0; }
  const double y = 1.0 / z / z;
  return value + 0.5 * y + (1.0 + y * (0.1666666667 + y * (-0.03333333333 + y * (0.02380952381 + y * -0.03333333333)))) / z;
}
double mock_tetragamma(double x) {
  if (x <= 1e-4) return -2.0 / (x * x * x);
  double z = x, value = 0.0;
  while (z < 8.0) { value -= 2.0 / (z * z * z); z += 1.0; }
  const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z6 = z4 * z2, z8 = z4 * z4, z10 = z8 * z2;
  return value + (-1.0 / z2 - 1.0 / z3 - 0.5 / z4 + 1.0 / (6.0 * z6) - 1.0 / (6.0 * z8) + 3.0 / (10.0 * z10));
}
void lik_terms(int link, double y, double x, double* first, double* info, double* dinfo) {
  if (link == 8) {       // gaussian_latent: FirstDerivLogLikGaussian (likelihoods.h:12514-12516), information 1 / aux
    *first = (y - x) / g_mock_aux; *info = 1. / g_mock_aux; *dinfo = 0.;
    return;
  }
  if (link == 7) {       // lognormal: FirstDerivLogLikLogNormal, SecondDerivNegLogLikLogNormal (likelihoods.h:12534-12538, :13384-13386); constant information
    *first = (std::log(y) - (x - 0.5 * g_mock_aux)) / g_mock_aux; *info = 1. / g_mock_aux; *dinfo = 0.;
    return;
  }
  if (link == 6) {       // t, fisher_laplace: FirstDerivLogLikT, FisherInformationT (likelihoods.h:12509-12512, :13358-13360); the information does not depend on the location
    const double sc = g_mock_aux, nu = g_mock_aux2, res = y - x;
    *first = (nu + 1.) * res / (nu * sc * sc + res * res); *info = (nu + 1.) / (nu + 3.) / (sc * sc); *dinfo = 0.;
    return;
  }
  if (link == 5) {       // likelihoods.h:12501-12507, :13336-13346, :13892-13917
    const double phi = g_mock_aux;
    double mu = sigmoid(x); if (mu < 1e-12) mu = 1e-12; if (mu > 1.0 - 1e-12) mu = 1.0 - 1e-12;
    const double d = mu * (1.0 - mu), logit_y = std::log(y) - std::log1p(-y);
    const double dig1 = mock_digamma((1.0 - mu) * phi), dig2 = mock_digamma(mu * phi), tri1 = mock_trigamma((1.0 - mu) * phi), tri2 = mock_trigamma(mu * phi);
    const double tet1 = mock_tetragamma((1.0 - mu) * phi), tet2 = mock_tetragamma(mu * phi);
    const double C = dig1 - dig2 + logit_y, S = tri1 + tri2;
    *first = phi * d * C;
    *info = -(-phi * phi * d * d * S + phi * d * (1.0 - 2.0 * mu) * C);
    *dinfo = 3.0 * phi * phi * d * d * (1.0 - 2.0 * mu) * S + phi * phi * phi * d * d * d * (tet2 - tet1) + -phi * (d * ((1.0 - 2.0 * mu) * (1.0 - 2.0 * mu) - 2.0 * d)) * C;
    return;
  }
  if (link == 3) { const double r = g_mock_aux, q = y * std::exp(-x); *first = r * (q - 1.); *info = r * q; *dinfo = -r * q; return; }
  if (link == 4) {
    const double r = g_mock_aux, mu = std::exp(x), mr = mu + r;
    *first = y - (y + r) / mr * mu; *info = (y + r) * mu * r / (mr * mr); *dinfo = -(y + r) * mu * r * (mu - r) / (mr * mr * mr); return;
  }
  if (link == 0) { const double p = sigmoid(x); *first = y - p; *info = p * (1. - p); *dinfo = p * (1. - p) * (1. - 2. * p); return; }
  if (link == 2) { const double e = std::exp(x); *first = y - e; *info = e; *dinfo = e; return; }
  if (y != 0. && y != 1.) {         // a proportion (binomial_probit / quasi_bernoulli_probit): y f(1) + (1 - y) f(0)
    double f1, i1, d1, f0, i0, d0;
    lik_terms(1, 1., x, &f1, &i1, &d1); lik_terms(1, 0., x, &f0, &i0, &d0);
    *first = y * f1 + (1. - y) * f0; *info = y * i1 + (1. - y) * i0; *dinfo = y * d1 + (1. - y) * d0;
    return;
  }
  const double z = y > 0 ? x : -x;
  const double r = std::exp(-0.5 * z * z - 0.5 * std::log(2 * M_PI) - normal_log_cdf(z));
  *first = y > 0 ? r : -r;
  *info = r * (z + r);
  const double dz = -r * (z + r) * (z + r) + r * (1. - r * (z + r));
  *dinfo = y > 0 ? dz : -dz;
}
}  // namespace

// GPB_MOCK_TIMING=1: seconds spent inside the restated calls, printed at exit (to separate the host code under test from the oracle's arithmetic)
#include <chrono>
#include <cstdio>
#include <cstdlib>
namespace {
struct MockTimes { double t[8] = {0}; long c[8] = {0}; const char* name[8] = {"factor", "nll_terms", "grad_terms", "yaux", "newton_leaf_values", "get_factor", "set_y", "laplace_eval / grad_current"};
  ~MockTimes() { if (std::getenv("GPB_MOCK_TIMING")) for (int i = 0; i < 8; ++i) if (c[i]) std::fprintf(stderr, "mock timing: %-20s %6ld calls %9.3f s\n", name[i], c[i], t[i]); } };
MockTimes g_times;
static const bool g_mock_trace = std::getenv("GPB_MOCK_TRACE") != nullptr;
#define MOCK_TRACE(name) do { if (g_mock_trace) std::fprintf(stderr, "mock call: %s\n", name); } while (0)
struct MockTimer { int k; std::chrono::steady_clock::time_point t0; explicit MockTimer(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
  ~MockTimer() { g_times.t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g_times.c[k]++; } };
}  // namespace
struct gpb_hip_vecchia {
  int n = 0, d = 0, m = 0;
  std::vector<double> coords;                  // column-major n x d
  std::vector<int> nn; bool has_nn = false;
  std::vector<double> y; bool has_y = false;   // the response the likelihood is evaluated at (y0, or y0 - X beta after set_resid)
  std::vector<double> y0;                      // the response last uploaded with set_y
  int p = 0; std::vector<double> X;            // covariates, column-major [p][n], Vecchia order
  std::vector<double> A, D; bool has_factor = false; int f_gauss = 1;
  // Laplace state
  int link = 0;
  std::vector<int> labels; std::vector<double> fe; bool has_fe = false;
  std::vector<double> weights;     // sample weights of the non-Gaussian likelihood (order of the labels); the oracle reads them through orc_set_weights
  double wv(int k) const { return weights.empty() ? 1.0 : weights[k]; }
  std::vector<double> resp_real; double aux = 1.0; double aux2 = 2.0; double aux_grad4[8] = {0., 0., 0., 0., 0., 0., 0., 0.};     // gamma's response, the shape, the last aux gradient
  bool real_resp = false, binomial = false;      // proportions under the logit / probit links (binomial_*, quasi_bernoulli_*)
  int pc_type = 0, pc_rank = 50;                  // cg_preconditioner_type: 0 = vadu, 1 = pivoted_cholesky with pc_rank columns, 2 = fitc with the inducing points pc_ip
  std::vector<double> pc_ip; int pc_nip = 0;       // k x d column-major
  double yv(int k) const { return (link == 3 || link == 5 || link == 6 || link == 7 || link == 8 || real_resp) ? resp_real[k] : (double)labels[k]; }
  std::vector<int> re_ptr;                     // empty: one datum per random effect
  std::vector<double> mode, mode_prev, dld, sv; double grad2[2] = {0., 0.};
  bool has_mode = false, grad_state = false;
  int l_cov = 0; double l_var = 0., l_a = 0.;
};
typedef gpb_hip_vecchia gpb_hip_vecchia_t;

namespace {
std::vector<int> sort_by_coordinate_sum(const double* c, int n, int d) {
  std::vector<double> cs(n);
  orc_coords_sum(c, n, d, cs.data());
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  const double* v = cs.data();
  std::sort(idx.begin(), idx.end(), [v](int a, int b) { return v[a] < v[b]; });
  return idx;
}
bool any_duplicate_neighbor(const std::vector<double>& c, int n_all, int d, const std::vector<int>& nn, int m, int row0) {
  for (int i = row0; i < n_all; ++i)
    for (int j = 0; j < m; ++j) {
      const int k = nn[(size_t)i * m + j];
      if (k < 0) continue;
      double s2 = 0.;
      for (int q = 0; q < d; ++q) { const double t = c[(size_t)q * n_all + i] - c[(size_t)q * n_all + k]; s2 += t * t; }
      if (s2 < 1e-20) return true;
    }
  return false;
}
// [observed; prediction] coordinates (column-major), neighbour search for the appended rows, factor of every row from row0 on
struct Appended { int n_all = 0, m = 0; std::vector<double> c; std::vector<int> nn; std::vector<double> A, D; bool dup = false; };
int appended_factor(const gpb_hip_vecchia* h, int n_pred, const double* cp, int mp, bool cond_all, bool all_rows, bool pred_first, int cov, double var, double a,
                    int gauss, Appended* out) {
  const int n = h->n, d = h->d, n_all = n + n_pred;
  const int obs0 = pred_first ? n_pred : 0, pred0 = pred_first ? 0 : n;
  out->n_all = n_all;
  out->c.assign((size_t)n_all * d, 0.);
  for (int q = 0; q < d; ++q) {
    std::copy(h->coords.begin() + (size_t)q * n, h->coords.begin() + (size_t)(q + 1) * n, out->c.begin() + (size_t)q * n_all + obs0);
    std::copy(cp + (size_t)q * n_pred, cp + (size_t)(q + 1) * n_pred, out->c.begin() + (size_t)q * n_all + pred0);
  }
  int m = mp;
  const int m_cap = cond_all ? n_all - 1 : n;
  if (m > m_cap) m = m_cap;
  if (m < 1) return fail("mock: num_neighbors_pred = %d", mp);
  out->m = m;
  out->nn.assign((size_t)n_all * m, -1);
  const std::vector<int> ss = sort_by_coordinate_sum(out->c.data(), n_all, d);
  const int start_at = all_rows ? 0 : n;
  orc_vecchia_neighbors_range(out->c.data(), n_all, d, m, ss.data(), start_at, cond_all ? n_all - 2 : n - 1, out->nn.data(), nullptr);
  out->A.assign((size_t)n_all * m, 0.); out->D.assign(n_all, 0.);
  orc_vecchia_factor(out->c.data(), n_all, d, out->nn.data(), m, cov, var, a, gauss, out->A.data(), out->D.data(), nullptr, nullptr);
  out->dup = any_duplicate_neighbor(out->c, n_all, d, out->nn, m, start_at);
  return 0;
}
// dense (Sigma^-1 + W) of the Laplace state at the current mode, its lower Cholesky factor in place
int dense_M_chol(const gpb_hip_vecchia* h, std::vector<double>* Mout) {
  const int n = h->n, m = h->m;
  std::vector<double>& M = *Mout;
  M.assign((size_t)n * n, 0.);
  std::vector<int> ec; std::vector<double> ev;
  for (int i = 0; i < n; ++i) {
    ec.assign(1, i); ev.assign(1, 1.);
    for (int j = 0; j < m; ++j) { const int c = h->nn[(size_t)i * m + j]; if (c >= 0) { ec.push_back(c); ev.push_back(-h->A[(size_t)i * m + j]); } }
    const double di = 1. / h->D[i];
    for (size_t a1 = 0; a1 < ec.size(); ++a1) for (size_t b1 = 0; b1 < ec.size(); ++b1) M[(size_t)ec[a1] * n + ec[b1]] += ev[a1] * di * ev[b1];
  }
  const bool mapped = !h->re_ptr.empty();
  g_mock_aux = h->aux; g_mock_aux2 = h->aux2;
  for (int i = 0; i < n; ++i) {
    const int d0 = mapped ? h->re_ptr[i] : i, d1 = mapped ? h->re_ptr[i + 1] : i + 1;
    double w = 0.;
    for (int k = d0; k < d1; ++k) { double f, inf, di; lik_terms(h->link, h->yv(k), h->mode[i] + (h->has_fe ? h->fe[k] : 0.), &f, &inf, &di); w += h->wv(k) * inf; }
    M[(size_t)i * n + i] += w;
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double acc = M[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) acc -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
      if (i == j) { if (!(acc > 0.)) return fail("mock: Sigma^-1 + W is not positive definite"); M[(size_t)i * n + i] = std::sqrt(acc); }
      else M[(size_t)i * n + j] = acc / M[(size_t)j * n + j];
    }
  return 0;
}
void chol_solve(const std::vector<double>& L, int n, std::vector<double>& b) {      // b <- (L L')^-1 b
  for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * b[k]; b[i] = v / L[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * b[k]; b[i] = v / L[(size_t)i * n + i]; }
}
int laplace_run(gpb_hip_vecchia* h, int cov, double var, double a, int nrv, int seed, int cg, int cgt, double cgd, double dcm, int reset, double* out9, double* mode_host) {
  const int n = h->n, m = h->m;
  if (!h->has_nn) return fail("mock: neighbours have not been determined");
  const bool mapped = !h->re_ptr.empty();
  const int nd = mapped ? h->re_ptr[n] : n;
  if ((int)h->labels.size() != nd) return fail("labels have not been set (call gpb_hip_vecchia_laplace_set_labels)");
  std::vector<double> Ag((size_t)2 * n * m), Dg((size_t)2 * n);
  h->A.assign((size_t)n * m, 0.); h->D.assign(n, 0.);
  orc_vecchia_factor(h->coords.data(), n, h->d, h->nn.data(), m, cov, var, a, 0, h->A.data(), h->D.data(), Ag.data(), Dg.data());
  h->has_factor = true; h->f_gauss = 0;
  std::vector<double> rv((size_t)n * nrv);
  // pivoted_cholesky: L_k at these parameters, rand_vec_trace_I2_ drawn first (generator counter 0), rand_vec_trace_I_ second (1)
  std::vector<double> pcL, rv2, pcV, pcSm;
  const bool pc = h->pc_type >= 1;
  if (h->pc_type == 2) {       // fitc: cross-covariance C, Sigma_m (diagonal x (1 + 1e-6)), V = (L_m^-1 C')' (Calc_FITC_Preconditioner_Vecchia, re_model_template.h:9577-9591)
    const int k = h->pc_nip, d = h->d;
    if (k < 1) return fail("the fitc preconditioner needs its inducing points (gpb_hip_vecchia_laplace_set_inducing_points)");
    auto covf = [&](double dist) { const double r = a * dist; return cov == 0 ? var * std::exp(-r) : (cov == 1 ? var * (1. + r) * std::exp(-r) : var * (1. + r + r * r / 3.) * std::exp(-r)); };
    pcL.assign((size_t)n * k, 0.); pcV.assign((size_t)n * k, 0.); pcSm.assign((size_t)k * k, 0.); rv2.assign((size_t)k * nrv, 0.);
    for (int p = 0; p < k; ++p) for (int q = 0; q < k; ++q) {
      double d2 = 0.; for (int c = 0; c < d; ++c) { const double t = h->pc_ip[(size_t)c * k + p] - h->pc_ip[(size_t)c * k + q]; d2 += t * t; }
      pcSm[(size_t)p * k + q] = covf(std::sqrt(d2)) * (p == q ? 1.0 + 1e-6 : 1.0);
    }
    std::vector<double> Lm((size_t)k * k, 0.);
    double ld = 0.;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j <= i; ++j) {
        double acc = pcSm[(size_t)i * k + j];
        for (int q = 0; q < j; ++q) acc -= Lm[(size_t)i * k + q] * Lm[(size_t)j * k + q];
        if (i == j) { if (!(acc > 0.)) return fail("mock: Sigma_m is not positive definite"); Lm[(size_t)i * k + i] = std::sqrt(acc); ld += std::log(Lm[(size_t)i * k + i]); }
        else Lm[(size_t)i * k + j] = acc / Lm[(size_t)j * k + j];
      }
    for (int i = 0; i < n; ++i) {
      for (int q = 0; q < k; ++q) {
        double d2 = 0.; for (int c = 0; c < d; ++c) { const double t = h->coords[(size_t)c * n + i] - h->pc_ip[(size_t)c * k + q]; d2 += t * t; }
        pcL[(size_t)q * n + i] = covf(std::sqrt(d2));
      }
      for (int q = 0; q < k; ++q) {       // V_i = L_m^-1 C_i'
        double acc = pcL[(size_t)q * n + i];
        for (int p = 0; p < q; ++p) acc -= Lm[(size_t)q * k + p] * pcV[(size_t)p * n + i];
        pcV[(size_t)q * n + i] = acc / Lm[(size_t)q * k + q];
      }
    }
    orc_gen_rand_normal(seed, 0ull, k, nrv, rv2.data());
    orc_set_fitc(pcL.data(), pcV.data(), pcSm.data(), 2. * ld, k, rv2.data());
  } else
  if (h->pc_type == 3) orc_set_vecchia_response(h->coords.data(), h->d, cov, var, a);      // vecchia_response: no low-rank part, no second set of normals
  else
  if (pc) {
    const int k = std::min(h->pc_rank, n);
    pcL.assign((size_t)n * k, 0.); rv2.assign((size_t)k * nrv, 0.);
    orc_pivoted_cholesky(h->coords.data(), n, h->d, cov, var, a, k, 1e-6, pcL.data());
    orc_gen_rand_normal(seed, 0ull, k, nrv, rv2.data());
    orc_set_pivchol(pcL.data(), k, rv2.data());
  }
  orc_gen_rand_normal(seed, (pc && h->pc_type != 3) ? 1ull : 0ull, n, nrv, rv.data());
  std::vector<int> dptr;
  if (mapped) dptr = h->re_ptr; else { dptr.resize(n + 1); std::iota(dptr.begin(), dptr.end(), 0); }
  const bool warm = !reset && h->has_mode;
  std::vector<double> mode = warm ? h->mode : std::vector<double>(n, 0.);
  h->mode_prev = warm ? h->mode : std::vector<double>(n, 0.);
  std::vector<double> dbg((size_t)2 * n + 8, 0.);
  double out6[6] = {0, 0, 0, 0, 0, 0};
  const bool ctx = h->link >= 3 || h->real_resp;
  if (ctx) orc_set_aux(h->aux, (h->link == 3 || h->link == 5 || h->link == 6 || h->link == 7 || h->link == 8 || h->real_resp) ? h->resp_real.data() : nullptr, h->link >= 3 ? h->aux_grad4 : nullptr);
  if (h->link == 6) orc_set_aux2(h->aux2);
  orc_set_binomial(h->binomial ? 1 : 0);
  const int rc = orc_vecchia_laplace_grad_map_dbg(h->link, h->A.data(), h->D.data(), Ag.data(), Dg.data(), h->nn.data(), n, m, dptr.data(), h->labels.data(),
                                                  h->has_fe ? h->fe.data() : nullptr, rv.data(), nrv, cg, cgt, cgd, dcm, out6, h->grad2, mode.data(), warm ? 1 : 0,
                                                  dbg.data());
  if (ctx) orc_clear_aux();
  orc_set_binomial(0);
  if (pc) orc_clear_pivchol();
  if (rc) return fail("NaN or Inf occurred in the mode finding algorithm for the Laplace approximation");
  h->mode = mode; h->has_mode = true; h->grad_state = h->pc_type != 3;
  h->dld.assign(dbg.begin(), dbg.begin() + n); h->sv.assign(dbg.begin() + n, dbg.begin() + 2 * n);
  h->l_cov = cov; h->l_var = var; h->l_a = a;
  for (int k = 0; k < 9; ++k) out9[k] = 0.;
  for (int k = 0; k < 6; ++k) out9[k] = out6[k];
  if (mode_host) std::copy(mode.begin(), mode.end(), mode_host);
  return 0;
}
}  // namespace

extern "C" {
#define EXPORT __attribute__((visibility("default")))
EXPORT const char* gpb_hip_get_last_error(void) { return g_err; }
EXPORT int gpb_hip_device_count(int* count) { if (count) *count = 1; return 0; }
EXPORT int gpb_hip_set_device(int) { return 0; }
EXPORT int gpb_hip_selftest(void) { return 0; }
EXPORT int gpb_hip_pinned_alloc(size_t bytes, void** out) { *out = std::malloc(bytes ? bytes : 1); return *out ? 0 : fail("mock: out of memory"); }
EXPORT int gpb_hip_pinned_free(void* p) { std::free(p); return 0; }
EXPORT int gpb_hip_vecchia_comm_info(gpb_hip_vecchia_t*, int* rank, int* world) { if (rank) *rank = 0; if (world) *world = 0; return 0; }

EXPORT int gpb_hip_vecchia_create(int32_t n, int32_t d, int32_t num_neighbors, const double* coords_colmajor, gpb_hip_vecchia_t** out) {
  if (!out || !coords_colmajor || n < 1 || d < 1) return fail("mock: gpb_hip_vecchia_create: invalid argument");
  int m = std::min(num_neighbors, n - 1);
  if (num_neighbors < 1 && n > 1) return fail("gpb_hip_vecchia_create: num_neighbors = %d", num_neighbors);
  if (m > 126) return fail("gpb_hip_vecchia_create: num_neighbors = %d exceeds the supported maximum %d", m, 126);
  auto* h = new gpb_hip_vecchia();
  h->n = n; h->d = d; h->m = m < 1 ? 1 : m;
  h->coords.assign(coords_colmajor, coords_colmajor + (size_t)n * d);
  *out = h;
  return 0;
}
EXPORT int gpb_hip_vecchia_free(gpb_hip_vecchia_t* h) { if (h && !h->weights.empty()) orc_set_weights(nullptr); delete h; return 0; }
EXPORT int gpb_hip_vecchia_find_neighbors(gpb_hip_vecchia_t* h, int* has_duplicates) {
  const int n = h->n, m = h->m;
  h->nn.assign((size_t)n * m, -1);
  if (n > 1) {
    const std::vector<int> ss = sort_by_coordinate_sum(h->coords.data(), n, h->d);
    orc_vecchia_neighbors_range(h->coords.data(), n, h->d, m, ss.data(), 0, -1, h->nn.data(), nullptr);
  }
  h->has_nn = true; h->has_factor = false;
  if (has_duplicates) *has_duplicates = any_duplicate_neighbor(h->coords, n, h->d, h->nn, m, 0) ? 1 : 0;
  return 0;
}
EXPORT int gpb_hip_vecchia_set_neighbors(gpb_hip_vecchia_t* h, const int32_t* nn) { h->nn.assign(nn, nn + (size_t)h->n * h->m); h->has_nn = true; h->has_factor = false; return 0; }
EXPORT int gpb_hip_vecchia_get_neighbors(gpb_hip_vecchia_t* h, int32_t* nn) {
  if (!h->has_nn) return fail("neighbours have not been determined");
  std::copy(h->nn.begin(), h->nn.end(), nn);
  return 0;
}
EXPORT int gpb_hip_vecchia_set_y(gpb_hip_vecchia_t* h, const double* y_host) { MockTimer mock_timer_(6); h->y.assign(y_host, y_host + h->n); h->y0 = h->y; h->has_y = true; return 0; }
EXPORT int gpb_hip_vecchia_set_covariates(gpb_hip_vecchia_t* h, int32_t p, const double* X) {
  if (p < 0 || (p > 0 && !X)) return fail("mock: set_covariates: bad arguments");
  h->p = p; h->X.assign(X, X + (size_t)p * h->n);
  return 0;
}
EXPORT int gpb_hip_vecchia_set_resid(gpb_hip_vecchia_t* h, const double* beta) {
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  h->y = h->y0;
  if (beta) for (int j = 0; j < h->p; ++j) for (int i = 0; i < h->n; ++i) h->y[i] -= h->X[(size_t)j * h->n + i] * beta[j];
  return 0;
}
// G = (B [X, y0])' D^-1 (B [X, y0]), (p + 1) x (p + 1) row-major, with the factor of the last gpb_hip_vecchia_factor
EXPORT int gpb_hip_vecchia_gram(gpb_hip_vecchia_t* h, double* G) {
  if (!h->has_factor) return fail("mock: gram before factor");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  const int n = h->n, q = h->p + 1;
  std::vector<std::vector<double>> u(q, std::vector<double>(n));
  for (int j = 0; j < q; ++j) orc_vecchia_By(h->A.data(), h->nn.data(), n, h->m, j < h->p ? h->X.data() + (size_t)j * n : h->y0.data(), u[j].data());
  for (int a = 0; a < q; ++a) for (int b = 0; b <= a; ++b) {
    double s2 = 0.;
    for (int i = 0; i < n; ++i) s2 += u[a][i] * u[b][i] / h->D[i];
    G[(size_t)a * q + b] = G[(size_t)b * q + a] = s2;
  }
  return 0;
}
EXPORT int gpb_hip_vecchia_factor(gpb_hip_vecchia_t* h, int cov, double var, double a, int gauss) { MOCK_TRACE("gpb_hip_vecchia_factor"); MockTimer mock_timer_(0);
  if (!h->has_nn) return fail("neighbours have not been determined");
  h->A.assign((size_t)h->n * h->m, 0.); h->D.assign(h->n, 0.);
  orc_vecchia_factor(h->coords.data(), h->n, h->d, h->nn.data(), h->m, cov, var, a, gauss, h->A.data(), h->D.data(), nullptr, nullptr);
  h->has_factor = true; h->f_gauss = gauss;
  return 0;
}
EXPORT int gpb_hip_vecchia_nll_terms(gpb_hip_vecchia_t* h, int cov, double var, double a, int gauss, double* out3) { MockTimer mock_timer_(1);
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  if (gpb_hip_vecchia_factor(h, cov, var, a, gauss)) return -1;
  std::vector<double> u(h->n);
  orc_vecchia_By(h->A.data(), h->nn.data(), h->n, h->m, h->y.data(), u.data());
  double q = 0., ld = 0.; int bad = 0;
  for (int i = 0; i < h->n; ++i) { q += u[i] * u[i] / h->D[i]; ld += std::log(h->D[i]); if (!(h->D[i] > 0.)) ++bad; }
  out3[0] = q; out3[1] = ld; out3[2] = bad;
  return 0;
}
EXPORT int gpb_hip_vecchia_nll_terms_batch(gpb_hip_vecchia_t* h, int cov, int32_t K, const double* var, const double* a, int gauss, double* out3K) {
  for (int k = 0; k < K; ++k) if (gpb_hip_vecchia_nll_terms(h, cov, var[k], a[k], gauss, out3K + 3 * (size_t)k)) return -1;
  return 0;
}
EXPORT int gpb_hip_vecchia_grad_terms(gpb_hip_vecchia_t* h, int cov, double var, double a, double* t7) { MockTimer mock_timer_(2);
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  if (!h->has_nn) return fail("neighbours have not been determined");
  const int n = h->n, m = h->m;
  std::vector<double> Ag((size_t)2 * n * m), Dg((size_t)2 * n);
  h->A.assign((size_t)n * m, 0.); h->D.assign(n, 0.);
  orc_vecchia_factor(h->coords.data(), n, h->d, h->nn.data(), m, cov, var, a, 1, h->A.data(), h->D.data(), Ag.data(), Dg.data());
  h->has_factor = true; h->f_gauss = 1;
  std::vector<double> u(n);
  orc_vecchia_By(h->A.data(), h->nn.data(), n, m, h->y.data(), u.data());
  double q = 0., ld = 0.; int bad = 0;
  for (int i = 0; i < n; ++i) { q += u[i] * u[i] / h->D[i]; ld += std::log(h->D[i]); if (!(h->D[i] > 0.)) ++bad; }
  t7[0] = q; t7[1] = ld; t7[2] = bad;
  for (int k = 0; k < 2; ++k) {       // g_k = t[3 + 2 k] / sigma2 + t[4 + 2 k] (re_model_template.h:1988-2011)
    double tB = 0., tD = 0.;
    for (int i = 0; i < n; ++i) {
      double dBy = 0.;
      for (int j = 0; j < m; ++j) { const int c = h->nn[(size_t)i * m + j]; if (c >= 0) dBy -= Ag[((size_t)k * n + i) * m + j] * h->y[c]; }
      const double ui = u[i] / h->D[i], dD = Dg[(size_t)k * n + i];
      tB += dBy * ui - 0.5 * ui * ui * dD;
      tD += 0.5 * dD / h->D[i];
    }
    t7[3 + 2 * k] = tB; t7[4 + 2 * k] = tD;
  }
  return 0;
}
EXPORT int gpb_hip_vecchia_yaux(gpb_hip_vecchia_t* h, double* yaux_host) { MockTimer mock_timer_(3);
  if (!h->has_factor || !h->has_y) return fail("mock: factor / response missing");
  orc_vecchia_yaux(h->A.data(), h->D.data(), h->nn.data(), h->n, h->m, h->y.data(), yaux_host);
  return 0;
}
EXPORT int gpb_hip_vecchia_psi_inv_diag(gpb_hip_vecchia_t* h, double* diag_host) {
  if (!h->has_factor) return fail("mock: factor missing");
  for (int i = 0; i < h->n; ++i) diag_host[i] = 1. / h->D[i];
  for (int i = 0; i < h->n; ++i)
    for (int j = 0; j < h->m; ++j) { const int c = h->nn[(size_t)i * h->m + j]; if (c >= 0) diag_host[c] += h->A[(size_t)i * h->m + j] * h->A[(size_t)i * h->m + j] / h->D[i]; }
  return 0;
}
EXPORT int gpb_hip_vecchia_predict_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* cp, int32_t mp, int cov, double var, double a, double* pred_mean,
                                            double* pred_D, int* has_duplicates) {
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  Appended ap;
  if (appended_factor(h, n_pred, cp, mp, false, false, false, cov, var, a, 1, &ap)) return -1;
  for (int k = 0; k < n_pred; ++k) {
    const int i = h->n + k;
    double s = 0.;
    for (int j = 0; j < ap.m; ++j) { const int c = ap.nn[(size_t)i * ap.m + j]; if (c >= 0) s += ap.A[(size_t)i * ap.m + j] * h->y[c]; }
    pred_mean[k] = s; pred_D[k] = ap.D[i];
  }
  if (has_duplicates) *has_duplicates = ap.dup ? 1 : 0;
  return 0;
}
EXPORT int gpb_hip_vecchia_predict_cond_all(gpb_hip_vecchia_t* h, int32_t n_pred, const double* cp, int32_t mp, int cov, double var, double a, int32_t* m_used,
                                            int32_t* nn_pred, double* A_pred, double* D_pred, int* has_duplicates) {
  Appended ap;
  if (appended_factor(h, n_pred, cp, mp, true, false, false, cov, var, a, 1, &ap)) return -1;
  *m_used = ap.m;
  std::copy(ap.nn.begin() + (size_t)h->n * ap.m, ap.nn.end(), nn_pred);
  std::copy(ap.A.begin() + (size_t)h->n * ap.m, ap.A.end(), A_pred);
  std::copy(ap.D.begin() + h->n, ap.D.end(), D_pred);
  if (has_duplicates) *has_duplicates = ap.dup ? 1 : 0;
  return 0;
}
EXPORT int gpb_hip_vecchia_predict_joint_factor(gpb_hip_vecchia_t* h, int32_t n_pred, const double* cp, int32_t mp, int layout_pred_first, int cond_all, int gauss,
                                                int cov, double var, double a, int32_t* m_used, int32_t* nn_all, double* A_all, double* D_all, double* u_all,
                                                int* has_duplicates) {
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  Appended ap;
  if (appended_factor(h, n_pred, cp, mp, cond_all != 0, true, layout_pred_first != 0, cov, var, a, gauss, &ap)) return -1;
  *m_used = ap.m;
  std::copy(ap.nn.begin(), ap.nn.end(), nn_all);
  std::copy(ap.A.begin(), ap.A.end(), A_all);
  std::copy(ap.D.begin(), ap.D.end(), D_all);
  std::vector<double> yall(ap.n_all, 0.);
  std::copy(h->y.begin(), h->y.end(), yall.begin() + (layout_pred_first ? n_pred : 0));
  orc_vecchia_By(ap.A.data(), ap.nn.data(), ap.n_all, ap.m, yall.data(), u_all);
  if (has_duplicates) *has_duplicates = ap.dup ? 1 : 0;
  return 0;
}
// dense symmetric positive definite solve (the exact-GP machinery on the device): M row-major lower triangle filled, x = M^-1 rhs, inv_sub = the
// (n - sub0) x (n - sub0) trailing block of M^-1
EXPORT int gpb_hip_dense_spd_solve(int32_t n, const double* M_host, const double* rhs, double* x, int32_t sub0, double* inv_sub) {
  std::vector<double> L((size_t)n * n, 0.);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) L[(size_t)i * n + j] = M_host[(size_t)i * n + j];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double acc = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) acc -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      if (i == j) { if (!(acc > 0.)) return fail("mock: the matrix is not positive definite"); L[(size_t)i * n + i] = std::sqrt(acc); }
      else L[(size_t)i * n + j] = acc / L[(size_t)j * n + j];
    }
  std::vector<double> b(rhs, rhs + n);
  chol_solve(L, n, b);
  std::copy(b.begin(), b.end(), x);
  if (inv_sub) {
    const int q = n - sub0;
    for (int c = 0; c < q; ++c) {
      std::vector<double> e(n, 0.); e[sub0 + c] = 1.;
      chol_solve(L, n, e);
      for (int r = 0; r < q; ++r) inv_sub[(size_t)r * q + c] = e[sub0 + r];
    }
  }
  return 0;
}

// ---- Laplace path ----
EXPORT int gpb_hip_vecchia_laplace_set_likelihood(gpb_hip_vecchia_t* h, int id) {
  if (id < 0 || id > 8) return fail("gpb_hip_vecchia_laplace_set_likelihood: id %d", id);
  if (h->link != id) { h->labels.clear(); h->grad_state = false; }
  h->link = id; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_data_map(gpb_hip_vecchia_t* h, const int32_t* re_ptr) {
  if (re_ptr) h->re_ptr.assign(re_ptr, re_ptr + h->n + 1); else h->re_ptr.clear();
  h->labels.clear(); h->fe.clear(); h->has_fe = false; h->has_mode = false; h->grad_state = false;
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_labels(gpb_hip_vecchia_t* h, const int32_t* y) { MOCK_TRACE("gpb_hip_vecchia_laplace_set_labels");
  const int nd = h->re_ptr.empty() ? h->n : h->re_ptr[h->n];
  h->labels.assign(y, y + nd); h->real_resp = false; h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_response_real(gpb_hip_vecchia_t* h, const double* y) {
  if (h->link != 3 && h->link != 5 && h->link != 6 && h->link != 7 && h->link != 8 && h->link > 1) return fail("gpb_hip_vecchia_laplace_set_response_real: a real-valued response is for gamma, beta and for proportions under the logit / probit links (likelihood id %d)", h->link);
  const int nd = h->re_ptr.empty() ? h->n : h->re_ptr[h->n];
  for (int i = 0; i < nd; ++i) {
    if (h->link == 3 || h->link == 7) { if (!(y[i] > 0.)) return fail("gamma / lognormal: the response must be > 0 (found %g at Vecchia position %d)", y[i], i); }
    else if (h->link == 5) { if (!(y[i] > 0. && y[i] < 1.)) return fail(" Must have 0 < y < 1 for the response variable ('y') for likelihood = 'beta', found %g ", y[i]); }
    else if (h->link == 6 || h->link == 8) { if (!std::isfinite(y[i])) return fail("t / gaussian_latent: the response must be finite"); }
    else if (!(y[i] >= 0. && y[i] <= 1.)) return fail(" Must have 0 <= y <= 1 for the response variable ('y') (found %g at Vecchia position %d)", y[i], i);
  }
  h->resp_real.assign(y, y + nd); h->labels.assign(nd, 0); h->real_resp = true; h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_binomial(gpb_hip_vecchia_t* h, int on) { h->binomial = on != 0; return 0; }
EXPORT int gpb_hip_vecchia_laplace_set_preconditioner(gpb_hip_vecchia_t* h, int type, int rank) {
  if (type < 0 || type > 3) return fail("preconditioner type %d is not on this path (0 = vadu, 1 = pivoted_cholesky, 2 = fitc, 3 = vecchia_response)", type);
  const int rk = rank > 0 ? rank : (type == 2 ? 200 : 50);
  if (type == 1 && rk > h->n) return fail("'fitc_piv_chol_preconditioner_rank' cannot be larger than the dimension of the mode (= number of unique locations) ");
  if (h->pc_type != type || h->pc_rank != rk) h->grad_state = false;
  h->pc_type = type; h->pc_rank = rk; return 0;
}
// Lloyd iterations of the inducing-point selection (kmeans_plusplus -> calculate_means, src/GPBoost/GP_utils.cpp:237-308) as gpb_hip.cpp: gpb_hip_kmeans_lloyd runs
// them: first mean at the smallest Euclidean distance, means updated per cluster over its rows in ascending order, until the means repeat or max_it
EXPORT int gpb_hip_kmeans_lloyd(int32_t n, int32_t d, const double* x, int32_t k, double* means_rowmajor, int32_t max_it, int32_t* iterations) {
  if (!x || !means_rowmajor || n < 1 || d < 1 || d > 3 || k < 1 || k > 256) return fail("gpb_hip_kmeans_lloyd: invalid argument (d <= 3, k <= 256)");
  std::vector<double> means(means_rowmajor, means_rowmajor + (size_t)k * d), old(means.size(), 0.), oldold(means.size(), 0.), mnew(means.size());
  std::vector<int> cl(n), cnt(k);
  int count = 0;
  do {
    oldold = old; old = means;
    for (int r = 0; r < n; ++r) {
      int best = 0; double bd = 0.;
      for (int j = 0; j < k; ++j) {
        double s2 = 0.; for (int c = 0; c < d; ++c) { const double t = x[(size_t)c * n + r] - means[(size_t)j * d + c]; s2 += t * t; }
        const double dd = std::sqrt(s2);
        if (j == 0 || dd < bd) { bd = dd; best = j; }
      }
      cl[r] = best;
    }
    std::fill(mnew.begin(), mnew.end(), 0.); std::fill(cnt.begin(), cnt.end(), 0);
    for (int r = 0; r < n; ++r) { for (int c = 0; c < d; ++c) mnew[(size_t)cl[r] * d + c] += x[(size_t)c * n + r]; cnt[cl[r]]++; }
    for (int j = 0; j < k; ++j) if (cnt[j] > 0) for (int c = 0; c < d; ++c) means[(size_t)j * d + c] = mnew[(size_t)j * d + c] / cnt[j];
    ++count;
  } while (means != old && means != oldold && count != max_it);
  std::copy(means.begin(), means.end(), means_rowmajor);
  if (iterations) *iterations = count;
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_inducing_points(gpb_hip_vecchia_t* h, int32_t k, const double* ip) {
  if (k < 1 || k >= h->n) return fail("Need to have less inducing points (currently fitc_piv_chol_preconditioner_rank = %d) than data points (%d) for cg_preconditioner_type = 'fitc' ", k, h->n);
  h->pc_ip.assign(ip, ip + (size_t)k * h->d); h->pc_nip = k; h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_weights(gpb_hip_vecchia_t* h, const double* w) {
  const int nd = h->re_ptr.empty() ? h->n : h->re_ptr[h->n];
  if (!w) { h->weights.clear(); orc_set_weights(nullptr); h->grad_state = false; return 0; }
  h->weights.assign(w, w + nd); orc_set_weights(h->weights.data());      // (one weighted model at a time: test infrastructure)
  h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_aux_pars(gpb_hip_vecchia_t* h, const double* aux, int32_t num_aux) {
  if (h->link < 3) return fail("gpb_hip_vecchia_laplace_set_aux_pars: likelihood id %d has no auxiliary parameters", h->link);
  if (num_aux != (h->link == 6 ? 2 : 1) || !(aux[0] > 0.)) return fail("The %s parameter is not > 0 (found %g)", h->link == 6 ? "scale" : "shape", aux[0]);
  if (h->link == 6 && !(aux[1] > 0.)) return fail("The df parameter is not > 0 (found %g)", aux[1]);
  if (aux[0] != h->aux) { h->aux = aux[0]; h->grad_state = false; }
  if (h->link == 6 && aux[1] != h->aux2) { h->aux2 = aux[1]; h->grad_state = false; }
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_get_aux_pars(gpb_hip_vecchia_t* h, double* aux_out, int32_t* num_aux) {
  *num_aux = h->link >= 3 ? (h->link == 6 ? 2 : 1) : 0;
  if (aux_out && *num_aux) aux_out[0] = h->aux;
  if (aux_out && *num_aux == 2) aux_out[1] = h->aux2;
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_grad_aux_current(gpb_hip_vecchia_t* h, double* out4) {
  if (h->link < 3) return fail("gpb_hip_vecchia_laplace_grad_aux_current: the likelihood has no auxiliary parameter");
  if (!h->grad_state) return fail("the gradient wrt the auxiliary parameter needs the state of gpb_hip_vecchia_laplace_grad_current");
  for (int k = 0; k < (h->link == 6 ? 8 : 4); ++k) out4[k] = h->aux_grad4[k];
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_set_fixed_effects(gpb_hip_vecchia_t* h, const double* fe) { MOCK_TRACE("gpb_hip_vecchia_laplace_set_fixed_effects");
  const int nd = h->re_ptr.empty() ? h->n : h->re_ptr[h->n];
  if (fe) { h->fe.assign(fe, fe + nd); h->has_fe = true; } else { h->fe.clear(); h->has_fe = false; }
  h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_eval(gpb_hip_vecchia_t* h, int cov, double var, double a, int nrv, int seed, int cg, int cgt, double cgd, double dcm, int reset,
                                        int /*keep*/, double* out9, double* mode_host) { MOCK_TRACE("gpb_hip_vecchia_laplace_eval"); MockTimer mock_timer_(7);
  return laplace_run(h, cov, var, a, nrv, seed, cg, cgt, cgd, dcm, reset, out9, mode_host);
}
EXPORT int gpb_hip_vecchia_laplace_logit(gpb_hip_vecchia_t* h, int cov, double var, double a, int nrv, int seed, int cg, int cgt, double cgd, double dcm, int reset,
                                         double* out9, double* mode_host) {
  return laplace_run(h, cov, var, a, nrv, seed, cg, cgt, cgd, dcm, reset, out9, mode_host);
}
EXPORT int gpb_hip_vecchia_laplace_grad_current(gpb_hip_vecchia_t* h, int, double, double* grad2, double*, double*) { MOCK_TRACE("gpb_hip_vecchia_laplace_grad_current"); MockTimer mock_timer_(7);
  if (h->pc_type == 3) return fail("Calculation of gradients is currently not correctly implemented for the '%s' preconditioner ", "vecchia_response");
  if (!h->grad_state) return fail("the gradient needs the state of an evaluation that kept it");
  grad2[0] = h->grad2[0]; grad2[1] = h->grad2[1]; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_reset_mode_to_previous(gpb_hip_vecchia_t* h) { MOCK_TRACE("gpb_hip_vecchia_laplace_reset_mode_to_previous");
  if (h->mode_prev.empty()) return fail("no mode has been found yet");
  h->mode = h->mode_prev; h->has_mode = true; h->grad_state = false; return 0;
}
EXPORT int gpb_hip_vecchia_laplace_grad_F_current(gpb_hip_vecchia_t* h, double* gF) { MOCK_TRACE("gpb_hip_vecchia_laplace_grad_F_current");
  if (!h->grad_state) return fail("the gradient wrt the fixed effects needs the state of gpb_hip_vecchia_laplace_grad_current");
  g_mock_aux = h->aux; g_mock_aux2 = h->aux2;
  const int n = h->n;
  const bool mapped = !h->re_ptr.empty();
  for (int i = 0; i < n; ++i) {
    const int d0 = mapped ? h->re_ptr[i] : i, d1 = mapped ? h->re_ptr[i + 1] : i + 1;
    double t3 = 0.;
    for (int k = d0; k < d1; ++k) { double f, inf, di; lik_terms(h->link, h->yv(k), h->mode[i] + (h->has_fe ? h->fe[k] : 0.), &f, &inf, &di); t3 += h->wv(k) * di; }
    const double diag = t3 == 0. ? 0. : h->dld[i] / t3;
    for (int k = d0; k < d1; ++k) {
      double f, inf, di; lik_terms(h->link, h->yv(k), h->mode[i] + (h->has_fe ? h->fe[k] : 0.), &f, &inf, &di);
      gF[k] = -(h->wv(k) * f) + 0.5 * (h->wv(k) * di) * diag - (h->wv(k) * inf) * h->sv[i];
    }
  }
  return 0;
}
// (full-scale Vecchia with a non-Gaussian likelihood is a device path only: tests/test_zz_vif_laplace_gpu.py; the oracle-backed shim says so)
EXPORT int gpb_hip_vecchia_vif_laplace_predict(gpb_hip_vecchia_t*, int32_t, const double*, int32_t, int, double, double, int, double, double*, double*, double*, int*, int*) {
  return fail("mock: full-scale Vecchia with a non-Gaussian likelihood is not restated in the oracle-backed shim (device tests cover it)");
}
EXPORT int gpb_hip_vecchia_laplace_predict(gpb_hip_vecchia_t* h, int32_t n_pred, const double* cp, int32_t mp, int cov, double var, double a, int, double,
                                           double* pred_mean, double* pred_var, double* pred_cov, int* has_duplicates, int* cg_iterations) {
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");       // the mode, handed over as the response
  Appended ap;
  if (appended_factor(h, n_pred, cp, mp, false, false, false, cov, var, a, 0, &ap)) return -1;
  const int n = h->n;
  for (int k = 0; k < n_pred; ++k) {
    double s = 0.;
    for (int j = 0; j < ap.m; ++j) { const int c = ap.nn[(size_t)(n + k) * ap.m + j]; if (c >= 0) s += ap.A[(size_t)(n + k) * ap.m + j] * h->y[c]; }
    pred_mean[k] = s;
  }
  if (has_duplicates) *has_duplicates = ap.dup ? 1 : 0;
  if (cg_iterations) *cg_iterations = 0;
  if (!pred_var && !pred_cov) return 0;
  if (!h->has_mode) return fail("predictive variances need the state of a likelihood evaluation (mode, information, factor)");
  std::vector<double> L;
  if (dense_M_chol(h, &L)) return -1;
  std::vector<std::vector<double>> X(n_pred);
  for (int k = 0; k < n_pred; ++k) {
    std::vector<double> b(n, 0.);
    for (int j = 0; j < ap.m; ++j) { const int c = ap.nn[(size_t)(n + k) * ap.m + j]; if (c >= 0) b[c] = -ap.A[(size_t)(n + k) * ap.m + j]; }
    X[k] = b;
    chol_solve(L, n, X[k]);
  }
  for (int r = 0; r < n_pred; ++r) {
    for (int k = (pred_cov ? 0 : r); k < (pred_cov ? n_pred : r + 1); ++k) {
      double q = 0.;
      for (int j = 0; j < ap.m; ++j) { const int c = ap.nn[(size_t)(n + r) * ap.m + j]; if (c >= 0) q += -ap.A[(size_t)(n + r) * ap.m + j] * X[k][c]; }
      if (pred_cov) pred_cov[(size_t)r * n_pred + k] = q + (r == k ? ap.D[n + r] : 0.);
      if (r == k && pred_var) pred_var[r] = ap.D[n + r] + q;
    }
  }
  if (pred_cov)      // symmetric to the last bit, as the device entry point returns it
    for (int r = 0; r < n_pred; ++r) for (int k = 0; k < r; ++k) { const double v = 0.5 * (pred_cov[(size_t)r * n_pred + k] + pred_cov[(size_t)k * n_pred + r]); pred_cov[(size_t)r * n_pred + k] = pred_cov[(size_t)k * n_pred + r] = v; }
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_mode_var(gpb_hip_vecchia_t* h, int, double, double* var_host, int* cg_iterations) {
  if (!h->has_mode) return fail("predictive variances need the state of a likelihood evaluation (mode, information, factor)");
  std::vector<double> L;
  if (dense_M_chol(h, &L)) return -1;
  for (int i = 0; i < h->n; ++i) { std::vector<double> e(h->n, 0.); e[i] = 1.; chol_solve(L, h->n, e); var_host[i] = e[i]; }
  if (cg_iterations) *cg_iterations = 0;
  return 0;
}

EXPORT int gpb_hip_vecchia_predict_cond_all_latent(gpb_hip_vecchia_t* h, int32_t n_pred, const double* cp, int32_t mp, int cov, double var, double a, int32_t* m_used,
                                                   int32_t* nn_pred, double* A_pred, double* D_pred, int* has_duplicates) {
  Appended ap;
  if (appended_factor(h, n_pred, cp, mp, true, false, false, cov, var, a, 0, &ap)) return -1;
  *m_used = ap.m;
  std::copy(ap.nn.begin() + (size_t)h->n * ap.m, ap.nn.end(), nn_pred);
  std::copy(ap.A.begin() + (size_t)h->n * ap.m, ap.A.end(), A_pred);
  std::copy(ap.D.begin() + h->n, ap.D.end(), D_pred);
  if (has_duplicates) *has_duplicates = ap.dup ? 1 : 0;
  return 0;
}
EXPORT int gpb_hip_vecchia_laplace_quad_forms(gpb_hip_vecchia_t* h, int32_t n_rows, int32_t mmax, const int32_t* cols, const double* vals, int, double, int want_cov,
                                              double* out, int* cg_iterations) {
  if (!h->has_mode) return fail("predictive variances need the state of a likelihood evaluation (mode, information, factor)");
  std::vector<double> L;
  if (dense_M_chol(h, &L)) return -1;
  const int n = h->n;
  std::vector<std::vector<double>> X(n_rows, std::vector<double>(n, 0.));
  for (int r = 0; r < n_rows; ++r) {
    for (int e = 0; e < mmax; ++e) { const int c = cols[(size_t)r * mmax + e]; if (c >= 0) X[r][c] = vals[(size_t)r * mmax + e]; }
    chol_solve(L, n, X[r]);
  }
  for (int r = 0; r < n_rows; ++r)
    for (int k = (want_cov ? 0 : r); k < (want_cov ? n_rows : r + 1); ++k) {
      double q = 0.;
      for (int e = 0; e < mmax; ++e) { const int c = cols[(size_t)r * mmax + e]; if (c >= 0) q += vals[(size_t)r * mmax + e] * X[k][c]; }
      if (want_cov) out[(size_t)r * n_rows + k] = q; else out[r] = q;
    }
  if (want_cov)
    for (int r = 0; r < n_rows; ++r) for (int k = 0; k < r; ++k) { const double v = 0.5 * (out[(size_t)r * n_rows + k] + out[(size_t)k * n_rows + r]); out[(size_t)r * n_rows + k] = out[(size_t)k * n_rows + r] = v; }
  if (cg_iterations) *cg_iterations = 0;
  return 0;
}

// ---- what this restatement leaves out ----
// route B on the CPU (tests/test_routeB_seams_cpu.py preloads this library under the route-B build of the reference): the neighbour search stays on
// the host (get = 0), the Newton leaf values come from the oracle
EXPORT int gpb_hip_route_b_set_device_search(int) { return 0; }
EXPORT int gpb_hip_route_b_get_device_search(void) { return 0; }
EXPORT int gpb_hip_vecchia_get_factor(gpb_hip_vecchia_t* h, double* A_host, double* D_host, double* u_host) { MockTimer mock_timer_(5);
  if (!h->has_factor) return fail("mock: factor missing");
  if (A_host) std::copy(h->A.begin(), h->A.end(), A_host);
  if (D_host) std::copy(h->D.begin(), h->D.end(), D_host);
  if (u_host) { if (!h->has_y) return fail("mock: response missing"); orc_vecchia_By(h->A.data(), h->nn.data(), h->n, h->m, h->y.data(), u_host); }
  return 0;
}
EXPORT int gpb_hip_vecchia_newton_leaf_values(gpb_hip_vecchia_t* h, const int32_t* leaf_index, int32_t num_leaves, double* leaf_values) { MockTimer mock_timer_(4);
  if (!h->has_factor || !h->has_y) return fail("mock: factor / response missing");
  std::vector<double> yaux(h->n);
  orc_vecchia_yaux(h->A.data(), h->D.data(), h->nn.data(), h->n, h->m, h->y.data(), yaux.data());
  if (orc_newton_leaf_values(h->A.data(), h->D.data(), h->nn.data(), h->n, h->m, yaux.data(), leaf_index, num_leaves, leaf_values)) return fail("mock: H' Psi^-1 H is not positive definite");
  return 0;
}
#define NOT_IN_MOCK(name) EXPORT int name() { return fail("mock shim (tests/mock_shim): " #name " is not restated on the CPU"); }
NOT_IN_MOCK(gpb_hip_exact_create) NOT_IN_MOCK(gpb_hip_exact_fisher_std_errors) NOT_IN_MOCK(gpb_hip_exact_free) NOT_IN_MOCK(gpb_hip_exact_grad_terms)
NOT_IN_MOCK(gpb_hip_exact_nll_terms) NOT_IN_MOCK(gpb_hip_exact_predict) NOT_IN_MOCK(gpb_hip_exact_psi_inv_diag) NOT_IN_MOCK(gpb_hip_exact_set_y)
NOT_IN_MOCK(gpb_hip_vecchia_fisher_std_errors) NOT_IN_MOCK(gpb_hip_vecchia_grad_terms_allreduce)
NOT_IN_MOCK(gpb_hip_vecchia_nll_terms_allreduce)
NOT_IN_MOCK(gpb_hip_vecchia_set_nugget_diag) NOT_IN_MOCK(gpb_hip_hist_register_host_buffers) NOT_IN_MOCK(gpb_hip_hist_unregister_host_buffers) NOT_IN_MOCK(gpb_hip_vecchia_timing) NOT_IN_MOCK(gpb_hip_mailbox_create) NOT_IN_MOCK(gpb_hip_vecchia_mailbox_attach) NOT_IN_MOCK(gpb_hip_vecchia_mailbox_info) NOT_IN_MOCK(gpb_hip_vecchia_mailbox_detach)
NOT_IN_MOCK(gpb_hip_vecchia_vif_factor) NOT_IN_MOCK(gpb_hip_vecchia_vif_grad_sums) NOT_IN_MOCK(gpb_hip_vecchia_vif_get_grad_factor) NOT_IN_MOCK(gpb_hip_vecchia_vif_predict_obs_only) NOT_IN_MOCK(gpb_hip_vecchia_vif_predict_cond_all) NOT_IN_MOCK(gpb_hip_vecchia_vif_set_inducing_points)
}  // extern "C"
