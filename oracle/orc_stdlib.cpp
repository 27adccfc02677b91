/*
 * oracle/orc_stdlib.cpp -- TEST INFRASTRUCTURE ONLY (companion of gpb_oracle.c).
 *
 * The two places where the reference's *result* depends on libstdc++
 * implementation details rather than on arithmetic; they are restated by calling
 * the same library routines, not re-implemented:
 *
 *  orc_sort_indices  include/GPBoost/utils.h:230-238 (SortIndeces: std::iota +
 *                    std::sort with comparator v[i1] < v[i2]; introsort is not
 *                    stable, so ties in the coordinate sums resolve in a
 *                    libstdc++-specific way).
 *  orc_shuffle       src/GPBoost/Vecchia_utils.cpp:1129-1131 (vecchia_ordering
 *                    == "random": std::shuffle(data_indices, RNG_t(seed)),
 *                    RNG_t = std::mt19937, include/GPBoost/type_defs.h:52,
 *                    seeded at include/GPBoost/re_model_template.h:161).
 */
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>
#include <cstdint>

extern "C" {

__attribute__((visibility("default")))
void orc_sort_indices(const double* v, int n, int* idx) {
  std::vector<int> t(n);
  std::iota(t.begin(), t.end(), 0);
  std::sort(t.begin(), t.end(), [v](int i1, int i2) { return v[i1] < v[i2]; });
  std::copy(t.begin(), t.end(), idx);
}

__attribute__((visibility("default")))
void orc_shuffle(int n, int seed, int* idx) {
  std::vector<int> t(n);
  std::iota(t.begin(), t.end(), 0);
  std::mt19937 rng(seed);
  std::shuffle(t.begin(), t.end(), rng);
  std::copy(t.begin(), t.end(), idx);
}

}  // extern "C"

/* GenRandVecNormalParallel -- src/GPBoost/CG_utils.cpp:978-994: column col_i of the n x t matrix is drawn from
 * std::normal_distribution<double>(0,1) on a std::mt19937 seeded with std::seed_seq{base_seed, run_id lo, run_id hi, col_i}.
 * out is column-major n x t. */
extern "C" __attribute__((visibility("default")))
void orc_gen_rand_normal(int base_seed, unsigned long long run_id, int n, int t, double* out) {
  const uint32_t b32 = static_cast<uint32_t>(base_seed);
  for (int col_i = 0; col_i < t; ++col_i) {
    std::normal_distribution<double> ndist(0.0, 1.0);
    std::seed_seq seq{ b32, static_cast<uint32_t>(run_id), static_cast<uint32_t>(run_id >> 32), static_cast<uint32_t>(col_i) };
    std::mt19937 generator(seq);
    for (int row_i = 0; row_i < n; ++row_i) out[(size_t)col_i * n + row_i] = ndist(generator);
  }
}

/* Inducing points of the full-scale-Vecchia ("VIF") approximation: the ONE generator of the model (RNG_t(seed),
 * include/GPBoost/re_model_template.h:161) first shuffles the data indices when vecchia_ordering == "random" (:351-355: for
 * gp_approx == "full_scale_vecchia" the shuffle happens before CreateREComponentsFITC_FSA) and then drives kmeans++ on the coordinates
 * in that order (CreateREComponentsFITC_FSA, :7714-7720 -> src/GPBoost/GP_utils.cpp: random_plusplus :208-235 -- a draw from
 * std::discrete_distribution weighted by the PLAIN distance to the closest chosen mean --, calculate_means :237-280, kmeans_plusplus
 * :282-308: Lloyd iterations until the means repeat, at most 1000).  libstdc++-specific like the shuffle: restated by calling the same
 * library routines.  coords: column-major n x d (data order); perm_out: Vecchia position -> data index; ip_out: column-major k x d. */
static int orc_kmeanspp(std::mt19937& rng, const std::vector<double>& x, int n, int d, int k, int max_it, double* ip_out) {
  auto dist = [&](const double* a, const double* b) { double s = 0.; for (int c = 0; c < d; ++c) { const double t = a[c] - b[c]; s += t * t; } return std::sqrt(s); };
  std::vector<double> means((size_t)k * d, 0.), w(n, 1.0);
  for (int i = 0; i < k; ++i) {                         // random_plusplus
    if (i == 1) for (auto& v : w) v *= -1.;
    if (i > 0) for (int r = 0; r < n; ++r) { const double dd = dist(&x[(size_t)r * d], &means[(size_t)(i - 1) * d]); if (w[r] > dd || w[r] < 0) w[r] = dd; }
    double sum = 0.; for (double v : w) sum += v;
    int v;
    if (sum > 0.) v = std::discrete_distribution<>(w.data(), w.data() + n)(rng);
    else v = std::uniform_int_distribution<>(0, n - 1)(rng);
    for (int c = 0; c < d; ++c) means[(size_t)i * d + c] = x[(size_t)v * d + c];
  }
  std::vector<double> old(means.size(), 0.), oldold(means.size(), 0.), mnew(means.size());
  std::vector<int> cl(n);
  int count = 0;
  do {                                                  // kmeans_plusplus
    oldold = old; old = means;
    std::fill(mnew.begin(), mnew.end(), 0.);
    for (int r = 0; r < n; ++r) {                       // calculate_means: nearest mean (first of equals)
      int best = 0; double bd = dist(&x[(size_t)r * d], &means[0]);
      for (int j = 1; j < k; ++j) { const double dd = dist(&x[(size_t)r * d], &means[(size_t)j * d]); if (dd < bd) { bd = dd; best = j; } }
      cl[r] = best;
    }
    std::vector<int> cnt(k, 0);
    for (int r = 0; r < n; ++r) { for (int c = 0; c < d; ++c) mnew[(size_t)cl[r] * d + c] += x[(size_t)r * d + c]; cnt[cl[r]]++; }   // (per mean: its rows in ascending order, as the reference's loop over j)
    for (int j = 0; j < k; ++j) if (cnt[j] > 0) for (int c = 0; c < d; ++c) means[(size_t)j * d + c] = mnew[(size_t)j * d + c] / cnt[j];
    ++count;
  } while (means != old && means != oldold && count != max_it);
  for (int j = 0; j < k; ++j) for (int c = 0; c < d; ++c) ip_out[(size_t)c * k + j] = means[(size_t)j * d + c];
  return count;
}
/* k2 > 0: a SECOND kmeans++ run with k2 means from the same generator -- the inducing points of the "fitc" preconditioner of a full-scale Vecchia model with a
 * non-Gaussian likelihood (Calc_FITC_Preconditioner_Vecchia, re_model_template.h:9502-9593, at the first covariance factor: ind_points_determined_for_preconditioner_
 * is never set for isotropic kernels, so the preconditioner gets its own points whatever fitc_piv_chol_preconditioner_rank is). */
extern "C" __attribute__((visibility("default")))
int orc_vif_setup2(int n, int d, const double* coords, int seed, int do_shuffle, int k, int max_it, int* perm_out, double* ip_out, int k2, double* ip2_out) {
  std::mt19937 rng(seed);
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  if (do_shuffle) std::shuffle(perm.begin(), perm.end(), rng);
  std::copy(perm.begin(), perm.end(), perm_out);
  if (k > n || k2 > n) return -1;
  std::vector<double> x((size_t)n * d);                 // row-major, Vecchia order
  for (int i = 0; i < n; ++i) for (int c = 0; c < d; ++c) x[(size_t)i * d + c] = coords[(size_t)c * n + perm[i]];
  const int count = orc_kmeanspp(rng, x, n, d, k, max_it, ip_out);
  if (k2 > 0) orc_kmeanspp(rng, x, n, d, k2, max_it, ip2_out);
  return count;
}
extern "C" __attribute__((visibility("default")))
int orc_vif_setup(int n, int d, const double* coords, int seed, int do_shuffle, int k, int max_it, int* perm_out, double* ip_out) {
  return orc_vif_setup2(n, d, coords, seed, do_shuffle, k, max_it, perm_out, ip_out, 0, nullptr);
}
