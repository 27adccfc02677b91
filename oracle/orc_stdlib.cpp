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
