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
