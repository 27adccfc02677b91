// gpboost_amd/csrc/nn_kernels.h -- launch interface of nn_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

struct NNKernelArgs {
  const double4* sorted_rec;  // [n] {x0,x1,x2,coords_sum} in coordinate-sum order
  const int* sorted_idx;      // [n] sort_sum: original index of the k-th smallest coordinate sum
  const double4* pts;         // [n] records in Vecchia order (head rows' duplicate check)
  int* nn;                    // [n][m] out
  int* has_duplicates;        // out flag (atomicOr)
  int n, m;
  int start_at, end_search_at; // rows [start_at, n) are searched; candidates have index < i and <= end_search_at (Vecchia_utils.cpp:739-754)
  int pos0, pos1;             // positions (coordinate-sum order) this launch searches for: [pos0, pos1) -- multi-GPU: a block per rank
  const double* sorted_nd = nullptr;   // d > 3: [n][d + 1] {x_0 .. x_{d-1}, coords_sum} in coordinate-sum order (sorted_rec unused)
  const double* coords_nd = nullptr;   // d > 3: [n][d] coordinates in Vecchia order (head rows' duplicate check)
  const int* qorder;          // [nq] positions of the queries in the order the lanes take them (nn_query_order), or nullptr: pos0 + lane id
  int nq;
};

// The queries (positions in [pos0, pos1) whose row index i is > m and >= start_at) grouped by floor(4 log2 i), smallest indices first,
// ascending position inside a group.  A query's cost is set by its index -- among n points only i are eligible, so the scan of row i
// visits ~ sqrt(n / i) times the candidates of row n -- and a wavefront runs as long as its slowest lane: with lanes taken in plain
// position order (random indices in every wavefront) 3/4 of the lane-time waited for the wavefront's smallest index.  Inside a
// group the lanes are still neighbours in coordinate-sum order, so their scan windows keep sharing cache lines.
void nn_query_order(const int* sorted_idx, int pos0, int pos1, int m, int start_at, int* out, int* nq);

hipError_t launch_vecchia_nn(int d, const NNKernelArgs& a, hipStream_t st);
// k-means assignment step (GP_utils.cpp:237-280): x column-major [d][n], means row-major [k][d] (k <= 256, d <= 3) -> cl[n]
hipError_t launch_kmeans_assign(const double* x, const double* means, int n, int d, int k, int* cl, hipStream_t st);

}  // namespace gpb
