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
};

hipError_t launch_vecchia_nn(int d, const NNKernelArgs& a, hipStream_t st);

}  // namespace gpb
