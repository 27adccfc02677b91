// gpboost_amd/csrc/leaf_kernels.h -- launch interface of leaf_kernels.hip (Newton update of the leaf values, row a9)
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

int leaf_num_workgroups(int n);
// LP in {16, 32, 64} >= number of leaves; partials: leaf_num_workgroups(n) * (LP*LP + LP) doubles; out: LP*LP + LP doubles =
// { M row-major LP x LP, rhs[LP] }
hipError_t launch_leaf_gram(int LP, const double* A, const double* D, const int* nn, const double* yaux, const int* leaf, int n, int m,
                            double* partials, double* out, hipStream_t st);

}  // namespace gpb
