// gpboost_amd/csrc/dense_kernels.h -- launch interface of dense_kernels.hip (exact GP path)
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {
// P: row-major np x np (np = n rounded up to 64), lower triangle significant
hipError_t launch_dense_cov(int cov, bool d3, const double4* pts, int n, int np, double var, double a, double nugget,
                            const double* gtab, double* P, hipStream_t st);
hipError_t launch_dense_cholesky(double* P, int np, int* info, hipStream_t st, hipStream_t st2 = nullptr, hipEvent_t ev_panels = nullptr,
                                 hipEvent_t ev_rest = nullptr);
hipError_t launch_dense_solve(const double* P, int n, int np, const double* y, double* z, double* out, double* x_out,
                              hipStream_t st);
}  // namespace gpb
