// gpboost_amd/csrc/dense_kernels.h -- launch interface of dense_kernels.hip (exact GP path)
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {
// P: row-major np x np (np = n rounded up to 64), lower triangle significant
// ld = leading dimension of P (np, or 2 np when Psi is the top-left block of the augmented matrix of the gradient)
hipError_t launch_dense_cov(int cov, bool d3, const double4* pts, int n, int np, int ld, double var, double a, double nugget,
                            const double* gtab, double* P, hipStream_t st);
hipError_t launch_dense_cholesky(double* P, int np, int* info, hipStream_t st, hipStream_t st2 = nullptr, hipEvent_t ev_panels = nullptr,
                                 hipEvent_t ev_rest = nullptr, int ncols = -1);
// work: np doubles of scratch -> one launch per 64-wide block step over many workgroups; nullptr -> the one-workgroup kernel
hipError_t launch_dense_solve(const double* P, int n, int np, int ld, const double* y, double* z, double* out, double* x_out,
                              hipStream_t st, double* work = nullptr);
// exact-GP gradient (re_model_template.h:2016-2040, CalcPsiInv :6586-6614): identity block of the augmented matrix, and the trace /
// quadratic-form sums over the lower tiles ([4][dense_grad_num_tiles(np)] partials, term-major)
hipError_t launch_dense_aug_identity(double* P2, int np, int ld, hipStream_t st);
int dense_grad_num_tiles(int np);
hipError_t launch_dense_grad(int cov, bool d3, const double4* pts, int n, int np, int ld, double var, double a, const double* gtab,
                             const double* P2, const double* ya, double* part, hipStream_t st);
}  // namespace gpb
