// gpboost_amd/csrc/vif_kernels.h -- launch interface of vif_kernels.hip (full-scale Vecchia / VIF approximation)
#pragma once
#include <hip/hip_runtime.h>
#include "vecchia_kernels.h"

namespace gpb {

// All n x k matrices are row-major [n][kq], kq = vif_kq(k): k + 1 (the response column) rounded up to a multiple of 8 doubles.
inline int vif_kq(int k) { return (k + 1 + 7) / 8 * 8; }

// rows [i0, i1): C[i][j] = var k(|x_i - ip_j|) (j < k), C[i][k] = y_i, 0 beyond; dC (may be NULL): d/d log a.  ip is [k][3] (unused coordinates 0)
hipError_t launch_vif_crosscov(int cov, const double4* pts, const double* ip, int i0, int i1, int k, int kq, int d, double var, double a, double* C, double* dC,
                               hipStream_t st);
// Out (+)= In * M, In / Out [n][kq], M [kq][kq] row-major (zero outside its k x k block), all in device memory
hipError_t launch_vif_gemm(const double* In, const double* M, int n, int kq, double* Out, bool accumulate, hipStream_t st);
// rows [i0, i1): Q = B X (and Q2 = B X2 if X2 != NULL) for the stored factor
hipError_t launch_vif_spmm(const double* A, const int* nn, int i0, int i1, int m, int kq, const double* X, double* Q, const double* X2, double* Q2, hipStream_t st);
// G [kq][kq] = Q' D^-1 Q; part: vif_gram_part_doubles(n, kq) doubles of workspace
size_t vif_gram_part_doubles(int n, int kq);
hipError_t launch_vif_gram(const double* Q, const double* D, int n, int kq, double* part, double* G, hipStream_t st);
// v = D^-1 (u - Q w), z = y - C w   (u, y: column k of Q, C; w: kq doubles, zero from k on)
hipError_t launch_vif_vec(const double* Q, const double* C, const double* D, const double* w, int n, int k, int kq, double* v, double* z, hipStream_t st);
// residual-process Vecchia factor: A, D, u (MODE_FACTOR outputs) + partials [GPB_P_*][npts] (one row of three sums per point)
hipError_t launch_vif_resid_factor(int cov, const VecchiaKernelArgs& args, const double* V, int kip, int kq, hipStream_t st);
size_t vif_resid_lds_bytes(int m, int kq_grad);      // dynamic LDS of the per-point kernels (kq_grad > 0: the derivative kernel)
// derivative of the residual-process factor + the per-point sums of the gradient: partials [12][npts] = {S1..S6} x {variance, range}
struct VifGradLaunch {
  const double* V; const double* C; const double* dC; const double* Q; const double* QdC;
  const double* X1; const double* V1; const double* X2r; const double* Hm;     // [n][kq] each
  const double* w; const double* v; const double* z;                         // kq, n, n
  double* dA0; double* dA1; double* dD0; double* dD1;                         // optional outputs (all or none): [n][m], [n]
  double* partials;
};
#define GPB_VIF_GRAD_TERMS 12
hipError_t launch_vif_resid_grad(int cov, const VecchiaKernelArgs& args, const VifGradLaunch& L, int kip, int kq, hipStream_t st);

}  // namespace gpb
