// gpboost_amd/csrc/pivchol_kernels.h -- launch interface of pivchol_kernels.hip: the "pivoted_cholesky" preconditioner of the
// Vecchia-Laplace iterative methods (reference: include/GPBoost/CG_utils.h:438-486, src/GPBoost/CG_utils.cpp:231-499,
// include/GPBoost/likelihoods.h:16277-16296, :16389-16465, :16554-16611).
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

// Layouts.  L: the rank-k factor [n][k] row-major, row = STORAGE slot of the point (the order of the Laplace vectors).  Block vectors:
// [chunk][row][nc] (nc = 1: one plain column per chunk; 4: the probe block).  Small operands: [chunk][k][nc].
int pc_parts(int n);                         // row slices of the tall-skinny reductions
// pivoted Cholesky of the NON-approximated covariance var * k(a * dist) (PivotedCholsekyFactorizationSigma): one argmax + one update launch per column
hipError_t pc_piv_init(int n, int k, double var, double* L, double* diag, int* pi, int* pos, int* done, hipStream_t st);
// out2 = { point chosen at step m (as a double), sum |diag| over the points not yet chosen }; swaps pi / pos as the reference does
hipError_t pc_piv_argmax(int n, int m, const double* diag, int* pi, int* pos, double* out2, hipStream_t st);
hipError_t pc_piv_update(const double4* pts, const int* sigma, int n, int k, int m, int p, int cov, int d3, double var, double a, double* L, double* diag, int* done,
                         hipStream_t st);
// G (lower triangle, packed by rows: e = p (p + 1) / 2 + q) = L^T diag(W) L;  part: pc_parts(n) * k (k + 1) / 2 doubles of scratch
hipError_t pc_gram(const double* L, const double* W, int n, int k, double* part, double* G, hipStream_t st);
// x2[chunk][k][nc] = M * (L^T (W .* X));  M k x k row-major (the inverse of I_k + L^T W L);  part: ncol * pc_parts(n) * k * nc doubles of scratch
hipError_t pc_ltwx(const double* L, const double* W, const double* M, const double* X, int n, int k, int ncol, int nc, double* part, double* x2, hipStream_t st);
// out = W .* (X - L x2) [mode 0: P^-1 X with x2 from pc_ltwx],  X - L x2 [mode 1: W^-1 P^-1 X],  L x2 + X ./ sqrt(W) [mode 2: probe vectors with x2 = the k x t normals]
// -W .* (L x2) [mode 3],  X + W .* (L x2) [mode 4]  (full-scale Vecchia: the Woodbury part of Sigma^-1, the vifdu preconditioner)
hipError_t pc_combine(const double* L, const double* W, const double* X, const double* x2, int n, int k, int ncol, int nc, int mode, double* out, hipStream_t st);
// out = x .* w (inv = 0) or x ./ w (inv = 1); v += h ./ w
hipError_t pc_rowscale(const double* x, const double* w, int n, int ncol, int nc, int inv, double* out, hipStream_t st);
hipError_t pc_add_div(double* v, const double* h, const double* w, int n, int ncol, int nc, hipStream_t st);
// ---- preconditioner "fitc": P = diag(W^-1 + Sigma_m[0][0] - ||V_i||^2) + C Sigma_m^-1 C' -- the same kernels with L := C (cross-covariance with the inducing points),
// the diagonal of the preconditioner's inverse wp := D^-1 in place of W, and M := (Sigma_m + C' D^-1 C)^-1 ----
// rows of src [n][ld] (Vecchia order) -> dst [n][k] (row sigma[i]); vnorm2 (optional): sum_q src[i][q]^2 into slot sigma[i]
hipError_t pc_pack_rows(const double* src, int ld, const int* sigma, int n, int k, double* dst, double* vnorm2, hipStream_t st);
// wp[i] = 1 / (1 / W[i] + sm00 - vnorm2[i])      (likelihoods.h:16302-16308)
hipError_t pc_fitc_diag(const double* W, const double* vnorm2, double sm00, int n, double* wp, hipStream_t st);
// ---- preconditioner "vecchia_response": P^-1 = B_p' D_p^-1 B_p, the Vecchia factor of W^-1 + Sigma (likelihoods.h:16315-16323) ----
// nug[i] (Vecchia order) = 1 / W[sigma[i]] + jit: the diagonal additions of the factor launch;  D2s[sigma[i]] = D2[i] - jit and its square root (storage order)
hipError_t pc_vr_nugget(const double* W, const int* sigma, int n, double jit, double* nug, hipStream_t st);
hipError_t pc_vr_diag(const double* D2, const int* sigma, int n, double jit, double* D2s, double* sqrtD2s, hipStream_t st);
// out2 = { sum_i sdiag_i W_i, 0 } (wp == nullptr: pivoted_cholesky) or { sum_i sdiag_i wp_i^2 / W_i, sum_i wp_i / W_i } (fitc), sdiag_i = L[i, :] M L[i, :]' -- the deterministic
// traces of CalcLogDetStochDerivAuxParVecchia (likelihoods.h:16810-16833) for dW = f W (a likelihood whose information is constant in the location parameter)
hipError_t pc_aux_sums(const double* L, const double* M, const double* W, const double* wp, int n, int k, double* out2, hipStream_t st);
hipError_t pc_wmax(const double* w, int n, double* out1, hipStream_t st);      // out1[0] = max_i w[i] (NaN propagates)
// d log|Sigma W + I| / d mode_i, pivoted_cholesky branch of CalcLogDetStochDerivModeVecchia (likelihoods.h:16554-16611): U = (W^-1 + Sigma)^-1 Z,
// WIPIZ = W^-1 P^-1 Z, row-wise optimal c (CalcOptimalCVectorized), deterministic part diag(L M L^T) dW - dW / W
// (wp != nullptr: the fitc branch, :16612-16633 -- deterministic part diag(C M C') wp (W^-1 dW W^-1 wp) - W^-1 dW W^-1 wp)
hipError_t pc_row_stats(const double* U, const double* WIPIZ, const double* L, const double* M, const double* W, const double* dW3, int n, int k, int t, int nc,
                        double* dld, hipStream_t st, const double* wp = nullptr, int det_centre = 0, double* sdiag_scratch = nullptr);      // sdiag_scratch: n doubles; then diag(L M L') comes from the tiled kernel (k <= 256)
// out[i] = c0 - 2 L_i' M1 L2_i + L_i' M2 L_i (the derivative of the fitc preconditioner's diagonal, likelihoods.h:5478-5486); out = a .* b (.* c)
hipError_t pc_row_quad(const double* L, const double* L2, const double* M1, const double* M2, int n, int k, double c0, double* out, hipStream_t st);
hipError_t pc_mul3(const double* a, const double* b, const double* c, int n, double* out, hipStream_t st);
// columns [col0, col0 + cnt) of L [n][k], times w[i], as a block vector of ncol chunks x nc columns (zero beyond cnt): right-hand sides W C of the VIF-Laplace prediction
hipError_t pc_cols_to_block(const double* L, const double* w, int n, int k, int col0, int cnt, int ncol, int nc, double* out, hipStream_t st);

}  // namespace gpb
