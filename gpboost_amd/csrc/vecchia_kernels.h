// gpboost_amd/csrc/vecchia_kernels.h -- host-visible launch interface of vecchia_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

enum : int { MODE_NLL = 0, MODE_FACTOR = 1, MODE_GRAD = 2 };

// per-workgroup partial sums (fixed layout, reduced by reduce_partials_kernel)
enum : int {
  GPB_P_LOGDET = 0,   // sum log D_i
  GPB_P_QUAD = 1,     // sum u_i^2 / D_i
  GPB_P_BAD = 2,      // #(D_i <= 0)
  GPB_P_G1_VAR = 3,   // sum (dB_var y)_i u'_i - 0.5 u'_i^2 dD_var,i
  GPB_P_G2_VAR = 4,   // sum 0.5 dD_var,i / D_i
  GPB_P_G1_RNG = 5,
  GPB_P_G2_RNG = 6,
};
#define GPB_NUM_PARTIALS 7

// Padded neighbour counts the kernels are instantiated for (m is rounded up to the next one).
#ifndef GPB_MT_LIST
#define GPB_MT_LIST 10, 20, 30, 40, 50, 62
#define GPB_MT_CASES GPB_CASE(10) GPB_CASE(20) GPB_CASE(30) GPB_CASE(40) GPB_CASE(50) GPB_CASE(62)
#endif
#define GPB_MAX_NEIGHBORS 62        // the register-resident point kernel (vecchia_kernels.hip)
#define GPB_MAX_NEIGHBORS_BIG 126   // the LDS-resident generality kernel (vecchia_big_kernels.hip): 62 < m <= 126
#ifndef GPB_EXP_TAB_SIZE
#define GPB_EXP_TAB_SIZE 256   // entries of the 2^(j/256) table (dev_common.h: exp_of_scaled)
#endif

struct VecchiaKernelArgs {
  const double4* pts;      // [n] {x0, x1, x2, y} in Vecchia order
  const int* nn;           // [n][m] neighbour indices, -1 padded
  const double* exp_tab;   // [GPB_EXP_TAB_SIZE] 2^(j/GPB_EXP_TAB_SIZE)
  double* partials;        // [GPB_NUM_PARTIALS][nblocks]  (term-major)
  double* A;               // MODE_FACTOR: [n][m]
  double* D;               // MODE_FACTOR: [n]
  double* u;               // MODE_FACTOR: [n]  (B y)
  int m;                   // actual number of neighbours per row
  int i_begin, i_end;      // points [i_begin, i_end) of the ordering handled by this launch (shard)
  double var;              // sigma1^2 / sigma^2
  double a;                // transformed range
  double diag_nn;          // diagonal of C_nn:   Gaussian var + 1;          else var * (1 + 1e-10)
  double diag_i;           // first summand of D: Gaussian var + 1;          else var
  double nugget;           // Gaussian 1, else 0
  double diag_mult = 1.0;  // vif_kernels.hip only: the neighbours' diagonal is (diag_nn - |V_a|^2) * diag_mult -- 1 + 1e-10 for the latent residual process (Vecchia_utils.cpp:1489-1500, :1608)
  const double* nug = nullptr;   // sample weights (Gaussian only): [n] observation-specific nugget 1 / w_i on the transformed scale, Vecchia order;
                                 // then the diagonals are var + nug[.] instead of diag_nn / diag_i
  const double* coords_nd = nullptr;   // d > 3 (generality path): [n][dim] coordinates in Vecchia order; pts then only carries the response (w)
  int dim = 0;
  // vecchia_point_kernel (persistent worker workgroups + one finisher workgroup that adds up the workers' sums inside the launch; not
  // used by the generality kernel).  `partials` then holds [GPB_NUM_PARTIALS][workers] 8-byte slots that are all-ones ("empty") between
  // launches: the host fills the buffer with 0xFF once, the finisher restores what it consumed.
  int ngroups = 0;                     // groups of 16 points of this launch = ceil((i_end - i_begin) / 16)
  double* out = nullptr;               // [GPB_NUM_PARTIALS] the launch's sums in GPB_P_* order
  double* out_user = nullptr;          // optional: the same in the caller-facing order {quad, logdet, bad, g1v, g2v, g1r, g2r}
  double* out_host = nullptr;          // optional: pinned + coherent host copy (GPB_P_* order), polled by the host
};
#define GPB_MAX_DIM 10

int vecchia_padded_m(int m);
hipError_t launch_vecchia_point_kernel(int mode, int cov, bool d3, const VecchiaKernelArgs& args, hipStream_t st);
// m > GPB_MAX_NEIGHBORS: one workgroup and one row of `partials` per point (vecchia_big_kernels.hip)
// (dk = 2: d <= 2, 3: d = 3 -- coordinates from the point records; 0: 3 < d <= GPB_MAX_DIM -- coordinates from args.coords_nd)
hipError_t launch_vecchia_point_big(int mode, int cov, int dk, const VecchiaKernelArgs& args, hipStream_t st);
hipError_t launch_reduce_partials(const double* partials, int nblocks, int nterms, double* out, double* out_user,
                                  hipStream_t st, double* out_host = nullptr);
hipError_t launch_publish(const double* src, double* dst_host, int n, hipStream_t st);
hipError_t launch_pack_y(double4* pts, const double* y, int n, hipStream_t st);
hipError_t launch_scatter_pts(double4* pts, const int* rank, int n, hipStream_t st);                      // pts[n + rank[k]] = pts[k]  (pts holds 2 n records)
hipError_t launch_remap_nn(const int* nn, const int* rank, size_t cnt, int n, int* nn2, hipStream_t st);  // nn2 = n + rank[nn]
hipError_t launch_By(const double* A, const int* nn, int n, int m, const double* y, double* u, hipStream_t st);
hipError_t launch_By_pts(const double* A, const int* nn, const double4* pts, int m, int i0, int i1, double* u, hipStream_t st);   // u = B y, y = pts[.].w
hipError_t launch_Bt(const double* A, const int* t_ptr, const int* t_pos, int n, int m, int i0, int i1, const double* v,
                     double* w, hipStream_t st);
hipError_t launch_scale_by_Dinv(const double* u, const double* D, int n, int i0, int i1, double* v, hipStream_t st);
// out_j = (B^T D^-1 B)_jj from a stored factor and the transposed neighbour index
hipError_t launch_BtDinvB_diag(const double* A, const double* D, const int* t_ptr, const int* t_pos, int n, int m, double* out, hipStream_t st);
hipError_t launch_gram(const double* U, const double* D, int n, int q, double* G, hipStream_t st);        // G = U diag(1/D) U^T, U: [q][n]
hipError_t launch_resid(double4* pts, const double* y0, const double* X, const double* beta, int n, int p, hipStream_t st);
hipError_t launch_dpp_selftest(const double* in, double* out, hipStream_t st);

}  // namespace gpb
