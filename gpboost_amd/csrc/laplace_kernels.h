// gpboost_amd/csrc/laplace_kernels.h -- launch interface of laplace_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

namespace gpb {

struct alignas(16) LapEnt { double val; int src; int pad; };   // one matrix entry: coefficient, source row (storage index)
struct LapTri {            // one level-scheduled triangular solve; matrix stored in LEVEL ORDER as slots (position q)
  const int* ptr;          // [nlev + 1] level boundaries (slot positions)
  const int* lsplit;       // [nlev + 1] 1 if the level holds a row that is split over several slots
  const int4* meta;        // [nslots] {row (storage index) of the slot's first... or -1, #slots of the row if first slot else 0,
                           //           begin, end of the slot's overflow entries}
  const LapEnt* hent;      // [nslots * 32] head entries (padding: source 0, coefficient 0); .val refreshed per evaluation
  const LapEnt* oent;      // [max(novf, 1)] overflow entries 33.. of a slot (a slot holds <= 64 unless its row has > 4096)
  int nlev, nslots, has_ovf;
};
constexpr int kTriThreads = 512;        // workgroup of the level-scheduled solves
constexpr int kTriRowsPerRound = 64;    // rows it handles per round (32 groups of 16 lanes x 2)
struct LapSeg { int L0, L1, nsplit, nrounds; };   // levels [L0, L1) in one launch, nrounds rounds per workgroup; nsplit > 1: one wide level, one round per workgroup
// Dense block of a solve: the rows of a contiguous range of NARROW levels (the head of the forward solve -- the first points of the
// ordering, which all depend on each other -- and the tail of the backward solve).  A run of such levels costs one dependent L2 round
// trip per level (~3 us, ~270 levels at n = 1e5); instead the inverse of the block's unit lower-triangular matrix I - A_blk is formed
// once per evaluation (row k = e_k + sum_j A_kj row j: 30 row updates, level by level) and a solve applies it as one dense product:
//     x_blk = inv * (rhs_blk [.* rdw] + A_(blk, outside) x_outside).
struct LapDense {
  int K, ld;                   // rows of the block (0: no block), leading dimension of inv
  const int* rows;             // [K] storage index of block row k (level order)
  const int* iptr; const int* icol; const int* ipos;   // entries inside the block: column = block-local index (< k), position in A
  const int* optr; const int* osrc; const int* opos;   // entries whose source row lies outside the block: storage index, position in A
  double* inv;                 // [K * ld] lower triangle of (I - A_blk)^-1 (the rest is never read), refreshed per evaluation
  double* tbuf;                // [K * kDenseCols] right-hand sides of the block, [k][column]; then kDenseSplit partial products of the same size
  const int* blev; int nblev;  // HOST array [nblev + 1]: block-local level boundaries (the order inv is built in)
};
constexpr int kDenseCols = 64;            // columns the dense block handles per pass
constexpr int kDenseSplit = 8;            // parts the column range of the block product is cut into (probe block)
struct LapLevels {                        // fwd: (D^-1 + W) B z = t;  bwd: B^T t = r;  launch segments of each (host arrays)
  LapTri fwd, bwd;
  const LapSeg* fseg; int n_fseg;
  const LapSeg* bseg; int n_bseg;
  LapDense fdense, bdense;                // head block of the forward solve (before fseg), tail block of the backward solve (after bseg)
  const double* A;                        // this evaluation's coefficients, Vecchia order [n][m] (the blocks' entries point into it)
  // barrier-free solves (lap_sptrsv_sf_kernel): one launch per triangular solve for all levels of fseg / bseg
  int syncfree = 0;                       // 1: use them
  const int* fwd_ptr_host = nullptr;      // HOST copies of fwd.ptr / bwd.ptr (slot range of a run of levels)
  const int* bwd_ptr_host = nullptr;
  int* err = nullptr;                     // device word: set when a bounded spin ran out (a dependency never arrived)
};
struct CgScalars {         // per-column CG scalars on the device; part / part2: lap_cg_parts(n) partial dot products per column
  double* a; double* a_old; double* b; double* rz_old; double* rnorm; double* Td; double* Ts; double* part; double* part2;
};
int lap_cg_parts(int n);

// the response as the likelihood kernels see it: int labels / counts (yi) or -- gamma -- real values (yd), and the likelihood's auxiliary parameter
// (shape of gamma / negative_binomial; unused by the others)
struct LikResp { const int* yi; const double* yd; double aux; const double* w = nullptr; double aux2 = 2.0; };     // aux2: the second auxiliary parameter (t: degrees of freedom)     // w: sample weights per datum (storage order of the response), or nullptr
hipError_t lap_newton_setup(int link, const double* mode, const LikResp& y, const double* fe, const double* D, int n, double* W, double* rhs, double* dw, double* rdw, hipStream_t st,
                            const int* dptr = nullptr);   // dptr (n + 1): repeated locations -- row i sums over the data y[dptr[i] .. dptr[i + 1]) (also below)
// ncol = number of column chunks, nc = columns per chunk (1: plain columns; 4: block vectors stored [chunk][row][4])
hipError_t lap_apply(const LapLevels& lv, int n, const double* D, const double* W, const double* h, double* v, double* tmp, int ncol, int nc, hipStream_t st);
hipError_t lap_B(const LapLevels& lv, int n, const double* x, double* out, int ncol, int nc, hipStream_t st);
hipError_t lap_Bt(const LapLevels& lv, int n, const double* x, double* out, int ncol, int nc, hipStream_t st);
hipError_t lap_scatter(const double* in, const int* sigma, int n, double* out, hipStream_t st);
hipError_t lap_objective(int link, const double* x, const LikResp& y, const double* fe, const double* Bx, const double* D, int n, double* out2, hipStream_t st,
                         const int* dptr = nullptr);
hipError_t lap_vadu(const LapLevels& lv, int n, const double* rdw, const double* r, double* z, double* t, int ncol, int nc, hipStream_t st);
hipError_t lap_fwd_solve(const LapLevels& lv, int n, const double* scale, const double* rhs, double* z, int ncol, int nc, hipStream_t st);   // z = B^-1 (scale .* rhs)
hipError_t lap_dense_build(const LapDense& d, const double* A, hipStream_t st);    // inv of the block from this evaluation's A (Vecchia order [n][m])
hipError_t lap_permute_factor(const double* A, const int* hpos, const int* opos, size_t nh, size_t novf, LapEnt* hent, LapEnt* oent, hipStream_t st);
hipError_t lap_cg_alpha(const double* r, const double* z, const double* h, const double* v, int n, int ncol, int nc, const CgScalars& sc, hipStream_t st);
hipError_t lap_cg_update(double* u, double* r, const double* h, const double* v, int n, int ncol, int nc, const CgScalars& sc, hipStream_t st);
hipError_t lap_cg_beta(const double* r, const double* z, double* h, int n, int ncol, int nc, const CgScalars& sc, int j, int p_max, hipStream_t st);
hipError_t lap_lincomb(double* out, const double* x, const double* y, double cx, double cy, int n, hipStream_t st);
hipError_t lap_scale_probes(const double* rv, const double* dw, int n, int ncol, int nc, double* out, hipStream_t st);
hipError_t lap_logsums(const double* D, const double* dw, int n, double* out2, hipStream_t st);
hipError_t lap_dot(const double* x, const double* y, int n, double* out2, hipStream_t st);
// ---- gradient of the approximate marginal likelihood (block vectors: ncol chunks of nc columns, as above) ----
hipError_t lap_third_deriv(int link, const double* mode, const LikResp& y, const double* fe, int n, double* dW3, hipStream_t st, const int* dptr = nullptr);
hipError_t lap_grad_F(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* sv, int n, double* out, hipStream_t st);
hipError_t lap_grad_F_map(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* dW3, const double* sv, int n,
                          const int* dptr, double* out, hipStream_t st);      // repeated locations: per datum, storage order of the data
// link 3 / 4: the three data sums of the gradient wrt log(aux) (laplace_kernels.hip: lik_aux_grad_kernel); dptr may be NULL (one datum per row)
hipError_t lap_aux_grad(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* dW3, const double* sv, int n,
                        const int* dptr, double* out3, hipStream_t st);
hipError_t lap_range_deriv(const double4* pts, const int* nn, const double* A, int n, int m, int cov, int d3, double var, double a, double* dA, double* dD, hipStream_t st);
hipError_t lap_factor_deriv(const double4* pts, const int* nn, const double* A, int n, int m, int cov, int d3, double var, double a, double diag_nn, double nug,
                            int which, double* dA, double* dD, hipStream_t st);   // which: 0 = d/dlog(range), 1 = d/dlog(variance ratio) with a nugget
hipError_t lap_fisher_mid(const double* P, const double* T, const double* D, const double* dD, int n, int ncol, int nc, double* H, hipStream_t st);
hipError_t lap_mul(const LapTri& T, int n, const double* x, double* out, int ncol, int nc, hipStream_t st);       // plain product with T's entries
hipError_t lap_row_stats(const double* U, const double* PIZ, const double* BPIZ, const double* dW3, const double* rdw, int n, int t, int nc, double* dld, hipStream_t st);
hipError_t lap_coldots(const double* X, const double* Y, const double* T, int n, int ncol, int nc, double* out, hipStream_t st);
hipError_t lap_deriv_mid(const double* R, const double* Z, const double* D, const double* dD, const double* W, int n, int ncol, int nc, int sel, double* H, double* V, hipStream_t st);
// ---- predictive (co)variances at new locations: right-hand sides Bpo' e_p of a block of prediction points and b_r' X for its solution ----
hipError_t lap_pred_rhs(const int* nn_p, const double* A_p, const int* sigma, int n, int m, int p0, int cnt, int ncol, int nc, double* out, hipStream_t st);
hipError_t lap_pred_quad(const int* nn_p, const double* A_p, const int* sigma, const double* X, int n, int m, int row0, int n_rows, int cnt, int nc,
                         int diag_only, double* out, hipStream_t st);
hipError_t lap_sums3(const double* rdw, const double* D, const double* dD, int n, double* out3, hipStream_t st);

}  // namespace gpb
