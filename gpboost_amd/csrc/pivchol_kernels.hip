// gpboost_amd/csrc/pivchol_kernels.hip -- device side of cg_preconditioner_type = "pivoted_cholesky" for the Vecchia-Laplace approximation
// (SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, include/GPBoost/re_model_template.h:5906).  The linear systems are solved in the form
//     (W^-1 + Sigma) u' = Sigma rhs,  u = W^-1 u',   Sigma = B^-1 D B^-T   (CG_utils.cpp:231-343, :345-499)
// preconditioned with P = W^-1 + L_k L_k^T, L_k the rank-k pivoted Cholesky factor of the non-approximated covariance matrix
// (CG_utils.h:438-486): P^-1 r = W r - W L_k (I_k + L_k^T W L_k)^-1 L_k^T W r.  The k x k matrix M = (I_k + L_k^T W L_k)^-1 is formed on the host from
// the Gram matrix these kernels reduce (k = 50 by default: a 20 KB problem), everything of size n stays on the device:
//   * tall-skinny reductions  L^T (W .* X)  with fixed-order partial sums (bit-reproducible, no atomics),
//   * rank-k updates          X - L x2      one row of L per thread, the small operand in LDS,
//   * the factorisation itself: per column one argmax launch (first maximum in the reference's pivot order) and one update launch.
// Sigma h is the existing pair of barrier-free triangular solves (lap_vadu with the diagonal D).
#include "pivchol_kernels.h"

#include <cmath>

namespace gpb {
namespace {

__device__ __forceinline__ double pc_cov(int cov, double d, double var, double a) {       // cov_fcts.h:2100-2118 on the transformed scale
  const double r = a * d;
  if (cov == 0) return var * exp(-r);
  if (cov == 1) return var * (1.0 + r) * exp(-r);
  return var * (1.0 + r + r * r / 3.0) * exp(-r);
}

__global__ void pc_piv_init_kernel(int n, double var, double* __restrict__ diag, int* __restrict__ pi, int* __restrict__ pos, int* __restrict__ done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  diag[i] = var; pi[i] = i; pos[i] = i; done[i] = 0;      // GetZSigmaZtij(h, h) = the marginal variance (re_comp.h:1393-1395)
}

// Step m: the first maximum of diag over pi[m..n) in the order of pi (Eigen's maxCoeff on diag(pi.tail(n - m)), CG_utils.h:457), the L1 norm of
// that tail (err of the previous step, :479), and the swap pi[m] <-> pi[i] (:459-461).  One workgroup.
__global__ __launch_bounds__(1024) void pc_piv_argmax_kernel(int n, int m, const double* __restrict__ diag, int* __restrict__ pi, int* __restrict__ pos,
                                                              double* __restrict__ out2) {
  __shared__ double s_val[1024];
  __shared__ double s_sum[1024];
  __shared__ int s_pos[1024];
  __shared__ int s_idx[1024];
  const int tid = threadIdx.x;
  double best = -INFINITY, sum = 0.0;
  int bpos = 0x7fffffff, bidx = -1;
  for (int i = tid; i < n; i += 1024) {
    const int ps = pos[i];
    if (ps < m) continue;
    const double v = diag[i];
    sum += fabs(v);
    if (v > best || (v == best && ps < bpos) || bidx < 0) { best = v; bpos = ps; bidx = i; }
  }
  s_val[tid] = best; s_sum[tid] = sum; s_pos[tid] = bpos; s_idx[tid] = bidx;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (tid < w) {
      s_sum[tid] += s_sum[tid + w];
      const double v = s_val[tid + w];
      const int ps = s_pos[tid + w], ix = s_idx[tid + w];
      if (ix >= 0 && (s_idx[tid] < 0 || v > s_val[tid] || (v == s_val[tid] && ps < s_pos[tid]))) { s_val[tid] = v; s_pos[tid] = ps; s_idx[tid] = ix; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int p = s_idx[0], i = s_pos[0], old = pi[m];
    pi[m] = p; pi[i] = old; pos[p] = m; pos[old] = i;
    out2[0] = (double)p; out2[1] = s_sum[0];
  }
}

// Column m of the factor (CG_utils.h:463-481): for every point j not yet chosen,
//   L_jm = Sigma(j, p) - L[j, :m] . L[p, :m];  unless |L_jm| < 1e-12: L_jm /= sqrt(diag[p]), stored;  diag[j] -= L_jm^2  (with the unscaled value in the other case, sic)
// and L[p][m] = sqrt(diag[p]).  j, p: Vecchia positions; rows of L: storage slots (sigma).
__global__ void pc_piv_update_kernel(const double4* __restrict__ pts, const int* __restrict__ sigma, int n, int k, int m, int p, int cov, int d3, double var, double a,
                                     double* __restrict__ L, double* __restrict__ diag, int* __restrict__ done) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (done[j]) return;
  const double dp = diag[p];
  if (j == p) { L[(size_t)sigma[p] * k + m] = sqrt(dp); done[p] = 1; return; }
  const double4 cj = pts[j], cp = pts[p];
  const double ex = cj.x - cp.x, ey = cj.y - cp.y, ez = d3 ? cj.z - cp.z : 0.0;
  double ljm = pc_cov(cov, sqrt(ex * ex + ey * ey + ez * ez), var, a);
  const double* Lj = L + (size_t)sigma[j] * k;
  const double* Lp = L + (size_t)sigma[p] * k;
  if (m > 0) {
    double sdot = 0.0;
    for (int q = 0; q < m; ++q) sdot += Lj[q] * Lp[q];
    ljm -= sdot;
  }
  if (!(fabs(ljm) < 1e-12)) { ljm /= sqrt(dp); L[(size_t)sigma[j] * k + m] = ljm; }
  diag[j] -= ljm * ljm;
}

__device__ __forceinline__ void pc_slice(int n, int parts, int b, int& lo, int& hi) {
  const int per = (n + parts - 1) / parts;
  lo = b * per; hi = lo + per < n ? lo + per : n;
  if (lo > n) lo = n;
}

// part[slice][e] = sum over the rows of the slice of L[i][p] W[i] L[i][q], e = p (p + 1) / 2 + q  (the lower triangle of L^T W L)
__global__ void pc_gram_kernel(const double* __restrict__ L, const double* __restrict__ W, int n, int k, int npairs, double* __restrict__ part) {
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  int p = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
  while ((p + 1) * (p + 2) / 2 <= e) ++p;
  while (p * (p + 1) / 2 > e) --p;
  const int q = e - p * (p + 1) / 2;
  int lo, hi;
  pc_slice(n, (int)gridDim.x, (int)blockIdx.x, lo, hi);
  double acc = 0.0;
  for (int i = lo; i < hi; ++i) { const double* Li = L + (size_t)i * k; acc = __builtin_fma(Li[p] * W[i], Li[q], acc); }
  part[(size_t)blockIdx.x * npairs + e] = acc;
}
// the same with 4 x 4 register tiles of (p, q): thread t of a slice owns the tile (tp, tq), tq <= tp, of the lower triangle -- 8 row values and the weight loaded for 16 fmas
// (the one-pair-per-thread form above: 3 loads per fma, 1.85 ms at n = 1e5, k = 200; profiles/r06_zz_trace_vif_non_gaussian_*)
__global__ __launch_bounds__(256) void pc_gram_tiled_kernel(const double* __restrict__ L, const double* __restrict__ W, int n, int k, int npairs, int ntile, double* __restrict__ part) {
  const int tl = blockIdx.y * blockDim.x + threadIdx.x;
  const int nt = (k + 3) / 4, ntiles = nt * (nt + 1) / 2;
  (void)ntile;
  if (tl >= ntiles) return;
  int tp = (int)((sqrt(8.0 * (double)tl + 1.0) - 1.0) * 0.5);
  while ((tp + 1) * (tp + 2) / 2 <= tl) ++tp;
  while (tp * (tp + 1) / 2 > tl) --tp;
  const int tq = tl - tp * (tp + 1) / 2;
  int lo, hi;
  pc_slice(n, (int)gridDim.x, (int)blockIdx.x, lo, hi);
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  const int p0 = 4 * tp, q0 = 4 * tq;
  for (int i = lo; i < hi; ++i) {
    const double* Li = L + (size_t)i * k;
    const double w = W[i];
    double lp[4], lq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { lp[a] = p0 + a < k ? Li[p0 + a] * w : 0.0; lq[a] = q0 + a < k ? Li[q0 + a] : 0.0; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(lp[a], lq[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pp = p0 + a, qq = q0 + b;
      if (pp < k && qq <= pp) part[(size_t)blockIdx.x * npairs + (size_t)pp * (pp + 1) / 2 + qq] = acc[a][b];
    }
}
__global__ void pc_gram_reduce_kernel(const double* __restrict__ part, int parts, int npairs, double* __restrict__ G) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npairs) return;
  double s = 0.0;
  for (int b = 0; b < parts; ++b) s += part[(size_t)b * npairs + e];      // fixed order
  G[e] = s;
}

// part[chunk][slice][q][c] = sum over the rows of the slice of L[i][q] W[i] X[i][c]; 512 threads = 8 row groups x 64 columns of L, every thread with UNR (8 / 2) independent
// accumulation chains (round 6: with 4 groups and one chain the kernel ran at 1 TB/s of the n x k matrix -- one workgroup per CU, one load in flight per thread;
// profiles/r06_z_trace_vif_non_gaussian_config4_size_rocprofv3_summary.txt).  The order of the additions is fixed: chain u of group g takes the rows lo + g + 8 (u + UNR j).
template <int NC>
__global__ __launch_bounds__(512) void pc_ltwx_kernel(const double* __restrict__ L, const double* __restrict__ W, const double* __restrict__ X, int n, int k,
                                                        double* __restrict__ part) {
  // NC == 1: 256 lanes span the columns (a row of L is read as ONE contiguous run: 64-column slabs made four strided passes over the rows), 2 row groups x 8 chains;
  // NC == 4: 64 lanes x 8 row groups x 2 chains (the block's 4 columns per row keep the register budget)
  constexpr int LW = NC == 1 ? 256 : 64, G = 512 / LW, UNR = NC == 1 ? 8 : 2;
  __shared__ double s[G][LW][NC];
  const int lane = threadIdx.x % LW, g = threadIdx.x / LW;
  const int chunk = blockIdx.y, parts = gridDim.x;
  const double* Xc = X + (size_t)chunk * n * NC;
  int lo, hi;
  pc_slice(n, parts, (int)blockIdx.x, lo, hi);
  for (int q0 = 0; q0 < k; q0 += LW) {
    const int q = q0 + lane;
    double acc[UNR][NC];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[u][c] = 0.0;
    if (q < k) {
      for (int i0 = lo + g; i0 < hi; i0 += G * UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = i0 + G * u;
          if (i < hi) {
            const double lw = L[(size_t)i * k + q] * W[i];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = __builtin_fma(lw, Xc[(size_t)i * NC + c], acc[u][c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double a = acc[0][c];
#pragma unroll
      for (int u = 1; u < UNR; ++u) a += acc[u][c];
      s[g][lane][c] = a;
    }
    __syncthreads();
    if (g == 0 && q < k) {
      double* dst = part + (((size_t)chunk * parts + blockIdx.x) * k + q) * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        double a = s[0][lane][c];
#pragma unroll
        for (int gg = 1; gg < G; ++gg) a += s[gg][lane][c];
        dst[c] = a;
      }
    }
    __syncthreads();
  }
}
// The block form as what it is -- a tall-skinny GEMM, P (k x 4 ncol) = L' (W X) -- on v_mfma_f64_16x16x4_f64 (round 6).  The kernel above re-reads the n x k matrix once per
// chunk of 4 columns (13 x 160 MB at n = 1e5, k = 200, t = 50: 626 us); a VALU form with all chunks per pass needs one operand per fma delivered as a scalar or an LDS
// broadcast (costed: the LDS port is 4x short, ~100 SGPRs of row data per step spill).  The matrix instruction reuses every loaded value 16 times: per step of 4 rows a
// wavefront loads 4 values of L (its 16-column tiles) and CT of W X and issues 4 CT MFMAs.  Workgroup = one row slice; wave w owns the column tiles w, w + 4, ... of L
// (k <= 256 per pass); CT column tiles of 16 block-vector columns (= 4 chunks each) per launch row.  Fragment maps (cdna_hip_programming.md section 3, as dense_kernels.hip):
// A[row = lane & 15][kk = lane >> 4], B[kk = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15].  Output part[chunk][slice][q][c] as above.
typedef double pc_d4 __attribute__((ext_vector_type(4)));
// Two steps of loads are in flight: issued before a step's MFMAs and CONSUMED two steps later (the masks and the product with W are applied when a value is used: a select right
// behind a load makes the wavefront wait for it).  (Tried: 8 waves x 2 tiles with two steps in flight, 3 waves per SIMD: 286 us against 185 us for this form with the
// select at the load -- more redundant loads of W X per step.)
template <int CT>
__global__ __launch_bounds__(256, 2) void pc_ltwx_mfma_kernel(const double* __restrict__ L, const double* __restrict__ W, const double* __restrict__ X, int n, int k, int nchunks,
                                                                double* __restrict__ part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
  const int parts = gridDim.x, col0 = blockIdx.y * 16 * CT;
  int lo, hi;
  pc_slice(n, parts, (int)blockIdx.x, lo, hi);
  size_t xoff[CT];
  bool xon[CT];
#pragma unroll
  for (int nj = 0; nj < CT; ++nj) {
    const int col = col0 + 16 * nj + fr, ch = col >> 2;
    xon[nj] = ch < nchunks;
    xoff[nj] = (size_t)(xon[nj] ? ch : 0) * n * 4 + (col & 3);
  }
  for (int qp = 0; qp < k; qp += 256) {
    int qa[4];
    bool qon[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int q = qp + 16 * (wave + 4 * t) + fr; qon[t] = q < k; qa[t] = qon[t] ? q : 0; }
    pc_d4 acc[4][CT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nj = 0; nj < CT; ++nj) acc[t][nj] = (pc_d4){0.0, 0.0, 0.0, 0.0};
    if (qp + 16 * wave < k && lo < hi) {
      double an[2][4], xn[2][CT], wn[2];
      bool rvn[2];
      auto fetch = [&](int i0, int b) {                // unconditional loads from clamped (valid) addresses
        const int i = i0 + fk;
        rvn[b] = i < hi;
        const size_t ir = (size_t)(rvn[b] ? i : lo);
        wn[b] = W[ir];
#pragma unroll
        for (int t = 0; t < 4; ++t) an[b][t] = L[ir * k + qa[t]];
#pragma unroll
        for (int nj = 0; nj < CT; ++nj) xn[b][nj] = X[xoff[nj] + ir * 4];
      };
      auto step = [&](int b, int inext) {             // consume buffer b, refill it with the rows two steps on, then the MFMAs
        double af[4], bf[CT];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = (rvn[b] && qon[t]) ? an[b][t] : 0.0;
#pragma unroll
        for (int nj = 0; nj < CT; ++nj) bf[nj] = (rvn[b] && xon[nj]) ? wn[b] * xn[b][nj] : 0.0;
        fetch(inext, b);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nj = 0; nj < CT; ++nj) acc[t][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t], bf[nj], acc[t][nj], 0, 0, 0);
      };
      fetch(lo, 0);
      fetch(lo + 4, 1);
      for (int i0 = lo; i0 < hi; i0 += 8) {            // (a step whose rows lie beyond the slice multiplies zeros)
        step(0, i0 + 8);
        step(1, i0 + 12);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nj = 0; nj < CT; ++nj) {
        const int col = col0 + 16 * nj + fr, ch = col >> 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qp + 16 * (wave + 4 * t) + fk + 4 * r;
          if (q < k && ch < nchunks) part[(((size_t)ch * parts + blockIdx.x) * k + q) * 4 + (col & 3)] = acc[t][nj][r];
        }
      }
  }
}
// x2[chunk][q][c] = sum_p M[q][p] y[p][c],  y = the slices' partial sums: four quarters of the slices added in slice order each (by four thread groups -- the sum over 256
// slices was a chain of 256 dependent loads per element, 77 us for 200 elements), then the quarters in order
template <int NC>
__global__ __launch_bounds__(1024) void pc_small_kernel(const double* __restrict__ part, int parts, const double* __restrict__ M, int k, double* __restrict__ x2) {
  extern __shared__ double s_y[];                      // k * NC, then 4 x k * NC quarter sums
  const int chunk = blockIdx.x;
  const int len = k * NC;
  double* s_q = s_y + len;
  const int sub = threadIdx.x >> 8, tl = threadIdx.x & 255;
  const int b0 = (int)((long long)parts * sub / 4), b1 = (int)((long long)parts * (sub + 1) / 4);
  for (int e = tl; e < len; e += 256) {
    double acc = 0.0;
    for (int b = b0; b < b1; ++b) acc += part[((size_t)chunk * parts + b) * len + e];
    s_q[sub * len + e] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < len; e += 1024) s_y[e] = ((s_q[e] + s_q[len + e]) + s_q[2 * len + e]) + s_q[3 * len + e];
  __syncthreads();
  for (int e = threadIdx.x; e < len; e += 1024) {
    const int q = e / NC, c = e % NC;
    const double* Mq = M + (size_t)q * k;
    double acc = 0.0;
    for (int p = 0; p < k; ++p) acc = __builtin_fma(Mq[p], s_y[p * NC + c], acc);
    x2[(size_t)chunk * len + e] = acc;
  }
}

// one row per thread: acc = L[i, :] x2[:, c];  mode 0: W (X - acc), 1: X - acc, 2: acc + X / sqrt(W), 3: -W acc (X not used beyond a load), 4: X + W acc
template <int NC>
__global__ __launch_bounds__(256) void pc_combine_kernel(const double* __restrict__ L, const double* __restrict__ W, const double* X,
                                                           const double* __restrict__ x2, int n, int k, int mode, double* out) {      // out may be X
  extern __shared__ double s_x2[];                     // k * NC
  const int chunk = blockIdx.y;
  const int len = k * NC;
  for (int e = threadIdx.x; e < len; e += 256) s_x2[e] = x2[(size_t)chunk * len + e];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* Li = L + (size_t)i * k;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  for (int q = 0; q < k; ++q) {
    const double l = Li[q];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_fma(l, s_x2[q * NC + c], acc[c]);
  }
  const size_t o = ((size_t)chunk * n + i) * NC;
  const double w = W[i];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double x = X[o + c];
    out[o + c] = mode == 0 ? w * x - w * acc[c] : (mode == 1 ? x - acc[c] : (mode == 2 ? acc[c] + sqrt(1.0 / w) * x : (mode == 3 ? -w * acc[c] : x + w * acc[c])));
  }
}

// the block form on the matrix cores (see pc_ltwx_mfma_kernel): acc (n x 4 ncol) = L x2 with one wavefront per 64 rows x 16 CT columns (rows <-> A, the k x 16 CT
// entries of x2 <-> B, steps of 4 over k), then the same epilogue as above.  (Tried: 32 rows per wave with L read as one 32-byte load per lane and group of 16 columns:
// 128 us against 117 us.)
template <int CT>
__global__ __launch_bounds__(256, 2) void pc_combine_mfma_kernel(const double* __restrict__ L, const double* __restrict__ W, const double* X, const double* __restrict__ x2, int n, int k,
                                                                   int nchunks, int mode, double* out) {      // out may be X
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
  const int r0 = blockIdx.x * 256 + 64 * wave, col0 = blockIdx.y * 16 * CT;
  if (r0 >= n) return;
  const double* La[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) { const int i = r0 + 16 * mi + fr; La[mi] = L + (size_t)(i < n ? i : n - 1) * k; }
  const double* xb[CT];
  bool xon[CT];
#pragma unroll
  for (int nj = 0; nj < CT; ++nj) {
    const int col = col0 + 16 * nj + fr, ch = col >> 2;
    xon[nj] = ch < nchunks;
    xb[nj] = x2 + (size_t)(xon[nj] ? ch : 0) * k * 4 + (col & 3);
  }
  pc_d4 acc[4][CT];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < CT; ++nj) acc[mi][nj] = (pc_d4){0.0, 0.0, 0.0, 0.0};
  double an[4], bn[CT];
  bool qvn;
  auto fetch = [&](int q0) {                             // unconditional loads from clamped addresses; masked when used (see pc_ltwx_mfma_kernel)
    const int q = q0 + fk;
    qvn = q < k;
    const int qc = qvn ? q : 0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) an[mi] = La[mi][qc];
#pragma unroll
    for (int nj = 0; nj < CT; ++nj) bn[nj] = xb[nj][(size_t)qc * 4];
  };
  fetch(0);
  for (int q0 = 0; q0 < k; q0 += 4) {
    double af[4], bf[CT];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[mi] = qvn ? an[mi] : 0.0;
#pragma unroll
    for (int nj = 0; nj < CT; ++nj) bf[nj] = (qvn && xon[nj]) ? bn[nj] : 0.0;
    if (q0 + 4 < k) fetch(q0 + 4);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < CT; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = r0 + 16 * mi + fk + 4 * r;
      if (i >= n) continue;
      const double w = W[i];
#pragma unroll
      for (int nj = 0; nj < CT; ++nj) {
        const int col = col0 + 16 * nj + fr, ch = col >> 2;
        if (ch >= nchunks) continue;
        const size_t o = ((size_t)ch * n + i) * 4 + (col & 3);
        const double x = X[o], a = acc[mi][nj][r];
        out[o] = mode == 0 ? w * x - w * a : (mode == 1 ? x - a : (mode == 2 ? a + sqrt(1.0 / w) * x : (mode == 3 ? -w * a : x + w * a)));
      }
    }
}

__global__ void pc_rowscale_kernel(const double* x, const double* __restrict__ w, int n, int nc, int inv, double* out) {       // x may be out
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const size_t o = (size_t)blockIdx.y * n * nc + g;
  const double wi = w[g / nc];
  out[o] = inv ? x[o] * (1.0 / wi) : wi * x[o];
}
__global__ void pc_add_div_kernel(double* __restrict__ v, const double* __restrict__ h, const double* __restrict__ w, int n, int nc) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const size_t o = (size_t)blockIdx.y * n * nc + g;
  v[o] += (1.0 / w[g / nc]) * h[o];
}

__global__ __launch_bounds__(1024) void pc_aux_sums_kernel(const double* __restrict__ L, const double* __restrict__ M, const double* __restrict__ W,
                                                            const double* __restrict__ wp, int n, int k, double* __restrict__ out2) {
  __shared__ double s[2048];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double* Li = L + (size_t)i * k;
    double sdiag = 0.0;
    for (int q = 0; q < k; ++q) {
      const double* Mq = M + (size_t)q * k;
      double acc = 0.0;
      for (int p = 0; p < k; ++p) acc = __builtin_fma(Mq[p], Li[p], acc);
      sdiag = __builtin_fma(Li[q], acc, sdiag);
    }
    if (wp) { a += sdiag * (wp[i] * (wp[i] / W[i])); b += wp[i] / W[i]; }
    else a += sdiag * W[i];
  }
  s[threadIdx.x] = a; s[1024 + threadIdx.x] = b;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (threadIdx.x < w) { s[threadIdx.x] += s[threadIdx.x + w]; s[1024 + threadIdx.x] += s[1024 + threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = s[0]; out2[1] = s[1024]; }
}
__global__ __launch_bounds__(1024) void pc_wmax_kernel(const double* __restrict__ w, int n, double* __restrict__ out1) {
  __shared__ double s[1024];
  double best = -INFINITY;
  bool nan = false;
  for (int i = threadIdx.x; i < n; i += 1024) { const double v = w[i]; if (v != v) nan = true; if (v > best) best = v; }
  s[threadIdx.x] = nan ? NAN : best;
  __syncthreads();
  for (int k = 512; k >= 1; k >>= 1) {
    if (threadIdx.x < k) { const double a = s[threadIdx.x], b = s[threadIdx.x + k]; s[threadIdx.x] = (a != a || b != b) ? NAN : (a > b ? a : b); }
    __syncthreads();
  }
  if (threadIdx.x == 0) out1[0] = s[0];
}

__global__ void pc_pack_rows_kernel(const double* __restrict__ src, int ld, const int* __restrict__ sigma, int n, int k, double* __restrict__ dst,
                                    double* __restrict__ vnorm2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* in = src + (size_t)i * ld;
  double* out = dst + (size_t)sigma[i] * k;
  double s2 = 0.0;
  for (int q = 0; q < k; ++q) { const double v = in[q]; out[q] = v; s2 = __builtin_fma(v, v, s2); }
  if (vnorm2) vnorm2[sigma[i]] = s2;
}
// "vecchia_response": the pseudo nugget of the preconditioner's factor, Vecchia order, from W in storage order -- 1 / W_i plus the jitter var * 1e-10 the reference
// multiplies into the NEIGHBOURS' diagonal entries (Vecchia_utils.cpp:1606-1614): the factor kernel takes ONE diagonal addition per point, so the point's own
// entry carries the jitter too and pc_vr_diag takes it out of D again (D_i = own entry - A_i c_i is linear in the own entry)
__global__ void pc_vr_nugget_kernel(const double* __restrict__ W, const int* __restrict__ sigma, int n, double jit, double* __restrict__ nug) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nug[i] = 1.0 / W[sigma[i]] + jit;
}
__global__ void pc_vr_diag_kernel(const double* __restrict__ D2, const int* __restrict__ sigma, int n, double jit, double* __restrict__ D2s,
                                  double* __restrict__ sqrtD2s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = D2[i] - jit;
  D2s[sigma[i]] = d;
  sqrtD2s[sigma[i]] = sqrt(d);
}
__global__ void pc_fitc_diag_kernel(const double* __restrict__ W, const double* __restrict__ vnorm2, double sm00, int n, double* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = 1.0 / W[i];
  d += sm00;
  d -= vnorm2[i];
  wp[i] = 1.0 / d;
}

// Row-wise quadratic forms with k x k symmetric matrices, k <= 256: one wavefront per 4 rows; lane l owns the columns q = l + 64 j of M, reads row p of M coalesced
// (M symmetric: M[p][q] = M[q][p]) once for its 4 rows; x[p] of the rows broadcast from LDS.  (round 6: the one-thread-per-row form below took 33 ms at n = 1e5, k = 200 --
// k^2 global loads per thread; profiles/r06_z_trace_vif_non_gaussian_config4_size_rocprofv3_summary.txt.)
//   TWO: out[i] = c0 - 2 L_i' M1 L2_i + L_i' M2 L_i;   !TWO: out[i] = L_i' M2 L_i
template <bool TWO>
__global__ __launch_bounds__(256) void pc_row_quad_tiled_kernel(const double* __restrict__ L, const double* __restrict__ L2, const double* __restrict__ M1,
                                                                const double* __restrict__ M2, int n, int k, double c0, double* __restrict__ out) {
  constexpr int RW = 4, JM = 4;
  __shared__ double s_x[4][RW][256], s_y[4][RW][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wv) * RW;
  if (row0 >= n) return;
  for (int r = 0; r < RW; ++r) {
    const int i = min(row0 + r, n - 1);
    for (int p = lane; p < k; p += 64) { s_x[wv][r][p] = L[(size_t)i * k + p]; if (TWO) s_y[wv][r][p] = L2[(size_t)i * k + p]; }
  }
  __builtin_amdgcn_wave_barrier();
  double a1[RW][JM], a2[RW][JM];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < JM; ++j) { a1[r][j] = 0.0; a2[r][j] = 0.0; }
  for (int p = 0; p < k; ++p) {
    double m1[JM], m2[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int q = lane + 64 * j;
      m2[j] = q < k ? M2[(size_t)p * k + q] : 0.0;
      m1[j] = (TWO && q < k) ? M1[(size_t)p * k + q] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const double xp = s_x[wv][r][p];
      const double yp = TWO ? s_y[wv][r][p] : 0.0;
#pragma unroll
      for (int j = 0; j < JM; ++j) { a2[r][j] = __builtin_fma(m2[j], xp, a2[r][j]); if (TWO) a1[r][j] = __builtin_fma(m1[j], yp, a1[r][j]); }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    double q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const int q = lane + 64 * j;
      const double xq = q < k ? s_x[wv][r][q] : 0.0;
      q2 = __builtin_fma(xq, a2[r][j], q2);
      if (TWO) q1 = __builtin_fma(xq, a1[r][j], q1);
    }
    for (int off = 32; off > 0; off >>= 1) { q1 += __shfl_xor(q1, off); q2 += __shfl_xor(q2, off); }
    if (lane == 0 && row0 + r < n) out[row0 + r] = TWO ? c0 - 2.0 * q1 + q2 : q2;
  }
}

__global__ void pc_row_stats_kernel(const double* __restrict__ U, const double* __restrict__ WIPIZ, const double* __restrict__ L, const double* __restrict__ M,
                                    const double* __restrict__ W, const double* __restrict__ dW3, int n, int k, int t, int nc, double* __restrict__ dld,
                                    const double* __restrict__ wp, int det_centre, const double* __restrict__ sdiag_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d3 = dW3[i], wi = 1.0 / W[i];
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < t; ++c) {
    const size_t o = ((size_t)(c / nc) * n + i) * nc + (c % nc);
    s1 += -1.0 * ((wi * U[o]) * d3 * WIPIZ[o]);
    s2 += -1.0 * (WIPIZ[o] * d3 * WIPIZ[o]);
  }
  const double tr1 = s1 / t, trP = s2 / t;
  // diag of L (I_k + L^T W L)^-1 L^T  (likelihoods.h:16583-16586)
  const double* Li = L + (size_t)i * k;
  double sdiag = 0.0;
  if (sdiag_in) sdiag = sdiag_in[i];                    // (pc_row_quad_tiled_kernel<false>)
  else
  for (int q = 0; q < k; ++q) {
    const double* Mq = M + (size_t)q * k;
    double acc = 0.0;
    for (int p = 0; p < k; ++p) acc = __builtin_fma(Mq[p], Li[p], acc);
    sdiag = __builtin_fma(Li[q], acc, sdiag);
  }
  const double trw = wi * d3;
  // det_centre (full-scale Vecchia, likelihoods.h:5376-5390): CalcOptimalCVectorized centres the control variate with its DETERMINISTIC trace, the Vecchia path
  // (:16630-16632) with the mean of the samples
  double b_centre = trP;
  if (det_centre && wp) { const double tDI0 = wi * (trw * wp[i]); b_centre = sdiag * (wp[i] * tDI0) - tDI0; }
  double cv = 0.0, vr = 0.0;
  for (int c = 0; c < t; ++c) {
    const size_t o = ((size_t)(c / nc) * n + i) * nc + (c % nc);
    const double a1 = -1.0 * ((wi * U[o]) * d3 * WIPIZ[o]) - tr1;
    const double b1 = -1.0 * (WIPIZ[o] * d3 * WIPIZ[o]) - b_centre;
    cv += a1 * b1; vr += b1 * b1;
  }
  cv /= t; vr /= t;
  const double copt = (vr == 0.0) ? 1.0 : cv / vr;
  if (wp) {
    const double tDI = wi * (trw * wp[i]), tDIDI = wp[i] * tDI;
    dld[i] = tr1 + trw + copt * (sdiag * tDIDI - tDI) - copt * trP;
    return;
  }
  dld[i] = tr1 + trw + copt * (sdiag * d3 - trw) - copt * trP;
}

// out[i] = c0 - 2 L_i' M1 L2_i + L_i' M2 L_i: the derivative of the fitc preconditioner's diagonal, Sigma_m[0][0]' - 2 C_i' Sigma_m^-1 dC_i + C_i' Sigma_m^-1 dSigma_m Sigma_m^-1 C_i
// (likelihoods.h:5478-5486); L / L2 [n][k] (rows in storage order), M1 / M2 k x k row-major (symmetric)
__global__ void pc_row_quad_kernel(const double* __restrict__ L, const double* __restrict__ L2, const double* __restrict__ M1, const double* __restrict__ M2,
                                   int n, int k, double c0, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* Li = L + (size_t)i * k;
  const double* L2i = L2 + (size_t)i * k;
  double q1 = 0.0, q2 = 0.0;
  for (int q = 0; q < k; ++q) {
    const double* M1q = M1 + (size_t)q * k;
    const double* M2q = M2 + (size_t)q * k;
    double a1 = 0.0, a2 = 0.0;
    for (int p = 0; p < k; ++p) { a1 = __builtin_fma(M1q[p], L2i[p], a1); a2 = __builtin_fma(M2q[p], Li[p], a2); }
    q1 = __builtin_fma(Li[q], a1, q1); q2 = __builtin_fma(Li[q], a2, q2);
  }
  out[i] = c0 - 2.0 * q1 + q2;
}
// columns [col0, col0 + cnt) of L [n][k] (rows in storage order), times w[i], into a block vector [chunk][row][nc] of ncol chunks (zero beyond cnt)
__global__ void pc_cols_to_block_kernel(const double* __restrict__ L, const double* __restrict__ w, int n, int k, int col0, int cnt, int nc, double* __restrict__ out) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const int i = (int)(g / nc), c = (int)(blockIdx.y * nc + g % nc);
  out[(size_t)blockIdx.y * n * nc + g] = c < cnt ? w[i] * L[(size_t)i * k + col0 + c] : 0.0;
}
// out[i] = a[i] * b[i] * (c ? c[i] : 1)
__global__ void pc_mul3_kernel(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i] * (c ? c[i] : 1.0);
}

}  // namespace

int pc_parts(int n) { const int p = (n + 255) / 256; return p < 1 ? 1 : (p > 1024 ? 1024 : p); }      // (cap 256 until round 6: one workgroup per CU left the tall-skinny reductions at 1.6 TB/s)

hipError_t pc_piv_init(int n, int k, double var, double* L, double* diag, int* pi, int* pos, int* done, hipStream_t st) {
  hipError_t e = hipMemsetAsync(L, 0, sizeof(double) * (size_t)n * k, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pc_piv_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, var, diag, pi, pos, done);
  return hipGetLastError();
}
hipError_t pc_piv_argmax(int n, int m, const double* diag, int* pi, int* pos, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(pc_piv_argmax_kernel, dim3(1), dim3(1024), 0, st, n, m, diag, pi, pos, out2);
  return hipGetLastError();
}
hipError_t pc_piv_update(const double4* pts, const int* sigma, int n, int k, int m, int p, int cov, int d3, double var, double a, double* L, double* diag, int* done,
                         hipStream_t st) {
  hipLaunchKernelGGL(pc_piv_update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, sigma, n, k, m, p, cov, d3, var, a, L, diag, done);
  return hipGetLastError();
}
hipError_t pc_gram(const double* L, const double* W, int n, int k, double* part, double* G, hipStream_t st) {
  const int npairs = k * (k + 1) / 2, parts = pc_parts(n);
  const int nt = (k + 3) / 4, ntiles = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(pc_gram_tiled_kernel, dim3(parts, (ntiles + 255) / 256), dim3(256), 0, st, L, W, n, k, npairs, ntiles, part);
  hipLaunchKernelGGL(pc_gram_reduce_kernel, dim3((npairs + 255) / 256), dim3(256), 0, st, part, parts, npairs, G);
  return hipGetLastError();
}
hipError_t pc_ltwx(const double* L, const double* W, const double* M, const double* X, int n, int k, int ncol, int nc, double* part, double* x2, hipStream_t st) {
  const int parts = pc_parts(n);
  const size_t lds = sizeof(double) * (size_t)k * nc * 5;
  if (nc == 4) {
    if (ncol > 8) {      // groups of 16 chunks (64 block-vector columns), one launch each: the pointers move, the kernel sees chunks 0 .. 15 of its group
      for (int g0 = 0; g0 < ncol; g0 += 16)
        hipLaunchKernelGGL(pc_ltwx_mfma_kernel<4>, dim3(parts, 1), dim3(256), 0, st, L, W, X + (size_t)g0 * n * 4, n, k, ncol - g0 < 16 ? ncol - g0 : 16,
                           part + (size_t)g0 * parts * k * 4);
    }
    else if (ncol > 4) hipLaunchKernelGGL(pc_ltwx_mfma_kernel<2>, dim3(parts, 1), dim3(256), 0, st, L, W, X, n, k, ncol, part);
    else hipLaunchKernelGGL(pc_ltwx_mfma_kernel<1>, dim3(parts, 1), dim3(256), 0, st, L, W, X, n, k, ncol, part);
    hipLaunchKernelGGL(pc_small_kernel<4>, dim3(ncol), dim3(1024), lds, st, part, parts, M, k, x2);
  } else {
    hipLaunchKernelGGL(pc_ltwx_kernel<1>, dim3(parts, ncol), dim3(512), 0, st, L, W, X, n, k, part);
    hipLaunchKernelGGL(pc_small_kernel<1>, dim3(ncol), dim3(1024), lds, st, part, parts, M, k, x2);
  }
  return hipGetLastError();
}
hipError_t pc_combine(const double* L, const double* W, const double* X, const double* x2, int n, int k, int ncol, int nc, int mode, double* out, hipStream_t st) {
  const size_t lds = sizeof(double) * (size_t)k * nc;
  if (nc == 4 && ncol > 8) {
    for (int g0 = 0; g0 < ncol; g0 += 16)
      hipLaunchKernelGGL(pc_combine_mfma_kernel<4>, dim3((n + 255) / 256, 1), dim3(256), 0, st, L, W, X + (size_t)g0 * n * 4, x2 + (size_t)g0 * k * 4, n, k,
                         ncol - g0 < 16 ? ncol - g0 : 16, mode, out + (size_t)g0 * n * 4);
  }
  else if (nc == 4 && ncol > 4) hipLaunchKernelGGL(pc_combine_mfma_kernel<2>, dim3((n + 255) / 256, 1), dim3(256), 0, st, L, W, X, x2, n, k, ncol, mode, out);
  else if (nc == 4) hipLaunchKernelGGL(pc_combine_mfma_kernel<1>, dim3((n + 255) / 256, 1), dim3(256), 0, st, L, W, X, x2, n, k, ncol, mode, out);
  else hipLaunchKernelGGL(pc_combine_kernel<1>, dim3((n + 255) / 256, ncol), dim3(256), lds, st, L, W, X, x2, n, k, mode, out);
  return hipGetLastError();
}
hipError_t pc_rowscale(const double* x, const double* w, int n, int ncol, int nc, int inv, double* out, hipStream_t st) {
  hipLaunchKernelGGL(pc_rowscale_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, x, w, n, nc, inv, out);
  return hipGetLastError();
}
hipError_t pc_add_div(double* v, const double* h, const double* w, int n, int ncol, int nc, hipStream_t st) {
  hipLaunchKernelGGL(pc_add_div_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, v, h, w, n, nc);
  return hipGetLastError();
}
hipError_t pc_aux_sums(const double* L, const double* M, const double* W, const double* wp, int n, int k, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(pc_aux_sums_kernel, dim3(1), dim3(1024), 0, st, L, M, W, wp, n, k, out2);
  return hipGetLastError();
}
hipError_t pc_wmax(const double* w, int n, double* out1, hipStream_t st) {
  hipLaunchKernelGGL(pc_wmax_kernel, dim3(1), dim3(1024), 0, st, w, n, out1);
  return hipGetLastError();
}
hipError_t pc_row_stats(const double* U, const double* WIPIZ, const double* L, const double* M, const double* W, const double* dW3, int n, int k, int t, int nc,
                        double* dld, hipStream_t st, const double* wp, int det_centre, double* sdiag_scratch) {
  const double* sd = nullptr;
  if (sdiag_scratch && k <= 256) {      // diag(L M L') by the tiled kernel first (M symmetric)
    hipLaunchKernelGGL(pc_row_quad_tiled_kernel<false>, dim3((n + 15) / 16), dim3(256), 0, st, L, L, M, M, n, k, 0.0, sdiag_scratch);
    sd = sdiag_scratch;
  }
  hipLaunchKernelGGL(pc_row_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, st, U, WIPIZ, L, M, W, dW3, n, k, t, nc, dld, wp, det_centre, sd);
  return hipGetLastError();
}
hipError_t pc_row_quad(const double* L, const double* L2, const double* M1, const double* M2, int n, int k, double c0, double* out, hipStream_t st) {
  if (k <= 256) hipLaunchKernelGGL(pc_row_quad_tiled_kernel<true>, dim3((n + 15) / 16), dim3(256), 0, st, L, L2, M1, M2, n, k, c0, out);
  else hipLaunchKernelGGL(pc_row_quad_kernel, dim3((n + 127) / 128), dim3(128), 0, st, L, L2, M1, M2, n, k, c0, out);
  return hipGetLastError();
}
hipError_t pc_cols_to_block(const double* L, const double* w, int n, int k, int col0, int cnt, int ncol, int nc, double* out, hipStream_t st) {
  hipLaunchKernelGGL(pc_cols_to_block_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, L, w, n, k, col0, cnt, nc, out);
  return hipGetLastError();
}
hipError_t pc_mul3(const double* a, const double* b, const double* c, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(pc_mul3_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, c, n, out);
  return hipGetLastError();
}
hipError_t pc_pack_rows(const double* src, int ld, const int* sigma, int n, int k, double* dst, double* vnorm2, hipStream_t st) {
  hipLaunchKernelGGL(pc_pack_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, ld, sigma, n, k, dst, vnorm2);
  return hipGetLastError();
}
hipError_t pc_vr_nugget(const double* W, const int* sigma, int n, double jit, double* nug, hipStream_t st) {
  hipLaunchKernelGGL(pc_vr_nugget_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W, sigma, n, jit, nug);
  return hipGetLastError();
}
hipError_t pc_vr_diag(const double* D2, const int* sigma, int n, double jit, double* D2s, double* sqrtD2s, hipStream_t st) {
  hipLaunchKernelGGL(pc_vr_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, D2, sigma, n, jit, D2s, sqrtD2s);
  return hipGetLastError();
}
hipError_t pc_fitc_diag(const double* W, const double* vnorm2, double sm00, int n, double* wp, hipStream_t st) {
  hipLaunchKernelGGL(pc_fitc_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W, vnorm2, sm00, n, wp);
  return hipGetLastError();
}

}  // namespace gpb
