// gpboost_amd/csrc/vif_kernels.hip
//
// Full-scale Vecchia ("VIF": Vecchia-inducing-points full-scale) approximation, Gaussian likelihood, Euclidean neighbours --
// SURVEY.md section 8, row f4.  The covariance is
//     Psi = C_nm Sigma_m^-1 C_mn  +  Vecchia approximation of (the residual process + nugget),
// with k inducing points (kmeans++ on the host).  What the reference does per evaluation and what stands in for it here:
//   CalcSigmaComps            include/GPBoost/re_model_template.h:8151-8200   Sigma_m (diagonal x (1 + 1e-6)), its Cholesky factor L_m (host: k <= 256)
//                                                                               C_nm and V = L_m^-1 C_mn          -> vif_crosscov_kernel, vif_gemm_kernel
//   CalcCovFactorGradientVecchia, full_scale_vecchia branches
//                             src/GPBoost/Vecchia_utils.cpp:1463-1500, 1599-1623
//                                                                               every covariance of the per-point system minus the predictive-process
//                                                                               part V_a . V_b, then A_i, D_i     -> vif_resid_factor_kernel
//   CalcCovFactorFITC_FSA     re_model_template.h:9646-9745                     B C_nm (vif_spmm_kernel) and the Woodbury matrix
//                                                                               Sigma_m + (B C_nm)' D^-1 (B C_nm): vif_gram_kernel (tiled, split over row
//                                                                               chunks, fixed-order sum), k x k Cholesky on the host
//   CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i  re_model_template.h:2205-2330 and the derivative branches of the residual-process factor
//                             Vecchia_utils.cpp:1503-1524, 1640-1656            -> vif_resid_grad_kernel + four n x k x k products
//
// Layout (round 4): every n x k matrix is ROW-major [n][kq], kq = k + 1 rounded up to a multiple of 8 doubles (64 bytes): a point's row is one
// contiguous, line-aligned segment -- what the gathers of a point's neighbours (vif_spmm_kernel, the derivative kernel) and the tiles of the
// products want.  Column k of C carries the RESPONSE y, so that Q = B [C, y] has u = B y in its column k and ONE Gram pass returns the Woodbury
// matrix, (B C)' D^-1 u and u' D^-1 u; columns beyond k are zero, and the k x k matrices the products multiply with are stored [kq][kq] with
// zero rows / columns from k on, so the response column never leaks into a product.
#include "dev_common.h"
#include "vecchia_kernels.h"
#include "vif_kernels.h"

namespace gpb {

namespace {

// Matern 0.5 / 1.5 / 2.5 on the transformed scale (include/GPBoost/cov_fcts.h:2100-2118): var * f(a * dist)
template <int COV>
__device__ __forceinline__ double matern_plain(double dist, double var, double a) {
  const double r = a * dist;
  const double e = var * exp(-r);
  if constexpr (COV == kMatern05) return e;
  else if constexpr (COV == kMatern15) return e * (1.0 + r);
  else return e * __builtin_fma(r, __builtin_fma(r, 1.0 / 3.0, 1.0), 1.0);
}
// the covariance and its derivative wrt log a (GradientRangeMaternShape0_5 / 1_5 / 2_5 with transf_scale, cov_fcts.h:2535-2554):
// -r e, -r^2 e, -r^2 (1 + r) e / 3 with e = var exp(-r)
template <int COV>
__device__ __forceinline__ void matern_with_grad(double dist, double var, double a, double& kv, double& dk) {
  const double r = a * dist;
  const double e = var * exp(-r);
  if constexpr (COV == kMatern05) { kv = e; dk = -r * e; }
  else if constexpr (COV == kMatern15) { kv = e * (1.0 + r); dk = -r * r * e; }
  else { kv = e * __builtin_fma(r, __builtin_fma(r, 1.0 / 3.0, 1.0), 1.0); dk = -r * r * (1.0 + r) * e * (1.0 / 3.0); }
}
__device__ __forceinline__ double dist3(const double4& p, const double* q, int d) {
  const double dx = p.x - q[0], dy = d > 1 ? p.y - q[1] : 0.0, dz = d > 2 ? p.z - q[2] : 0.0;
  return sqrt(dx * dx + dy * dy + dz * dz);
}
__device__ __forceinline__ double dist4(const double4& p, const double4& q) {
  const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
  return sqrt(dx * dx + dy * dy + dz * dz);
}
// sum over the 16 lanes of an aligned group, result in every lane (fixed butterfly: bit-reproducible)
__device__ __forceinline__ double sum16(double s) {
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  return s;
}
}  // namespace

// C[i][j] = var k(|x_i - ip_j|) for j < k, C[i][k] = y_i (the response column), 0 beyond; dC (optional): d/d log a of the same, 0 from k on
template <int COV, bool GRAD>
__global__ __launch_bounds__(256) void vif_crosscov_kernel(const double4* __restrict__ pts, const double* __restrict__ ip, int i0, int i1, int k, int kq, int d,
                                                           double var, double a, double* __restrict__ C, double* __restrict__ dC) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t i = (size_t)i0 + e / kq;
  const int j = (int)(e % kq);
  if (i >= (size_t)i1) return;
  const double4 p = pts[i];
  double kv = 0.0, dk = 0.0;
  if (j < k) {
    const double dist = dist3(p, ip + (size_t)j * 3, d);
    if constexpr (GRAD) matern_with_grad<COV>(dist, var, a, kv, dk);
    else kv = matern_plain<COV>(dist, var, a);
  } else if (j == k) kv = p.w;
  C[i * kq + j] = kv;
  if constexpr (GRAD) dC[i * kq + j] = dk;
}

// Out[n][kq] (+)= In[n][kq] * M[kq][kq] (M row-major, zero outside its k x k block): 64 x 64 output tile per workgroup, 16-wide K steps through
// LDS, 4 x 4 outputs per thread.  ACC: Out += ...
template <bool ACC>
__global__ __launch_bounds__(256) void vif_gemm_kernel(const double* __restrict__ In, const double* __restrict__ M, int n, int kq, double* __restrict__ Out) {
  __shared__ double sA[16][65];     // [kk][row]
  __shared__ double sB[16][64];     // [kk][col]
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
  const size_t row0 = (size_t)blockIdx.x * 64;
  const int col0 = blockIdx.y * 64;
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  for (int k0 = 0; k0 < kq; k0 += 16) {
#pragma unroll
    for (int e = tid; e < 64 * 16; e += 256) {             // A tile: 64 rows x 16 columns of In (a row's 16 doubles contiguous)
      const int r = e >> 4, kk = e & 15;
      const size_t gi = row0 + r;
      sA[kk][r] = (gi < (size_t)n && k0 + kk < kq) ? In[gi * kq + k0 + kk] : 0.0;
    }
#pragma unroll
    for (int e = tid; e < 16 * 64; e += 256) {             // B tile: 16 rows x 64 columns of M
      const int kk = e >> 6, c = e & 63;
      sB[kk][c] = (k0 + kk < kq && col0 + c < kq) ? M[(size_t)(k0 + kk) * kq + col0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = sA[kk][tr + 16 * r];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = sB[kk][tc + 16 * c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fma(av[r], bv[c], acc[r][c]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t gi = row0 + tr + 16 * r;
    if (gi >= (size_t)n) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gc = col0 + tc + 16 * c;
      if (gc >= kq) continue;
      if constexpr (ACC) Out[gi * kq + gc] += acc[r][c];
      else Out[gi * kq + gc] = acc[r][c];
    }
  }
}

// ---- the point's k x k system in REGISTERS (one-wavefront kernels, NT > 0) -----------------------------------------------------------
// Lane r holds row r of C_nn / of its lower factor L in R = 16 NT registers (static indices: every loop below is unrolled over R and guarded by the
// uniform j < k); a pivot or a multiplier travels by v_readlane (SGPR pair -> operand of the fma), so a column step of the right-looking Cholesky is
// 2 readlanes + 1 fma per trailing column instead of a read-modify-write chain through LDS with three barriers (the LDS form cost ~3500 cycles per
// column: 44 of the ~98 us a point took).  Entries above the diagonal are never read (lane c's row[j], j < c, is the multiplier l_cj), so the
// updates run unpredicated and leave garbage there.  Lanes >= k carry identity rows: their multipliers are zero.
__device__ __forceinline__ double vif_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int I> struct VifIC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void vif_sfor(F&& f) {
  if constexpr (B < E) { f(VifIC<B>{}); vif_sfor<B + 1, E>(f); }
}
template <int B, int E, class F> __device__ __forceinline__ void vif_sfor_down(F&& f) {      // E - 1, E - 2, ..., B
  if constexpr (B < E) { f(VifIC<E - 1>{}); vif_sfor_down<B, E - 1>(f); }
}
// in: rows of the SPD matrix (lower triangle); out: rows of L (diagonal included), dinv = 1 / L_rr of this lane's row (1 for lanes >= k)
template <int R>
__device__ __forceinline__ void vif_reg_cholesky(double (&row)[R], int k, int lane, double& dinv) {
  dinv = 1.0;
  vif_sfor<0, R>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    if (j < k) {
      const double d = vif_readlane(row[j], j);
      const double sd = sqrt(d), inv = 1.0 / sd;
      const double t = row[j] * inv;
      row[j] = lane == j ? sd : t;
      if (lane == j) dinv = inv;
      vif_sfor<j + 1, R>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        const double lc = vif_readlane(row[j], c);
        row[c] = __builtin_fma(-row[j], lc, row[c]);
      });
    }
  });
}
// L z = b for two right-hand sides held one entry per lane (b -> z in place); row = rows of L
template <int R>
__device__ __forceinline__ void vif_reg_forward2(const double (&row)[R], int k, int lane, double dinv, double& b1, double& b2) {
  vif_sfor<0, R>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    if (j < k) {
      const double t1 = b1 * dinv, t2 = b2 * dinv;
      const double z1 = vif_readlane(t1, j), z2 = vif_readlane(t2, j);
      b1 = lane > j ? __builtin_fma(-row[j], z1, b1) : (lane == j ? t1 : b1);
      b2 = lane > j ? __builtin_fma(-row[j], z2, b2) : (lane == j ? t2 : b2);
    }
  });
}
// L' x = z (z -> x in place); col = COLUMNS of L: col[r] of lane c is L[r][c]
template <int R, bool TWO>
__device__ __forceinline__ void vif_reg_backward(const double (&col)[R], int k, int lane, double dinv, double& z1, double& z2) {
  vif_sfor_down<0, R>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    if (j < k) {
      const double t1 = z1 * dinv;
      const double x1 = vif_readlane(t1, j);
      z1 = lane < j ? __builtin_fma(-col[j], x1, z1) : (lane == j ? t1 : z1);
      if constexpr (TWO) {
        const double t2 = z2 * dinv;
        const double x2 = vif_readlane(t2, j);
        z2 = lane < j ? __builtin_fma(-col[j], x2, z2) : (lane == j ? t2 : z2);
      }
    }
  });
}
// rows / columns of the factor kept in s_C (lower triangle) -> registers; lanes >= k: identity
template <int R>
__device__ __forceinline__ void vif_load_rows(const double* s_C, int ld, int k, int lane, double (&row)[R]) {
  const int rl = min(lane, k);                                   // (keeps the address inside s_C for the idle lanes)
  vif_sfor<0, R>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    const double v = (c < k) ? s_C[rl * ld + c] : 0.0;
    row[c] = lane < k ? v : (c == lane ? 1.0 : 0.0);
  });
}
template <int R>
__device__ __forceinline__ void vif_load_cols(const double* s_C, int ld, int k, int lane, double (&col)[R]) {
  const int cl = min(lane, k);
  vif_sfor<0, R>([&](auto r_) {
    constexpr int r = decltype(r_)::value;
    const double v = (r < k) ? s_C[r * ld + cl] : 0.0;           // r < lane: above the diagonal, never used
    col[r] = lane < k ? v : (r == lane ? 1.0 : 0.0);
  });
}
// sum over the wavefront (fixed butterfly: every lane ends with the same total)
__device__ __forceinline__ double vif_wave_sum(double v) {
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) v += __shfl_xor(v, w, 64);
  return v;
}

// Common prefix of the factor and the derivative kernel of the residual process: one workgroup of T lanes per point -- T = 64, ONE wavefront, for
// m <= 62 (every __syncthreads of the Cholesky / substitution chains is then a barrier of a single wavefront: the 128-lane form spent most of its
// 16 ms at n = 1e5 in ~210 two-wavefront barriers per point), T = 128 beyond.  The whitened cross-covariances of the point and its neighbours pass
// through LDS in chunks of kVifKC = 64 columns (round 4: 16 KB instead of 50 KB per workgroup at m = 30, k = 200 -> six workgroups per CU instead of
// two); a chunk is staged with one lane per column and sixteen rows' loads in flight (the first form's element loop issued one dependent
// load -> store pair at a time: ~30 exposed memory latencies per chunk).  The Gram matrix G = V_S V_S' of the staged rows S accumulates across the
// chunks
//   NT > 0 (T = 64, NT = ceil((m + 1) / 16) <= 4): on the matrix cores -- 16 x 16 tiles (I >= J) of v_mfma_f64_16x16x4_f64, the A operand of tile
//           row I doubling as the B operand of tile column I (lane (fr, fk) holds V[16 I + fr][4 kk + fk] for both), NT (NT + 1) / 2 accumulator
//           tiles in registers over all chunks; rows beyond the point's k + 1 are zero;
//   NT = 0 (T = 128, m > 62): pair (r, c <= r) by pair on the lanes, one fma chain per pair in ascending column order, accumulated in s_C.
// Then
//   C_nn = var k(.) - G + nugget I,   c = var k(.) - G[., i]                      (Vecchia_utils.cpp:1489-1500, 1601)
// and C_nn is factorised in place (right-looking Cholesky in LDS, stands in for Eigen's LLT, :1617).  Returns k = number of neighbours and
// G[i][i]; s_c = c; s_C rows 0..k-1 = the lower factor; s_idx = the staged points (row k: the point itself).
constexpr int kVifKC = 64;            // columns of V per staged chunk
constexpr int kVifKCP = kVifKC + 1;   // LDS row stride of a chunk (odd: the rows start on different banks)
constexpr int kVifStageRows = 16;     // rows of a chunk whose loads a lane keeps in flight
typedef double vif_double4v __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int vif_staged_rows(int m, int nt) { return nt > 0 ? 16 * nt : m + 1; }
template <int COV, int T, int NT>
__device__ __forceinline__ int vif_point_setup(const VecchiaKernelArgs& args, const double* __restrict__ V, int kip, int kq, int ld, int i,
                                               double* s_C, double* s_V, double* s_c, int* s_idx, double4* s_pts, double4& ctr, double4& own, double& gii,
                                               double& dinv) {
  static_assert(NT == 0 || T == 64, "the MFMA Gram path is the one-wavefront form");
  const int m = args.m, tid = threadIdx.x;
  const int idx = tid < m ? args.nn[(size_t)i * m + tid] : -1;
  const int k = __builtin_amdgcn_readfirstlane(__syncthreads_count(idx >= 0));   // the valid neighbours are a prefix of the row (short rows: i < m)
  s_idx[tid] = tid < k ? idx : (tid == k ? i : -1);      // row k of the staged block is the point itself
  dinv = 1.0;
  const int npair = (k + 1) * (k + 2) / 2;
  if constexpr (NT == 0) {
    for (int p = tid; p < npair; p += T) {               // zero the Gram accumulators (pair p -> (r, c <= r))
      int r = (int)((__fsqrt_rn(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);   // p < 2^13: exact up to the two corrections below
      while (r * (r + 1) / 2 > p) --r;
      while ((r + 1) * (r + 2) / 2 <= p) ++r;
      s_C[r * ld + (p - r * (r + 1) / 2)] = 0.0;
    }
  }
  __syncthreads();
  ctr = args.pts[i];
  own = args.pts[tid < k ? idx : i];                       // (an index select: a select between two double4 objects goes through the stack)
  constexpr int kAcc = NT > 0 ? NT * (NT + 1) / 2 : 1;
  vif_double4v acc[kAcc];
#pragma unroll
  for (int q = 0; q < kAcc; ++q) acc[q] = (vif_double4v){0.0, 0.0, 0.0, 0.0};
  const int cl = tid & 63, rp = tid >> 6;               // staging: lane = column of the chunk; T = 128: two rows per pass
  constexpr int RP = T / 64;
  const int rows_staged = NT > 0 ? 16 * NT : k + 1;      // NT > 0: the tiles' rows beyond k are written as zeros
  for (int c0 = 0; c0 < kip; c0 += kVifKC) {
    const int cw = min(kVifKC, kip - c0);
    for (int r0 = 0; r0 * RP < rows_staged; r0 += kVifStageRows) {
      double v[kVifStageRows];
#pragma unroll
      for (int j = 0; j < kVifStageRows; ++j) {
        const int r = (r0 + j) * RP + rp;
        v[j] = (r <= k && cl < cw) ? V[(size_t)s_idx[r] * kq + c0 + cl] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < kVifStageRows; ++j) {
        const int r = (r0 + j) * RP + rp;
        if (r < rows_staged) s_V[r * kVifKCP + cl] = v[j];
      }
    }
    __syncthreads();
    if constexpr (NT > 0) {
      const int fr = tid & 15, fk = tid >> 4;
      const int ksteps = (cw + 3) >> 2;                  // columns cw .. 63 of the chunk hold zeros
      for (int kk = 0; kk < ksteps; ++kk) {
        double a[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) a[q] = s_V[(16 * q + fr) * kVifKCP + 4 * kk + fk];
#pragma unroll
        for (int mi = 0; mi < NT; ++mi)
#pragma unroll
          for (int nj = 0; nj <= mi; ++nj)
            acc[mi * (mi + 1) / 2 + nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], a[nj], acc[mi * (mi + 1) / 2 + nj], 0, 0, 0);
      }
    } else {
      for (int p = tid; p < npair; p += T) {
        int r = (int)((__fsqrt_rn(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > p) --r;
        while ((r + 1) * (r + 2) / 2 <= p) ++r;
        const int c = p - r * (r + 1) / 2;
        const double* vr = s_V + r * kVifKCP;
        const double* vc = s_V + c * kVifKCP;
        double sum = s_C[r * ld + c];
        for (int j = 0; j < cw; ++j) sum = __builtin_fma(vr[j], vc[j], sum);
        s_C[r * ld + c] = sum;
      }
    }
    __syncthreads();
  }
  if constexpr (NT > 0) {                                // accumulator entry r of lane (fr, fk) of tile (mi, nj): G[16 mi + fk + 4 r][16 nj + fr]
    const int fr = tid & 15, fk = tid >> 4;
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
      for (int nj = 0; nj <= mi; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = 16 * mi + fk + 4 * r, gj = 16 * nj + fr;
          if (gj <= gi && gi <= k) s_C[gi * ld + gj] = acc[mi * (mi + 1) / 2 + nj][r];
        }
    __syncthreads();
  }
  gii = s_C[k * ld + k];
  if constexpr (NT > 0) {
    // C_nn row by row in registers (lane r = row r; the staged points come from LDS), factorised there, then the factor goes to s_C for the
    // solves that need its columns (vif_load_cols) and for the derivative kernel
    constexpr int R = 16 * NT;
    s_pts[tid] = own;                                          // lanes >= k hold the point itself (row k of the staged block)
    __syncthreads();
    double row[R];
    const int rl = min(tid, k);
    vif_sfor<0, R>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      double v = 0.0;
      if (c < k) {
        const double g = s_C[rl * ld + c];
        const double kv = matern_plain<COV>(dist4(own, s_pts[c]), args.var, args.a);
        v = (c == tid ? args.diag_nn : kv) - g;               // diagonal: var + nugget - |V_a|^2
        if (c == tid) v *= args.diag_mult;
      }
      row[c] = tid < k ? v : (c == tid ? 1.0 : 0.0);
    });
    const double cv = tid < k ? matern_plain<COV>(dist4(own, ctr), args.var, args.a) - s_C[k * ld + rl] : 0.0;
    s_c[tid] = cv;
    vif_reg_cholesky<R>(row, k, tid, dinv);
    __syncthreads();                                           // (every lane has read its Gram row)
    vif_sfor<0, R>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      if (c < k && c <= tid && tid < k) s_C[tid * ld + c] = row[c];
    });
    __syncthreads();
    return k;
  }
  if (tid < k) {
    for (int q = 0; q < tid; ++q) s_C[tid * ld + q] = matern_plain<COV>(dist4(own, args.pts[s_idx[q]]), args.var, args.a) - s_C[tid * ld + q];
    s_c[tid] = matern_plain<COV>(dist4(own, ctr), args.var, args.a) - s_C[k * ld + tid];
  }
  __syncthreads();
  if (tid < k) s_C[tid * ld + tid] = (args.diag_nn - s_C[tid * ld + tid]) * args.diag_mult;        // (var + nugget - |V_a|^2) x the latent jitter
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    if (tid == j) s_C[j * ld + j] = sqrt(s_C[j * ld + j]);
    __syncthreads();
    if (tid > j && tid < k) s_C[tid * ld + j] /= s_C[j * ld + j];
    __syncthreads();
    if (tid > j && tid < k) {
      const double lj = s_C[tid * ld + j];
      for (int c = j + 1; c <= tid; ++c) s_C[tid * ld + c] = __builtin_fma(-lj, s_C[c * ld + j], s_C[tid * ld + c]);
    }
    __syncthreads();
  }
  return k;
}

// L L' x = b for two right-hand sides at once (s_z1, s_z2: in b, out x), L = the factor in s_C; all 128 lanes take part (barriers)
__device__ __forceinline__ void vif_chol_solve2(const double* s_C, int ld, int k, double* s_z1, double* s_z2) {
  const int tid = threadIdx.x;
  for (int j = 0; j < k; ++j) {
    if (tid == j) { const double inv = 1.0 / s_C[j * ld + j]; s_z1[j] *= inv; s_z2[j] *= inv; }
    __syncthreads();
    if (tid > j && tid < k) {
      const double l = s_C[tid * ld + j];
      s_z1[tid] = __builtin_fma(-l, s_z1[j], s_z1[tid]);
      s_z2[tid] = __builtin_fma(-l, s_z2[j], s_z2[tid]);
    }
    __syncthreads();
  }
  for (int j = k - 1; j >= 0; --j) {
    if (tid == j) { const double inv = 1.0 / s_C[j * ld + j]; s_z1[j] *= inv; s_z2[j] *= inv; }
    __syncthreads();
    if (tid < j) {
      const double l = s_C[j * ld + tid];
      s_z1[tid] = __builtin_fma(-l, s_z1[j], s_z1[tid]);
      s_z2[tid] = __builtin_fma(-l, s_z2[j], s_z2[tid]);
    }
    __syncthreads();
  }
}

// Residual-process factor: D_i = var + nugget - G[i][i] - A_i . c,  A_i = C_nn^-1 c,  u_i = y_i - A_i . y_nn.  Outputs as MODE_FACTOR:
// A [n][m], D [n], u [n], and the three partial sums {log D_i, u_i^2 / D_i, D_i <= 0} per point.
template <int COV, int T, int NT>
__global__ __launch_bounds__(T) void vif_resid_factor_kernel(VecchiaKernelArgs args, const double* __restrict__ V, int kip, int kq, int ld) {
  extern __shared__ double s_dyn[];
  const int m = args.m;
  double* s_C = s_dyn;                                   // [m + 1][ld]: Gram matrix of the staged rows, then C_nn and its factor (row m: the point)
  double* s_V = s_dyn + (size_t)(m + 1) * ld;            // [vif_staged_rows(m, NT)][kVifKCP]: one chunk of the whitened rows
  __shared__ double s_c[T], s_z1[T], s_z2[T], s_red[T];
  __shared__ int s_idx[T];
  __shared__ double4 s_pts[NT > 0 ? T : 1];
  const int tid = threadIdx.x;
  const int i = args.i_begin + blockIdx.x;
  double4 ctr, own; double gii, dinv;
  const int k = vif_point_setup<COV, T, NT>(args, V, kip, kq, ld, i, s_C, s_V, s_c, s_idx, s_pts, ctr, own, gii, dinv);
  if constexpr (NT > 0) {                                // the solves in registers (see vif_reg_cholesky)
    constexpr int R = 16 * NT;
    double reg[R];
    vif_load_rows<R>(s_C, ld, k, tid, reg);
    double z1 = tid < k ? s_c[tid] : 0.0, z2 = tid < k ? own.w : 0.0;
    vif_reg_forward2<R>(reg, k, tid, dinv, z1, z2);      // L z1 = c, L z2 = y_nn
    if (tid >= k) { z1 = 0.0; z2 = 0.0; }
    const double Dv = args.diag_i - gii - vif_wave_sum(z1 * z1);      // D_i (Vecchia_utils.cpp:1463-1465, :1623)
    const double uv = ctr.w - vif_wave_sum(z1 * z2);                  // u_i = (B y)_i
    vif_load_cols<R>(s_C, ld, k, tid, reg);
    double unused = 0.0;
    vif_reg_backward<R, false>(reg, k, tid, dinv, z1, unused);        // L' A = z1
    if (tid < m) args.A[(size_t)i * m + tid] = tid < k ? z1 : 0.0;
    if (tid == 0) {
      args.D[i] = Dv; args.u[i] = uv;
      const size_t nb = gridDim.x;
      args.partials[(size_t)GPB_P_LOGDET * nb + blockIdx.x] = log(Dv);
      args.partials[(size_t)GPB_P_QUAD * nb + blockIdx.x] = uv * uv / Dv;
      args.partials[(size_t)GPB_P_BAD * nb + blockIdx.x] = (Dv > 0.0) ? 0.0 : 1.0;
    }
    return;
  }
  s_z1[tid] = tid < k ? s_c[tid] : 0.0;
  s_z2[tid] = tid < k ? own.w : 0.0;
  __syncthreads();
  for (int j = 0; j < k; ++j) {                          // L z1 = c, L z2 = y_nn
    if (tid == j) { const double inv = 1.0 / s_C[j * ld + j]; s_z1[j] *= inv; s_z2[j] *= inv; }
    __syncthreads();
    if (tid > j && tid < k) {
      const double l = s_C[tid * ld + j];
      s_z1[tid] = __builtin_fma(-l, s_z1[j], s_z1[tid]);
      s_z2[tid] = __builtin_fma(-l, s_z2[j], s_z2[tid]);
    }
    __syncthreads();
  }
  const double z1 = tid < k ? s_z1[tid] : 0.0, z2 = tid < k ? s_z2[tid] : 0.0;
  s_red[tid] = z1 * z1; s_c[tid] = z1 * z2;
  __syncthreads();
  // fixed tree over 128 slots whatever T (T = 64: the upper half is zero): the same additions in the same order as the 128-lane form
  for (int w = 64; w >= 1; w >>= 1) {
    if (tid < w) {
      const double a2 = (tid + w < T) ? s_red[tid + w] : 0.0, b2 = (tid + w < T) ? s_c[tid + w] : 0.0;
      s_red[tid] += a2; s_c[tid] += b2;
    }
    __syncthreads();
  }
  const double Dv = args.diag_i - gii - s_red[0];        // D_i (Vecchia_utils.cpp:1463-1465, :1623)
  const double uv = ctr.w - s_c[0];                      // u_i = (B y)_i
  __syncthreads();
  for (int j = k - 1; j >= 0; --j) {                     // L' A = z1
    if (tid == j) s_z1[j] /= s_C[j * ld + j];
    __syncthreads();
    if (tid < j) s_z1[tid] = __builtin_fma(-s_C[j * ld + tid], s_z1[j], s_z1[tid]);
    __syncthreads();
  }
  if (tid < m) args.A[(size_t)i * m + tid] = tid < k ? s_z1[tid] : 0.0;
  if (tid == 0) {
    args.D[i] = Dv; args.u[i] = uv;
    const size_t nb = gridDim.x;
    args.partials[(size_t)GPB_P_LOGDET * nb + blockIdx.x] = log(Dv);
    args.partials[(size_t)GPB_P_QUAD * nb + blockIdx.x] = uv * uv / Dv;
    args.partials[(size_t)GPB_P_BAD * nb + blockIdx.x] = (Dv > 0.0) ? 0.0 : 1.0;
  }
}

// Q[i][:] = X[i][:] - sum_j A[i][j] X[nn[i][j]][:]   (B X for the row-major n x kq matrix X; TWO: the same for a second matrix with the
// same factor row -- C and dC share the index / coefficient loads).  One workgroup = 8 rows, 32 lanes per row striding over the columns.
template <bool TWO>
__global__ __launch_bounds__(256) void vif_spmm_kernel(const double* __restrict__ A, const int* __restrict__ nn, int i0, int i1, int m, int kq,
                                                       const double* __restrict__ X, double* __restrict__ Q, const double* __restrict__ X2, double* __restrict__ Q2) {
  __shared__ double s_a[8][128];
  __shared__ int s_n[8][128];
  const int r = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const int i = i0 + blockIdx.x * 8 + r;
  const bool live = i < i1;
  for (int j = lane; j < m; j += 32) {
    s_n[r][j] = live ? nn[(size_t)i * m + j] : -1;
    s_a[r][j] = live ? A[(size_t)i * m + j] : 0.0;
  }
  __syncthreads();
  if (!live) return;
  for (int c = lane; c < kq; c += 32) {
    double acc = X[(size_t)i * kq + c], acc2 = 0.0;
    if constexpr (TWO) acc2 = X2[(size_t)i * kq + c];
    for (int j = 0; j < m; ++j) {
      const int s = s_n[r][j];
      if (s < 0) break;                                  // the valid neighbours are a prefix of the row
      const double av = s_a[r][j];
      acc = __builtin_fma(-av, X[(size_t)s * kq + c], acc);
      if constexpr (TWO) acc2 = __builtin_fma(-av, X2[(size_t)s * kq + c], acc2);
    }
    Q[(size_t)i * kq + c] = acc;
    if constexpr (TWO) Q2[(size_t)i * kq + c] = acc2;
  }
}

// Gram matrix G = Q' D^-1 Q of the first kq columns (k + 1 of them meaningful: the Woodbury part, (B C)' D^-1 u and u' D^-1 u), tiled:
// workgroup (tile pair (ti >= tj), row chunk) accumulates its 64 x 64 tile over the chunk's rows, 4 x 4 entries per thread, 16 rows per LDS
// step; part[chunk][ti][tj][64][64].  vif_gram_sum_kernel adds the chunks in a fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void vif_gram_kernel(const double* __restrict__ Q, const double* __restrict__ D, int n, int kq, int nt, int rows_per_chunk,
                                                       double* __restrict__ part) {
  __shared__ double sI[16][65], sJ[16][65];
  int t = blockIdx.x, ti = 0;
  while (t >= ti + 1) { t -= ti + 1; ++ti; }            // pair index -> (ti, tj = t), tj <= ti
  const int tj = t;
  const int chunk = blockIdx.y;
  const int r0 = chunk * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  for (int rb = r0; rb < r1; rb += 16) {
#pragma unroll
    for (int e = tid; e < 16 * 64; e += 256) {
      const int rr = e >> 6, c = e & 63;
      const int gi = rb + rr;
      double vi = 0.0, vj = 0.0;
      if (gi < r1) {
        const double dinv = 1.0 / D[gi];
        if (ti * 64 + c < kq) vi = Q[(size_t)gi * kq + ti * 64 + c] * dinv;
        if (tj * 64 + c < kq) vj = Q[(size_t)gi * kq + tj * 64 + c];
      }
      sI[rr][c] = vi; sJ[rr][c] = vj;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      double av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = sI[rr][tr + 16 * r];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = sJ[rr][tc + 16 * c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fma(av[r], bv[c], acc[r][c]);
    }
    __syncthreads();
  }
  double* out = part + ((size_t)chunk * (nt * (nt + 1) / 2) + blockIdx.x) * 4096;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) out[(tr + 16 * r) * 64 + tc + 16 * c] = acc[r][c];
}
// G[a][b] = G[b][a] = sum over the chunks (ascending) of the tile entries; G: [kq][kq] row-major
__global__ __launch_bounds__(256) void vif_gram_sum_kernel(const double* __restrict__ part, int nchunks, int nt, int kq, double* __restrict__ G) {
  const int pair = blockIdx.x;
  int t = pair, ti = 0;
  while (t >= ti + 1) { t -= ti + 1; ++ti; }
  const int tj = t;
  const int npairs = nt * (nt + 1) / 2;
  for (int e = threadIdx.x; e < 4096; e += 256) {
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += part[((size_t)c * npairs + pair) * 4096 + e];
    const int a = ti * 64 + (e >> 6), b = tj * 64 + (e & 63);
    if (a < kq && b < kq) {
      G[(size_t)a * kq + b] = s;
      if (ti != tj) G[(size_t)b * kq + a] = s;
    }
  }
}

// per row, 16 lanes: v_i = (u_i - Q_i . w) / D_i  (u_i = Q[i][k]),  z_i = y_i - C_i . w  (y_i = C[i][k]);  w: kq doubles, zero from k on
__global__ __launch_bounds__(256) void vif_vec_kernel(const double* __restrict__ Q, const double* __restrict__ C, const double* __restrict__ D,
                                                      const double* __restrict__ w, int n, int k, int kq, double* __restrict__ v, double* __restrict__ z) {
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, lane = threadIdx.x & 15;
  const int ii = i < n ? i : n - 1;
  double sq = 0.0, sc = 0.0;
  for (int c = lane; c < k; c += 16) {
    const double wc = w[c];
    sq = __builtin_fma(Q[(size_t)ii * kq + c], wc, sq);
    sc = __builtin_fma(C[(size_t)ii * kq + c], wc, sc);
  }
  sq = sum16(sq); sc = sum16(sc);
  if (lane == 0 && i < n) {
    v[i] = (Q[(size_t)i * kq + k] - sq) / D[i];
    z[i] = C[(size_t)i * kq + k] - sc;
  }
}

// ---- derivative of the residual-process factor and the per-point terms of the gradient ----------------------------------------------
// For parameter p in {variance, range} (log scale) and Atilde = (A_i, -1) over (neighbours, point):
//   h^p_a = sum_b dK^p_ab Atilde_b + dc^p_a . X1_i + c_a . X2^p_i          [the second and third term: minus the derivative of the
//                                                                            predictive-process part V_a . V_b contracted with Atilde,
//                                                                            Vecchia_utils.cpp:1511-1516 -- X1 = Q Si, X2^p = (B dC^p) Si - Q Si dSm^p Si]
//   dA^p_i = -C_nn^-1 h^p[nn]   (:1640-1641),   dD^p_i = A_i . h^p[nn] - h^p[i]   (:1646-1655, 1668-1679)
// with dK^0 = K (no nugget), dC^0 = C, so the variance parameter reads   c_a . V1_i,  V1 = X1 + X2^0.   The per-point sums of
// CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i (re_model_template.h:2284-2317, 2447-2452), rewritten as row-wise dot products:
//   T_S1 dD/D | T_S2 2 (x . z_nn) v - v^2 dD | T_S3 (x . g)/D, g_a = c_a . Hm_i | T_S4 dD kappa / D^2, kappa = Q_i . Hm_i | T_S5 (B dC^p)_i . Hm_i / D
//   | T_S6 v (B dC^p)_i . w      (x = C_nn^-1 h = -dA; Hm = Q W^-1; v = D^-1 B (y - C w); z = y - C w)
// partials: [12][npts] = {S1..S6} x {variance, range}.  Optional outputs dA [2][n][m], dD [2][n] (tests).
struct VifGradArgs {
  const double* V; const double* C; const double* dC; const double* Q; const double* QdC;
  const double* X1; const double* V1; const double* X2r; const double* Hm;
  const double* w; const double* v; const double* z;
  double* dA0; double* dA1; double* dD0; double* dD1;
  double* partials;
  int kip, kq, kp, ld;
};
enum : int { VIF_S1 = 0, VIF_S2 = 1, VIF_S3 = 2, VIF_S4 = 3, VIF_S5 = 4, VIF_S6 = 5 };

template <int COV, int T, int NT>
__global__ __launch_bounds__(T) void vif_resid_grad_kernel(VecchiaKernelArgs args, VifGradArgs g) {
  extern __shared__ double s_dyn[];
  const int m = args.m, ld = g.ld, kq = g.kq, kip = g.kip;
  double* s_C = s_dyn;
  double* s_V = s_dyn + (size_t)(m + 1) * ld;            // [vif_staged_rows(m, NT)][kVifKCP] chunks of V during the set-up; afterwards the five k-vectors of the point
  __shared__ double s_c[T], s_h0[T], s_h1[T], s_at[T], s_g[T], s_red[T];
  __shared__ double s_pk[4][T];                          // partial sums of the kernel-derivative contraction: [sub][row] for K and dK
  __shared__ double s_pk2[4][T];
  __shared__ double s_self[4];
  __shared__ int s_idx[T];
  __shared__ double4 s_pts[NT > 0 ? T : 1];
  const int tid = threadIdx.x;
  const int i = args.i_begin + blockIdx.x;
  double4 ctr, own; double gii, dinv;
  const int k = vif_point_setup<COV, T, NT>(args, g.V, kip, kq, ld, i, s_C, s_V, s_c, s_idx, s_pts, ctr, own, gii, dinv);
  // Atilde = (A_i, -1)
  s_at[tid] = tid < k ? args.A[(size_t)i * m + tid] : (tid == k ? -1.0 : 0.0);
  // the point's k-vectors (after the set-up the staged rows of V are dead): X1_i, V1_i, X2r_i, Hm_i, w
  double* s_x1 = s_V; double* s_v1 = s_V + kq; double* s_x2 = s_V + 2 * kq; double* s_hm = s_V + 3 * kq; double* s_w = s_V + 4 * kq;
  for (int c = tid; c < kq; c += T) {
    s_x1[c] = g.X1[(size_t)i * kq + c]; s_v1[c] = g.V1[(size_t)i * kq + c]; s_x2[c] = g.X2r[(size_t)i * kq + c];
    s_hm[c] = g.Hm[(size_t)i * kq + c]; s_w[c] = g.w[c];
  }
  __syncthreads();
  // ---- sum_b K_ab Atilde_b and sum_b dK_ab Atilde_b, rows a = 0..k: LPR lanes per row, each a strided part of the columns ----------
  const int rows = k + 1;
  const int lpr = rows <= T / 4 ? 4 : (rows <= T / 2 ? 2 : 1);
  {
    const int a = tid / lpr, sub = tid - a * lpr;
    double sk = 0.0, sd = 0.0;
    if (a < rows) {
      const double4 pa = NT > 0 ? s_pts[a] : args.pts[s_idx[a]];      // row k is the point itself (s_idx[k] = i, s_pts[k] = the point)
      for (int b = sub; b < rows; b += lpr) {
        const double4 pb = NT > 0 ? s_pts[b] : args.pts[s_idx[b]];
        double kv, dk;
        matern_with_grad<COV>(dist4(pa, pb), args.var, args.a, kv, dk);
        const double at = s_at[b];
        sk = __builtin_fma(kv, at, sk); sd = __builtin_fma(dk, at, sd);
      }
    }
    if (a < rows) { s_pk[sub][a] = sk; s_pk2[sub][a] = sd; }
    __syncthreads();
  }
  // ---- low-rank parts: 16 lanes per staged row (8 rows per pass), row a <= k: c_a . V1_i, dc_a . X1_i + c_a . X2r_i, c_a . Hm_i ---------
  for (int a0 = 0; a0 < rows; a0 += T / 16) {
    const int a = a0 + (tid >> 4), lane = tid & 15;
    const bool live = a < rows;
    const size_t src = (size_t)s_idx[live ? a : 0] * kq;
    double t0 = 0.0, t1 = 0.0, tg = 0.0;
    for (int c0 = lane; c0 < kip; c0 += 64) {            // four columns per lane with their eight loads in flight (same order of the fma chains)
      double cv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        const bool ok = live && c < kip;
        cv[u] = ok ? g.C[src + c] : 0.0; dv[u] = ok ? g.dC[src + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        if (c < kip) {
          t0 = __builtin_fma(cv[u], s_v1[c], t0);
          t1 = __builtin_fma(dv[u], s_x1[c], __builtin_fma(cv[u], s_x2[c], t1));
          tg = __builtin_fma(cv[u], s_hm[c], tg);
        }
      }
    }
    t0 = sum16(t0); t1 = sum16(t1); tg = sum16(tg);
    if (live && lane == 0) {
      double k0 = 0.0, k1 = 0.0;
      for (int s = 0; s < lpr; ++s) { k0 += s_pk[s][a]; k1 += s_pk2[s][a]; }
      s_h0[a] = k0 + t0; s_h1[a] = k1 + t1; s_g[a] = tg;
    }
  }
  // the point's own row-wise dot products: kappa = Q_i . Hm_i, (B dC)_i . Hm_i, Q_i . w, (B dC)_i . w   (first 16 lanes)
  if (tid < 16) {
    double q_hm = 0.0, d_hm = 0.0, q_w = 0.0, d_w = 0.0;
    for (int c0 = tid; c0 < kip; c0 += 64) {
      double qv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        qv[u] = c < kip ? g.Q[(size_t)i * kq + c] : 0.0; dv[u] = c < kip ? g.QdC[(size_t)i * kq + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u;
        if (c < kip) {
          q_hm = __builtin_fma(qv[u], s_hm[c], q_hm); d_hm = __builtin_fma(dv[u], s_hm[c], d_hm);
          q_w = __builtin_fma(qv[u], s_w[c], q_w); d_w = __builtin_fma(dv[u], s_w[c], d_w);
        }
      }
    }
    q_hm = sum16(q_hm); d_hm = sum16(d_hm); q_w = sum16(q_w); d_w = sum16(d_w);
    if (tid == 0) { s_self[0] = q_hm; s_self[1] = d_hm; s_self[2] = q_w; s_self[3] = d_w; }
  }
  __syncthreads();
  const double h0_self = s_h0[k], h1_self = s_h1[k];
  const double a_own = tid < k ? s_at[tid] : 0.0;
  const double h0_own = tid < k ? s_h0[tid] : 0.0, h1_own = tid < k ? s_h1[tid] : 0.0;
  const double g_own = tid < k ? s_g[tid] : 0.0;
  const double z_own = tid < k ? g.z[s_idx[tid]] : 0.0;
  double x0, x1;                                          // x^p = C_nn^-1 h^p[nn]
  if constexpr (NT > 0) {
    constexpr int R = 16 * NT;
    double reg[R];
    vif_load_rows<R>(s_C, ld, k, tid, reg);
    x0 = h0_own; x1 = h1_own;
    vif_reg_forward2<R>(reg, k, tid, dinv, x0, x1);
    vif_load_cols<R>(s_C, ld, k, tid, reg);
    vif_reg_backward<R, true>(reg, k, tid, dinv, x0, x1);
    if (tid >= k) { x0 = 0.0; x1 = 0.0; }
  } else {
    __syncthreads();
    if (tid >= k) { s_h0[tid] = 0.0; s_h1[tid] = 0.0; }
    __syncthreads();
    vif_chol_solve2(s_C, ld, k, s_h0, s_h1);
    x0 = tid < k ? s_h0[tid] : 0.0; x1 = tid < k ? s_h1[tid] : 0.0;
  }
  if (g.dA0 && tid < m) { g.dA0[(size_t)i * m + tid] = -x0; g.dA1[(size_t)i * m + tid] = -x1; }
  // six sums over the neighbours in one tree: A . h^p, x^p . z_nn, x^p . g   (s_pk / s_pk2 are dead by now)
  __syncthreads();
  s_pk[0][tid] = a_own * h0_own; s_pk[1][tid] = a_own * h1_own; s_pk[2][tid] = x0 * z_own; s_pk[3][tid] = x1 * z_own;
  s_pk2[0][tid] = x0 * g_own; s_pk2[1][tid] = x1 * g_own;
  __syncthreads();
  for (int w = 64; w >= 1; w >>= 1) {                   // fixed tree over 128 slots whatever T (T = 64: the upper half is zero)
    if (tid < w && tid + w < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s_pk[q][tid] += s_pk[q][tid + w];
      s_pk2[0][tid] += s_pk2[0][tid + w]; s_pk2[1][tid] += s_pk2[1][tid + w];
    }
    __syncthreads();
  }
  const double tot0 = s_pk[0][0], tot1 = s_pk[1][0], tot2 = s_pk[2][0], tot3 = s_pk[3][0], tot4 = s_pk2[0][0], tot5 = s_pk2[1][0];
  if (tid == 0) {
    const double Di = args.D[i], di = 1.0 / Di, vi = g.v[i];
    const double dD0 = tot0 - h0_self, dD1 = tot1 - h1_self;
    if (g.dD0) { g.dD0[i] = dD0; g.dD1[i] = dD1; }
    const double kappa = s_self[0];
    const size_t nb = gridDim.x, b = blockIdx.x;
    double* P = g.partials;
    P[(size_t)(2 * VIF_S1 + 0) * nb + b] = dD0 * di;                       P[(size_t)(2 * VIF_S1 + 1) * nb + b] = dD1 * di;
    P[(size_t)(2 * VIF_S2 + 0) * nb + b] = 2.0 * tot2 * vi - vi * vi * dD0; P[(size_t)(2 * VIF_S2 + 1) * nb + b] = 2.0 * tot3 * vi - vi * vi * dD1;
    P[(size_t)(2 * VIF_S3 + 0) * nb + b] = tot4 * di;                    P[(size_t)(2 * VIF_S3 + 1) * nb + b] = tot5 * di;
    P[(size_t)(2 * VIF_S4 + 0) * nb + b] = dD0 * di * di * kappa;          P[(size_t)(2 * VIF_S4 + 1) * nb + b] = dD1 * di * di * kappa;
    P[(size_t)(2 * VIF_S5 + 0) * nb + b] = kappa * di;                     P[(size_t)(2 * VIF_S5 + 1) * nb + b] = s_self[1] * di;
    P[(size_t)(2 * VIF_S6 + 0) * nb + b] = vi * s_self[2];                 P[(size_t)(2 * VIF_S6 + 1) * nb + b] = vi * s_self[3];
  }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------
hipError_t launch_vif_crosscov(int cov, const double4* pts, const double* ip, int i0, int i1, int k, int kq, int d, double var, double a, double* C, double* dC,
                               hipStream_t st) {
  if (i1 <= i0) return hipSuccess;
  const dim3 grid((unsigned)(((size_t)(i1 - i0) * kq + 255) / 256));
#define GPB_VIF_CC(C_)                                                                                                               \
  do {                                                                                                                               \
    if (dC) hipLaunchKernelGGL((vif_crosscov_kernel<C_, true>), grid, dim3(256), 0, st, pts, ip, i0, i1, k, kq, d, var, a, C, dC);    \
    else hipLaunchKernelGGL((vif_crosscov_kernel<C_, false>), grid, dim3(256), 0, st, pts, ip, i0, i1, k, kq, d, var, a, C, dC);      \
  } while (0)
  switch (cov) {
    case kMatern05: GPB_VIF_CC(kMatern05); break;
    case kMatern15: GPB_VIF_CC(kMatern15); break;
    case kMatern25: GPB_VIF_CC(kMatern25); break;
    default: return hipErrorInvalidValue;
  }
#undef GPB_VIF_CC
  return hipGetLastError();
}
hipError_t launch_vif_gemm(const double* In, const double* M, int n, int kq, double* Out, bool accumulate, hipStream_t st) {
  const dim3 grid((n + 63) / 64, (kq + 63) / 64);
  if (accumulate) hipLaunchKernelGGL(vif_gemm_kernel<true>, grid, dim3(256), 0, st, In, M, n, kq, Out);
  else hipLaunchKernelGGL(vif_gemm_kernel<false>, grid, dim3(256), 0, st, In, M, n, kq, Out);
  return hipGetLastError();
}
hipError_t launch_vif_spmm(const double* A, const int* nn, int i0, int i1, int m, int kq, const double* X, double* Q, const double* X2, double* Q2, hipStream_t st) {
  if (i1 <= i0) return hipSuccess;
  if (m > 128) return hipErrorInvalidValue;
  const dim3 grid((i1 - i0 + 7) / 8);
  if (X2) hipLaunchKernelGGL(vif_spmm_kernel<true>, grid, dim3(256), 0, st, A, nn, i0, i1, m, kq, X, Q, X2, Q2);
  else hipLaunchKernelGGL(vif_spmm_kernel<false>, grid, dim3(256), 0, st, A, nn, i0, i1, m, kq, X, Q, X2, Q2);
  return hipGetLastError();
}
int vif_gram_chunks(int n, int kq) {
  const int nt = (kq + 63) / 64, pairs = nt * (nt + 1) / 2;
  int chunks = (1024 + pairs - 1) / pairs;                        // ~ four workgroups per CU
  const int maxc = (n + 255) / 256;                               // at least 256 rows per chunk
  if (chunks > maxc) chunks = maxc;
  return chunks < 1 ? 1 : chunks;
}
size_t vif_gram_part_doubles(int n, int kq) {
  const int nt = (kq + 63) / 64;
  return (size_t)vif_gram_chunks(n, kq) * (nt * (nt + 1) / 2) * 4096;
}
hipError_t launch_vif_gram(const double* Q, const double* D, int n, int kq, double* part, double* G, hipStream_t st) {
  const int nt = (kq + 63) / 64, pairs = nt * (nt + 1) / 2, chunks = vif_gram_chunks(n, kq);
  const int rpc = (((n + chunks - 1) / chunks) + 15) / 16 * 16;
  hipLaunchKernelGGL(vif_gram_kernel, dim3(pairs, chunks), dim3(256), 0, st, Q, D, n, kq, nt, rpc, part);
  hipLaunchKernelGGL(vif_gram_sum_kernel, dim3(pairs), dim3(256), 0, st, part, chunks, nt, kq, G);
  return hipGetLastError();
}
hipError_t launch_vif_vec(const double* Q, const double* C, const double* D, const double* w, int n, int k, int kq, double* v, double* z, hipStream_t st) {
  hipLaunchKernelGGL(vif_vec_kernel, dim3((unsigned)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, st, Q, C, D, w, n, k, kq, v, z);
  return hipGetLastError();
}
static int vif_tiles(int m) { return m <= 62 ? (m + 1 + 15) / 16 : 0; }      // NT of the per-point kernels: 1..4 (one wavefront, MFMA Gram) or 0
size_t vif_resid_lds_bytes(int m, int kq_grad) {      // kq_grad > 0: the derivative kernel (its five k-vectors alias the chunk buffer)
  const size_t chunk = (size_t)vif_staged_rows(m, vif_tiles(m)) * kVifKCP, vecs = 5 * (size_t)kq_grad;
  return sizeof(double) * ((size_t)(m + 1) * ((m + 1) | 1) + (chunk > vecs ? chunk : vecs));
}
hipError_t launch_vif_resid_factor(int cov, const VecchiaKernelArgs& args, const double* V, int kip, int kq, hipStream_t st) {
  const int npts = args.i_end - args.i_begin;
  if (npts <= 0 || args.m < 1 || args.m > GPB_MAX_NEIGHBORS_BIG) return hipErrorInvalidValue;
  const int ld = (args.m + 1) | 1;
  const size_t lds = vif_resid_lds_bytes(args.m, 0);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
#define GPB_VIF_LAUNCH_T(C_, T_, NT_)                                                                                                   \
  do {                                                                                                                              \
    auto kern = vif_resid_factor_kernel<C_, T_, NT_>;                                                                                   \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e_ != hipSuccess) return e_;                                                                                                \
    hipLaunchKernelGGL(kern, dim3(npts), dim3(T_), lds, st, args, V, kip, kq, ld);                                                  \
  } while (0)
#define GPB_VIF_LAUNCH(C_)                                                                                                           \
  do {                                                                                                                              \
    switch (vif_tiles(args.m)) {                                                                                                    \
      case 1: GPB_VIF_LAUNCH_T(C_, 64, 1); break;                                                                                   \
      case 2: GPB_VIF_LAUNCH_T(C_, 64, 2); break;                                                                                   \
      case 3: GPB_VIF_LAUNCH_T(C_, 64, 3); break;                                                                                   \
      case 4: GPB_VIF_LAUNCH_T(C_, 64, 4); break;                                                                                   \
      default: GPB_VIF_LAUNCH_T(C_, 128, 0); break;                                                                                 \
    }                                                                                                                               \
  } while (0)
  switch (cov) {
    case kMatern05: GPB_VIF_LAUNCH(kMatern05); break;
    case kMatern15: GPB_VIF_LAUNCH(kMatern15); break;
    case kMatern25: GPB_VIF_LAUNCH(kMatern25); break;
    default: return hipErrorInvalidValue;
  }
#undef GPB_VIF_LAUNCH
#undef GPB_VIF_LAUNCH_T
  return hipGetLastError();
}
hipError_t launch_vif_resid_grad(int cov, const VecchiaKernelArgs& args, const VifGradLaunch& L, int kip, int kq, hipStream_t st) {
  const int npts = args.i_end - args.i_begin;
  if (npts <= 0 || args.m < 1 || args.m > GPB_MAX_NEIGHBORS_BIG) return hipErrorInvalidValue;
  const int ld = (args.m + 1) | 1;
  const size_t lds = vif_resid_lds_bytes(args.m, kq);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  VifGradArgs g;
  g.V = L.V; g.C = L.C; g.dC = L.dC; g.Q = L.Q; g.QdC = L.QdC; g.X1 = L.X1; g.V1 = L.V1; g.X2r = L.X2r; g.Hm = L.Hm; g.w = L.w; g.v = L.v; g.z = L.z;
  g.dA0 = L.dA0; g.dA1 = L.dA1; g.dD0 = L.dD0; g.dD1 = L.dD1; g.partials = L.partials; g.kip = kip; g.kq = kq; g.kp = 0; g.ld = ld;
#define GPB_VIF_LAUNCH_T(C_, T_, NT_)                                                                                                   \
  do {                                                                                                                              \
    auto kern = vif_resid_grad_kernel<C_, T_, NT_>;                                                                                   \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e_ != hipSuccess) return e_;                                                                                                \
    hipLaunchKernelGGL(kern, dim3(npts), dim3(T_), lds, st, args, g);                                                               \
  } while (0)
#define GPB_VIF_LAUNCH(C_)                                                                                                           \
  do {                                                                                                                              \
    switch (vif_tiles(args.m)) {                                                                                                    \
      case 1: GPB_VIF_LAUNCH_T(C_, 64, 1); break;                                                                                   \
      case 2: GPB_VIF_LAUNCH_T(C_, 64, 2); break;                                                                                   \
      case 3: GPB_VIF_LAUNCH_T(C_, 64, 3); break;                                                                                   \
      case 4: GPB_VIF_LAUNCH_T(C_, 64, 4); break;                                                                                   \
      default: GPB_VIF_LAUNCH_T(C_, 128, 0); break;                                                                                 \
    }                                                                                                                               \
  } while (0)
  switch (cov) {
    case kMatern05: GPB_VIF_LAUNCH(kMatern05); break;
    case kMatern15: GPB_VIF_LAUNCH(kMatern15); break;
    case kMatern25: GPB_VIF_LAUNCH(kMatern25); break;
    default: return hipErrorInvalidValue;
  }
#undef GPB_VIF_LAUNCH
#undef GPB_VIF_LAUNCH_T
  return hipGetLastError();
}

}  // namespace gpb
