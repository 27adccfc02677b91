// gpboost_amd/csrc/dense_kernels.hip
//
// Exact (non-approximated) GP path for gfx950 -- SURVEY.md section 8 row a10 / BASELINE config 1:
//   Psi = Sigma + I on the transformed scale   include/GPBoost/re_model_template.h:8151 (CalcSigmaComps),
//                                              :9273-9287 (CalcZSigmaZt), cov_fcts.h:634-755
//   Psi = L L^T (dense, fp64)                  :6491-6494 (CalcChol, Eigen::LLT)
//   y^T Psi^-1 y, log|Psi| = 2 sum log L_ii    :9894, :3127, :3132
//   y_aux = Psi^-1 y                           :9894
//
// Kernels (matrix row-major, lower triangle, leading dimension np = n rounded up to 64; the padding
// rows/columns are the identity so they change neither the log-determinant nor the solves):
//   dense_cov_lower_kernel   covariance assembly, one 128x128 tile per workgroup, 8x8 entries per lane, 32-byte
//                            stores: the HBM-write-bound kernel of the path (8 B written per Matern evaluation)
//   potrf_diag_v2_kernel     64x64 diagonal block, one wavefront, row-per-lane in registers, v_readlane broadcasts
//   trsm_panel_v2_kernel     L21 = A21 L11^-T, one lane per row, L11 staged in LDS (wave-uniform reads)
//   syrk_mfma_db_kernel      C -= L L^T on lower 128x128 tiles with v_mfma_f64_16x16x4_f64 (one wavefront per 64x64
//                            quadrant = 16 accumulator tiles kept across the K loop, K staged through LDS in chunks of
//                            64); used "narrow" (K = 64, inside a 512-wide block column) and "wide" (K = 512)
//   trsv_lower_kernel        forward (and optionally backward) substitution + y^T Psi^-1 y + log-det
#include "dev_common.h"
#include "dense_kernels.h"

namespace gpb {

namespace {
constexpr int TB = 64;            // tile / panel width
constexpr int LDSS = 66;          // LDS row stride (doubles): 16 rows x 4 k of an MFMA fragment hit 64 distinct banks
typedef double double4v __attribute__((ext_vector_type(4)));

}  // namespace

// ---- covariance assembly ----------------------------------------------------------------
// grid = lower 128x128 tiles (ti >= tj) flattened; block = 256 threads, thread (tx, ty) computes an 8x8 sub-block
// (64 kernel evaluations per lane amortise the prologue).  Coordinates are staged in LDS already multiplied by
// a * 64/ln2 (dev_common.h), the 64-entry table carries the variance.  Stores: 2 x 32 bytes per lane and row; each
// store instruction covers 512 contiguous bytes of a row with its 16 lanes (write-combining friendly).
constexpr int CT = 128;
template <int COV, bool D3>
__global__ __launch_bounds__(256) void dense_cov_lower_kernel(const double4* __restrict__ pts, int n, int np, int ld, double var,
                                                              double a, double nugget, const double* __restrict__ gtab,
                                                              double* __restrict__ P) {
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ double s_rx[CT], s_ry[CT], s_rz[CT], s_cx[CT], s_cy[CT], s_cz[CT];
  const int tid = threadIdx.x;
  if (tid < GPB_EXP_TAB_SIZE) s_tab[tid] = gtab[tid] * var;
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while ((long long)ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - (int)((long long)ti * (ti + 1) / 2);
  const double sc = a * kCoordScale;   // half-scaled coordinates: d2 below is (rho/2)^2, see exp_of_scaled
  if (tid < CT) {
    const int r = ti * CT + tid;
    const double4 p = r < n ? pts[r] : make_double4(0, 0, 0, 0);
    s_rx[tid] = p.x * sc; s_ry[tid] = p.y * sc; s_rz[tid] = p.z * sc;
  } else {
    const int c = tj * CT + tid - CT;
    const double4 p = c < n ? pts[c] : make_double4(0, 0, 0, 0);
    s_cx[tid - CT] = p.x * sc; s_cy[tid - CT] = p.y * sc; s_cz[tid - CT] = p.z * sc;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  double qx[8], qy[8], qz[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {   // columns tx*4..+3 and 64+tx*4..+3: every store instruction is contiguous across the 16 lanes
    const int lc = (cc < 4) ? tx * 4 + cc : 64 + tx * 4 + (cc - 4);
    qx[cc] = s_cx[lc]; qy[cc] = s_cy[lc]; qz[cc] = D3 ? s_cz[lc] : 0.0;
  }
  const double diag = var + nugget;
#pragma unroll 2
  for (int rr = 0; rr < 8; ++rr) {
    const int lr = ty * 8 + rr, r = ti * CT + lr;
    if (r >= np) break;
    const double px = s_rx[lr], py = s_ry[lr], pz = D3 ? s_rz[lr] : 0.0;
    double v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = tj * CT + ((cc < 4) ? tx * 4 + cc : 64 + tx * 4 + (cc - 4));
      const double dx = px - qx[cc], dy = py - qy[cc];
      double d2 = __builtin_fma(dx, dx, 1e-300);
      d2 = __builtin_fma(dy, dy, d2);
      if (D3) { const double dz = pz - qz[cc]; d2 = __builtin_fma(dz, dz, d2); }
      double val = matern_cov_s<COV>(d2, s_tab);
      if (r == c) val = diag;
      if (r >= n || c >= n) val = (r == c) ? 1.0 : 0.0;     // identity padding
      v[cc] = val;
    }
    double* rowp = P + (size_t)r * ld + (size_t)tj * CT + tx * 4;   // ld > np: Psi is the top-left block of the augmented matrix (dense_grad)
    if (tj * CT + tx * 4 < np) *reinterpret_cast<double4*>(rowp) = make_double4(v[0], v[1], v[2], v[3]);
    if (tj * CT + 64 + tx * 4 < np) *reinterpret_cast<double4*>(rowp + 64) = make_double4(v[4], v[5], v[6], v[7]);
  }
}

// ---- diagonal block factorisation -----------------------------------------------------------
// One wavefront, row-per-lane: lane r holds row r of the 64x64 block in registers (M[c], c <= r).  The pivot and the
// column multipliers are wave-uniform, so they travel through v_readlane into SGPRs and feed v_fma_f64 directly
// (2016 readlane pairs + FMAs, ~7 us) -- no LDS, no barriers.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned int)lo);
}

// ---- trailing update with fp64 MFMA: C[r][c] -= sum_{k in [kp0, kp0 + K)} L[r][k] L[c][k] for r in [r_base, np), c in [c_base, c_lim), c <= r ----
// (syrk_mfma_db_kernel / syrk_mfma_small_kernel below.  The factorisation calls the update twice per step: "narrow" (K = 64, only the columns of the
// current 512-wide block column) and "wide" (K = 512, the whole trailing matrix, once per block column) -- the wide call reads and writes every C tile
// once per 512 columns instead of once per 64, which lifts the arithmetic intensity from ~5 to ~40 flop/B.  Round 4 removed the superseded generations:
// the single-buffered 128 x 128 update of round 2 and the first forms of the two panel kernels; their measurements are in DESIGN.md section 4.5.)

// ---- round 3: the panel chain and the update, second forms --------------------------------------------------------------------
// (2) syrk_mfma_db_kernel: the same tiles and MFMA schedule as syrk_mfma_kernel with the K loop double-buffered -- the next 32-column
//     chunk travels global -> registers while the MFMAs of the current one run, then registers -> the other LDS buffer, one barrier
//     per chunk.  (r03_pmc.json for syrk_mfma_kernel: MFMA busy 0.38 of the kernel's cycles, 41 % of the wave cycles waiting: the
//     staging of the single-buffered form is exposed.)
// 1 / sqrt(x): v_rsq_f64 and one Newton step (relative error 1.5 delta^2 ~ 1e-16 for the instruction's delta ~ 2^-26), the form the point
// kernel uses -- sqrt() followed by a division is two long dependent software sequences (~500 cycles per pivot, 64 pivots per block:
// 13 of the 27 us of potrf_diag_kernel)
__device__ __forceinline__ double inv_sqrt_newton(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-0.5 * x * y0, y0, 0.5);
  return __builtin_fma(y0, e, y0);
}
// Second forms of the two panel kernels.  potrf: the pivot's inverse square root by inv_sqrt_newton (27.7 -> 21.3 us
// per block under the kernel trace).  trsm: the reciprocals of L11's diagonal once per workgroup (one division per lane, in parallel)
// instead of 64 divisions in every lane's chain, and four running sums per entry instead of one dependent chain of fmas.
// (Tried: the COLUMN-oriented recurrence -- once x_j is final the other 63 - j entries take x_j L[q][j] as independent fmas, L11 staged
// transposed: the compiler keeps the LDS operands in AGPRs and the solve went from 20 to 36 us per panel: not kept.)
// (Tried: both in ONE launch, every workgroup factoring the diagonal block redundantly in registers -- 23 000 instructions with the SGPR
// traffic of the readlanes spilled through v_writelane / AGPRs, 29 us per panel SLOWER than the two launches: not kept.)
__global__ __launch_bounds__(64) void potrf_diag_v2_kernel(double* __restrict__ P, int np, int k0, int* __restrict__ info) {
  __builtin_amdgcn_s_setprio(3);                         // the panel chain is the critical path: win issue arbitration against the update's waves
  const int r = threadIdx.x;
  double* row = P + (size_t)(k0 + r) * np + k0;
  double M[TB];
#pragma unroll
  for (int c = 0; c < TB; ++c) M[c] = row[c];
  bool bad = false;
  static_for<0, TB>([&](auto k_) {
    constexpr int k = decltype(k_)::value;
    const double piv = readlane_f64(M[k], k);
    if (!(piv > 0.0)) bad = true;
    const double inv = inv_sqrt_newton(piv);             // wave-uniform
    M[k] *= inv;                                         // lane k: piv / sqrt(piv) = L[k][k]
    static_for<k + 1, TB>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      const double lck = readlane_f64(M[k], c);
      M[c] = __builtin_fma(-M[k], lck, M[c]);
    });
  });
#pragma unroll
  for (int c = 0; c < TB; ++c) if (c <= r) row[c] = M[c];
  if (bad && r == 0) atomicOr(info, 1);
}

__global__ __launch_bounds__(64) void trsm_panel_v2_kernel(double* __restrict__ P, int np, int k0) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ double sL[TB][TB + 1];
  __shared__ double sInv[TB];
  const int tid = threadIdx.x;
  const double* __restrict__ L11 = P + (size_t)k0 * np + k0;
#pragma unroll 8
  for (int r = 0; r < TB; ++r) {
    sL[r][tid] = L11[(size_t)r * np + tid];
  }
  __syncthreads();
  sInv[tid] = 1.0 / sL[tid][tid];                        // one division per lane, all 64 in parallel (NOT inside the loop above: a
  __syncthreads();                                       // predicated division per iteration is 64 sequential divisions)
  const int i = k0 + TB + blockIdx.x * 64 + tid;
  const bool live = i < np;
  double* row = P + (size_t)(live ? i : k0 + TB) * np + k0;
  double x[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) x[j] = row[j];
  static_for<0, TB>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    double t[4] = {x[j], 0.0, 0.0, 0.0};                 // four running sums: the dependent chain is j / 4 fmas long
    static_for<0, j>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      t[p & 3] = __builtin_fma(-x[p], sL[j][p], t[p & 3]);
    });
    x[j] = ((t[0] + t[1]) + (t[2] + t[3])) * sInv[j];
  });
  if (live) {
#pragma unroll
    for (int j = 0; j < TB; ++j) row[j] = x[j];
  }
}

//     KCT = 16 (used when an update has more than 256 tiles): 16-column chunks, 74 KB of LDS and -- with
//     __launch_bounds__(256, 2) -- 224 VGPRs without spills, so that TWO workgroups share a CU: two wavefronts per SIMD, one issuing MFMAs
//     while the other waits at its barrier or for its operands, and room for the panel kernels of the next block column (trsm 33 KB, 64 x 64
//     update 68 KB of LDS) beside a workgroup of the look-ahead update instead of queueing behind it.  Measured at n = 16 384
//     (profiles/r03_e_dense_ab.log): factorisation 53.6 -> 44.3 ms, 27.3 -> 33.1 TFLOP/s = 0.42 of the fp64 MFMA peak; with KCT = 16 and ONE
//     workgroup per CU it was 57.2 ms (more barriers, nothing to overlap them with).  PMC of the KCT = 32 form: MFMA busy 0.50 of the CU-busy
//     cycles, 49 % of the wave cycles waiting.
//     (Tried: an XCD-aware tile order -- supertiles of 8 x 4 tiles per XCD so that 32 CUs share 12 panel chunks in their L2: 53.0 -> 55.3 ms
//     at n = 16 384.  The update is not short of L2 bandwidth: not kept.)
template <int KCT>
__global__ __launch_bounds__(256, KCT == 16 ? 2 : 1) void syrk_mfma_db_kernel(double* __restrict__ P, int np, int kp0, int K, int r_base,
                                                           int c_base, int c_lim, int ntj) {
  constexpr int LDKT = KCT + 2;                          // 34 / 18: 16 rows x 2 k of a fragment half hit 32 distinct 8-byte banks
  constexpr int TPR = KCT / 2;                           // threads per staged row (a double2 each)
  constexpr int RPP = 256 / TPR;                         // rows per staging pass
  constexpr int NT = 128 / RPP;                          // passes
  __shared__ double sbuf[2][2][128 * LDKT];              // [buffer][A | B]: 139,264 B (KCT = 32) / 73,728 B (16)
  const int ti = blockIdx.x / ntj, tj = blockIdx.x % ntj;
  const int r0 = r_base + ti * 128, c0 = c_base + tj * 128;
  if (c0 > r0 + 127) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wi = wave >> 1, wj = wave & 1;
  const int gr0 = r0 + 64 * wi, gc0 = c0 + 64 * wj;
  const bool quad_live = (gc0 <= gr0 + 63) && gr0 < np && gc0 < c_lim;
  double4v acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = (double4v){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  // staging: 128 rows x KCT columns per operand, NT double2 per thread and operand: thread tid takes rows (tid / TPR) + RPP t
  const int si = tid / TPR, sj = (tid % TPR) * 2;
  double2 ra[NT], rb[NT];
  auto gload = [&](int kc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = si + RPP * t;
      ra[t] = (r0 + i < np) ? *reinterpret_cast<const double2*>(P + (size_t)(r0 + i) * np + kp0 + kc + sj) : make_double2(0.0, 0.0);
      rb[t] = (c0 + i < np) ? *reinterpret_cast<const double2*>(P + (size_t)(c0 + i) * np + kp0 + kc + sj) : make_double2(0.0, 0.0);
    }
  };
  auto sstore = [&](int b) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = si + RPP * t;
      *reinterpret_cast<double2*>(&sbuf[b][0][i * LDKT + sj]) = ra[t];
      *reinterpret_cast<double2*>(&sbuf[b][1][i * LDKT + sj]) = rb[t];
    }
  };
  const int nch = K / KCT;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < nch) gload((ch + 1) * KCT);            // in flight during the MFMAs below
    if (quad_live) {
      const double* qA = &sbuf[cur][0][(64 * wi) * LDKT];
      const double* qB = &sbuf[cur][1][(64 * wj) * LDKT];
#pragma unroll
      for (int kk = 0; kk < KCT / 4; ++kk) {
        double af[4], bf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          af[q] = qA[(16 * q + fr) * LDKT + 4 * kk + fk];
          bf[q] = qB[(16 * q + fr) * LDKT + 4 * kk + fk];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
      }
    }
    if (ch + 1 < nch) sstore(cur ^ 1);                   // the other buffer: last read before the previous barrier
    __syncthreads();
  }
  if (!quad_live) return;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = gr0 + 16 * mi + fk + 4 * r, gj = gc0 + 16 * nj + fr;
        if (gj <= gi && gi < np && gj < c_lim) P[(size_t)gi * np + gj] -= acc[mi][nj][r];
      }
}

// (3) syrk_mfma_small_kernel: the narrow (K = 64) update inside a 512-wide block column on 64 x 64 tiles -- one wavefront per 32 x 32
//     quadrant, 2 x 2 MFMA tiles.  The 128 x 128 form spends 32 us per panel step at n = 2000 (16 accumulator tiles' worth of MFMA and
//     64 read-modify-writes per lane on at most 64 workgroups); a quarter of the work per workgroup on four times the workgroups is a
//     shorter critical path, and the narrow updates ARE the critical path of the panel chain.
__global__ __launch_bounds__(256) void syrk_mfma_small_kernel(double* __restrict__ P, int np, int kp0, int K, int r_base,
                                                              int c_base, int c_lim, int ntj) {
  __shared__ double sA[64 * LDSS], sB[64 * LDSS];       // 67,584 B: two workgroups per CU
  __builtin_amdgcn_s_setprio(2);
  const int ti = blockIdx.x / ntj, tj = blockIdx.x % ntj;
  const int r0 = r_base + ti * 64, c0 = c_base + tj * 64;
  if (c0 > r0 + 63) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wi = wave >> 1, wj = wave & 1;
  const int gr0 = r0 + 32 * wi, gc0 = c0 + 32 * wj;
  const bool quad_live = (gc0 <= gr0 + 31) && gr0 < np && gc0 < c_lim;
  double4v acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = (double4v){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  const int si = tid >> 5, sj = (tid & 31) * 2;          // staging: thread takes rows si + 8 t, columns sj, sj + 1
  for (int kc = 0; kc < K; kc += TB) {
    if (kc) __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = si + 8 * t;
      const double2 va = (r0 + i < np) ? *reinterpret_cast<const double2*>(P + (size_t)(r0 + i) * np + kp0 + kc + sj) : make_double2(0.0, 0.0);
      const double2 vb = (c0 + i < np) ? *reinterpret_cast<const double2*>(P + (size_t)(c0 + i) * np + kp0 + kc + sj) : make_double2(0.0, 0.0);
      *reinterpret_cast<double2*>(&sA[i * LDSS + sj]) = va;
      *reinterpret_cast<double2*>(&sB[i * LDSS + sj]) = vb;
    }
    __syncthreads();
    if (quad_live) {
      const double* qA = sA + (32 * wi) * LDSS;
      const double* qB = sB + (32 * wj) * LDSS;
#pragma unroll
      for (int kk = 0; kk < TB / 4; ++kk) {
        double af[2], bf[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          af[q] = qA[(16 * q + fr) * LDSS + 4 * kk + fk];
          bf[q] = qB[(16 * q + fr) * LDSS + 4 * kk + fk];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
      }
    }
  }
  if (!quad_live) return;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = gr0 + 16 * mi + fk + 4 * r, gj = gc0 + 16 * nj + fr;
        if (gj <= gi && gi < np && gj < c_lim) P[(size_t)gi * np + gj] -= acc[mi][nj][r];
      }
}

// NLL without a forward substitution: with y as row np of the (np + 64)-row matrix the panel solves leave z = L^-1 y in that row and
// the Schur complement leaves -z'z at [np][np].  out[0] = y' Psi^-1 y, out[1] = log|Psi|.
__global__ void dense_set_yrow_kernel(double* __restrict__ P, int n, int np, int ld, const double* __restrict__ y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ld) P[(size_t)np * ld + j] = j < n ? y[j] : 0.0;
}
__global__ __launch_bounds__(1024) void dense_yrow_sums_kernel(const double* __restrict__ P, int n, int np, int ld, double* __restrict__ out) {
  __shared__ double sred[1024];
  const int tid = threadIdx.x;
  double lgd = 0.0;
  for (int i = tid; i < n; i += 1024) lgd += log(P[(size_t)i * ld + i]);
  sred[tid] = lgd; __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) { if (tid < w) sred[tid] += sred[tid + w]; __syncthreads(); }
  if (tid == 0) { out[0] = -P[(size_t)np * ld + np]; out[1] = 2.0 * sred[0]; }
}

// ---- triangular solves + reductions, one workgroup ----------------------------------------------
// z = L^-1 y; out[0] = z^T z (= y^T Psi^-1 y), out[1] = 2 sum log L_ii; if x_out: x = L^-T z (= Psi^-1 y)
__global__ __launch_bounds__(1024) void trsv_lower_kernel(const double* __restrict__ P, int n, int np, int ld,
                                                          const double* __restrict__ y, double* __restrict__ z,
                                                          double* __restrict__ out, double* __restrict__ x_out) {
  __shared__ double sz[TB];
  __shared__ double sred[1024];
  __shared__ double sD[TB][TB + 1];       // the diagonal block of the current step (its 64 sequential steps read LDS, not HBM)
  const int tid = threadIdx.x;
  for (int i = tid; i < np; i += 1024) z[i] = i < n ? y[i] : 0.0;
  __syncthreads();
  for (int b0 = 0; b0 < np; b0 += TB) {
    for (int e = tid; e < TB * TB; e += 1024) sD[e >> 6][e & 63] = P[(size_t)(b0 + (e >> 6)) * ld + b0 + (e & 63)];
    __syncthreads();
    // diagonal block: 64 sequential steps by the first wavefront
    if (tid < TB) {
      double zi = z[b0 + tid];
      for (int k = 0; k < TB; ++k) {
        const double lkk = sD[k][k];
        const double zk = __shfl(zi, k, 64) / lkk;
        if (tid == k) zi = zk;
        if (tid > k) zi = __builtin_fma(-sD[tid][k], zk, zi);
      }
      sz[tid] = zi; z[b0 + tid] = zi;
    }
    __syncthreads();
    // rows below: z[i] -= L[i][b0:b0+64] . z_block   (4 threads per row, 16 columns each)
    const int sub = tid & 3;
    for (int i = b0 + TB + (tid >> 2); i < np; i += 256) {
      const double* row = P + (size_t)i * ld + b0 + sub * 16;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_fma(row[k], sz[sub * 16 + k], acc);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (sub == 0) z[i] -= acc;
    }
    __syncthreads();
  }
  double q = 0.0, lgd = 0.0;
  for (int i = tid; i < n; i += 1024) { q = __builtin_fma(z[i], z[i], q); lgd += log(P[(size_t)i * ld + i]); }
  sred[tid] = q; __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) { if (tid < w) sred[tid] += sred[tid + w]; __syncthreads(); }
  if (tid == 0) out[0] = sred[0];
  __syncthreads();
  sred[tid] = lgd; __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) { if (tid < w) sred[tid] += sred[tid + w]; __syncthreads(); }
  if (tid == 0) out[1] = 2.0 * sred[0];
  if (x_out == nullptr) return;
  __syncthreads();
  // backward: L^T x = z, blocks from the bottom; x overwrites z
  for (int b0 = np - TB; b0 >= 0; b0 -= TB) {
    for (int e = tid; e < TB * TB; e += 1024) sD[e >> 6][e & 63] = P[(size_t)(b0 + (e >> 6)) * ld + b0 + (e & 63)];
    __syncthreads();
    if (tid < TB) {
      double xi = z[b0 + tid];
      for (int k = TB - 1; k >= 0; --k) {
        const double lkk = sD[k][k];
        const double xk = __shfl(xi, k, 64) / lkk;
        if (tid == k) xi = xk;
        if (tid < k) xi = __builtin_fma(-sD[k][tid], xk, xi);
      }
      sz[tid] = xi; z[b0 + tid] = xi;
    }
    __syncthreads();
    // rows above: z[i] -= sum_k L[b0+k][i] x[b0+k]  (column access of L: coalesced across i)
    for (int i = tid; i < b0; i += 1024) {
      double acc = 0.0;
#pragma unroll 8
      for (int k = 0; k < TB; ++k) acc = __builtin_fma(P[(size_t)(b0 + k) * ld + i], sz[k], acc);
      z[i] -= acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 1024) x_out[i] = z[i];
}

// ---- exact-GP gradient: the trace and the quadratic forms of CalcGradPars (re_model_template.h:2016-2040) in one pass --------
// For the lower 128x128 tiles of the n x n problem: Sigma_ij = var k(d_ij) (no nugget) and dSigma_ij = d/dlog(a) of it are evaluated
// again from the coordinates (never stored), W_ij = -(Psi^-1)_ij is read from the Schur-complement block of the augmented matrix
// (rows / columns np.. of P2, leading dimension ld), ya = Psi^-1 y.  Per tile, with weight 2 off the diagonal:
//   part[0] = sum Sigma_ij ya_i ya_j      part[1] = sum Sigma_ij W_ij      part[2] = sum dSigma_ij ya_i ya_j      part[3] = sum dSigma_ij W_ij
// which give  g_k = -1/2 ya' dPsi_k ya / sigma2 + 1/2 tr(Psi^-1 dPsi_k)  for k = variance, range  (:2031-2034).
template <int COV, bool D3>
__global__ __launch_bounds__(256) void dense_grad_kernel(const double4* __restrict__ pts, int n, int np, int ld, double var, double a,
                                                         const double* __restrict__ gtab, const double* __restrict__ P2,
                                                         const double* __restrict__ ya, double* __restrict__ part, int ntiles) {
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ double s_rx[CT], s_ry[CT], s_rz[CT], s_ra[CT], s_cx[CT], s_cy[CT], s_cz[CT], s_ca[CT];
  __shared__ double s_red[4][256];
  const int tid = threadIdx.x;
  s_tab[tid] = gtab[tid] * var;
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while ((long long)ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - (int)((long long)ti * (ti + 1) / 2);
  const double sc = a * kCoordScale;
  if (tid < CT) {
    const int r = ti * CT + tid;
    const double4 p = r < n ? pts[r] : make_double4(0, 0, 0, 0);
    s_rx[tid] = p.x * sc; s_ry[tid] = p.y * sc; s_rz[tid] = p.z * sc; s_ra[tid] = r < n ? ya[r] : 0.0;
  } else {
    const int c = tj * CT + tid - CT;
    const double4 p = c < n ? pts[c] : make_double4(0, 0, 0, 0);
    s_cx[tid - CT] = p.x * sc; s_cy[tid - CT] = p.y * sc; s_cz[tid - CT] = p.z * sc; s_ca[tid - CT] = c < n ? ya[c] : 0.0;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int rr = 0; rr < 8; ++rr) {
    const int lr = ty * 8 + rr, r = ti * CT + lr;
    if (r >= n) break;
    const double px = s_rx[lr], py = s_ry[lr], pz = D3 ? s_rz[lr] : 0.0, yr = s_ra[lr];
    const double* wrow = P2 + (size_t)(np + r) * ld + np;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int lc = (cc < 4) ? tx * 4 + cc : 64 + tx * 4 + (cc - 4);
      const int c = tj * CT + lc;
      if (c > r || c >= n) continue;
      const double dx = px - s_cx[lc], dy = py - s_cy[lc];
      double d2 = __builtin_fma(dx, dx, 1e-300);
      d2 = __builtin_fma(dy, dy, d2);
      if (D3) { const double dz = pz - s_cz[lc]; d2 = __builtin_fma(dz, dz, d2); }
      double dk;
      double k = matern_cov_dlog_s<COV>(d2, s_tab, dk);
      if (r == c) { k = var; dk = 0.0; }
      const double w = (r == c) ? 1.0 : 2.0;
      const double yy = w * yr * s_ca[lc], ww = w * wrow[c];
      acc[0] = __builtin_fma(k, yy, acc[0]); acc[1] = __builtin_fma(k, ww, acc[1]);
      acc[2] = __builtin_fma(dk, yy, acc[2]); acc[3] = __builtin_fma(dk, ww, acc[3]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) s_red[q][tid] = acc[q];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s_red[q][tid] += s_red[q][tid + w];
    }
    __syncthreads();
  }
  if (tid < 4) part[(size_t)tid * ntiles + t] = s_red[tid][0];
}

// ---- exact-GP Fisher information (CalcFisherInformation, dense branch, re_model_template.h:10066-10127) ------------------------------
// FI_ab = 1/2 tr(Psi^-1 dPsi_a Psi^-1 dPsi_b) over (error variance, marginal variance, range).  Every product Psi^-1 dPsi_a Psi^-1 dPsi_b the
// reference forms with dense GEMMs is here a block of ONE Schur complement: the partial factorisation (first np columns) of
//     [[Psi, ., ., .], [I, 0, ., .], [E1, 0, 0, .], [E2, 0, 0, 0]],  E1 = Sigma (no nugget),  E2 = dSigma / dlog(a)
// leaves S_ab = -E_a Psi^-1 E_b (E0 = I) in block (a + 1, b + 1), a >= b -- the same MFMA trailing updates as the factorisation itself --
// and tr(Psi^-1 E_a Psi^-1 E_b) = sum_ij W_ij (S_ab)_ji with W = S_00 = -Psi^-1.
// This kernel writes E1 and E2 as FULL blocks (rows row1.. / row2.., columns 0..np) -- grid = all 128 x 128 tiles.
template <int COV, bool D3>
__global__ __launch_bounds__(256) void dense_deriv_blocks_kernel(const double4* __restrict__ pts, int n, int ld, double var, double a,
                                                                 const double* __restrict__ gtab, double* __restrict__ P, int row1, int row2, int nt) {
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ double s_rx[CT], s_ry[CT], s_rz[CT], s_cx[CT], s_cy[CT], s_cz[CT];
  const int tid = threadIdx.x;
  s_tab[tid] = gtab[tid] * var;
  const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
  const double sc = a * kCoordScale;
  if (tid < CT) {
    const int r = ti * CT + tid;
    const double4 p = r < n ? pts[r] : make_double4(0, 0, 0, 0);
    s_rx[tid] = p.x * sc; s_ry[tid] = p.y * sc; s_rz[tid] = p.z * sc;
  } else {
    const int c = tj * CT + tid - CT;
    const double4 p = c < n ? pts[c] : make_double4(0, 0, 0, 0);
    s_cx[tid - CT] = p.x * sc; s_cy[tid - CT] = p.y * sc; s_cz[tid - CT] = p.z * sc;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  for (int rr = 0; rr < 8; ++rr) {
    const int lr = ty * 8 + rr, r = ti * CT + lr;
    if (r >= n) break;
    const double px = s_rx[lr], py = s_ry[lr], pz = D3 ? s_rz[lr] : 0.0;
    double v1[8], v2[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int lc = (cc < 4) ? tx * 4 + cc : 64 + tx * 4 + (cc - 4);
      const int c = tj * CT + lc;
      const double dx = px - s_cx[lc], dy = py - s_cy[lc];
      double d2 = __builtin_fma(dx, dx, 1e-300);
      d2 = __builtin_fma(dy, dy, d2);
      if (D3) { const double dz = pz - s_cz[lc]; d2 = __builtin_fma(dz, dz, d2); }
      double dk;
      double k = matern_cov_dlog_s<COV>(d2, s_tab, dk);
      if (r == c) { k = var; dk = 0.0; }
      if (c >= n) { k = 0.0; dk = 0.0; }
      v1[cc] = k; v2[cc] = dk;
    }
    const size_t col = (size_t)tj * CT + tx * 4;
    double* p1 = P + (size_t)(row1 + r) * ld + col;
    double* p2 = P + (size_t)(row2 + r) * ld + col;
    *reinterpret_cast<double4*>(p1) = make_double4(v1[0], v1[1], v1[2], v1[3]);
    *reinterpret_cast<double4*>(p1 + 64) = make_double4(v1[4], v1[5], v1[6], v1[7]);
    *reinterpret_cast<double4*>(p2) = make_double4(v2[0], v2[1], v2[2], v2[3]);
    *reinterpret_cast<double4*>(p2 + 64) = make_double4(v2[4], v2[5], v2[6], v2[7]);
  }
}

// The six traces T_ab = sum_{i,j < n} W_ij (S_ab)_ji, (a, b) = 00, 10, 20, 11, 21, 22, over the lower 128 x 128 tiles ([6][ntiles] partials,
// term-major).  W and the diagonal blocks are symmetric and stored as lower triangles; the off-diagonal blocks are full.
__global__ __launch_bounds__(256) void dense_fisher_sums_kernel(const double* __restrict__ P, int n, int np, int ld, double* __restrict__ part, int ntiles) {
  __shared__ double s_red[6][256];
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while ((long long)ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - (int)((long long)ti * (ti + 1) / 2);
  const int tx = tid & 15, ty = tid >> 4;
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const size_t b1 = (size_t)np, b2 = (size_t)2 * np, b3 = (size_t)3 * np;
  for (int rr = 0; rr < 8; ++rr) {
    const int i = ti * CT + ty * 8 + rr;
    if (i >= n) break;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int j = tj * CT + ((cc < 4) ? tx * 4 + cc : 64 + tx * 4 + (cc - 4));
      if (j > i || j >= n) continue;
      const double w = P[(b1 + i) * ld + b1 + j];
      const double s10 = P[(b2 + i) * ld + b1 + j], s20 = P[(b3 + i) * ld + b1 + j], s21 = P[(b3 + i) * ld + b2 + j];
      const double s11 = P[(b2 + i) * ld + b2 + j], s22 = P[(b3 + i) * ld + b3 + j];
      if (i == j) {
        acc[0] = __builtin_fma(w, w, acc[0]); acc[1] = __builtin_fma(w, s10, acc[1]); acc[2] = __builtin_fma(w, s20, acc[2]);
        acc[3] = __builtin_fma(w, s11, acc[3]); acc[4] = __builtin_fma(w, s21, acc[4]); acc[5] = __builtin_fma(w, s22, acc[5]);
      } else {
        const double t10 = P[(b2 + j) * ld + b1 + i], t20 = P[(b3 + j) * ld + b1 + i], t21 = P[(b3 + j) * ld + b2 + i];
        acc[0] = __builtin_fma(2.0 * w, w, acc[0]); acc[1] = __builtin_fma(w, s10 + t10, acc[1]); acc[2] = __builtin_fma(w, s20 + t20, acc[2]);
        acc[3] = __builtin_fma(2.0 * w, s11, acc[3]); acc[4] = __builtin_fma(w, s21 + t21, acc[4]); acc[5] = __builtin_fma(2.0 * w, s22, acc[5]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) s_red[q][tid] = acc[q];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
#pragma unroll
      for (int q = 0; q < 6; ++q) s_red[q][tid] += s_red[q][tid + w];
    }
    __syncthreads();
  }
  if (tid < 6) part[(size_t)tid * ntiles + t] = s_red[tid][0];
}

// ---- exact-GP prediction: the cross-covariance block --------------------------------------------------------------------------------
// C[i][j] = var k(|x*_i - x_j|) for prediction point i and observed point j, written as rows row0.. of the augmented matrix
// [[Psi, ., .], [C, 0, .], [y', 0, 0]]: its partial factorisation (first np columns) leaves -C Psi^-1 C' (the reduction of the predictive
// covariance) and -C Psi^-1 y (minus the predictive mean) in the Schur complement -- no triangular solve with n_pred right-hand sides.
template <int COV, bool D3>
__global__ __launch_bounds__(256) void dense_cross_cov_kernel(const double4* __restrict__ pts, int n, const double4* __restrict__ pred, int n_pred, int ld,
                                                              double var, double a, const double* __restrict__ gtab, double* __restrict__ P, int row0) {
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  const int tid = threadIdx.x;
  if (tid < GPB_EXP_TAB_SIZE) s_tab[tid] = gtab[tid] * var;
  __syncthreads();
  const int j = blockIdx.x * 256 + tid, i = blockIdx.y;
  if (j >= n || i >= n_pred) return;
  const double sc = a * kCoordScale;
  const double4 p = pred[i], q = pts[j];
  const double dx = (p.x - q.x) * sc, dy = (p.y - q.y) * sc;
  double d2 = __builtin_fma(dx, dx, 1e-300);
  d2 = __builtin_fma(dy, dy, d2);
  if (D3) { const double dz = (p.z - q.z) * sc; d2 = __builtin_fma(dz, dz, d2); }
  P[(size_t)(row0 + i) * ld + j] = matern_cov_s<COV>(d2, s_tab);
}

// bottom-left block of the augmented matrix := identity (the rest was zeroed by a memset)
__global__ void dense_aug_identity_kernel(double* __restrict__ P2, int np, int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) P2[(size_t)(np + i) * ld + i] = 1.0;
}

// ---- launchers ---------------------------------------------------------------------------------------
template <int COV>
static void launch_cov(bool d3, const double4* pts, int n, int np, int ld, double var, double a, double nugget, const double* gtab,
                       double* P, hipStream_t st) {
  const int nt = (np + CT - 1) / CT;
  const int ntiles = nt * (nt + 1) / 2;
  if (d3) hipLaunchKernelGGL((dense_cov_lower_kernel<COV, true>), dim3(ntiles), dim3(256), 0, st, pts, n, np, ld, var, a, nugget, gtab, P);
  else hipLaunchKernelGGL((dense_cov_lower_kernel<COV, false>), dim3(ntiles), dim3(256), 0, st, pts, n, np, ld, var, a, nugget, gtab, P);
}

hipError_t launch_dense_cov(int cov, bool d3, const double4* pts, int n, int np, int ld, double var, double a, double nugget,
                            const double* gtab, double* P, hipStream_t st) {
  switch (cov) {
    case kMatern05: launch_cov<kMatern05>(d3, pts, n, np, ld, var, a, nugget, gtab, P, st); break;
    case kMatern15: launch_cov<kMatern15>(d3, pts, n, np, ld, var, a, nugget, gtab, P, st); break;
    case kMatern25: launch_cov<kMatern25>(d3, pts, n, np, ld, var, a, nugget, gtab, P, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static void launch_update_narrow(double* P, int np, int kp0, int K, int r_base, int c_base, int c_lim, hipStream_t st) {
  if (r_base >= np || c_base >= c_lim) return;
  const int nti = (np - r_base + 63) / 64, ntj = (c_lim - c_base + 63) / 64;
  hipLaunchKernelGGL(syrk_mfma_small_kernel, dim3(nti * ntj), dim3(256), 0, st, P, np, kp0, K, r_base, c_base, c_lim, ntj);
}
static void launch_update(double* P, int np, int kp0, int K, int r_base, int c_base, int c_lim, hipStream_t st) {
  if (r_base >= np || c_base >= c_lim) return;
  const int nti = (np - r_base + 127) / 128, ntj = (c_lim - c_base + 127) / 128;
  // a trailing matrix of at most 256 tiles of 128 x 128 does not fill the CUs with one workgroup each: 64 x 64 tiles (two workgroups per CU, a
  // quarter of the K loop per workgroup) shorten the update that sits on the critical path of small factorisations (n = 2000: three wide updates
  // of ~95 us each) and of the last block columns of large ones.
  if (nti * ntj <= 256) { launch_update_narrow(P, np, kp0, K, r_base, c_base, c_lim, st); return; }
  // more than 256 tiles: 16-column chunks and two workgroups per CU (74 KB of LDS each: one issues MFMAs while the other waits at its barrier)
  hipLaunchKernelGGL(syrk_mfma_db_kernel<16>, dim3(nti * ntj), dim3(256), 0, st, P, np, kp0, K, r_base, c_base, c_lim, ntj);
}

// Blocked right-looking Cholesky with two levels: 64-column panels (potrf + trsm + narrow update inside the current
// 512-wide block column) and one wide K = 512 update of the remaining trailing matrix per block column.
// Look-ahead (st2 != nullptr): the wide update of block column J is split into the STRIP that the next block column needs
// (columns [J1, J1 + 512), stream st) and the REST (columns >= J1 + 512, stream st2); the latency-bound panel kernels of block
// column J + 1 then run concurrently with REST_J.  Hazards, all read-modify-write on C tiles:
//   REST_J  reads the L columns of block J                      -> waits for the panels of J (event ev_panels)
//   STRIP_J touches columns REST_{J-1} also updates             -> waits for REST_{J-1}     (event ev_rest)
//   panels of J + 1 touch only columns [J1, J1 + 512)           -> disjoint from REST_J, ordered after STRIP_J on st
//   REST_J and REST_{J-1} overlap                               -> same stream st2, in order
// ncols < np (default: all): only the first ncols columns are factorised; the rest of the matrix then holds the Schur complement
// C22 - L21 L21^T of the trailing block (lower triangle) -- the exact-GP gradient uses it on [[Psi, .], [I, 0]] to get -Psi^-1.
hipError_t launch_dense_cholesky(double* P, int np, int* info, hipStream_t st, hipStream_t st2, hipEvent_t ev_panels, hipEvent_t ev_rest,
                                 int ncols) {
  constexpr int OB = 512;
  if (ncols <= 0 || ncols > np) ncols = np;
  const bool lookahead = st2 != nullptr && ev_panels != nullptr && ev_rest != nullptr && np > 2 * OB;
  bool rest_pending = false;
  for (int J0 = 0; J0 < ncols; J0 += OB) {
    const int Jend = (J0 + OB < ncols) ? J0 + OB : ncols;
    for (int k0 = J0; k0 < Jend; k0 += TB) {
      const int rows_below = np - k0 - TB;
      hipLaunchKernelGGL(potrf_diag_v2_kernel, dim3(1), dim3(64), 0, st, P, np, k0, info);
      if (rows_below <= 0) break;
      hipLaunchKernelGGL(trsm_panel_v2_kernel, dim3((rows_below + 63) / 64), dim3(64), 0, st, P, np, k0);
      launch_update_narrow(P, np, k0, TB, k0 + TB, k0 + TB, Jend, st);     // narrow (K = 64): columns of this block column only, 64 x 64 tiles
    }
    if (Jend >= np) break;
    if (!lookahead) {
      launch_update(P, np, J0, Jend - J0, Jend, Jend, np, st);             // wide: K = 512, whole trailing matrix
      continue;
    }
    const int Send = (Jend + OB < np) ? Jend + OB : np;                     // strip = the next block column
    (void)hipEventRecord(ev_panels, st);
    if (rest_pending) (void)hipStreamWaitEvent(st, ev_rest, 0);             // STRIP_J after REST_{J-1}
    launch_update(P, np, J0, Jend - J0, Jend, Jend, Send, st);
    if (Send < np) {
      (void)hipStreamWaitEvent(st2, ev_panels, 0);                          // REST_J after the panels of J
      launch_update(P, np, J0, Jend - J0, Send, Send, np, st2);
      (void)hipEventRecord(ev_rest, st2);
      rest_pending = true;
    }
  }
  if (rest_pending) (void)hipStreamWaitEvent(st, ev_rest, 0);               // everything is ordered on st again
  return hipGetLastError();
}

hipError_t launch_dense_cross_cov(int cov, bool d3, const double4* pts, int n, const double4* pred, int n_pred, int ld, double var, double a,
                                  const double* gtab, double* P, int row0, hipStream_t st) {
  const dim3 grid((n + 255) / 256, n_pred);
#define GPB_CC(COV_) do { if (d3) hipLaunchKernelGGL((dense_cross_cov_kernel<COV_, true>), grid, dim3(256), 0, st, pts, n, pred, n_pred, ld, var, a, gtab, P, row0); \
                          else hipLaunchKernelGGL((dense_cross_cov_kernel<COV_, false>), grid, dim3(256), 0, st, pts, n, pred, n_pred, ld, var, a, gtab, P, row0); } while (0)
  switch (cov) {
    case kMatern05: GPB_CC(kMatern05); break;
    case kMatern15: GPB_CC(kMatern15); break;
    case kMatern25: GPB_CC(kMatern25); break;
    default: return hipErrorInvalidValue;
  }
#undef GPB_CC
  return hipGetLastError();
}
// y as row `row` of the matrix (generalises launch_dense_set_yrow, whose row is np)
__global__ void dense_set_row_kernel(double* __restrict__ P, int n, int ld, int row, const double* __restrict__ y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ld) P[(size_t)row * ld + j] = j < n ? y[j] : 0.0;
}
hipError_t launch_dense_set_row(double* P, int n, int ld, int row, const double* y, hipStream_t st) {
  hipLaunchKernelGGL(dense_set_row_kernel, dim3((ld + 255) / 256), dim3(256), 0, st, P, n, ld, row, y);
  return hipGetLastError();
}
hipError_t launch_dense_set_yrow(double* P, int n, int np, int ld, const double* y, hipStream_t st) {
  hipLaunchKernelGGL(dense_set_yrow_kernel, dim3((ld + 255) / 256), dim3(256), 0, st, P, n, np, ld, y);
  return hipGetLastError();
}
hipError_t launch_dense_yrow_sums(const double* P, int n, int np, int ld, double* out, hipStream_t st) {
  hipLaunchKernelGGL(dense_yrow_sums_kernel, dim3(1), dim3(1024), 0, st, P, n, np, ld, out);
  return hipGetLastError();
}
// x = L^-T z for z given in `work` (np doubles, overwritten), the backward half of launch_dense_solve
hipError_t launch_dense_solve_backward(const double* P, int np, int ld, double* work, double* x_out, hipStream_t st);

hipError_t launch_dense_aug_identity(double* P2, int np, int ld, hipStream_t st) {
  hipLaunchKernelGGL(dense_aug_identity_kernel, dim3((np + 255) / 256), dim3(256), 0, st, P2, np, ld);
  return hipGetLastError();
}

template <int COV>
static void launch_grad(bool d3, const double4* pts, int n, int np, int ld, double var, double a, const double* gtab, const double* P2,
                        const double* ya, double* part, int ntiles, hipStream_t st) {
  if (d3) hipLaunchKernelGGL((dense_grad_kernel<COV, true>), dim3(ntiles), dim3(256), 0, st, pts, n, np, ld, var, a, gtab, P2, ya, part, ntiles);
  else hipLaunchKernelGGL((dense_grad_kernel<COV, false>), dim3(ntiles), dim3(256), 0, st, pts, n, np, ld, var, a, gtab, P2, ya, part, ntiles);
}
template <int COV>
static void launch_deriv_blocks(bool d3, const double4* pts, int n, int ld, double var, double a, const double* gtab, double* P, int row1, int row2,
                                hipStream_t st) {
  const int nt = (n + CT - 1) / CT;
  if (d3) hipLaunchKernelGGL((dense_deriv_blocks_kernel<COV, true>), dim3(nt * nt), dim3(256), 0, st, pts, n, ld, var, a, gtab, P, row1, row2, nt);
  else hipLaunchKernelGGL((dense_deriv_blocks_kernel<COV, false>), dim3(nt * nt), dim3(256), 0, st, pts, n, ld, var, a, gtab, P, row1, row2, nt);
}
hipError_t launch_dense_deriv_blocks(int cov, bool d3, const double4* pts, int n, int ld, double var, double a, const double* gtab, double* P,
                                     int row1, int row2, hipStream_t st) {
  switch (cov) {
    case kMatern05: launch_deriv_blocks<kMatern05>(d3, pts, n, ld, var, a, gtab, P, row1, row2, st); break;
    case kMatern15: launch_deriv_blocks<kMatern15>(d3, pts, n, ld, var, a, gtab, P, row1, row2, st); break;
    case kMatern25: launch_deriv_blocks<kMatern25>(d3, pts, n, ld, var, a, gtab, P, row1, row2, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_dense_fisher_sums(const double* P, int n, int np, int ld, double* part, hipStream_t st) {
  const int ntiles = dense_grad_num_tiles(np);
  hipLaunchKernelGGL(dense_fisher_sums_kernel, dim3(ntiles), dim3(256), 0, st, P, n, np, ld, part, ntiles);
  return hipGetLastError();
}
int dense_grad_num_tiles(int np) { const int nt = (np + CT - 1) / CT; return nt * (nt + 1) / 2; }
hipError_t launch_dense_grad(int cov, bool d3, const double4* pts, int n, int np, int ld, double var, double a, const double* gtab,
                             const double* P2, const double* ya, double* part, hipStream_t st) {
  const int ntiles = dense_grad_num_tiles(np);
  switch (cov) {
    case kMatern05: launch_grad<kMatern05>(d3, pts, n, np, ld, var, a, gtab, P2, ya, part, ntiles, st); break;
    case kMatern15: launch_grad<kMatern15>(d3, pts, n, np, ld, var, a, gtab, P2, ya, part, ntiles, st); break;
    case kMatern25: launch_grad<kMatern25>(d3, pts, n, np, ld, var, a, gtab, P2, ya, part, ntiles, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- the same solves over many workgroups: one launch per 64-wide block step -------------------------------------------------
// trsv_lower_kernel walks the whole triangle with ONE workgroup (49 ms at n = 16 384 next to a 68 ms factorisation).  Here step b of the
// forward substitution is one launch: every workgroup solves the 64 x 64 diagonal block against the current right-hand side block
// redundantly (64 sequential steps of one wavefront out of LDS, the same arithmetic everywhere) and then subtracts L[i, block b] z_b
// from ITS 256 rows below; the solved block goes to a separate vector, so no workgroup ever reads what another one writes in the same
// launch.  The backward substitution mirrors it (column access of L: coalesced across the rows above).  n / 64 launches each way.
__global__ __launch_bounds__(1024) void trsv_fwd_step_kernel(const double* __restrict__ P, int np, int ld, int b0, double* __restrict__ t,
                                                             double* __restrict__ z) {
  __shared__ double sz[TB];
  __shared__ double sD[TB][TB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 1024) sD[e >> 6][e & 63] = P[(size_t)(b0 + (e >> 6)) * ld + b0 + (e & 63)];
  __syncthreads();
  if (tid < TB) {
    double zi = t[b0 + tid];
    for (int k = 0; k < TB; ++k) {
      const double lkk = sD[k][k];
      const double zk = __shfl(zi, k, 64) / lkk;
      if (tid == k) zi = zk;
      if (tid > k) zi = __builtin_fma(-sD[tid][k], zk, zi);
    }
    sz[tid] = zi;
    if (blockIdx.x == 0) z[b0 + tid] = zi;
  }
  __syncthreads();
  // rows below: t[i] -= L[i][b0:b0+64] . z_block   (4 threads per row, 16 columns each; 256 rows per workgroup)
  const int sub = tid & 3;
  const int i = b0 + TB + blockIdx.x * 256 + (tid >> 2);
  if (i < np) {
    const double* row = P + (size_t)i * ld + b0 + sub * 16;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_fma(row[k], sz[sub * 16 + k], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (sub == 0) t[i] -= acc;
  }
}
__global__ __launch_bounds__(1024) void trsv_bwd_step_kernel(const double* __restrict__ P, int ld, int b0, double* __restrict__ t,
                                                             double* __restrict__ x) {
  __shared__ double sx[TB];
  __shared__ double sD[TB][TB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 1024) sD[e >> 6][e & 63] = P[(size_t)(b0 + (e >> 6)) * ld + b0 + (e & 63)];
  __syncthreads();
  if (tid < TB) {
    double xi = t[b0 + tid];
    for (int k = TB - 1; k >= 0; --k) {
      const double lkk = sD[k][k];
      const double xk = __shfl(xi, k, 64) / lkk;
      if (tid == k) xi = xk;
      if (tid < k) xi = __builtin_fma(-sD[k][tid], xk, xi);
    }
    sx[tid] = xi;
    if (blockIdx.x == 0) x[b0 + tid] = xi;
  }
  __syncthreads();
  const int i = blockIdx.x * 1024 + tid;                 // rows above: t[i] -= sum_k L[b0+k][i] x[b0+k]
  if (i < b0) {
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < TB; ++k) acc = __builtin_fma(P[(size_t)(b0 + k) * ld + i], sx[k], acc);
    t[i] -= acc;
  }
}
// t = y (zero beyond n);   out[0] = z'z, out[1] = 2 sum log L_ii over the first n rows (one workgroup, fixed order)
__global__ void trsv_init_kernel(const double* __restrict__ y, int n, int np, double* __restrict__ t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) t[i] = i < n ? y[i] : 0.0;
}
__global__ __launch_bounds__(1024) void trsv_sums_kernel(const double* __restrict__ P, int n, int ld, const double* __restrict__ z, double* __restrict__ out) {
  __shared__ double sred[1024];
  const int tid = threadIdx.x;
  double q = 0.0, lgd = 0.0;
  for (int i = tid; i < n; i += 1024) { q = __builtin_fma(z[i], z[i], q); lgd += log(P[(size_t)i * ld + i]); }
  sred[tid] = q; __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) { if (tid < w) sred[tid] += sred[tid + w]; __syncthreads(); }
  if (tid == 0) out[0] = sred[0];
  __syncthreads();
  sred[tid] = lgd; __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) { if (tid < w) sred[tid] += sred[tid + w]; __syncthreads(); }
  if (tid == 0) out[1] = 2.0 * sred[0];
}

// z (np) = L^-1 y, out = {z'z, log-det}; x_out (np, optional) = L^-T z; work: np doubles of scratch
hipError_t launch_dense_solve(const double* P, int n, int np, int ld, const double* y, double* z, double* out, double* x_out,
                              hipStream_t st, double* work) {
  if (work == nullptr) {          // no scratch: the one-workgroup form
    hipLaunchKernelGGL(trsv_lower_kernel, dim3(1), dim3(1024), 0, st, P, n, np, ld, y, z, out, x_out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(trsv_init_kernel, dim3((np + 255) / 256), dim3(256), 0, st, y, n, np, work);
  for (int b0 = 0; b0 < np; b0 += TB) {
    const int below = np - b0 - TB;
    hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3(below > 0 ? (below + 255) / 256 : 1), dim3(1024), 0, st, P, np, ld, b0, work, z);
  }
  hipLaunchKernelGGL(trsv_sums_kernel, dim3(1), dim3(1024), 0, st, P, n, ld, (const double*)z, out);
  if (x_out) {
    (void)hipMemcpyAsync(work, z, sizeof(double) * (size_t)np, hipMemcpyDeviceToDevice, st);
    for (int b0 = np - TB; b0 >= 0; b0 -= TB)
      hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(b0 > 0 ? (b0 + 1023) / 1024 : 1), dim3(1024), 0, st, P, ld, b0, work, x_out);
  }
  return hipGetLastError();
}

hipError_t launch_dense_solve_backward(const double* P, int np, int ld, double* work, double* x_out, hipStream_t st) {
  for (int b0 = np - TB; b0 >= 0; b0 -= TB)
    hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(b0 > 0 ? (b0 + 1023) / 1024 : 1), dim3(1024), 0, st, P, ld, b0, work, x_out);
  return hipGetLastError();
}

}  // namespace gpb
