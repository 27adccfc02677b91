// gpboost_amd/csrc/vecchia_kernels.hip
//
// Fused Vecchia factor kernels for gfx950 (CDNA4).  One launch does, per point i of the
// Vecchia ordering, everything the reference's per-point loop does
// (src/GPBoost/Vecchia_utils.cpp:1461-1683, CalcCovFactorGradientVecchia) *and* the
// reductions that follow it (include/GPBoost/re_model_template.h:9960-9968 u = B y,
// y^T Psi^-1 y = sum u_i^2 / D_i; :2946-2948 log|Psi| = sum log D_i; :1988-2011 the
// covariance-parameter gradient), so B, D^-1, dB, dD never exist in memory unless a
// caller asks for A and D (MODE_FACTOR).
//
// Mapping (this is the MI355X design, not the reference's):
//   * 16 lanes (one DPP row) per point, 4 points per 64-lane wavefront, 16 per workgroup.
//   * The point's augmented system is held row-per-lane in registers:
//       rows 0..MT-1   the (padded) neighbours            C_nn + nugget
//       row  MT        the point itself                   [c^T, sigma1^2 + nugget]
//       row  MT+1      the responses                      [y_nn^T, y_i]
//     row r lives in lane r%16, register slot r/16.  Right-looking elimination of
//     columns 0..MT-1 leaves D_i in entry (MT, MT) and u_i = (B y)_i in entry (MT+1, MT):
//     no triangular solves, no A_i, for the likelihood.
//   * The rank-1 updates use v_fmac_f64 with the DPP row_newbcast modifier: the
//     multiplier L[c][k] is broadcast from the lane that owns row c *inside* the FMA, so
//     the elimination costs one fp64 VALU op per (slot, c, k) and no LDS traffic.
//   * Short rows (i < m) and m < MT are padded with decoupled dummy neighbours placed
//     1e30 apart (their covariances underflow to exactly 0), so there is no divergence.
//   * Neighbour records {x0,x1,x2,y} (32 B) are gathered once per point into LDS; column
//     operands of the kernel evaluations are LDS broadcast reads.
//   * Block partial sums are written per workgroup and reduced by a second, single-block
//     kernel in a fixed order: results are bit-reproducible run to run.
//
// Roofline notes: algorithmic HBM bytes per point are
// 4m + 8d(m+1) + 8(m+1); the kernel is fp64-VALU bound (m(m+1)/2 exp+sqrt and ~m^3/3
// FMAs per point), see DESIGN.md.
#include "dev_common.h"
#include "vecchia_kernels.h"

namespace gpb {

namespace {

// Dummy neighbours sit kDummySpacing apart: far enough that every kernel value underflows to exactly 0
// (a * 1e30 >> 745, checked on the host), small enough that r^3 stays finite in the Matern-2.5 derivative.
constexpr double kDummySpacing = 1e30;

template <int MT>
struct Layout {
  static_assert(MT >= 1 && MT <= 62, "1 <= MT <= 62");
  static_assert(MT % 16 != 15, "row MT and row MT+1 must share a register slot");
  static constexpr int R = MT + 2;                 // rows
  static constexpr int NS = (R + 15) / 16;         // register slots per lane
  static constexpr int PS = MT / 16, PL = MT % 16; // slot / lane of the point's own row
  static constexpr int YS = (MT + 1) / 16, YL = (MT + 1) % 16;  // y-row
  static constexpr int NCOL = MT + 1;              // columns 0..MT
  __host__ __device__ static constexpr int cmax(int s) { return (16 * s + 15 < MT) ? 16 * s + 15 : MT; }
};

template <bool D3>
__device__ __forceinline__ double sq_dist(const double4& p, const double4& q) {
  const double dx = p.x - q.x, dy = p.y - q.y;
  double d2 = dx * dx;
  d2 = __builtin_fma(dy, dy, d2);
  if constexpr (D3) {
    const double dz = p.z - q.z;
    d2 = __builtin_fma(dz, dz, d2);
  }
  return d2;
}

// d/d log(a) of the kernel, transformed scale (transf_scale == true):
// include/GPBoost/cov_fcts.h:2182-2193 (cm) and :2535-2554.
template <int COV>
__device__ __forceinline__ double matern_dlog_range(double dist, double var, double a, const double* tab) {
  const double r = a * dist;
  const double e = fast_exp_neg(-r, tab);
  if constexpr (COV == kMatern05) return -r * var * e;                       // cm d sigma, cm = -a
  else if constexpr (COV == kMatern15) return -var * r * r * e;              // cm d^2 e^{-ad}, cm = -var a^2
  else return -var * (1.0 / 3.0) * r * r * __builtin_fma(1.0, r, 1.0) * e;   // cm/3 d^2 (1+ad) e^{-ad}
}

}  // namespace

// MODE_NLL    : partial sums {sum log D, sum u^2/D, #(D<=0)} only
// MODE_FACTOR : additionally A[n][m], D[n], u[n] to HBM
// MODE_GRAD   : partial sums for the nll terms and the two parameter gradients
template <int MT, int COV, bool D3, int MODE>
__global__ __launch_bounds__(256) void vecchia_point_kernel(VecchiaKernelArgs args) {
  using L = Layout<MT>;
  constexpr int NS = L::NS;
  constexpr bool kNeedSolve = (MODE != MODE_NLL);

  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ double4 s_pts[16][NS * 16];
  __shared__ double s_red[GPB_NUM_PARTIALS][16];
  __shared__ double s_inv[kNeedSolve ? 16 : 1][kNeedSolve ? MT : 1];
  __shared__ double s_A[(MODE == MODE_GRAD) ? 16 : 1][(MODE == MODE_GRAD) ? MT + 2 : 1];
  __shared__ double s_b[(MODE == MODE_GRAD) ? 16 : 1][(MODE == MODE_GRAD) ? MT + 2 : 1];

  const int tid = threadIdx.x;
  const int g = tid >> 4;   // point within the workgroup
  const int l = tid & 15;   // lane within the point's DPP row
  fill_exp_table(s_tab, args.exp_tab);

  const long long i_raw = (long long)args.i_begin + (long long)blockIdx.x * 16 + g;
  const bool active = i_raw < (long long)args.i_end;
  const int i = active ? (int)i_raw : args.i_end - 1;   // inactive groups redo the last point, contribute 0
  const int m = args.m;
  const double var = args.var, a = args.a;

  // ---- gather the rows' records ------------------------------------------------
  double4 own[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = 16 * s + l;
    int idx = -1;
    if (r < m) idx = args.nn[(size_t)i * m + r];
    else if (r == MT) idx = i;
    double4 p;
    if (idx >= 0) p = args.pts[idx];
    else p = make_double4(kDummySpacing * (double)(r + 1), 0.0, 0.0, 0.0);
    own[s] = p;
    s_pts[g][r] = p;
  }
  __syncthreads();

  // ---- assemble the augmented matrix, row-per-lane, in registers ----------------
  // include/GPBoost/cov_fcts.h:634-755 (CalculateCovMat) + Vecchia_utils.cpp:1599-1609
  double M[NS][L::NCOL];
  static_for<0, NS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    static_for<0, L::cmax(s) + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      const double4 q = s_pts[g][c];
      double v = matern_cov<COV>(fast_sqrt(sq_dist<D3>(own[s], q)), var, a, s_tab);
      if constexpr (c >= 16 * s) v = (l == c - 16 * s) ? ((c == MT) ? args.diag_i : args.diag_nn) : v;
      if constexpr (s == L::YS) v = (l == L::YL) ? q.w : v;
      M[s][c] = v;
    });
  });

  // ---- right-looking elimination of columns 0..MT-1 -----------------------------
  // stands in for Eigen LLT + solve (Vecchia_utils.cpp:1617-1623)
  static_for<0, MT>([&](auto k_) {
    constexpr int k = decltype(k_)::value;
    constexpr int sk = k / 16, lk = k % 16;
    const double piv = GPB_ROW_BCAST(lk, M[sk][k]);
    const double inv = fast_rsqrt(piv);
    if constexpr (kNeedSolve) { if (l == 0) s_inv[g][k] = inv; }
    static_for<sk, NS>([&](auto s_) { M[decltype(s_)::value][k] *= inv; });
    static_for<k + 1, MT + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      constexpr int sc = c / 16, lc = c % 16;
      static_for<sc, NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        GPB_ROW_FNMA(lc, M[s][c], M[sc][k], M[s][k]);
      });
    });
  });

  const double Dv = GPB_ROW_BCAST(L::PL, M[L::PS][MT]);   // D_i  (Vecchia_utils.cpp:1623; the reference stores 1/D_i, :1682)
  const double uv = GPB_ROW_BCAST(L::YL, M[L::YS][MT]);   // u_i = (B y)_i
  const double Dinv = 1.0 / Dv;

  double red[GPB_NUM_PARTIALS];
#pragma unroll
  for (int t = 0; t < GPB_NUM_PARTIALS; ++t) red[t] = 0.0;
  red[GPB_P_LOGDET] = log(Dv);
  red[GPB_P_QUAD] = uv * uv * Dinv;
  red[GPB_P_BAD] = (Dv > 0.0) ? 0.0 : 1.0;

  if constexpr (kNeedSolve) {
    // ---- back-substitution x = L^-T (row), for the point's row (-> A_i) and the y-row (-> b_i = C^-1 y_nn)
    double X[MT];
    static_for<0, MT>([&](auto k_) { X[decltype(k_)::value] = M[L::PS][decltype(k_)::value]; });
    __syncthreads();   // s_inv visible
    static_for_down<0, MT>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      constexpr int sj = j / 16, lj = j % 16;
      const double xj = X[j] * s_inv[g][j];
      X[j] = xj;
      static_for<0, j>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        GPB_ROW_FNMA(lj, X[k], M[sj][k], xj);
      });
    });
    // lane PL now holds A_i (Vecchia_utils.cpp:1618), lane YL holds b_i.  No DPP below this line.
    if constexpr (MODE == MODE_FACTOR) {
      if (active && l == L::PL) {
        double* Arow = args.A + (size_t)i * m;
        static_for<0, MT>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if (k < m) Arow[k] = X[k];
        });
      }
      if (active && l == 0) { args.D[i] = Dv; args.u[i] = uv; }
    }
    if constexpr (MODE == MODE_GRAD) {
      // extended vectors over rows 0..MT: At = (A, -1), bt = (b, 0); y-row gets (0, 0)
      if (l == L::PL) { static_for<0, MT>([&](auto k_) { s_A[g][decltype(k_)::value] = X[decltype(k_)::value]; }); s_A[g][MT] = -1.0; s_A[g][MT + 1] = 0.0; }
      if (l == L::YL) { static_for<0, MT>([&](auto k_) { s_b[g][decltype(k_)::value] = X[decltype(k_)::value]; }); s_b[g][MT] = 0.0; s_b[g][MT + 1] = 0.0; }
      __syncthreads();
      // range parameter: accD = sum_{c<r<=MT} dK_rc At_r At_c ; accU = sum dK_rc (bt_r At_c + bt_c At_r)
      // (dD_range = 2 accD, (dB_range y)_i = accU; derivation in DESIGN.md, restating
      //  Vecchia_utils.cpp:1640-1652 without forming dA_i)
      double accD = 0.0, accU = 0.0, sAA = 0.0, sbA = 0.0;
      static_for<0, NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        const int r = 16 * s + l;
        const double Ar = s_A[g][r < MT + 2 ? r : MT + 1];
        const double br = s_b[g][r < MT + 2 ? r : MT + 1];
        if (r < MT) { sAA = __builtin_fma(Ar, Ar, sAA); sbA = __builtin_fma(br, Ar, sbA); }
        constexpr int CM = (16 * s + 14 < MT - 1) ? 16 * s + 14 : MT - 1;   // strictly-lower columns
        static_for<0, CM + 1>([&](auto c_) {
          constexpr int c = decltype(c_)::value;
          const double4 q = s_pts[g][c];
          double dk = matern_dlog_range<COV>(fast_sqrt(sq_dist<D3>(own[s], q)), var, a, s_tab);
          if constexpr (c >= 16 * s) dk = (l > c - 16 * s) ? dk : 0.0;      // keep c < r only
          const double Ac = s_A[g][c], bc = s_b[g][c];
          accD = __builtin_fma(dk * Ar, Ac, accD);
          accU = __builtin_fma(dk, __builtin_fma(br, Ac, bc * Ar), accU);
        });
      });
      // reduce the four accumulators over the 16 lanes of the row (xor butterflies stay inside the row)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        accD += __shfl_xor(accD, off, 16);
        accU += __shfl_xor(accU, off, 16);
        sAA += __shfl_xor(sAA, off, 16);
        sbA += __shfl_xor(sbA, off, 16);
      }
      const double up = uv * Dinv;                       // u' = D^-1 B y  (re_model_template.h:1999)
      // variance (ipar 0): dD = D - nugget - sum A^2 (Gaussian: nugget = 1), (dB y)_i = -sum b_r A_r
      const double dD_var = Dv - args.nugget - sAA;
      const double uk_var = -sbA;
      const double dD_rng = 2.0 * accD;
      const double uk_rng = accU;
      red[GPB_P_G1_VAR] = uk_var * up - 0.5 * up * up * dD_var;   // (uk.u - 0.5 u^T dD u) pieces (:2004)
      red[GPB_P_G2_VAR] = 0.5 * Dinv * dD_var;                    // 0.5 sum D^-1 dD
      red[GPB_P_G1_RNG] = uk_rng * up - 0.5 * up * up * dD_rng;
      red[GPB_P_G2_RNG] = 0.5 * Dinv * dD_rng;
    }
  }

  // ---- workgroup partial sums, fixed order ---------------------------------------
  if (l == 0) {
#pragma unroll
    for (int t = 0; t < GPB_NUM_PARTIALS; ++t) s_red[t][g] = active ? red[t] : 0.0;
  }
  __syncthreads();
  if (tid < GPB_NUM_PARTIALS) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += s_red[tid][q];
    args.partials[(size_t)blockIdx.x * GPB_NUM_PARTIALS + tid] = acc;
  }
}

#ifndef GPB_INSTANTIATE_MT
// Deterministic final reduction: one workgroup, each thread strides over the block partials,
// then a fixed-shape tree.  out[t] = sum_b partials[b][t].
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
                                                              double* __restrict__ out) {
  __shared__ double s[256];
  for (int t = 0; t < GPB_NUM_PARTIALS; ++t) {
    double acc = 0.0, comp = 0.0;   // Kahan: 62,500 block partials at n = 1e6
    for (int b = threadIdx.x; b < nblocks; b += 256) {
      const double v = partials[(size_t)b * GPB_NUM_PARTIALS + t] - comp;
      const double tmp = acc + v;
      comp = (tmp - acc) - v;
      acc = tmp;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[t] = s[0];
    __syncthreads();
  }
}

// pts[i].w = y[i]
__global__ void pack_y_kernel(double4* __restrict__ pts, const double* __restrict__ y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pts[i].w = y[i];
}

// u = B y from a stored factor (re_model_template.h:9965)
__global__ void vecchia_By_kernel(const double* __restrict__ A, const int* __restrict__ nn, int n, int m,
                                  const double* __restrict__ y, double* __restrict__ u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = y[i];
  for (int j = 0; j < m; ++j) {
    const int c = nn[(size_t)i * m + j];
    if (c >= 0) s = __builtin_fma(-A[(size_t)i * m + j], y[c], s);
  }
  u[i] = s;
}

// w = B^T v via the transposed index (CSR over columns): w_j = v_j - sum_{e in T[j]} A_flat[e] v[e / m]
__global__ void vecchia_Bt_kernel(const double* __restrict__ A, const int* __restrict__ t_ptr,
                                  const int* __restrict__ t_pos, int n, int m, const double* __restrict__ v,
                                  double* __restrict__ w) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = v[j];
  const int e0 = t_ptr[j], e1 = t_ptr[j + 1];
  for (int e = e0; e < e1; ++e) {
    const int pos = t_pos[e];
    s = __builtin_fma(-A[pos], v[pos / m], s);
  }
  w[j] = s;
}

// v = u / D elementwise
__global__ void scale_by_Dinv_kernel(const double* __restrict__ u, const double* __restrict__ D, int n,
                                     double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = u[i] / D[i];
}

// ---- on-device self-test of the fp64 DPP primitives -----------------------------
// out[lane] = {asm bcast, builtin bcast, asm fnma, fma-with-builtin-bcast} for LANE = 5 and 11
__global__ void dpp_selftest_kernel(const double* __restrict__ in, double* __restrict__ out) {
  const int t = threadIdx.x;
  const double x = in[t], y = in[64 + t], z = in[128 + t];
  const double b1 = row_bcast<5>(x);
  const double b2 = row_bcast_builtin<5>(x);
  double acc1 = z;
  row_fnma<11>(acc1, x, y);
  const double acc2 = __builtin_fma(-row_bcast_builtin<11>(x), y, z);
  out[t * 4 + 0] = b1; out[t * 4 + 1] = b2; out[t * 4 + 2] = acc1; out[t * 4 + 3] = acc2;
}

#endif  // !GPB_INSTANTIATE_MT

// ---- launchers --------------------------------------------------------------------
// The heavy template is compiled once per padded neighbour count in its own translation unit
// (-DGPB_INSTANTIATE_MT=<MT>), so the build parallelises; the dispatcher TU has no template code.
#ifdef GPB_INSTANTIATE_MT
#ifndef GPB_INSTANTIATE_MODE
#error "define GPB_INSTANTIATE_MODE (0 nll, 1 factor, 2 grad) together with GPB_INSTANTIATE_MT"
#endif
template <int MT, bool D3>
static hipError_t launch_cov(int cov, const VecchiaKernelArgs& args, int nblocks, hipStream_t st) {
  switch (cov) {
    case kMatern05: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern05, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    case kMatern15: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern15, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    case kMatern25: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern25, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#define GPB_CAT4_(a, b, c, d) a##b##c##d
#define GPB_CAT4(a, b, c, d) GPB_CAT4_(a, b, c, d)
hipError_t GPB_CAT4(launch_vecchia_mt, GPB_INSTANTIATE_MT, _mode, GPB_INSTANTIATE_MODE)(bool d3, int cov, const VecchiaKernelArgs& args,
                                                                                      int nblocks, hipStream_t st) {
  return d3 ? launch_cov<GPB_INSTANTIATE_MT, true>(cov, args, nblocks, st)
            : launch_cov<GPB_INSTANTIATE_MT, false>(cov, args, nblocks, st);
}
#else   // dispatcher TU
#define GPB_CASE(MTV)                                                                                   \
  hipError_t launch_vecchia_mt##MTV##_mode0(bool, int, const VecchiaKernelArgs&, int, hipStream_t);       \
  hipError_t launch_vecchia_mt##MTV##_mode1(bool, int, const VecchiaKernelArgs&, int, hipStream_t);       \
  hipError_t launch_vecchia_mt##MTV##_mode2(bool, int, const VecchiaKernelArgs&, int, hipStream_t);
GPB_MT_CASES
#undef GPB_CASE

int vecchia_padded_m(int m) {
  const int sizes[] = {GPB_MT_LIST};
  for (int s : sizes) if (m <= s) return s;
  return -1;
}

hipError_t launch_vecchia_point_kernel(int mode, int cov, bool d3, const VecchiaKernelArgs& args, hipStream_t st) {
  const int npts = args.i_end - args.i_begin;
  if (npts <= 0) return hipErrorInvalidValue;
  const int nblocks = (npts + 15) / 16;
  const int mt = vecchia_padded_m(args.m);
  switch (mt) {
#define GPB_CASE(MTV)                                                                       \
  case MTV:                                                                                   \
    if (mode == MODE_NLL) return launch_vecchia_mt##MTV##_mode0(d3, cov, args, nblocks, st);  \
    if (mode == MODE_FACTOR) return launch_vecchia_mt##MTV##_mode1(d3, cov, args, nblocks, st); \
    if (mode == MODE_GRAD) return launch_vecchia_mt##MTV##_mode2(d3, cov, args, nblocks, st);  \
    return hipErrorInvalidValue;
    GPB_MT_CASES
#undef GPB_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_reduce_partials(const double* partials, int nblocks, double* out, hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, out);
  return hipGetLastError();
}
hipError_t launch_pack_y(double4* pts, const double* y, int n, hipStream_t st) {
  hipLaunchKernelGGL(pack_y_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, y, n);
  return hipGetLastError();
}
hipError_t launch_By(const double* A, const int* nn, int n, int m, const double* y, double* u, hipStream_t st) {
  hipLaunchKernelGGL(vecchia_By_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, nn, n, m, y, u);
  return hipGetLastError();
}
hipError_t launch_Bt(const double* A, const int* t_ptr, const int* t_pos, int n, int m, const double* v, double* w,
                     hipStream_t st) {
  hipLaunchKernelGGL(vecchia_Bt_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, t_ptr, t_pos, n, m, v, w);
  return hipGetLastError();
}
hipError_t launch_scale_by_Dinv(const double* u, const double* D, int n, double* v, hipStream_t st) {
  hipLaunchKernelGGL(scale_by_Dinv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, u, D, n, v);
  return hipGetLastError();
}
hipError_t launch_dpp_selftest(const double* in, double* out, hipStream_t st) {
  hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, st, in, out);
  return hipGetLastError();
}

#endif  // GPB_INSTANTIATE_MT

}  // namespace gpb
