// gpboost_amd/csrc/laplace_kernels.hip
//
// Device kernels of the Vecchia-Laplace approximation for non-Gaussian likelihoods (Bernoulli-logit first):
// BASELINE config 4 / SURVEY.md section 8 row a13.  They implement the building blocks of
//   FindModePostRandEffCalcMLLVecchia        include/GPBoost/likelihoods.h:3773-4059
//   CGVecchiaLaplaceVec / CGTridiagVecchiaLaplace   src/GPBoost/CG_utils.cpp:21-229   ("vadu" preconditioner)
// on vectors / n x t column-major blocks that live in HBM; the (short) control flow stays on the host
// (gpb_laplace.inc).  Sigma^-1 = B^T D^-1 B with B = I - A from the MODE_FACTOR output of vecchia_point_kernel.
//
// MI355X mapping:
//   * every reduction is "one workgroup per column": 1024 lanes stride over the column in a fixed order and finish with
//     a fixed tree, so dot products are bit-reproducible and the CG scalars (a, b, Lanczos coefficients) are produced
//     on the device by the same kernel that consumes them -- one host sync per CG iteration (the convergence test);
//   * the two sparse triangular solves of the VADU preconditioner P^-1 = B^-1 (D^-1 + W)^-1 B^-T are level-scheduled:
//     rows are grouped by dependency depth once per neighbour table (depth ~ 400 at n = 1e5, m = 30), one workgroup per
//     right-hand side walks the levels with a barrier in between; the 50 probe vectors of the stochastic Lanczos
//     quadrature are 50 concurrent workgroups;
//   * B x is a row gather (A row contiguous, 30 x 8 B), B^T x uses the transposed neighbour index built for y_aux.
#include <hip/hip_runtime.h>
#include <math.h>
#include "laplace_kernels.h"

namespace gpb {

namespace {
__device__ __forceinline__ double sigmoid_stable(double x) {   // include/GPBoost/DF_utils.h:37-46
  if (x >= 0.0) { const double t = exp(-x); return 1.0 / (1.0 + t); }
  const double t = exp(x);
  return t / (1.0 + t);
}
__device__ __forceinline__ double softplus(double x) {         // DF_utils.h:57-60
  return log1p(exp(-fabs(x))) + fmax(x, 0.0);
}
// fixed-order block reduction of two values; result valid in thread 0
__device__ __forceinline__ void block_reduce2(double& a, double& b, double* s) {
  const int tid = threadIdx.x;
  s[tid] = a; s[1024 + tid] = b;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (tid < w) { s[tid] += s[tid + w]; s[1024 + tid] += s[1024 + tid + w]; }
    __syncthreads();
  }
  a = s[0]; b = s[1024];
  __syncthreads();
}
}  // namespace

// W = p (1 - p), grad = y - p, rhs = W mode + grad, dw = 1/D + W     (likelihoods.h:3882-3891, :12477, :13307, :16330)
__global__ void logit_newton_setup_kernel(const double* __restrict__ mode, const int* __restrict__ y, const double* __restrict__ D,
                                          int n, double* __restrict__ W, double* __restrict__ rhs, double* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = sigmoid_stable(mode[i]);
  const double w = p * (1.0 - p);
  W[i] = w;
  if (rhs) rhs[i] = w * mode[i] + ((double)y[i] - p);
  dw[i] = 1.0 / D[i] + w;
}

// out(:, c) = B x(:, c) [optionally scaled by 1/D]
__global__ void lap_B_kernel(const double* __restrict__ A, const int* __restrict__ nn, const double* __restrict__ D, int n, int m,
                             const double* __restrict__ x, double* __restrict__ out, int scale_dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t off = (size_t)blockIdx.y * n;
  const double* xc = x + off;
  double s = xc[i];
  const double* Ai = A + (size_t)i * m;
  const int* ni = nn + (size_t)i * m;
  for (int j = 0; j < m; ++j) { const int c = ni[j]; if (c >= 0) s = __builtin_fma(-Ai[j], xc[c], s); }
  out[off + i] = scale_dinv ? s * (1.0 / D[i]) : s;
}

// v(:, c) = B^T tmp(:, c) + W .* h(:, c)     (W may be NULL)
__global__ void lap_Bt_plus_kernel(const double* __restrict__ A, const int* __restrict__ t_ptr, const int* __restrict__ t_pos, int n, int m,
                                   const double* __restrict__ tmp, const double* __restrict__ W, const double* __restrict__ h,
                                   double* __restrict__ v) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t off = (size_t)blockIdx.y * n;
  const double* tc = tmp + off;
  double s = tc[j];
  for (int e = t_ptr[j]; e < t_ptr[j + 1]; ++e) { const int pos = t_pos[e]; s = __builtin_fma(-A[pos], tc[pos / m], s); }
  if (W) s = __builtin_fma(W[j], h[off + j], s);
  v[off + j] = s;
}

// one workgroup: out2 = { sum_i y_i x_i - softplus(x_i),  sum_i Bx_i^2 / D_i }   (likelihoods.h:3808-3812, :3955-3959)
__global__ __launch_bounds__(1024) void logit_objective_kernel(const double* __restrict__ x, const int* __restrict__ y, const double* __restrict__ Bx,
                                                               const double* __restrict__ D, int n, double* __restrict__ out2) {
  __shared__ double s[2048];
  double ll = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    ll += (double)y[i] * x[i] - softplus(x[i]);
    if (Bx) q = __builtin_fma(Bx[i] * (1.0 / D[i]), Bx[i], q);
  }
  block_reduce2(ll, q, s);
  if (threadIdx.x == 0) { out2[0] = ll; out2[1] = q; }
}

// column c: rz = r.z, hv = h.v, a = rz / hv; keeps a_old, rz_old          (CG_utils.cpp:73-75 / :170-171)
__global__ __launch_bounds__(1024) void cg_alpha_kernel(const double* __restrict__ r, const double* __restrict__ z, const double* __restrict__ h,
                                                        const double* __restrict__ v, int n, CgScalars sc) {
  __shared__ double s[2048];
  const int c = blockIdx.x;
  const size_t off = (size_t)c * n;
  double rz = 0.0, hv = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { rz = __builtin_fma(r[off + i], z[off + i], rz); hv = __builtin_fma(h[off + i], v[off + i], hv); }
  block_reduce2(rz, hv, s);
  if (threadIdx.x == 0) { sc.a_old[c] = sc.a[c]; sc.a[c] = rz / hv; sc.rz_old[c] = rz; }
}

// column c: u += a h, r -= a v, rnorm[c] = ||r||                            (CG_utils.cpp:76-79 / :172-175)
__global__ __launch_bounds__(1024) void cg_update_kernel(double* __restrict__ u, double* __restrict__ r, const double* __restrict__ h,
                                                         const double* __restrict__ v, int n, CgScalars sc) {
  __shared__ double s[2048];
  const int c = blockIdx.x;
  const size_t off = (size_t)c * n;
  const double a = sc.a[c];
  double rr = 0.0, dummy = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    if (u) u[off + i] = __builtin_fma(a, h[off + i], u[off + i]);
    const double ri = __builtin_fma(-a, v[off + i], r[off + i]);
    r[off + i] = ri;
    rr = __builtin_fma(ri, ri, rr);
  }
  block_reduce2(rr, dummy, s);
  if (threadIdx.x == 0) sc.rnorm[c] = sqrt(rr);
}

// column c: b = (r.z) / rz_old, h = z + b h; Lanczos coefficients of iteration j (CG_utils.cpp:97-99 / :205-213)
__global__ __launch_bounds__(1024) void cg_beta_kernel(const double* __restrict__ r, const double* __restrict__ z, double* __restrict__ h, int n,
                                                       CgScalars sc, int j, int p_max) {
  __shared__ double s[2048];
  __shared__ double s_b;
  const int c = blockIdx.x;
  const size_t off = (size_t)c * n;
  double rz = 0.0, dummy = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) rz = __builtin_fma(r[off + i], z[off + i], rz);
  block_reduce2(rz, dummy, s);
  if (threadIdx.x == 0) {
    const double b_old = sc.b[c];
    const double b = rz / sc.rz_old[c];
    sc.b[c] = b;
    s_b = b;
    if (sc.Td) {
      sc.Td[(size_t)c * p_max + j] = 1.0 / sc.a[c] + b_old / sc.a_old[c];
      if (j > 0) sc.Ts[(size_t)c * p_max + j - 1] = sqrt(b_old) / sc.a_old[c];
    }
  }
  __syncthreads();
  const double b = s_b;
  for (int i = threadIdx.x; i < n; i += 1024) h[off + i] = __builtin_fma(b, h[off + i], z[off + i]);
}

// ---- level-scheduled sparse triangular solves of the VADU preconditioner -----------------------------------------
// x_row = rhs_row [/ dw_row] + sum_e val_e x[src_e] with rows grouped by dependency depth.  The depth is ~ 400 at n = 1e5
// (the first m points form a dense chain) and more than half of the levels hold fewer than 64 rows, so the solves are
// LATENCY bound: what matters is the number of dependent memory round trips per level.  Layout and schedule are built for
// exactly one:
//   * rows are stored in LEVEL ORDER (position q): a 32-wide head (src = -1 padded) plus an overflow CSR for longer rows, so
//     the address of all matrix data depends on q only, never on a row-index load;
//   * 16 lanes (one DPP row) share a matrix row: its gathers are all in flight at once and the partial products are summed
//     with 4 DPP steps in a fixed order (bit-reproducible);
//   * a 1024-lane workgroup (64 groups x 2 rows) walks the rounds of all levels as a 3-stage software pipeline: while round r
//     gathers the solution entries it depends on, the matrix data / right-hand sides of round r+1 and the row indices of
//     round r+2 are already being fetched; after a level's barrier only the gather of freshly written entries remains.
// One workgroup per right-hand side; the 50 probe vectors of the stochastic Lanczos quadrature run as 50 workgroups.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
// sum over the 16 lanes of a DPP row, result in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}

namespace {
constexpr int TRI_R = 2;                 // rows per 16-lane group and round
constexpr int TRI_ROWS = 64 * TRI_R;     // rows per round of the workgroup
struct TriRound { int L, qb, b1; };      // level, first position of the round, end of the level
struct TriA { int i[TRI_R], ob[TRI_R], oe[TRI_R]; };
struct TriB { int hs[TRI_R][2], os[TRI_R][2]; double ha[TRI_R][2], oa[TRI_R][2], num[TRI_R], den[TRI_R]; };
}  // namespace

template <bool SCALE>
__global__ __launch_bounds__(1024) void lap_sptrsv_kernel(LapTri T, int n, const double* __restrict__ rhs, const double* __restrict__ dw,
                                                          double* x) {
  const size_t off = (size_t)blockIdx.x * n;
  const double* __restrict__ rc = rhs + off;
  double* xc = x + off;
  const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int nlev = T.nlev;
  // next round; ptr_next = T.ptr[d.L + 2] (fetched by the caller a whole round earlier: no load on the critical path)
  auto advance = [&](TriRound d, int ptr_next) -> TriRound {
    TriRound o = d;
    o.qb = d.qb + TRI_ROWS;
    if (o.qb >= d.b1 && d.L < nlev) { o.L = d.L + 1; o.qb = d.b1; o.b1 = (o.L < nlev) ? ptr_next : d.b1; }
    return o;
  };
  auto ptr_at = [&](int l) -> int { return T.ptr[l < nlev ? l : nlev]; };
  auto slot = [&](const TriRound& d, int s) -> int { const int q = d.qb + grp + 64 * s; return q < n ? q : n - 1; };
  auto issueA = [&](const TriRound& d, TriA& a) {
#pragma unroll
    for (int s = 0; s < TRI_R; ++s) {
      const int q = slot(d, s);
      a.i[s] = T.rows[q]; a.ob[s] = T.optr[q]; a.oe[s] = T.optr[q + 1];
    }
  };
  auto issueB = [&](const TriRound& d, const TriA& a, TriB& b) {
#pragma unroll
    for (int s = 0; s < TRI_R; ++s) {
      const int q = slot(d, s);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        b.hs[s][k] = T.hsrc[(size_t)q * 32 + lane + 16 * k];
        b.ha[s][k] = T.hval[(size_t)q * 32 + lane + 16 * k];
        const int e = a.ob[s] + lane + 16 * k;
        const bool in = e < a.oe[s];
        const int ec = in ? e : 0;
        const int src = T.osrc[ec];
        const double val = T.oval[ec];
        b.os[s][k] = in ? src : -1;
        b.oa[s][k] = in ? val : 0.0;
      }
      b.num[s] = rc[a.i[s]];
      b.den[s] = SCALE ? dw[a.i[s]] : 1.0;
    }
  };
  auto finish = [&](const TriRound& d, const TriA& a, const TriB& b) {
    double acc[TRI_R];
#pragma unroll
    for (int s = 0; s < TRI_R; ++s) {
      double g[4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        g[k] = xc[b.hs[s][k] >= 0 ? b.hs[s][k] : 0];
        g[2 + k] = xc[b.os[s][k] >= 0 ? b.os[s][k] : 0];
      }
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {        // masked slots read x[0], which may hold anything (even NaN): select, don't multiply by 0
        sum = __builtin_fma(b.ha[s][k], b.hs[s][k] >= 0 ? g[k] : 0.0, sum);
        sum = __builtin_fma(b.oa[s][k], b.os[s][k] >= 0 ? g[2 + k] : 0.0, sum);
      }
      acc[s] = sum;
    }
#pragma unroll
    for (int s = 0; s < TRI_R; ++s) {          // rows longer than 64 entries (rare): the remaining overflow entries
      for (int e = a.ob[s] + 32 + lane; e < a.oe[s]; e += 16) acc[s] = __builtin_fma(T.oval[e], xc[T.osrc[e]], acc[s]);
    }
#pragma unroll
    for (int s = 0; s < TRI_R; ++s) {
      const double tot = row16_sum(acc[s]);
      const bool live = d.L < nlev && d.qb + grp + 64 * s < d.b1;
      if (lane == 0 && live) xc[a.i[s]] = (SCALE ? b.num[s] / b.den[s] : b.num[s]) + tot;
    }
  };
  TriRound d0{0, T.ptr[0], T.ptr[1]};
  TriRound d1 = advance(d0, ptr_at(2));
  TriRound d2 = advance(d1, ptr_at(d1.L + 2));
  TriA a0, a1, a2;
  TriB b0, b1;
  issueA(d0, a0);
  issueB(d0, a0, b0);
  issueA(d1, a1);
  while (d0.L < nlev) {
    const int ptr_next = ptr_at(d2.L + 2);
    issueA(d2, a2);
    issueB(d1, a1, b1);
    finish(d0, a0, b0);
    if (d0.qb + TRI_ROWS >= d0.b1) __syncthreads();      // last round of its level
    a0 = a1; b0 = b1; a1 = a2;
    d0 = d1; d1 = d2; d2 = advance(d2, ptr_next);
  }
}

// once per evaluation: the factor's A in the level-ordered head / overflow layout of a solve
__global__ void lap_permute_factor_kernel(const double* __restrict__ A, const int* __restrict__ hpos, const int* __restrict__ opos, size_t nh,
                                          size_t novf, double* __restrict__ hval, double* __restrict__ oval) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < nh) { const int pos = hpos[g]; hval[g] = pos >= 0 ? A[pos] : 0.0; }
  if (g < novf) oval[g] = A[opos[g]];
}

// misc elementwise
__global__ void lap_lincomb_kernel(double* __restrict__ out, const double* __restrict__ x, const double* __restrict__ y, double cx, double cy, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cx * x[i] + cy * y[i];
}
// probes: R(:, c) <- sqrt(dw) .* randvec(:, c)      (likelihoods.h:16481-16487, before the B^T product)
__global__ void lap_scale_probes_kernel(const double* __restrict__ rv, const double* __restrict__ dw, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t off = (size_t)blockIdx.y * n;
  out[off + i] = sqrt(dw[i]) * rv[off + i];
}
// out2 = { sum log(1/D), sum log(dw) }
__global__ __launch_bounds__(1024) void lap_logsums_kernel(const double* __restrict__ D, const double* __restrict__ dw, int n, double* __restrict__ out2) {
  __shared__ double s[2048];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { a += log(1.0 / D[i]); b += log(dw[i]); }
  block_reduce2(a, b, s);
  if (threadIdx.x == 0) { out2[0] = a; out2[1] = b; }
}
// out2 = { x.y, sum |x| }
__global__ __launch_bounds__(1024) void lap_dot_kernel(const double* __restrict__ x, const double* __restrict__ y, int n, double* __restrict__ out2) {
  __shared__ double s[2048];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { a = __builtin_fma(x[i], y[i], a); b += fabs(x[i]); }
  block_reduce2(a, b, s);
  if (threadIdx.x == 0) { out2[0] = a; out2[1] = b; }
}

// ---- launchers --------------------------------------------------------------------------------------------
#define GRID1(n) dim3(((n) + 255) / 256), dim3(256)
hipError_t lap_newton_setup(const double* mode, const int* y, const double* D, int n, double* W, double* rhs, double* dw, hipStream_t st) {
  hipLaunchKernelGGL(logit_newton_setup_kernel, GRID1(n), 0, st, mode, y, D, n, W, rhs, dw);
  return hipGetLastError();
}
hipError_t lap_apply(const LapMat& B, const double* W, const double* h, double* v, double* tmp, int ncol, hipStream_t st) {
  hipLaunchKernelGGL(lap_B_kernel, dim3((B.n + 255) / 256, ncol), dim3(256), 0, st, B.A, B.nn, B.D, B.n, B.m, h, tmp, 1);
  hipLaunchKernelGGL(lap_Bt_plus_kernel, dim3((B.n + 255) / 256, ncol), dim3(256), 0, st, B.A, B.t_ptr, B.t_pos, B.n, B.m, tmp, W, h, v);
  return hipGetLastError();
}
hipError_t lap_B(const LapMat& B, const double* x, double* out, int ncol, hipStream_t st) {
  hipLaunchKernelGGL(lap_B_kernel, dim3((B.n + 255) / 256, ncol), dim3(256), 0, st, B.A, B.nn, B.D, B.n, B.m, x, out, 0);
  return hipGetLastError();
}
hipError_t lap_Bt(const LapMat& B, const double* x, double* out, int ncol, hipStream_t st) {
  hipLaunchKernelGGL(lap_Bt_plus_kernel, dim3((B.n + 255) / 256, ncol), dim3(256), 0, st, B.A, B.t_ptr, B.t_pos, B.n, B.m, x,
                     (const double*)nullptr, (const double*)nullptr, out);
  return hipGetLastError();
}
hipError_t lap_objective(const double* x, const int* y, const double* Bx, const double* D, int n, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(logit_objective_kernel, dim3(1), dim3(1024), 0, st, x, y, Bx, D, n, out2);
  return hipGetLastError();
}
hipError_t lap_vadu(const LapMat& B, const LapLevels& lv, const double* dw, const double* r, double* z, double* t, int ncol, hipStream_t st) {
  hipLaunchKernelGGL(lap_sptrsv_kernel<false>, dim3(ncol), dim3(1024), 0, st, lv.bwd, B.n, r, (const double*)nullptr, t);   // B^T t = r
  hipLaunchKernelGGL(lap_sptrsv_kernel<true>, dim3(ncol), dim3(1024), 0, st, lv.fwd, B.n, t, dw, z);                         // (D^-1 + W) B z = t
  return hipGetLastError();
}
hipError_t lap_permute_factor(const double* A, const int* hpos, const int* opos, size_t nh, size_t novf, double* hval, double* oval, hipStream_t st) {
  const size_t cnt = nh > novf ? nh : novf;
  hipLaunchKernelGGL(lap_permute_factor_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, A, hpos, opos, nh, novf, hval, oval);
  return hipGetLastError();
}
hipError_t lap_cg_alpha(const double* r, const double* z, const double* h, const double* v, int n, int ncol, const CgScalars& sc, hipStream_t st) {
  hipLaunchKernelGGL(cg_alpha_kernel, dim3(ncol), dim3(1024), 0, st, r, z, h, v, n, sc);
  return hipGetLastError();
}
hipError_t lap_cg_update(double* u, double* r, const double* h, const double* v, int n, int ncol, const CgScalars& sc, hipStream_t st) {
  hipLaunchKernelGGL(cg_update_kernel, dim3(ncol), dim3(1024), 0, st, u, r, h, v, n, sc);
  return hipGetLastError();
}
hipError_t lap_cg_beta(const double* r, const double* z, double* h, int n, int ncol, const CgScalars& sc, int j, int p_max, hipStream_t st) {
  hipLaunchKernelGGL(cg_beta_kernel, dim3(ncol), dim3(1024), 0, st, r, z, h, n, sc, j, p_max);
  return hipGetLastError();
}
hipError_t lap_lincomb(double* out, const double* x, const double* y, double cx, double cy, int n, hipStream_t st) {
  hipLaunchKernelGGL(lap_lincomb_kernel, GRID1(n), 0, st, out, x, y, cx, cy, n);
  return hipGetLastError();
}
hipError_t lap_scale_probes(const double* rv, const double* dw, int n, int ncol, double* out, hipStream_t st) {
  hipLaunchKernelGGL(lap_scale_probes_kernel, dim3((n + 255) / 256, ncol), dim3(256), 0, st, rv, dw, n, out);
  return hipGetLastError();
}
hipError_t lap_logsums(const double* D, const double* dw, int n, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(lap_logsums_kernel, dim3(1), dim3(1024), 0, st, D, dw, n, out2);
  return hipGetLastError();
}
hipError_t lap_dot(const double* x, const double* y, int n, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(lap_dot_kernel, dim3(1), dim3(1024), 0, st, x, y, n, out2);
  return hipGetLastError();
}

}  // namespace gpb
