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
//   * B x and B^T x use the same level-ordered head / overflow storage as the solves (16 lanes per row);
//   * all vectors of this workspace live in a STORAGE ORDER sigma = Morton rank of the coordinates (not the random Vecchia
//     ordering): a point's neighbours are spatially close, so their entries of x share cache lines -- the gathers of one
//     row touch a few lines instead of m, which is what bounds these kernels (one line per clock and CU).
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include "laplace_kernels.h"

namespace gpb {

namespace {
// NC doubles of one row of an NC-column chunk (block vectors of the probe block are stored [chunk][row][NC]: the NC columns
// of a chunk share every index / coefficient load of the solves and products, and one 32-byte sector serves a whole gather)
template <int NC> struct alignas(NC >= 2 ? 16 : 8) VecN { double v[NC]; };
template <int NC, int LS = NC>        // LS = doubles per row of the layout (NC < LS: one column of a wider chunk)
__device__ __forceinline__ VecN<NC> ldvec(const double* base, unsigned row) {
  return *reinterpret_cast<const VecN<NC>*>(reinterpret_cast<const char*>(base) + row * (unsigned)(8 * LS));
}
__device__ __forceinline__ double sigmoid_stable(double x) {   // include/GPBoost/DF_utils.h:37-46
  if (x >= 0.0) { const double t = exp(-x); return 1.0 / (1.0 + t); }
  const double t = exp(x);
  return t / (1.0 + t);
}
__device__ __forceinline__ double softplus(double x) {         // DF_utils.h:57-60
  return log1p(exp(-fabs(x))) + fmax(x, 0.0);
}
// Bernoulli-probit (likelihood id 1): log Phi(x) exactly as GPBoost::normalLogCDF (DF_utils.h:74-92) and the inverse Mills ratio
// phi(z) / Phi(z) as InvMillsRatioNormalPhi (:94-98); with z = x for y = 1 and z = -x for y = 0:
//   log p(y | x) = log Phi(z)                          (LogLikBernoulliProbit, likelihoods.h:11385-11392)
//   d/dx         = +- phi(z) / Phi(z)                   (FirstDerivLogLikBernoulliProbit, :12459-12466)
//   -d2/dx2      = r (z + r), r = phi(z) / Phi(z)       (SecondDerivNegLogLikBernoulliProbit, :13282-13291)
__device__ __forceinline__ double normal_log_cdf(double x) {
  if (x < 0.0) {
    const double e = erfc(-x * 0.70710678118654752440);
    if (e > 0.0) return log(0.5) + log(e);
    const double u = -x, u2 = u * u;
    const double series = 1.0 - 1.0 / u2 + 3.0 / (u2 * u2);
    return -0.5 * u2 - log(u) - 0.5 * log(2 * 3.14159265358979323846) + log(series);
  }
  const double Q = 0.5 * erfc(x * 0.70710678118654752440);
  if (Q == 0.0) return 0.0;
  return log1p(-Q);
}
__device__ __forceinline__ double inv_mills_phi(double z) {
  return exp((-z * z / 2. - 0.91893853320467274178) - normal_log_cdf(z));
}
// per-observation pieces of the supported likelihoods: LINK 0 = Bernoulli-logit, 1 = Bernoulli-probit, 2 = Poisson (log link:
// LogLikPoisson without the normalising constant, FirstDerivLogLikPoisson, SecondDerivNegLogLikPoisson; likelihoods.h:11407-11415,
// :12481-12483, :13315-13317); round 5, likelihoods with an auxiliary parameter `aux` (shape): 3 = gamma (log link, real-valued response:
// LogLikGamma :11872-11880, FirstDerivLogLikGamma :12485-12487, SecondDerivNegLogLikGamma :13319-13321), 4 = negative_binomial (LogLikNegBin
// :11882-11890, FirstDerivLogLikNegBin :12489-12492, SecondDerivNegLogLikNegBin :13323-13327).  The response reaches them as a double
// (LikResp::at: the int label, or gamma's real value).
// (round 5: the logit / probit links also take a REAL response in [0, 1] -- binomial_logit / binomial_probit (y = successes / trials, the trials are the sample
//  weights) and quasi_bernoulli_logit / _probit: LogLikBernoulliLogit<double> is linear in y; the probit terms are y f(1) + (1 - y) f(0) with the two Bernoulli
//  branches, exactly the end points at y = 0 and y = 1: LogLikBinomialProbit :11394-11398, FirstDeriv :12468-12474, SecondDeriv :13293-13305, third :13800-13820)
template <int LINK>
__device__ __forceinline__ double resp_at(const LikResp& r, int d) {
  if constexpr (LINK == 3 || LINK == 5 || LINK == 6 || LINK == 7 || LINK == 8) return r.yd[d];
  else if constexpr (LINK == 0 || LINK == 1) return r.yd ? r.yd[d] : (double)r.yi[d];
  else return (double)r.yi[d];
}
// LINK 5 = beta (round 5, second slice; mean = sigmoid(location), precision = aux, real-valued response in (0, 1)): LogLikBeta likelihoods.h:11903-11913,
// FirstDerivLogLikBeta :12501-12507, SecondDerivNegLogLikBeta :13336-13346, third derivative :13892-13917.  digamma / trigamma / tetragamma as
// src/GPBoost/DF_utils.cpp:82-201 (recurrence to >= 8.5 / 5 / 8, then the asymptotic series), sigmoid_stable_clamped as include/GPBoost/DF_utils.h:48-55.
__device__ __forceinline__ double digamma_dev(double x);
__device__ __forceinline__ double trigamma_dev(double x) {
  if (x <= 0.0001) return 1.0 / x / x;
  double value = 0.0, z = x;
  while (z < 5.0) { value = value + 1.0 / z / z; z = z + 1.0; }
  const double y = 1.0 / z / z;
  return value + 0.5 * y + (1.0 + y * (0.1666666667 + y * (-0.03333333333 + y * (0.02380952381 + y * -0.03333333333)))) / z;
}
__device__ __forceinline__ double tetragamma_dev(double x) {
  if (x <= 1e-4) return -2.0 / (x * x * x);
  double z = x, value = 0.0;
  while (z < 8.0) { value -= 2.0 / (z * z * z); z += 1.0; }
  const double z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z6 = z4 * z2, z8 = z4 * z4, z10 = z8 * z2;
  value += -1.0 / z2 - 1.0 / z3 - 0.5 / z4 + 1.0 / (6.0 * z6) - 1.0 / (6.0 * z8) + 3.0 / (10.0 * z10);
  return value;
}
__device__ __forceinline__ double sigmoid_clamped(double x) { double mu = sigmoid_stable(x); if (mu < 1e-12) mu = 1e-12; if (mu > 1.0 - 1e-12) mu = 1.0 - 1e-12; return mu; }
// sample weight of datum d (round 5; likelihoods.h:666-668 weights_): every per-datum term -- log-likelihood and its derivatives -- is multiplied by it
__device__ __forceinline__ double wt_at(const LikResp& r, int d) { return r.w ? r.w[d] : 1.0; }
template <int LINK>
__device__ __forceinline__ void lik_grad_info(double y, double x, double aux, double& grad, double& w, double aux2 = 0.0) {
  if constexpr (LINK == 0) {
    const double p = sigmoid_stable(x);
    grad = y - p;                         // likelihoods.h:12477
    w = p * (1.0 - p);                    // :13307
  } else if constexpr (LINK == 1) {
    if (y == 0.0 || y == 1.0) {
      const double z = y != 0.0 ? x : -x;
      const double r = inv_mills_phi(z);
      grad = y != 0.0 ? r : -r;
      w = r * (z + r);
    } else {                                // a proportion: both Bernoulli branches, mixed
      const double r1 = inv_mills_phi(x), r0 = inv_mills_phi(-x);
      grad = y * r1 + (1.0 - y) * -r0;
      w = y * r1 * (x + r1) + (1.0 - y) * -r0 * (x - r0);
    }
  } else if constexpr (LINK == 8) {       // gaussian_latent (round 6: the Gaussian likelihood through the Laplace machinery, aux = error variance): FirstDerivLogLikGaussian (likelihoods.h:12514-12516), information 1 / aux (:12956-12962)
    grad = (y - x) / aux;
    w = 1.0 / aux;
  } else if constexpr (LINK == 7) {       // lognormal (round 5, fourth slice; mean of y = exp(location), aux = variance of log y): FirstDerivLogLikLogNormal (likelihoods.h:12534-12538), SecondDerivNegLogLikLogNormal (:13384-13386)
    grad = (log(y) - (x - 0.5 * aux)) / aux;
    w = 1.0 / aux;
  } else if constexpr (LINK == 6) {       // t, fisher_laplace: FirstDerivLogLikT (likelihoods.h:12509-12512), FisherInformationT (:13358-13360): aux = scale, aux2 = df
    const double res = y - x;
    grad = (aux2 + 1.0) * res / (aux2 * aux * aux + res * res);
    w = (aux2 + 1.0) / (aux2 + 3.0) / (aux * aux);
  } else if constexpr (LINK == 5) {
    const double mu = sigmoid_clamped(x), logit_y = log(y) - log1p(-y);
    const double dig1 = digamma_dev((1.0 - mu) * aux), dig2 = digamma_dev(mu * aux);
    const double tri1 = trigamma_dev((1.0 - mu) * aux), tri2 = trigamma_dev(mu * aux);
    grad = aux * mu * (1.0 - mu) * (dig1 - dig2 + logit_y);
    const double h1 = -aux * aux * mu * mu * (1.0 - mu) * (1.0 - mu) * (tri1 + tri2);
    const double h2 = aux * mu * (1.0 - mu) * (1.0 - 2.0 * mu) * (dig1 - dig2 + logit_y);
    w = -(h1 + h2);
  } else if constexpr (LINK == 3) {
    const double q = y * exp(-x);
    grad = aux * (q - 1.0);
    w = aux * q;
  } else if constexpr (LINK == 4) {
    const double mu = exp(x), mr = mu + aux;
    grad = y - (y + aux) / mr * mu;
    w = (y + aux) * mu * aux / (mr * mr);
  } else {
    const double e = exp(x);
    grad = y - e;
    w = e;
  }
}
template <int LINK>
__device__ __forceinline__ double lik_loglik(double y, double x, double aux, double aux2 = 0.0) {
  if constexpr (LINK == 0) return y * x - softplus(x);      // likelihoods.h:11401-11403
  else if constexpr (LINK == 1) {
    if (y == 0.0 || y == 1.0) return normal_log_cdf(y != 0.0 ? x : -x);
    return y * normal_log_cdf(x) + (1.0 - y) * normal_log_cdf(-x);
  }
  else if constexpr (LINK == 3) return -aux * (x + y * exp(-x));
  else if constexpr (LINK == 4) return y * x - (y + aux) * log(exp(x) + aux);
  else if constexpr (LINK == 7) { const double z = log(y) - (x - 0.5 * aux); return -0.5 * z * z / aux; }      // LogLikLogNormal (:11950-11958) without its constant
  else if constexpr (LINK == 8) { const double res = y - x; return -res * res / 2.0 / aux; }                     // LogLikGaussian (:11927-11936) without its constant
  else if constexpr (LINK == 6) return -(aux2 + 1.0) / 2.0 * log(1.0 + (y - x) * (y - x) / (aux2 * aux * aux));       // LogLikT (:11915-11925) without its constant
  else if constexpr (LINK == 5) {
    const double mu = sigmoid_clamped(x);
    return -lgamma(mu * aux) - lgamma((1.0 - mu) * aux) + (mu * aux - 1.0) * log(y) + ((1.0 - mu) * aux - 1.0) * log1p(-y);
  }
  else return y * x - exp(x);
}
// digamma as GPBoost::digamma (src/GPBoost/DF_utils.cpp:82-125): small-argument approximation, recurrence up to x >= 8.5, de Moivre's expansion
__device__ __forceinline__ double digamma_dev(double x) {
  if (x <= 0.000001) return -0.57721566490153286060 - 1.0 / x + 1.6449340668482264365 * x;
  double v = 0.0;
  while (x < 8.5) { v -= 1.0 / x; x += 1.0; }
  double r = 1.0 / x;
  v += log(x) - 0.5 * r;
  r = r * r;
  return v - r * (1.0 / 12.0 - r * (1.0 / 120.0 - r * (1.0 / 252.0 - r * (1.0 / 240.0 - r * (1.0 / 132.0)))));
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

// W = information, grad = first derivative, rhs = W mode + grad, dw = 1/D + W, rdw = 1/dw     (likelihoods.h:3882-3891, :16330)
// Repeated locations (dptr != nullptr; Vecchia_utils.cpp:1156-1168, re_comp.h:863-885): the latent process lives on the unique locations, row i
// is a RANDOM EFFECT whose data are y[dptr[i] .. dptr[i + 1]) (grouped by random effect in the storage order of the rows), and every
// likelihood term of the row is the sum over its data (first_deriv_ll_ / information_ll_ on the random-effect scale, CalcZtVGivenIndices).
template <int LINK>
__global__ void lik_newton_setup_kernel(const double* __restrict__ mode, const LikResp y, const double* __restrict__ fe, const double* __restrict__ D,
                                          int n, double* __restrict__ W, double* __restrict__ rhs, double* __restrict__ dw, double* __restrict__ rdw,
                                          const int* __restrict__ dptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double gr, w;
  if (dptr) {
    gr = 0.0; w = 0.0;
    for (int d = dptr[i]; d < dptr[i + 1]; ++d) { double g1, w1; lik_grad_info<LINK>(resp_at<LINK>(y, d), fe ? mode[i] + fe[d] : mode[i], y.aux, g1, w1, y.aux2); const double wd = wt_at(y, d); gr += wd * g1; w += wd * w1; }
  } else {
    lik_grad_info<LINK>(resp_at<LINK>(y, i), fe ? mode[i] + fe[i] : mode[i], y.aux, gr, w, y.aux2);       // location parameter = mode + fixed effects (likelihoods.h:3861-3870)
    if (y.w) { gr *= y.w[i]; w *= y.w[i]; }
  }
  W[i] = w;
  if (rhs) rhs[i] = w * mode[i] + gr;
  const double v = 1.0 / D[i] + w;
  dw[i] = v;
  rdw[i] = 1.0 / v;       // the preconditioner's diagonal solve multiplies by this
}

// one workgroup: out2 = { sum_i log p(y_i | x_i),  sum_i Bx_i^2 / D_i }   (likelihoods.h:3808-3812, :3955-3959)
template <int LINK>
__global__ __launch_bounds__(1024) void lik_objective_kernel(const double* __restrict__ x, const LikResp y, const double* __restrict__ fe, const double* __restrict__ Bx,
                                                               const double* __restrict__ D, int n, double* __restrict__ out2, const int* __restrict__ dptr) {
  __shared__ double s[2048];
  double ll = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    if (dptr) { for (int d = dptr[i]; d < dptr[i + 1]; ++d) ll += wt_at(y, d) * lik_loglik<LINK>(resp_at<LINK>(y, d), fe ? x[i] + fe[d] : x[i], y.aux, y.aux2); }
    else ll += wt_at(y, i) * lik_loglik<LINK>(resp_at<LINK>(y, i), fe ? x[i] + fe[i] : x[i], y.aux, y.aux2);
    if (Bx) q = __builtin_fma(Bx[i] * (1.0 / D[i]), Bx[i], q);
  }
  block_reduce2(ll, q, s);
  if (threadIdx.x == 0) { out2[0] = ll; out2[1] = q; }
}

// ---- CG vector kernels --------------------------------------------------------------------------------------------
// Block vectors are stored [chunk][row][NC] (NC = 1: plain columns).  Every pass over the vectors is split over P workgroups per
// chunk (grid = (P, chunks)); dot products leave P partial sums per column in sc.part / sc.part2, and the NEXT kernel of the
// sequence adds them up in a fixed order before it uses the scalar (no atomics, no grid barrier, bit-reproducible):
//   cg_dots_kernel      partial r.z and h.v                                       (CG_utils.cpp:73-75 / :170-171)
//   cg_update_kernel    a = rz / hv from the partials; u += a h, r -= a v; partial r.r   (:76-79 / :172-175)
//   cg_rnorm_kernel     rnorm = sqrt(sum of the partial r.r)  (one value per column, read by the host)
//   cg_dots_kernel      partial r.z (after the preconditioner)
//   cg_hupdate_kernel   b = rz / rz_old from the partials, Lanczos coefficients, h = z + b h      (:97-99 / :205-213)
namespace {
__device__ __forceinline__ void slice_of(int n, int& lo, int& hi) {     // this workgroup's rows
  const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  lo = (int)blockIdx.x * per; hi = lo + per < n ? lo + per : n;
}
}  // namespace

template <int NC, bool HV>
__global__ __launch_bounds__(1024) void cg_dots_kernel(const double* __restrict__ r, const double* __restrict__ z, const double* __restrict__ h,
                                                       const double* __restrict__ v, int n, CgScalars sc) {
  __shared__ double s[2048];
  const size_t off = (size_t)blockIdx.y * n * NC;
  int lo, hi; slice_of(n, lo, hi);
  double rz[NC], hv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { rz[c] = 0.0; hv[c] = 0.0; }
  for (int i = lo + threadIdx.x; i < hi; i += 1024) {
    const VecN<NC> a = ldvec<NC>(r + off, i), b = ldvec<NC>(z + off, i);
#pragma unroll
    for (int c = 0; c < NC; ++c) rz[c] = __builtin_fma(a.v[c], b.v[c], rz[c]);
    if (HV) {
      const VecN<NC> e = ldvec<NC>(h + off, i), f = ldvec<NC>(v + off, i);
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = __builtin_fma(e.v[c], f.v[c], hv[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    block_reduce2(rz[c], hv[c], s);
    if (threadIdx.x == 0) {
      double* p = sc.part + ((size_t)(blockIdx.y * NC + c) * gridDim.x + blockIdx.x) * 2;
      p[0] = rz[c]; p[1] = hv[c];
    }
  }
}

template <int NC>
__global__ __launch_bounds__(1024) void cg_update_kernel(double* __restrict__ u, double* __restrict__ r, const double* __restrict__ h,
                                                         const double* __restrict__ v, int n, CgScalars sc) {
  __shared__ double s[2048];
  __shared__ double s_a[NC];
  const size_t off = (size_t)blockIdx.y * n * NC;
  if (threadIdx.x < NC) {
    const int col = blockIdx.y * NC + threadIdx.x;
    const double* p = sc.part + (size_t)col * gridDim.x * 2;
    double rz = 0.0, hv = 0.0;
    for (unsigned k = 0; k < gridDim.x; ++k) { rz += p[2 * k]; hv += p[2 * k + 1]; }
    const double a = rz / hv;
    s_a[threadIdx.x] = a;
    if (blockIdx.x == 0) { sc.a_old[col] = sc.a[col]; sc.a[col] = a; sc.rz_old[col] = rz; }
  }
  __syncthreads();
  double a[NC], rr[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { a[c] = s_a[c]; rr[c] = 0.0; }
  int lo, hi; slice_of(n, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += 1024) {
    const VecN<NC> hv = ldvec<NC>(h + off, i), vv = ldvec<NC>(v + off, i);
    VecN<NC> rv = ldvec<NC>(r + off, i);
    if (u) {
      VecN<NC> uv = ldvec<NC>(u + off, i);
#pragma unroll
      for (int c = 0; c < NC; ++c) uv.v[c] = __builtin_fma(a[c], hv.v[c], uv.v[c]);
      *reinterpret_cast<VecN<NC>*>(u + off + (size_t)i * NC) = uv;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) { rv.v[c] = __builtin_fma(-a[c], vv.v[c], rv.v[c]); rr[c] = __builtin_fma(rv.v[c], rv.v[c], rr[c]); }
    *reinterpret_cast<VecN<NC>*>(r + off + (size_t)i * NC) = rv;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    double dummy = 0.0;
    block_reduce2(rr[c], dummy, s);
    if (threadIdx.x == 0) sc.part2[(size_t)(blockIdx.y * NC + c) * gridDim.x + blockIdx.x] = rr[c];
  }
}

__global__ void cg_rnorm_kernel(CgScalars sc, int ncols, int P) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  double rr = 0.0;
  for (int k = 0; k < P; ++k) rr += sc.part2[(size_t)col * P + k];
  sc.rnorm[col] = sqrt(rr);
}

template <int NC>
__global__ __launch_bounds__(1024) void cg_hupdate_kernel(const double* __restrict__ z, double* __restrict__ h, int n, CgScalars sc, int j, int p_max) {
  __shared__ double s_b[NC];
  const size_t off = (size_t)blockIdx.y * n * NC;
  if (threadIdx.x < NC) {
    const int col = blockIdx.y * NC + threadIdx.x;
    const double* p = sc.part + (size_t)col * gridDim.x * 2;
    double rz = 0.0;
    for (unsigned k = 0; k < gridDim.x; ++k) rz += p[2 * k];
    const double b = rz / sc.rz_old[col];
    s_b[threadIdx.x] = b;
    if (blockIdx.x == 0) {
      const double b_old = sc.b[col];
      sc.b[col] = b;
      if (sc.Td) {
        sc.Td[(size_t)col * p_max + j] = 1.0 / sc.a[col] + b_old / sc.a_old[col];
        if (j > 0) sc.Ts[(size_t)col * p_max + j - 1] = sqrt(b_old) / sc.a_old[col];
      }
    }
  }
  __syncthreads();
  double b[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) b[c] = s_b[c];
  int lo, hi; slice_of(n, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += 1024) {
    const VecN<NC> zv = ldvec<NC>(z + off, i);
    VecN<NC> hv = ldvec<NC>(h + off, i);
#pragma unroll
    for (int c = 0; c < NC; ++c) hv.v[c] = __builtin_fma(b[c], hv.v[c], zv.v[c]);
    *reinterpret_cast<VecN<NC>*>(h + off + (size_t)i * NC) = hv;
  }
}

// ---- level-scheduled sparse triangular solves of the VADU preconditioner -----------------------------------------
// x_row = rhs_row [* rdw_row] + sum_e val_e x[src_e] with rows grouped by dependency depth.  The depth is ~ 400 at n = 1e5
// (the first m points form a dense chain) and more than half of the levels hold fewer than 64 rows, so the solves are
// LATENCY bound: what matters is the number of dependent memory round trips per level.  Layout and schedule are built for
// exactly one:
//   * the matrix is stored in LEVEL ORDER as SLOTS (position q): a 32-wide head (src = -1 padded) plus an overflow CSR, so the
//     address of all matrix data depends on q only, never on a row-index load.  A row with more than 64 entries (the early
//     points of B^T have hundreds) is split into consecutive slots of <= 64 entries that always share a round; their
//     partial sums meet in LDS, so no lane ever walks a long row serially;
//   * 16 lanes (one DPP row) share a slot: its gathers are all in flight at once and the partial products are summed with 4
//     DPP steps in a fixed order (bit-reproducible);
//   * a 512-lane workgroup (32 groups x 2 slots; 8 wavefronts, so each may hold the ~150 VGPRs of the pipeline state) walks the
//     rounds of all levels as a 3-stage software pipeline: while round r gathers the solution entries it depends on, the
//     matrix data / right-hand sides of round r+1 and the row indices of round r+2 are already being fetched; after a level's
//     barrier only the gather of freshly written entries remains.  The loop is unrolled over the pipeline's period (6) so
//     that loaded values are never copied between registers (a copy would wait for the prefetches just issued).
// Runs of narrow levels: one workgroup per right-hand side (50 for the probe block of the stochastic Lanczos quadrature);
// wide levels: own launch, ceil(slots / 64) workgroups per right-hand side (the kernel boundary is the level barrier).
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
// base[idx] with the BYTE offset formed in 32 bits: the load then takes its base from SGPRs and one VGPR of offset
// (global_load ... v_off, s[base:base+1]) instead of a 64-bit VGPR address computed per load (host guards the 4 GB range)
template <class Tp>
__device__ __forceinline__ Tp ldu(const Tp* __restrict__ base, unsigned idx) {
  return *reinterpret_cast<const Tp*>(reinterpret_cast<const char*>(base) + (idx * (unsigned)sizeof(Tp)));
}
template <int V> struct IntC { static constexpr int value = V; };
template <class F> __device__ __forceinline__ void static_for6(F&& f) { f(IntC<0>{}); f(IntC<1>{}); f(IntC<2>{}); f(IntC<3>{}); f(IntC<4>{}); f(IntC<5>{}); }
constexpr int TRI_R = 2;                       // slots per 16-lane group and round
constexpr int TRI_GROUPS = kTriThreads / 16;   // 16-lane groups per workgroup
constexpr int TRI_ROWS = kTriRowsPerRound;     // slots per round of the workgroup
static_assert(TRI_ROWS == TRI_GROUPS * TRI_R, "round size");
struct TriRound { int L, qb, b1, split; };   // level, first position of the round, end of the level, level has split rows
struct TriA { int i[TRI_R], ns[TRI_R], ob[TRI_R], oe[TRI_R]; };
template <int NC> struct TriB { int hs[TRI_R][2], os[TRI_R][2]; double ha[TRI_R][2], oa[TRI_R][2], num[TRI_R][NC], den[TRI_R]; };
}  // namespace

// Levels [L0, L1) of the solve, `nrounds` rounds for this workgroup.  nsplit == 1: one workgroup per right-hand side walks all
// rounds of these levels (barrier after each level).  nsplit > 1 (then L1 == L0 + 1, a "wide" level launched on its own so
// the kernel boundary orders it against its neighbours): workgroup `part` of the nsplit takes round `part` of that level.
//
// What bounds a round is the number of vector-memory INSTRUCTIONS the CU issues (its address unit takes ~16 clocks per
// 64-lane load, shared by all wavefronts), not latency.  Hence: the slot descriptor is one 16-byte record (one load instead
// of four), matrix entries are 16-byte {coefficient, source} records (one load instead of two), a wavefront whose groups
// have no slot in rounds r .. r+2 issues nothing, and rounds of at most 32 slots run a one-slot-per-group body.
template <bool SCALE, bool OVF, int NC, int LS>
__global__ __launch_bounds__(kTriThreads) void lap_sptrsv_kernel(LapTri T, const int* __restrict__ lv_ptr, const int* __restrict__ lv_split,
                                                                 int n, int L0, int L1, int nsplit, int nrounds,
                                                                 const double* __restrict__ rhs, const double* __restrict__ rdw, double* x) {
  // lv_ptr / lv_split (= T.ptr / T.lsplit) are separate __restrict__ arguments so that the level bookkeeping compiles to
  // scalar loads: as vector loads their results would have to be waited for with vmcnt(0), draining the prefetches
  __shared__ double s_part[2][TRI_ROWS][NC];
  // col = index of the NC-column unit this workgroup solves for; the vectors are stored [chunk][row][LS] (NC == LS: a whole
  // chunk; NC = 1 < LS: one column of a chunk -- the narrow-level runs of the probe block, where workgroups run side by side)
  const int col = blockIdx.x / nsplit, part = blockIdx.x - col * nsplit;
  const size_t off = (size_t)(col * NC / LS) * n * LS + (size_t)(col * NC % LS);
  const double* __restrict__ rc = rhs + off;
  double* xc = x + off;
  const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int wgrp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 4;     // first group of this wavefront (scalar)
  const int nlev = T.nlev, nslots = T.nslots;
  const int stride = TRI_ROWS * nsplit, first = TRI_ROWS * part;
  // next round; ptr_next = T.ptr[d.L + 2] (fetched by the caller a whole round earlier: no load on the critical path)
  auto advance = [&](TriRound d, int ptr_next, int split_next) -> TriRound {
    TriRound o = d;
    o.qb = d.qb + stride;
    if (o.qb >= d.b1 && d.L < L1) { o.L = d.L + 1; o.qb = d.b1 + first; o.b1 = (o.L < nlev) ? ptr_next : d.b1; o.split = split_next; }
    return o;
  };
  auto ptr_at = [&](int l) -> int { return lv_ptr[l < nlev ? l : nlev]; };
  auto split_at = [&](int l) -> int { return lv_split[l < nlev ? l : nlev]; };
  // does this wavefront hold any slot of round d?  (scalar: a wavefront without slots in rounds r, r+1, r+2 skips the whole
  // phase except its barriers -- no loads at all; most levels of the narrow runs occupy one or two wavefronts)
  auto wave_live = [&](const TriRound& d) -> bool { return d.L < L1 && d.qb + wgrp < d.b1; };
  // does round d use the second slot of the groups (more than 32 slots)?  Phases whose three rounds do not run the one-slot body.
  auto two_slots = [&](const TriRound& d) -> bool { return d.L < L1 && d.qb + TRI_GROUPS < d.b1; };
  // all device arrays are indexed with 32-bit unsigned offsets from uniform bases (one VGPR of address per load)
  auto slot = [&](const TriRound& d, int s) -> unsigned { const int q = d.qb + grp + TRI_GROUPS * s; return (unsigned)(q < nslots ? q : nslots - 1); };
  // Inside a body everything is straight-line (clamped, unconditional loads): the compiler's wait counts are exact there, and
  // the branches between bodies only merge at phase boundaries, where the next gathers need all prefetches anyway.
  auto issueA = [&](auto NS, const TriRound& d, TriA& a) {
#pragma unroll
    for (int s = 0; s < decltype(NS)::value; ++s) {
      const int4 m = ldu(T.meta, slot(d, s));
      a.i[s] = m.x; a.ns[s] = m.y; a.ob[s] = m.z; a.oe[s] = m.w;
    }
  };
  auto issueB = [&](auto NS, const TriRound& d, const TriA& a, TriB<NC>& b) {
#pragma unroll
    for (int s = 0; s < decltype(NS)::value; ++s) {
      const unsigned hq = slot(d, s) * 32u + (unsigned)lane;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const LapEnt h = ldu(T.hent, hq + 16u * k);
        b.hs[s][k] = h.src; b.ha[s][k] = h.val;
        if (OVF) {
          const int e = a.ob[s] + lane + 16 * k;
          const bool in = e < a.oe[s];
          const LapEnt o = ldu(T.oent, in ? (unsigned)e : 0u);      // masked: any valid entry, coefficient forced to 0
          b.os[s][k] = o.src;
          b.oa[s][k] = in ? o.val : 0.0;
        }
      }
      const unsigned row = a.i[s] >= 0 ? (unsigned)a.i[s] : 0u;      // padding slots and continuation slots never use it
      const VecN<NC> nv = ldvec<NC, LS>(rc, row);
#pragma unroll
      for (int c = 0; c < NC; ++c) b.num[s][c] = nv.v[c];
      b.den[s] = SCALE ? ldu(rdw, row) : 1.0;
    }
  };
  // part 1 of a round: gathers, prefetches, slot sums, partial sums to LDS
  auto body1 = [&](auto NS, const TriRound& d, const TriA& a, const TriB<NC>& b, int parity, const TriRound& dn, const TriA& an,
                   TriB<NC>& bn, const TriRound& dnn, TriA& ann, double (&tot)[TRI_R][NC]) {
    constexpr int ns = decltype(NS)::value;
    // (1) the gathers this round waits for go first ...
    VecN<NC> g[TRI_R][4];
#pragma unroll
    for (int s = 0; s < ns; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        g[s][k] = ldvec<NC, LS>((const double*)xc, (unsigned)b.hs[s][k]);
        if (OVF) g[s][2 + k] = ldvec<NC, LS>((const double*)xc, (unsigned)b.os[s][k]);
      }
    }
    // (2) ... then the prefetches of the next two rounds (independent of x)
    issueA(NS, dnn, ann);
    issueB(NS, dn, an, bn);
    // (3) slot sums; padding entries: coefficient 0 times a finite x (the workspace is zero-initialised)
#pragma unroll
    for (int s = 0; s < ns; ++s) {
      double sum[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          sum[c] = __builtin_fma(b.ha[s][k], g[s][k].v[c], sum[c]);
          if (OVF) sum[c] = __builtin_fma(b.oa[s][k], g[s][2 + k].v[c], sum[c]);
        }
      }
      // slots longer than 64 entries exist only when a row has > 4096 entries
      if (OVF) for (int e0 = a.ob[s] + 32 + lane; e0 - lane < a.oe[s]; e0 += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = e0 + 16 * k;
          const bool in = e < a.oe[s];
          LapEnt o = ldu(T.oent, in ? (unsigned)e : (unsigned)a.ob[s]);
          if (!in) o.val = 0.0;
          const VecN<NC> gg = ldvec<NC, LS>((const double*)xc, (unsigned)o.src);
#pragma unroll
          for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(o.val, gg.v[c], sum[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) tot[s][c] = row16_sum(sum[c]);
    }
    // (4) partial sums of split rows meet in LDS (only in levels that have any; double-buffered by round parity)
    if (d.split) {
#pragma unroll
      for (int s = 0; s < ns; ++s) if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) s_part[parity][grp + TRI_GROUPS * s][c] = tot[s][c];
      }
    }
  };
  // part 2 (after the exchange barrier): the first slot of a row adds the partial sums of its other slots and stores
  auto body2 = [&](auto NS, const TriRound& d, const TriA& a, const TriB<NC>& b, int parity, const double (&tot)[TRI_R][NC]) {
#pragma unroll
    for (int s = 0; s < decltype(NS)::value; ++s) {
      const bool live = d.L < L1 && d.qb + grp + TRI_GROUPS * s < d.b1 && a.ns[s] > 0;
      if (lane == 0 && live) {
        VecN<NC> out;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          double v = tot[s][c];
          // the continuation slots of a row follow its first slot: positions (q - qb) + 1 ... inside the same round
          for (int k = 1; k < a.ns[s]; ++k) v += s_part[parity][grp + TRI_GROUPS * s + k][c];
          out.v[c] = (SCALE ? __builtin_fma(b.num[s][c], b.den[s], v) : b.num[s][c] + v);
        }
        *reinterpret_cast<VecN<NC>*>(xc + (size_t)(unsigned)a.i[s] * LS) = out;
      }
    }
  };
  // 3-stage software pipeline, unrolled over its period (3 descriptor / stage-A buffers x 2 stage-B buffers = 6 phases) so that
  // no loaded value is ever copied between registers: a copy would make the wave wait for the prefetches it just issued
  // (zero-initialised: a wavefront that joins in executes the bodies of rounds it holds no slot of with whatever its
  // buffers contain -- zeros or the data of an older round, always valid indices)
  TriRound d[3];
  TriA a[3] = {};
  TriB<NC> b[2] = {};
  d[0] = TriRound{L0, lv_ptr[L0] + first, lv_ptr[L0 + 1], lv_split[L0]};
  d[1] = advance(d[0], ptr_at(L0 + 2), split_at(L0 + 1));
  d[2] = advance(d[1], ptr_at(d[1].L + 2), split_at(d[1].L + 1));
  issueA(IntC<2>{}, d[0], a[0]);
  issueB(IntC<2>{}, d[0], a[0], b[0]);
  issueA(IntC<2>{}, d[1], a[1]);
  for (int r0 = 0; r0 < nrounds; r0 += 6) {
    static_for6([&](auto P) {
      constexpr int p = decltype(P)::value;
      if (r0 + p < nrounds) {
        const TriRound& dc = d[p % 3];
        const TriRound& dn = d[(p + 1) % 3];
        const TriRound& dnn = d[(p + 2) % 3];
        const int ptr_next = ptr_at(dnn.L + 2), split_next = split_at(dnn.L + 1);
        const bool mine = wave_live(dc) || wave_live(dn) || wave_live(dnn);
        const bool wide = two_slots(dc) || two_slots(dn) || two_slots(dnn);
        double tot[TRI_R][NC] = {};
        if (mine) {
          if (wide) body1(IntC<2>{}, dc, a[p % 3], b[p % 2], p & 1, dn, a[(p + 1) % 3], b[(p + 1) % 2], dnn, a[(p + 2) % 3], tot);
          else body1(IntC<1>{}, dc, a[p % 3], b[p % 2], p & 1, dn, a[(p + 1) % 3], b[(p + 1) % 2], dnn, a[(p + 2) % 3], tot);
        }
        if (dc.split) __syncthreads();
        if (mine) {
          if (wide) body2(IntC<2>{}, dc, a[p % 3], b[p % 2], p & 1, tot);
          else body2(IntC<1>{}, dc, a[p % 3], b[p % 2], p & 1, tot);
        }
        if (dc.qb + stride >= dc.b1) __syncthreads();      // last round of its level
        d[p % 3] = advance(dnn, ptr_next, split_next);
      }
    });
  }
}

// Row-parallel products with the same level-ordered storage: 16 lanes per row (the group of a row's first slot walks all its
// slots), 16 slots per workgroup.
//   MODE 0: out = B x      MODE 1: out = D^-1 B x      MODE 2: out = B^T x + W .* h  (W may be NULL)
//   MODE 3: out = the plain product with the stored entries, no identity part: with the entries of dA/dtheta this is (dA) x (T = fwd
//           layout) or (dA)^T x (T = bwd layout) -- the B_grad = -dA products of the Laplace gradient
// (B x)_i = x_i - sum_j A_ij x[nn_ij] with T = fwd;  (B^T x)_j = x_j - sum_{i : j in N(i)} A_ij x_i with T = bwd.
template <int MODE, int NC>
__global__ __launch_bounds__(256) void lap_tri_spmv_kernel(LapTri T, int n, const double* __restrict__ x, const double* __restrict__ D,
                                                           const double* __restrict__ W, const double* __restrict__ h, double* __restrict__ out) {
  const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int q0 = blockIdx.x * 16 + grp;
  if (q0 >= T.nslots) return;              // whole 16-lane groups leave together
  const int4 m0 = T.meta[q0];
  const int row = m0.x, nseg = m0.y;
  if (nseg <= 0) return;                   // continuation or padding slot
  const size_t off = (size_t)blockIdx.y * n * NC;      // blockIdx.y = chunk of NC columns
  const double* __restrict__ xc = x + off;
  double sum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) sum[c] = 0.0;
  for (int q = q0; q < q0 + nseg; ++q) {
    LapEnt e[2]; VecN<NC> g[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) e[k] = ldu(T.hent, (unsigned)q * 32u + lane + 16u * k);
    const int4 mq = T.meta[q];
#pragma unroll
    for (int k = 0; k < 2; ++k) g[k] = ldvec<NC>(xc, (unsigned)e[k].src);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(e[k].val, g[k].v[c], sum[c]);
    for (int j = mq.z + lane; j < mq.w; j += 16) {
      const LapEnt o = T.oent[j];
      const VecN<NC> go = ldvec<NC>(xc, (unsigned)o.src);
#pragma unroll
      for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(o.val, go.v[c], sum[c]);
    }
  }
  double tot[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) tot[c] = row16_sum(sum[c]);
  if (lane == 0) {
    const VecN<NC> xr = ldvec<NC>(xc, (unsigned)row);
    VecN<NC> hr;
    if (MODE == 2 && W) hr = ldvec<NC>(h + off, (unsigned)row);
    VecN<NC> o;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double v = (MODE == 3) ? tot[c] : xr.v[c] - tot[c];
      if (MODE == 1) v *= 1.0 / D[row];
      if (MODE == 2 && W) v = __builtin_fma(W[row], hr.v[c], v);
      o.v[c] = v;
    }
    *reinterpret_cast<VecN<NC>*>(out + off + (size_t)row * NC) = o;
  }
}

// once per evaluation: the factor's A into the level-ordered entry records of a solve
// ---- dense blocks of the solves (laplace_kernels.h: LapDense) -------------------------------------------------------------------
namespace {
__device__ __forceinline__ double wave_sum(double v) {          // sum over the 64 lanes, result in every lane
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
}  // namespace
// One level of the block: row k of inv = e_k + sum_e A[ipos[e]] * (row icol[e] of inv); the rows it reads belong to earlier levels
// (earlier launches).  Thread = one column c <= k; entries of inv right of the diagonal are never read (c <= j guards them).
__global__ __launch_bounds__(256) void lap_dense_inv_kernel(LapDense d, const double* __restrict__ A, int k0) {
  const int k = k0 + (int)blockIdx.y;
  const int c = (int)(blockIdx.x * 256 + threadIdx.x);
  if (c > k) return;
  double acc = (c == k) ? 1.0 : 0.0;
  const int e1 = d.iptr[k + 1];
  for (int e = d.iptr[k]; e < e1; ++e) {
    const int j = d.icol[e];
    if (c <= j) acc = __builtin_fma(A[d.ipos[e]], d.inv[(size_t)j * d.ld + c], acc);
  }
  d.inv[(size_t)k * d.ld + c] = acc;
}
// Right-hand side of the block, one wavefront per block row and chunk: t = rhs [* rdw] + sum over the row's entries with a source
// OUTSIDE the block (already solved: earlier levels) of A * x[source].   tbuf[k][(chunk - c0) * NC + c], cn chunks per pass.
template <int NC, bool SCALE>
__global__ __launch_bounds__(256) void lap_dense_rhs_kernel(LapDense d, const double* __restrict__ A, int n, int c0, int cn, const double* __restrict__ rhs,
                                                            const double* __restrict__ rdw, const double* __restrict__ x) {
  const int k = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (k >= d.K) return;
  const int lane = threadIdx.x & 63, chunk = c0 + (int)blockIdx.y;
  const size_t off = (size_t)chunk * n * NC;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  const int e1 = d.optr[k + 1];
  for (int e = d.optr[k] + lane; e < e1; e += 64) {
    const double a = A[d.opos[e]];
    const VecN<NC> v = ldvec<NC>(x + off, (unsigned)d.osrc[e]);
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = __builtin_fma(a, v.v[c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = wave_sum(acc[c]);
  if (lane == 0) {
    const unsigned row = (unsigned)d.rows[k];
    const VecN<NC> num = ldvec<NC>(rhs + off, row);
    const double den = SCALE ? rdw[row] : 1.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) d.tbuf[(size_t)k * (cn * NC) + (size_t)(chunk - c0) * NC + c] = SCALE ? __builtin_fma(num.v[c], den, acc[c]) : num.v[c] + acc[c];
  }
}
// x_blk = inv * t for plain columns (NC = 1): one wavefront per block row (long rows first) and column, four independent partial sums
__global__ __launch_bounds__(256) void lap_dense_matvec_kernel(LapDense d, int n, int c0, int cn, double* __restrict__ x) {
  const int w = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (w >= d.K) return;
  const int k = d.K - 1 - w, lane = threadIdx.x & 63, col = (int)blockIdx.y;
  const double* __restrict__ row = d.inv + (size_t)k * d.ld;
  const double* __restrict__ t = d.tbuf + col;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int j = lane;
  for (; j + 192 <= k; j += 256) {
    a0 = __builtin_fma(row[j], t[(size_t)j * cn], a0);
    a1 = __builtin_fma(row[j + 64], t[(size_t)(j + 64) * cn], a1);
    a2 = __builtin_fma(row[j + 128], t[(size_t)(j + 128) * cn], a2);
    a3 = __builtin_fma(row[j + 192], t[(size_t)(j + 192) * cn], a3);
  }
  for (; j <= k; j += 64) a0 = __builtin_fma(row[j], t[(size_t)j * cn], a0);
  const double sum = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) x[(size_t)(c0 + col) * n + (unsigned)d.rows[k]] = sum;
}
// x_blk = inv * t for the probe block (chunks of 4 columns; cn <= 16 chunks per pass): a workgroup takes 32 block rows (long tiles
// first) and walks the columns of inv 64 at a time: the 32 x 64 tile of inv and the 64 x (4 cn) tile of tbuf go through LDS (coalesced
// loads, fetched into registers one tile ahead); thread (r, cg) keeps the 4 columns of chunk cg for rows r and r + 16.
// The column range of a row tile is cut into kDenseSplit equal parts (blockIdx.y) whose partial products go to `part`
// ([split][k][4 cn]) and are added in a fixed order by lap_dense_gemm4_sum_kernel: the longest row tile alone would otherwise take
// K / 64 column tiles in sequence.
__global__ __launch_bounds__(256) void lap_dense_gemm4_kernel(LapDense d, int cn, double* __restrict__ part) {
  __shared__ double s_inv[32][65];
  __shared__ alignas(32) double s_t[64 * 64];
  const int tile = (int)(gridDim.x - 1 - blockIdx.x);
  const int tid = threadIdx.x, r = tid >> 4, cg = tid & 15;
  const int k0 = tile * 32;
  const int kmax = (k0 + 31 < d.K ? k0 + 31 : d.K - 1);
  const int nt_all = kmax / 64 + 1, per = (nt_all + kDenseSplit - 1) / kDenseSplit;
  const int jt0 = (int)blockIdx.y * per, ntiles = (jt0 + per < nt_all ? jt0 + per : nt_all);      // column tiles [jt0, ntiles)
  const int ts = cn * 4;
  const size_t tend = (size_t)d.K * ts;
  double pinv[8], pt[16];
  auto fetch = [&](int jt) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = q * 256 + tid, rr = idx >> 6, j = jt * 64 + (idx & 63), kk = k0 + rr;
      pinv[q] = (kk < d.K && j <= kk) ? d.inv[(size_t)kk * d.ld + j] : 0.0;        // zero right of the diagonal: the products below run unmasked
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const size_t g = (size_t)jt * 64 * ts + (size_t)(q * 256 + tid);
      pt[q] = (q < cn && g < tend) ? d.tbuf[g] : 0.0;
    }
  };
  double acc0[4] = { 0.0, 0.0, 0.0, 0.0 }, acc1[4] = { 0.0, 0.0, 0.0, 0.0 };
  if (jt0 < ntiles) fetch(jt0);
  for (int jt = jt0; jt < ntiles; ++jt) {
    __syncthreads();                      // the previous tile has been consumed
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int idx = q * 256 + tid; s_inv[idx >> 6][idx & 63] = pinv[q]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < cn) s_t[q * 256 + tid] = pt[q];
    __syncthreads();
    if (jt + 1 < ntiles) fetch(jt + 1);
    if (cg < cn) {
#pragma unroll 8
      for (int jj = 0; jj < 64; ++jj) {
        const double a0 = s_inv[r][jj], a1 = s_inv[r + 16][jj];
        const VecN<4> v = *reinterpret_cast<const VecN<4>*>(&s_t[jj * ts + cg * 4]);
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc0[c] = __builtin_fma(a0, v.v[c], acc0[c]); acc1[c] = __builtin_fma(a1, v.v[c], acc1[c]); }
      }
    }
  }
  if (cg < cn) {
    double* pc = part + (size_t)blockIdx.y * d.K * ts + (size_t)cg * 4;
    VecN<4> o;
    if (k0 + r < d.K) {
#pragma unroll
      for (int c = 0; c < 4; ++c) o.v[c] = acc0[c];
      *reinterpret_cast<VecN<4>*>(pc + (size_t)(k0 + r) * ts) = o;
    }
    if (k0 + r + 16 < d.K) {
#pragma unroll
      for (int c = 0; c < 4; ++c) o.v[c] = acc1[c];
      *reinterpret_cast<VecN<4>*>(pc + (size_t)(k0 + r + 16) * ts) = o;
    }
  }
}
__global__ __launch_bounds__(256) void lap_dense_gemm4_sum_kernel(LapDense d, int n, int c0, int cn, const double* __restrict__ part, double* __restrict__ x) {
  const int g = (int)(blockIdx.x * 256 + threadIdx.x);          // (block row, chunk)
  if (g >= d.K * cn) return;
  const int k = g / cn, cg = g - k * cn;
  const size_t ts = (size_t)cn * 4;
  VecN<4> o = *reinterpret_cast<const VecN<4>*>(part + (size_t)k * ts + (size_t)cg * 4);
  for (int sp = 1; sp < kDenseSplit; ++sp) {
    const VecN<4> v = *reinterpret_cast<const VecN<4>*>(part + ((size_t)sp * d.K + k) * ts + (size_t)cg * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) o.v[c] += v.v[c];
  }
  *reinterpret_cast<VecN<4>*>(x + (size_t)(c0 + cg) * n * 4 + (size_t)(unsigned)d.rows[k] * 4) = o;
}

__global__ void lap_permute_factor_kernel(const double* __restrict__ A, const int* __restrict__ hpos, const int* __restrict__ opos, size_t nh,
                                          size_t novf, LapEnt* __restrict__ hent, LapEnt* __restrict__ oent) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < nh) { const int pos = hpos[g]; hent[g].val = pos >= 0 ? A[pos] : 0.0; }
  if (g < novf) oent[g].val = A[opos[g]];
}

// out[sigma[i]] = in[i]: Vecchia order -> storage order of the Laplace workspace
__global__ void lap_scatter_kernel(const double* __restrict__ in, const int* __restrict__ sigma, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[sigma[i]] = in[i];
}

// ---- predictive (co)variances of the latent process at new locations (PredictLaplaceApproxVecchia, likelihoods.h:8563-8824) ----
// Bpo has the rows b_p = -A_p of the prediction points (neighbours among the observed points, Vecchia positions nn_p, -1 padded).
// lap_pred_rhs: column c of a block (layout [chunk][storage slot][nc], zeroed by the caller) <- Bpo' e_p, p = p0 + c % cnt -- the columns
// beyond cnt repeat the first ones: finite and never read.  A point's neighbours are distinct, so every store has its own address.
__global__ void lap_pred_rhs_kernel(const int* __restrict__ nn_p, const double* __restrict__ A_p, const int* __restrict__ sigma, int n, int m, int p0,
                                    int cnt, int ncols, int nc, double* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ncols * m) return;
  const int c = g / m, j = g - c * m;
  const size_t e = (size_t)(p0 + c % cnt) * m + j;
  const int nb = nn_p[e];
  if (nb < 0) return;
  out[(size_t)(c / nc) * n * nc + (size_t)sigma[nb] * nc + (c % nc)] = -A_p[e];
}
// lap_pred_quad: with X = (Sigma^-1 + W)^-1 [the block's right-hand sides]:  out[r * cnt + c] = b_{row0 + r}' X(:, c), r < n_rows, c < cnt;
// diag_only: out[c] = b_{row0 + c}' X(:, c) (the quadratic forms of the block's own points).  One thread per output, fixed summation order.
__global__ void lap_pred_quad_kernel(const int* __restrict__ nn_p, const double* __restrict__ A_p, const int* __restrict__ sigma,
                                     const double* __restrict__ X, int n, int m, int row0, int n_rows, int cnt, int nc, int diag_only,
                                     double* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = diag_only ? cnt : n_rows * cnt;
  if (g >= total) return;
  const int r = diag_only ? g : g / cnt, c = diag_only ? g : g - r * cnt;
  const size_t e0 = (size_t)(row0 + r) * m;
  const double* Xc = X + (size_t)(c / nc) * n * nc + (c % nc);
  double acc = 0.0;
  for (int j = 0; j < m; ++j) {
    const int nb = nn_p[e0 + j];
    if (nb >= 0) acc = __builtin_fma(-A_p[e0 + j], Xc[(size_t)sigma[nb] * nc], acc);
  }
  out[g] = acc;
}

// misc elementwise
__global__ void lap_lincomb_kernel(double* __restrict__ out, const double* __restrict__ x, const double* __restrict__ y, double cx, double cy, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cx * x[i] + cy * y[i];
}
// probes: R(:, c) <- sqrt(dw) .* randvec(:, c)      (likelihoods.h:16481-16487, before the B^T product); chunked layout, nc = NC
__global__ void lap_scale_probes_kernel(const double* __restrict__ rv, const double* __restrict__ dw, int n, int nc, double* __restrict__ out) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const size_t off = (size_t)blockIdx.y * n * nc;
  out[off + g] = sqrt(dw[g / nc]) * rv[off + g];
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


// ---- gradient of the Laplace approximation (likelihoods.h:6521-6700; oracle/gpb_oracle.c: orc_vecchia_laplace_grad) ---------------
// third derivative of the log-likelihood = d information / d location parameter (CalcFirstDerivInformationLocPar, likelihoods.h:13772-13800)
template <int LINK>
__device__ __forceinline__ double lik_third(double y, double x, double aux, double aux2 = 0.0) {
  if constexpr (LINK == 0) { const double p = sigmoid_stable(x); return -p * (1.0 - p) * (2.0 * p - 1.0); }
  else if constexpr (LINK == 2) return exp(x);
  else if constexpr (LINK == 3) return -aux * y * exp(-x);                                            // likelihoods.h:13843-13849
  else if constexpr (LINK == 4) { const double mu = exp(x), mr = mu + aux; return -(y + aux) * mu * aux * (mu - aux) / (mr * mr * mr); }   // :13870-13878
  else if constexpr (LINK == 6 || LINK == 7 || LINK == 8) return 0.0;                                               // lognormal: constant information (:13929-13933); t, fisher_laplace: the information does not depend on the mode (:410-415)
  else if constexpr (LINK == 5) {                                                                      // :13892-13917
    const double mu = sigmoid_clamped(x), d = mu * (1.0 - mu), logit_y = log(y) - log1p(-y);
    const double dig1 = digamma_dev((1.0 - mu) * aux), dig2 = digamma_dev(mu * aux), tri1 = trigamma_dev((1.0 - mu) * aux), tri2 = trigamma_dev(mu * aux);
    const double tet1 = tetragamma_dev((1.0 - mu) * aux), tet2 = tetragamma_dev(mu * aux);
    const double C = dig1 - dig2 + logit_y, S = tri1 + tri2, Dlt = tet2 - tet1;
    const double term_trigam = 3.0 * aux * aux * d * d * (1.0 - 2.0 * mu) * S;
    const double term_tetragam = aux * aux * aux * d * d * d * Dlt;
    const double gp = d * ((1.0 - 2.0 * mu) * (1.0 - 2.0 * mu) - 2.0 * d);
    return term_trigam + term_tetragam + -aux * gp * C;
  }
  else {
    const double x2 = x * x;
    if (y == 0.0) { const double q = inv_mills_phi(-x); return -q * (1.0 - x2 + q * (3.0 * x - 2.0 * q)); }
    const double r = inv_mills_phi(x);
    if (y == 1.0) return -r * (x2 - 1.0 + r * (3.0 * x + 2.0 * r));
    const double q = inv_mills_phi(-x);
    return y * (-r * (x2 - 1.0 + r * (3.0 * x + 2.0 * r))) + (1.0 - y) * (-q * (1.0 - x2 + q * (3.0 * x - 2.0 * q)));
  }
}
template <int LINK>
__global__ void lik_third_kernel(const double* __restrict__ mode, const LikResp y, const double* __restrict__ fe, int n, double* __restrict__ dW3,
                                 const int* __restrict__ dptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dptr) { double t3 = 0.0; for (int d = dptr[i]; d < dptr[i + 1]; ++d) t3 += wt_at(y, d) * lik_third<LINK>(resp_at<LINK>(y, d), fe ? mode[i] + fe[d] : mode[i], y.aux, y.aux2); dW3[i] = t3; }
  else dW3[i] = wt_at(y, i) * lik_third<LINK>(resp_at<LINK>(y, i), fe ? mode[i] + fe[i] : mode[i], y.aux, y.aux2);
}

// boosting gradient for non-Gaussian data, d(-mll) / dF = -d log p / d loc + 0.5 d logdet / d mode - W .* (Sigma^-1 + W)^-1 d_mll_d_mode
// (likelihoods.h:6996-7001), from the vectors the covariance-parameter gradient leaves behind
template <int LINK>
__global__ void lik_grad_F_kernel(const double* __restrict__ mode, const LikResp y, const double* __restrict__ fe, const double* __restrict__ dld,
                                  const double* __restrict__ sv, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double gr, w;
  lik_grad_info<LINK>(resp_at<LINK>(y, i), fe ? mode[i] + fe[i] : mode[i], y.aux, gr, w, y.aux2);
  if (y.w) { gr *= y.w[i]; w *= y.w[i]; }
  out[i] = -gr + 0.5 * dld[i] - w * sv[i];
}

// the same on the DATA scale for repeated locations (use_random_effects_indices_of_data_, likelihoods.h:6944-6966 with the iterative methods'
// estimate diag((Sigma^-1 + W)^-1)_r = (d logdet / d mode)_r / (d information / d loc summed over the data of r), :6700-6703):
//   out_d = -d log p_d / d loc + 0.5 (d information_d / d loc) diag_r - information_d [(Sigma^-1 + W)^-1 d_mll_d_mode]_r,   r = random effect of d
// one thread per random effect (storage order), a handful of data each; out per datum in the storage order of the data
template <int LINK>
__global__ void lik_grad_F_map_kernel(const double* __restrict__ mode, const LikResp y, const double* __restrict__ fe, const double* __restrict__ dld,
                                      const double* __restrict__ dW3, const double* __restrict__ sv, int n, const int* __restrict__ dptr,
                                      double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t3 = dW3[i];
  const double diag = t3 == 0.0 ? 0.0 : dld[i] / t3;
  const double mi = mode[i], svi = sv[i];
  for (int d = dptr[i]; d < dptr[i + 1]; ++d) {
    const double loc = fe ? mi + fe[d] : mi;
    double gr, w;
    const double yd = resp_at<LINK>(y, d);
    lik_grad_info<LINK>(yd, loc, y.aux, gr, w, y.aux2);
    const double wd = wt_at(y, d);
    out[d] = -(wd * gr) + 0.5 * (wd * lik_third<LINK>(yd, loc, y.aux, y.aux2)) * diag - (wd * w) * svi;
  }
}

// Gradient wrt log(auxiliary parameter) of the likelihoods that have one (CalcGradNegMargLikelihoodLaplaceApproxVecchia, iterative branch,
// likelihoods.h:6743-6808): three sums over the data, from the vectors the covariance-parameter gradient leaves behind --
//   out3[0] = the data-dependent part of CalcGradNegLogLikAuxPars (:14185-14215; the host adds the terms that depend on aux and n only),
//   out3[1] = sum_d (d information_d / d log aux) diag_r(d),   diag_r = (d logdet / d mode)_r / (d information / d loc)_r  (:6700-6703),
//   out3[2] = sum_d (d^2 log p_d / d loc d log aux) s_r(d),    s = (Sigma^-1 + W)^-1 d_mll_d_mode  (CalcSecondDerivLogLikFirstDerivInformationAuxPar, :14777-14799).
// One workgroup, fixed reduction order: bit-reproducible.
template <int LINK>
__global__ __launch_bounds__(1024) void lik_aux_grad_kernel(const double* __restrict__ mode, const LikResp y, const double* __restrict__ fe,
                                                              const double* __restrict__ dld, const double* __restrict__ dW3, const double* __restrict__ sv,
                                                              int n, const int* __restrict__ dptr, double* __restrict__ out3) {
  __shared__ double s[2048];
  double e = 0.0, dsum = 0.0, isum = 0.0;
  const double r = y.aux;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const int d0 = dptr ? dptr[i] : i, d1 = dptr ? dptr[i + 1] : i + 1;
    const double t3 = dW3[i];
    const double diag = t3 == 0.0 ? 0.0 : dld[i] / t3;
    const double mi = mode[i], svi = sv[i];
    for (int d = d0; d < d1; ++d) {
      const double x = fe ? mi + fe[d] : mi, yv = resp_at<LINK>(y, d), wd = wt_at(y, d);
      if constexpr (LINK == 7) {       // lognormal: the data sum of CalcGradNegLogLikAuxPars (:14275-14286) -> e (dsum, isum unused)
        const double z = log(yv) - (x - 0.5 * r);
        e += wd * ((z + 1.0) * 0.5 - (z * z) / (2.0 * r));
      } else if constexpr (LINK == 8) {       // gaussian_latent: sum w resid^2 of CalcGradNegLogLikAuxPars (:14262-14274) -> e; the host scales it by -0.5 / aux and adds 0.5 n
        const double res = yv - x;
        e += wd * res * res;
      } else if constexpr (LINK == 6) {       // t: the two data sums of CalcGradNegLogLikAuxPars (:14241-14262): e -> log scale, dsum -> log df (isum unused)
        const double nu = y.aux2, nu_sigma2 = nu * r * r, res_sq = (yv - x) * (yv - x);
        e -= wd * (nu + 1.0) / (nu_sigma2 / res_sq + 1.0);
        dsum += wd * (-nu * log(1.0 + res_sq / nu_sigma2) + (nu + 1.0) / (1.0 + nu_sigma2 / res_sq));
      } else if constexpr (LINK == 5) {       // CalcGradNegLogLikAuxPars beta (:14229-14241; the host multiplies by -precision), CalcSecondDerivLogLikFirstDerivInformationAuxPar beta (:14816-14845)
        const double mu = sigmoid_clamped(x), dd = mu * (1.0 - mu), logit_y = log(yv) - log1p(-yv);
        const double dig1 = digamma_dev((1.0 - mu) * r), dig2 = digamma_dev(mu * r), tri1 = trigamma_dev((1.0 - mu) * r), tri2 = trigamma_dev(mu * r);
        const double tet1 = tetragamma_dev((1.0 - mu) * r), tet2 = tetragamma_dev(mu * r);
        e += wd * (digamma_dev(r) - mu * dig2 - (1.0 - mu) * dig1 + mu * log(yv) + (1.0 - mu) * log1p(-yv));
        const double C = dig1 - dig2 + logit_y, S = tri1 + tri2, Dlt_tri = (1.0 - mu) * tri1 - mu * tri2, Dlt_tet = (1.0 - mu) * tet1 + mu * tet2;
        const double cross_deriv = -(r * dd * C + r * r * dd * Dlt_tri);
        const double term1 = 2.0 * r * r * dd * dd * S, term2 = r * r * r * dd * dd * Dlt_tet;
        const double term3 = -r * dd * (1.0 - 2.0 * mu) * C, term4 = -r * r * dd * (1.0 - 2.0 * mu) * Dlt_tri;
        dsum = __builtin_fma(wd * (term1 + term2 + term3 + term4), diag, dsum);
        isum = __builtin_fma(wd * cross_deriv, svi, isum);
      } else if constexpr (LINK == 3) {
        const double q = yv * exp(-x);
        e += wd * (x + q);
        const double s2 = wd * (r * (q - 1.0));
        dsum = __builtin_fma(wd * (s2 + r), diag, dsum);       // (sic: the reference weights the first summand twice, likelihoods.h:14782-14783: w (w s2 + r))
        isum = __builtin_fma(s2, svi, isum);
      } else {
        const double mu = exp(x), mr = mu + r, yr = yv + r;
        e += wd * (r * (-digamma_dev(yr) + log(mr) + yr / mr));
        const double q = mu * r / (mr * mr);
        dsum = __builtin_fma(wd * (-q * (yv * (r - mu) - 2.0 * r * mu) / mr), diag, dsum);
        isum = __builtin_fma(wd * (q * (yv - mu)), svi, isum);
      }
    }
  }
  double z = 0.0;
  block_reduce2(e, dsum, s);
  block_reduce2(isum, z, s);
  if (threadIdx.x == 0) { out3[0] = e; out3[1] = dsum; out3[2] = isum; }
}

// dA_i / d log(a) and dD_i / d log(a) of the Vecchia factor WITHOUT nugget (Vecchia_utils.cpp:1640-1652; the range parameter is the only
// one whose derivative of A is not zero): one wavefront per point, C_nn in LDS, lane = row.
//   t = dc - dC A_i,  dA_i = C^-1 t (Cholesky of the jittered C_nn),  dD_i = -(dA_i . c + A_i . dc)
// Covariance and its range derivative on the transformed scale: cov_fcts.h:2100-2118, :2535-2554.  Once per gradient evaluation --
// the block CG around it is what the time goes to -- so this is the plain formulation, not the register-blocked one of vecchia_kernels.hip.
namespace {
__device__ __forceinline__ double cov_plain(int cov, double d, double var, double a) {
  const double r = a * d;
  if (cov == 0) return var * exp(-r);
  if (cov == 1) return var * (1.0 + r) * exp(-r);
  return var * (1.0 + r + r * r / 3.0) * exp(-r);
}
__device__ __forceinline__ double dcov_dlog_range_plain(int cov, double d, double var, double a) {
  if (cov == 0) return -a * d * var * exp(-a * d);
  const double cm = -var * a * a;
  if (cov == 1) return cm * d * d * exp(-a * d);
  const double r = a * d;
  return cm / 3.0 * d * d * (1.0 + r) * exp(-r);
}
}  // namespace
// (T lanes = rows of C_nn: 64 for m <= 62 -- 31 KB of LDS -- or 128 for m <= 126 -- 132 KB of dynamic LDS, one workgroup per CU;
//  leading dimension T - 1: odd, rows of one column fall into different banks)
// which = 0: d/dlog(range) (the Laplace gradient and the Fisher information); which = 1: d/dlog(variance ratio) of a model WITH a nugget
// (Fisher information of the Gaussian model: dC = C - nug I, dc = c  =>  dA = nug C^-1 A exactly, dD = var - dA'c - A'c).
// diag_nn = diagonal of C_nn (Gaussian: var + 1; otherwise var (1 + 1e-10), Vecchia_utils.cpp:1599-1609); nug = its nugget part.
template <int T>
__global__ __launch_bounds__(T) void lap_range_deriv_kernel(const double4* __restrict__ pts, const int* __restrict__ nn, const double* __restrict__ A, int n, int m,
                                                            int cov, int d3, double var, double a, double diag_nn, double nug, int which,
                                                            double* __restrict__ dA, double* __restrict__ dD) {
  constexpr int kDerivLd = T - 1;
  extern __shared__ double s_deriv[];                        // C[(T - 2) * kDerivLd], then seven vectors of T
  double* C = s_deriv;
  double *px = C + (T - 2) * kDerivLd, *py = px + T, *pz = py + T, *Ai = pz + T, *tv = Ai + T, *cv = tv + T, *dcv = cv + T;
  __shared__ int s_k;
  const int i = blockIdx.x, lane = threadIdx.x;
  const int idx = lane < m ? nn[(size_t)i * m + lane] : -1;
  if (lane == 0) s_k = 0;
  __syncthreads();
  if (idx >= 0) atomicAdd(&s_k, 1);                           // the valid neighbours are a prefix of the row
  __syncthreads();
  const int k = s_k;
  const double4 ctr = pts[i];
  double ox = 0.0, oy = 0.0, oz = 0.0, ai = 0.0;
  if (lane < k) {
    const double4 q = pts[idx];
    ox = q.x - ctr.x; oy = q.y - ctr.y; oz = d3 ? q.z - ctr.z : 0.0;
    ai = A[(size_t)i * m + lane];
  }
  px[lane] = ox; py[lane] = oy; pz[lane] = oz; Ai[lane] = ai;
  __syncthreads();
  if (k == 0) {
    if (lane < m) dA[(size_t)i * m + lane] = 0.0;
    if (lane == 0) dD[i] = which == 1 ? var : 0.0;
    return;
  }
  if (lane < k) {
    const int r = lane;
    const double di = sqrt(ox * ox + oy * oy + oz * oz);
    const double dcr = dcov_dlog_range_plain(cov, di, var, a);
    double tr = dcr;
    for (int q = 0; q < k; ++q) {
      if (q == r) continue;
      const double ex = ox - px[q], ey = oy - py[q], ez = oz - pz[q];
      const double dq = sqrt(ex * ex + ey * ey + ez * ez);
      tr -= dcov_dlog_range_plain(cov, dq, var, a) * Ai[q];
      if (q < r) C[r * kDerivLd + q] = cov_plain(cov, dq, var, a);
    }
    C[r * kDerivLd + r] = diag_nn;                             // Vecchia_utils.cpp:1599-1609
    cv[r] = cov_plain(cov, di, var, a);
    if (which == 1) { tv[r] = nug * Ai[r]; dcv[r] = cv[r]; }   // dc - dC A = c - (C - nug I) A = nug A;  dc = c
    else { tv[r] = tr; dcv[r] = dcr; }
  }
  __syncthreads();
  for (int j = 0; j < k; ++j) {                                // right-looking Cholesky, lower, in place
    if (lane == j) C[j * kDerivLd + j] = sqrt(C[j * kDerivLd + j]);
    __syncthreads();
    if (lane > j && lane < k) C[lane * kDerivLd + j] /= C[j * kDerivLd + j];
    __syncthreads();
    if (lane > j && lane < k) {
      const double lj = C[lane * kDerivLd + j];
      for (int c = j + 1; c <= lane; ++c) C[lane * kDerivLd + c] -= lj * C[c * kDerivLd + j];
    }
    __syncthreads();
  }
  for (int j = 0; j < k; ++j) {                                // L z = t (in tv)
    if (lane == j) tv[j] /= C[j * kDerivLd + j];
    __syncthreads();
    if (lane > j && lane < k) tv[lane] -= C[lane * kDerivLd + j] * tv[j];
    __syncthreads();
  }
  for (int j = k - 1; j >= 0; --j) {                           // L^T x = z
    if (lane == j) tv[j] /= C[j * kDerivLd + j];
    __syncthreads();
    if (lane < j) tv[lane] -= C[j * kDerivLd + lane] * tv[j];
    __syncthreads();
  }
  const double xr = lane < k ? tv[lane] : 0.0;
  if (lane < m) dA[(size_t)i * m + lane] = xr;
  double part = lane < k ? xr * cv[lane] + Ai[lane] * dcv[lane] : 0.0;
  __syncthreads();
  tv[lane] = part;
  __syncthreads();
  if (lane == 0) {
    double sacc = 0.0;
    for (int r = 0; r < k; ++r) sacc += tv[r];
    dD[i] = (which == 1 ? var : 0.0) - sacc;
  }
}

// d log|Sigma W + I| / d mode_i with the per-row control variate of the vadu preconditioner (CalcLogDetStochDerivModeVecchia,
// likelihoods.h:16660-16688; CalcOptimalCVectorized, CG_utils.cpp:1070-1090).  Block vectors in the chunk layout [chunk][row][nc].
__global__ void lap_row_stats_kernel(const double* __restrict__ U, const double* __restrict__ PIZ, const double* __restrict__ BPIZ, const double* __restrict__ dW3,
                                     const double* __restrict__ rdw, int n, int t, int nc, double* __restrict__ dld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d3 = dW3[i];
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < t; ++c) {
    const size_t o = ((size_t)(c / nc) * n + i) * nc + (c % nc);
    s1 += U[o] * d3 * PIZ[o];
    s2 += BPIZ[o] * d3 * BPIZ[o];
  }
  const double tr1 = s1 / t, trP = s2 / t;
  double cv = 0.0, vr = 0.0;
  for (int c = 0; c < t; ++c) {
    const size_t o = ((size_t)(c / nc) * n + i) * nc + (c % nc);
    const double a1 = U[o] * d3 * PIZ[o] - tr1;
    const double b1 = BPIZ[o] * d3 * BPIZ[o] - trP;
    cv += a1 * b1; vr += b1 * b1;
  }
  cv /= t; vr /= t;
  const double copt = (vr == 0.0) ? 1.0 : cv / vr;
  dld[i] = tr1 + copt * (rdw[i] * d3) - copt * trP;
}

// per column c of a block: out[c] = X(:, c) . T(:, c), out[ncols + c] = Y(:, c) . T(:, c); one workgroup per chunk, fixed order
template <int NC>
__global__ __launch_bounds__(1024) void lap_coldots_kernel(const double* __restrict__ X, const double* __restrict__ Y, const double* __restrict__ T, int n,
                                                           int ncols, double* __restrict__ out) {
  __shared__ double s[2048];
  const size_t off = (size_t)blockIdx.x * n * NC;
  double ax[NC], ay[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { ax[c] = 0.0; ay[c] = 0.0; }
  for (int i = threadIdx.x; i < n; i += 1024) {
    const VecN<NC> x = ldvec<NC>(X + off, i), y = ldvec<NC>(Y + off, i), tt = ldvec<NC>(T + off, i);
#pragma unroll
    for (int c = 0; c < NC; ++c) { ax[c] = __builtin_fma(x.v[c], tt.v[c], ax[c]); ay[c] = __builtin_fma(y.v[c], tt.v[c], ay[c]); }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    block_reduce2(ax[c], ay[c], s);
    if (threadIdx.x == 0) { out[blockIdx.x * NC + c] = ax[c]; out[ncols + blockIdx.x * NC + c] = ay[c]; }
  }
}

// the diagonal scalings between the products of SigmaI_deriv / P_deriv (likelihoods.h:6613-6640, :16760-16775), R = B v, Z = (dA) v:
//   sel 0: H = -R / D                                     (variance: SigmaI_deriv = -Sigma^-1)
//   sel 1: H = -Z / D - R dD / D^2,  V = R / D            (range:  B^T H - (dA)^T V = SigmaI_deriv v)
//   sel 2: H = -W Z,                 V = W R              (range:  B^T H - (dA)^T V = the extra part of P_deriv v)
__global__ void lap_deriv_mid_kernel(const double* __restrict__ R, const double* __restrict__ Z, const double* __restrict__ D, const double* __restrict__ dD,
                                     const double* __restrict__ W, int n, int nc, int sel, double* __restrict__ H, double* __restrict__ V) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const size_t o = (size_t)blockIdx.y * n * nc + g;
  const int i = (int)(g / nc);
  if (sel == 0) H[o] = -R[o] / D[i];
  else if (sel == 1) { H[o] = -Z[o] / D[i] - R[o] * dD[i] / (D[i] * D[i]); V[o] = R[o] / D[i]; }
  else { H[o] = -W[i] * Z[o]; V[o] = W[i] * R[o]; }
}
// out3 = { sum rdw / D, sum rdw dD / D^2, sum dD / D }  (the deterministic traces of the control variates and of d log|Sigma|)
__global__ __launch_bounds__(1024) void lap_sums3_kernel(const double* __restrict__ rdw, const double* __restrict__ D, const double* __restrict__ dD, int n, double* __restrict__ out3) {
  __shared__ double s[2048];
  double a = 0.0, b = 0.0, c = 0.0, z = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { a += rdw[i] / D[i]; b += rdw[i] * dD[i] / (D[i] * D[i]); c += dD[i] / D[i]; }
  block_reduce2(a, b, s);
  block_reduce2(c, z, s);
  if (threadIdx.x == 0) { out3[0] = a; out3[1] = b; out3[2] = c; }
}

// ---- launchers --------------------------------------------------------------------------------------------
#define GRID1(n) dim3(((n) + 255) / 256), dim3(256)
hipError_t lap_newton_setup(int link, const double* mode, const LikResp& y, const double* fe, const double* D, int n, double* W, double* rhs, double* dw, double* rdw, hipStream_t st,
                            const int* dptr) {
  switch (link) {
    case 0: hipLaunchKernelGGL(lik_newton_setup_kernel<0>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 1: hipLaunchKernelGGL(lik_newton_setup_kernel<1>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 2: hipLaunchKernelGGL(lik_newton_setup_kernel<2>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 3: hipLaunchKernelGGL(lik_newton_setup_kernel<3>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 4: hipLaunchKernelGGL(lik_newton_setup_kernel<4>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 5: hipLaunchKernelGGL(lik_newton_setup_kernel<5>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 6: hipLaunchKernelGGL(lik_newton_setup_kernel<6>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 7: hipLaunchKernelGGL(lik_newton_setup_kernel<7>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    case 8: hipLaunchKernelGGL(lik_newton_setup_kernel<8>, GRID1(n), 0, st, mode, y, fe, D, n, W, rhs, dw, rdw, dptr); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// nc = columns per chunk of the block layout (1: plain column-major; 4: the probe block), ncol = number of chunks
#define LAP_SPMV(MODE, T_, X, D_, W_, H_, OUT)                                                                                              \
  do {                                                                                                                                      \
    const dim3 grid_((T_.nslots + 15) / 16, ncol);                                                                                          \
    if (nc == 4) hipLaunchKernelGGL((lap_tri_spmv_kernel<MODE, 4>), grid_, dim3(256), 0, st, T_, n, X, D_, W_, H_, OUT);                    \
    else hipLaunchKernelGGL((lap_tri_spmv_kernel<MODE, 1>), grid_, dim3(256), 0, st, T_, n, X, D_, W_, H_, OUT);                            \
  } while (0)
hipError_t lap_apply(const LapLevels& lv, int n, const double* D, const double* W, const double* h, double* v, double* tmp, int ncol, int nc, hipStream_t st) {
  LAP_SPMV(1, lv.fwd, h, D, (const double*)nullptr, (const double*)nullptr, tmp);
  LAP_SPMV(2, lv.bwd, (const double*)tmp, (const double*)nullptr, W, h, v);
  return hipGetLastError();
}
hipError_t lap_B(const LapLevels& lv, int n, const double* x, double* out, int ncol, int nc, hipStream_t st) {
  LAP_SPMV(0, lv.fwd, x, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, out);
  return hipGetLastError();
}
hipError_t lap_Bt(const LapLevels& lv, int n, const double* x, double* out, int ncol, int nc, hipStream_t st) {
  LAP_SPMV(2, lv.bwd, x, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, out);
  return hipGetLastError();
}
hipError_t lap_pred_rhs(const int* nn_p, const double* A_p, const int* sigma, int n, int m, int p0, int cnt, int ncol, int nc, double* out, hipStream_t st) {
  hipLaunchKernelGGL(lap_pred_rhs_kernel, GRID1(ncol * nc * m), 0, st, nn_p, A_p, sigma, n, m, p0, cnt, ncol * nc, nc, out);
  return hipGetLastError();
}
hipError_t lap_pred_quad(const int* nn_p, const double* A_p, const int* sigma, const double* X, int n, int m, int row0, int n_rows, int cnt, int nc,
                         int diag_only, double* out, hipStream_t st) {
  const int total = diag_only ? cnt : n_rows * cnt;
  hipLaunchKernelGGL(lap_pred_quad_kernel, GRID1(total), 0, st, nn_p, A_p, sigma, X, n, m, row0, n_rows, cnt, nc, diag_only, out);
  return hipGetLastError();
}
hipError_t lap_scatter(const double* in, const int* sigma, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(lap_scatter_kernel, GRID1(n), 0, st, in, sigma, n, out);
  return hipGetLastError();
}
hipError_t lap_objective(int link, const double* x, const LikResp& y, const double* fe, const double* Bx, const double* D, int n, double* out2, hipStream_t st,
                         const int* dptr) {
  switch (link) {
    case 0: hipLaunchKernelGGL(lik_objective_kernel<0>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 1: hipLaunchKernelGGL(lik_objective_kernel<1>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 2: hipLaunchKernelGGL(lik_objective_kernel<2>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 3: hipLaunchKernelGGL(lik_objective_kernel<3>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 4: hipLaunchKernelGGL(lik_objective_kernel<4>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 5: hipLaunchKernelGGL(lik_objective_kernel<5>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 6: hipLaunchKernelGGL(lik_objective_kernel<6>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 7: hipLaunchKernelGGL(lik_objective_kernel<7>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    case 8: hipLaunchKernelGGL(lik_objective_kernel<8>, dim3(1), dim3(1024), 0, st, x, y, fe, Bx, D, n, out2, dptr); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// ---- the same solve WITHOUT level barriers: one launch for all levels [L0, L1) -------------------------------------------------------
// The level-per-launch schedule above pays one kernel boundary (5 - 8 us) per dependency level, ~240 times per CG iteration at n = 1e5
// (profiles/r02_p_*: 116 310 launches per evaluation).  Here the solution vector itself carries the dependencies: every entry of x in
// the launch's row range is pre-set to a sentinel (kLapEmpty, a NaN pattern no arithmetic produces; lap_sf_prefill_kernel), a row's
// 16 lanes gather their sources with L1-bypassing (sc1) loads and simply re-read a source while it still holds the sentinel, and the
// row's value is published with ONE write-through (sc1) 8-byte store per column -- the "data is the flag" granule hand-off of
// cdna_hip_programming.md Guideline 16 (per-XCD L2s are not coherent; static matrix entries keep their plain, cached loads, which the
// grid-barrier variant tried in round 2 lost to its agent-scope acquires).  A dependency level then costs one visibility round trip,
// not a launch.
// Forward progress: the grid is at most one RESIDENT round of workgroups (host: occupancy x CUs); 16-lane group g of the G groups of a
// column unit takes the slots qa + g, qa + g + G, ... in increasing order.  Slots are stored in level order, so every source of a slot
// has a SMALLER position: it is either final, or owned by a resident group that reaches it before anything that could wait for this
// one -- the smallest unfinished slot never waits.  A spin is bounded all the same (kLapSpinLimit re-reads, then the error word is set
// and the row is given up): a mistake must not hang the device.
// Arithmetic: per slot the same fma order and the same 16-lane butterfly as lap_sptrsv_kernel, slot sums of a split row added in
// slot order -- results are bit-identical to the level-scheduled solve.
constexpr unsigned long long kLapEmpty = 0xFFF8DEAD0000BEEFull;
constexpr int kLapSpinLimit = 1 << 20;
constexpr int kSfThreads = 256;

template <int NC, int LS>
__global__ void lap_sf_prefill_kernel(LapTri T, int qa, int qb, int n, double* x) {
  const int q = qa + blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= qb) return;
  const int4 m = T.meta[q];
  if (m.y <= 0) return;                                  // continuation / padding slot
  const size_t off = (size_t)(blockIdx.y * NC / LS) * n * LS + (size_t)(blockIdx.y * NC % LS);
  unsigned long long* xr = reinterpret_cast<unsigned long long*>(x + off + (size_t)(unsigned)m.x * LS);
#pragma unroll
  for (int c = 0; c < NC; ++c) xr[c] = kLapEmpty;
}

// one source entry, without waiting: ok = false when any of its NC values is still the sentinel
template <int NC, int LS>
__device__ __forceinline__ VecN<NC> lap_sf_peek(const double* xc, unsigned src, bool& ok) {
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(xc) + src * (unsigned)(8 * LS));
  VecN<NC> o;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const unsigned long long raw = __hip_atomic_load(p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = ok && raw != kLapEmpty;
    o.v[c] = __longlong_as_double((long long)raw);
  }
  return o;
}

template <bool SCALE, bool OVF, int NC, int LS>
__global__ __launch_bounds__(kSfThreads) void lap_sptrsv_sf_kernel(LapTri T, int n, int qa, int qb, const double* __restrict__ rhs,
                                                                  const double* __restrict__ rdw, double* x, int* err) {
  const int col = blockIdx.y;
  const size_t off = (size_t)(col * NC / LS) * n * LS + (size_t)(col * NC % LS);
  const double* __restrict__ rc = rhs + off;
  double* xc = x + off;
  const int lane = threadIdx.x & 15;
  const int gsh = 16 * ((threadIdx.x >> 4) & 3);         // this 16-lane group's bits in the wavefront's ballot
  const int G = gridDim.x * (kSfThreads / 16);
  // the descriptor and the head entries of a group's NEXT slot are fetched while it works on the current one: per slot one dependent
  // round trip (the gathers) instead of three (descriptor -> entries -> gathers)
  int q = qa + blockIdx.x * (kSfThreads / 16) + (threadIdx.x >> 4);
  int4 m_nx = T.meta[q < qb ? q : qa];
  LapEnt h_nx[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) h_nx[k] = T.hent[(size_t)(q < qb ? q : qa) * 32 + lane + 16 * k];
  for (; q < qb; q += G) {
    const int4 m0 = m_nx;
    LapEnt h0[2] = {h_nx[0], h_nx[1]};
    {
      const int qn = q + G < qb ? q + G : q;
      m_nx = T.meta[qn];
#pragma unroll
      for (int k = 0; k < 2; ++k) h_nx[k] = T.hent[(size_t)qn * 32 + lane + 16 * k];
    }
    if (m0.y <= 0) continue;                             // continuation / padding slot: its row's first group walks it
    const unsigned row = (unsigned)m0.x;
    // The four groups of a wavefront run in lockstep: a group must never WAIT inside a loop the others cannot leave (one of them may
    // own a source of this row -- the last slot of a level and the first of the next sit side by side).  So a pass only PEEKS at the
    // sources; the groups whose sources are all there finish and publish in that pass, the others take another pass.
    // (round 5) the row's right-hand side and scale are fetched BEFORE the wait for the sources, not after it: one dependent memory round trip
    // less on the critical path of every dependency level (the probe-block kernel below has done so since round 4)
    VecN<NC> num_pre;
    double den_pre = 1.0;
    if (lane == 0) {
      num_pre = ldvec<NC, LS>(rc, row);
      if (SCALE) den_pre = rdw[row];
    }
    bool done = false;
    int passes = 0;
    while (!done) {
      bool ok = true;
      double v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] = 0.0;
      for (int sl = 0; sl < m0.y; ++sl) {
        const int qq = q + sl;
        int ob = m0.z, oe = m0.w;
        if (sl > 0) { const int4 ms = T.meta[qq]; ob = ms.z; oe = ms.w; }
        double sum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) sum[c] = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const LapEnt h = sl == 0 ? h0[k] : T.hent[(size_t)qq * 32 + lane + 16 * k];
          if (h.val != 0.0) {                            // (padding: coefficient 0 -- not a dependency)
            const VecN<NC> g = lap_sf_peek<NC, LS>(xc, (unsigned)h.src, ok);
#pragma unroll
            for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(h.val, g.v[c], sum[c]);
          }
          if (OVF) {
            const int e = ob + lane + 16 * k;
            if (e < oe) {
              const LapEnt o = T.oent[e];
              if (o.val != 0.0) {
                const VecN<NC> g = lap_sf_peek<NC, LS>(xc, (unsigned)o.src, ok);
#pragma unroll
                for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(o.val, g.v[c], sum[c]);
              }
            }
          }
        }
        if (OVF) for (int e0 = ob + 32 + lane; e0 - lane < oe; e0 += 64) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = e0 + 16 * k;
            if (e < oe) {
              const LapEnt o = T.oent[e];
              if (o.val != 0.0) {
                const VecN<NC> g = lap_sf_peek<NC, LS>(xc, (unsigned)o.src, ok);
#pragma unroll
                for (int c = 0; c < NC; ++c) sum[c] = __builtin_fma(o.val, g.v[c], sum[c]);
              }
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] += row16_sum(sum[c]);          // (first slot: 0 + its sum, exact)
      }
      const unsigned long long bal = __ballot(ok);
      const bool ready = ((bal >> gsh) & 0xFFFFull) == 0xFFFFull;        // all 16 lanes of this group saw all their sources
      if (ready) {
        if (lane == 0) {
          const VecN<NC>& num = num_pre;
          const double den = den_pre;
          unsigned long long* xr = reinterpret_cast<unsigned long long*>(xc + (size_t)row * LS);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const double out = SCALE ? __builtin_fma(num.v[c], den, v[c]) : num.v[c] + v[c];
            __hip_atomic_store(xr + c, (unsigned long long)__double_as_longlong(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        done = true;
      } else if (++passes > kLapSpinLimit) {
        if (lane == 0) *err = 1;                         // give the row up: never hang the device
        done = true;
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
}

// ---- barrier-free solve of the PROBE BLOCK (round 4): one WAVEFRONT per row, all column chunks -------------------------------------------
// lap_sptrsv_sf_kernel gives every (row, chunk of 4 columns) pair its own 16-lane group: at n = 1e5 the 13 chunks of the 50 probe vectors are
// ~9600 groups per dependency level, each polling its 32 sources with 128 8-byte loads per pass -- measured SLOWER than one launch per level
// (profiles/r03_c_*: 442 vs 362 ms per log-determinant), the pollers crowd out the stores they wait for.  Here a row is owned by ONE wavefront
// for all its chunks: lane = (cg, e), e = lane % 16 the entry slot (entries e and e + 16 of the slot's 32-wide head, as in every other solve
// kernel), cg = lane / 16 takes the chunks cg, cg + 4, cg + 8, cg + 12.  The slot descriptor and the matrix entries are loaded ONCE per row
// (not once per chunk: 13x less index / coefficient traffic), the readiness of a source row is discovered once for all its chunks, and a
// level has ~740 pollers, not 9600.  Same arithmetic per (row, column) as lap_sptrsv_kernel / lap_sptrsv_sf_kernel -- per slot the fma chain over
// the lane's entries in the same order, the same 16-lane butterfly, split rows added slot by slot: results are bit-identical to the
// level-scheduled solve.  Forward progress as in lap_sptrsv_sf_kernel: the grid is resident (host: occupancy x CUs), a wavefront takes its slots
// in increasing (= level) order, a wavefront handles ONE row at a time (no lane ever waits for another lane of its own wavefront); spins are
// bounded (kLapSpinLimit) and a give-up sets the error word instead of hanging the device.
constexpr int kSfwThreads = 256;

// v1 of this kernel gathered with 8-byte agent-scope atomic loads (as lap_sptrsv_sf_kernel does): four instructions and four 32-byte sector
// requests per 32-byte row of a chunk -- 1.8e8 sector requests per solve at n = 1e5, request-rate bound (profiles/r04_a_*: 3.3 / 2.0 ms per solve
// against 1.1 / 0.9 ms of level launches).  v2: a row of a chunk is gathered as TWO L1-bypassing 16-byte loads (global_load_dwordx4 sc1) and
// published as two 16-byte write-through stores; the sentinel is still checked per 8-byte value (16-byte sc1 halves are observed untorn on
// gfx950, MI355X_MICROARCH.md -- and a torn half would only be seen as "not there yet").  The loads of a batch (two entries x J chunks of a
// lane) and their s_waitcnt are ONE asm block: the compiler never sees a register that a load in flight is still going to write.
// v3: J = chunks per 16-lane group is a template parameter.  With J = 4 one wavefront owns all (<= 16) chunks of a row and the per-level
// chain is gather (16 loads per lane) -> 16 butterflies -> 16 stores: ~7 us per dependency level measured (profiles/r04_b_*), 118 levels deep.
// With J = 1 a wavefront owns 4 chunks of a row, the block is solved as ceil(chunks / 4) independent column units (blockIdx.y) whose chains
// are as short as a single vector's; J = 2 in between.
typedef double lap_v2d __attribute__((ext_vector_type(2)));
// NL rows of 32 bytes (one row of one chunk each) -> d[2 r], d[2 r + 1] = the two halves of row r; loads + wait in one asm block
template <int NL> struct SfwRows { lap_v2d d[2 * NL]; };
__device__ __forceinline__ void lap_sfw_gather(const void* const (&p)[2], SfwRows<2>& b) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %5, off sc1\n\t"
      "global_load_dwordx4 %3, %5, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(b.d[0]), "=&v"(b.d[1]), "=&v"(b.d[2]), "=&v"(b.d[3])
      : "v"(p[0]), "v"(p[1])
      : "memory");
}
__device__ __forceinline__ void lap_sfw_gather(const void* const (&p)[4], SfwRows<4>& b) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %9, off sc1\n\t"
      "global_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
      "global_load_dwordx4 %4, %10, off sc1\n\t"
      "global_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
      "global_load_dwordx4 %6, %11, off sc1\n\t"
      "global_load_dwordx4 %7, %11, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(b.d[0]), "=&v"(b.d[1]), "=&v"(b.d[2]), "=&v"(b.d[3]), "=&v"(b.d[4]), "=&v"(b.d[5]), "=&v"(b.d[6]), "=&v"(b.d[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
      : "memory");
}
__device__ __forceinline__ void lap_sfw_gather(const void* const (&p)[8], SfwRows<8>& b) {
  asm volatile(
      "global_load_dwordx4 %0, %16, off sc1\n\t"
      "global_load_dwordx4 %1, %16, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %17, off sc1\n\t"
      "global_load_dwordx4 %3, %17, off offset:16 sc1\n\t"
      "global_load_dwordx4 %4, %18, off sc1\n\t"
      "global_load_dwordx4 %5, %18, off offset:16 sc1\n\t"
      "global_load_dwordx4 %6, %19, off sc1\n\t"
      "global_load_dwordx4 %7, %19, off offset:16 sc1\n\t"
      "global_load_dwordx4 %8, %20, off sc1\n\t"
      "global_load_dwordx4 %9, %20, off offset:16 sc1\n\t"
      "global_load_dwordx4 %10, %21, off sc1\n\t"
      "global_load_dwordx4 %11, %21, off offset:16 sc1\n\t"
      "global_load_dwordx4 %12, %22, off sc1\n\t"
      "global_load_dwordx4 %13, %22, off offset:16 sc1\n\t"
      "global_load_dwordx4 %14, %23, off sc1\n\t"
      "global_load_dwordx4 %15, %23, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(b.d[0]), "=&v"(b.d[1]), "=&v"(b.d[2]), "=&v"(b.d[3]), "=&v"(b.d[4]), "=&v"(b.d[5]), "=&v"(b.d[6]), "=&v"(b.d[7]),
        "=&v"(b.d[8]), "=&v"(b.d[9]), "=&v"(b.d[10]), "=&v"(b.d[11]), "=&v"(b.d[12]), "=&v"(b.d[13]), "=&v"(b.d[14]), "=&v"(b.d[15])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ __forceinline__ void lap_sfw_store32(void* p, lap_v2d lo, lap_v2d hi) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" : : "v"(p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ bool lap_sfw_present(double v) { return (unsigned long long)__double_as_longlong(v) != kLapEmpty; }

// one wavefront per (row, column unit): the unit (blockIdx.y) covers the chunks [4 J unit, 4 J (unit + 1)) of the block; lane = (cg, e) takes
// the entries e, e + 16 of the slot for the unit's chunks cg, cg + 4, .., cg + 4 (J - 1)
template <bool SCALE, bool OVF, int J>
__global__ __launch_bounds__(kSfwThreads) void lap_sptrsv_sfw_kernel(LapTri T, int n, int qa, int qb, int ncol, const double* __restrict__ rhs,
                                                                   const double* __restrict__ rdw, double* x, int* err) {
  const int lane = threadIdx.x & 63, e = lane & 15, cg = lane >> 4;
  const int NW = gridDim.x * (kSfwThreads / 64);
  const size_t cstride = (size_t)n * 4;                  // doubles per chunk of the [chunk][row][4] layout
  const int ch0 = (int)blockIdx.y * 4 * J;               // first chunk of this column unit
  // byte address of row 0 of this lane's j-th chunk (a chunk beyond ncol: the unit's first chunk, never consumed)
  const char* xb[J];
  bool chunk_on[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int ch = ch0 + cg + 4 * j;
    chunk_on[j] = ch < ncol;
    xb[j] = reinterpret_cast<const char*>(x + (size_t)(chunk_on[j] ? ch : ch0) * cstride);
  }
  int q = qa + blockIdx.x * (kSfwThreads / 64) + (threadIdx.x >> 6);
  int4 m_nx = T.meta[q < qb ? q : qa];
  LapEnt h_nx[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) h_nx[k] = T.hent[(size_t)(q < qb ? q : qa) * 32 + e + 16 * k];
  for (; q < qb; q += NW) {
    const int4 m0 = m_nx;
    LapEnt h0[2] = {h_nx[0], h_nx[1]};
    {
      const int qn = q + NW < qb ? q + NW : q;
      m_nx = T.meta[qn];
#pragma unroll
      for (int k = 0; k < 2; ++k) h_nx[k] = T.hent[(size_t)qn * 32 + e + 16 * k];
    }
    if (m0.y <= 0) continue;                             // continuation / padding slot (uniform over the wavefront)
    const unsigned row = (unsigned)m0.x;
    // the row's right-hand side (and scale) travel with the first gathers instead of after the last one: one dependent round trip less per row
    VecN<4> num[J];
#pragma unroll
    for (int j = 0; j < J; ++j) num[j] = ldvec<4, 4>(rhs + (size_t)(chunk_on[j] ? ch0 + cg + 4 * j : ch0) * cstride, row);
    const double den = SCALE ? rdw[row] : 1.0;
    double v[J][4];
    int passes = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[j][c] = 0.0;
      for (int sl = 0; sl < m0.y; ++sl) {
        const int qq = q + sl;
        int ob = m0.z, oe = m0.w;
        if (sl > 0) { const int4 ms = T.meta[qq]; ob = ms.z; oe = ms.w; }
        double sum[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) sum[j][c] = 0.0;
        // NE entries of this lane in ONE round trip (NE x J rows of 32 bytes); an entry with coefficient 0 is padding (not a dependency): its gather
        // goes to row 0 and is ignored.  Accumulation order per (chunk, column): the entries in the order given -- callers pass them in the order
        // of lap_sptrsv_kernel's fma chain.
        auto take = [&](auto ne_, const LapEnt* en) {
          constexpr int NE = decltype(ne_)::value;
          bool used[NE], any = false;
#pragma unroll
          for (int q = 0; q < NE; ++q) { used[q] = en[q].val != 0.0; any = any || used[q]; }
          if (!__any(any)) return;                       // (uniform) nothing to gather for the whole wavefront
          const void* p[NE * J];
#pragma unroll
          for (int q = 0; q < NE; ++q) {
            const unsigned off = used[q] ? (unsigned)en[q].src * 32u : 0u;
#pragma unroll
            for (int j = 0; j < J; ++j) p[q * J + j] = xb[j] + off;
          }
          SfwRows<NE * J> b;
          lap_sfw_gather(p, b);
#pragma unroll
          for (int q = 0; q < NE; ++q) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
              if (chunk_on[j] && used[q]) {
                const int r = q * J + j;
                const double g0 = b.d[2 * r][0], g1 = b.d[2 * r][1], g2 = b.d[2 * r + 1][0], g3 = b.d[2 * r + 1][1];
                ok = ok && lap_sfw_present(g0) && lap_sfw_present(g1) && lap_sfw_present(g2) && lap_sfw_present(g3);
                sum[j][0] = __builtin_fma(en[q].val, g0, sum[j][0]); sum[j][1] = __builtin_fma(en[q].val, g1, sum[j][1]);
                sum[j][2] = __builtin_fma(en[q].val, g2, sum[j][2]); sum[j][3] = __builtin_fma(en[q].val, g3, sum[j][3]);
              }
            }
          }
        };
        const LapEnt none = {0.0, 0, 0};
        // fma chain of lap_sptrsv_kernel / lap_sptrsv_sf_kernel per lane: head k = 0, overflow k = 0, head k = 1, overflow k = 1, then the rest of the overflow
        const LapEnt hA = sl == 0 ? h0[0] : T.hent[(size_t)qq * 32 + e];
        const LapEnt hB = sl == 0 ? h0[1] : T.hent[(size_t)qq * 32 + e + 16];
        if (OVF) {
          const int eo0 = ob + e, eo1 = ob + e + 16;
          const LapEnt first[4] = {hA, eo0 < oe ? T.oent[eo0] : none, hB, eo1 < oe ? T.oent[eo1] : none};
          take(IntC<4>{}, first);                          // the head and the first 32 overflow entries of the slot: one round trip (v3: two)
          for (int e0 = ob + 32 + e; e0 - e < oe; e0 += 64) {
            const LapEnt more[4] = {e0 < oe ? T.oent[e0] : none, e0 + 16 < oe ? T.oent[e0 + 16] : none,
                                    e0 + 32 < oe ? T.oent[e0 + 32] : none, e0 + 48 < oe ? T.oent[e0 + 48] : none};
            take(IntC<4>{}, more);
          }
        } else {
          const LapEnt two[2] = {hA, hB};
          take(IntC<2>{}, two);
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[j][c] += row16_sum(sum[j][c]);       // (first slot: 0 + its sum, exact)
      }
      if (__all(ok)) break;                              // every source of every chunk of this row has arrived
      if (++passes > kLapSpinLimit) { if (lane == 0) *err = 1; break; }       // give the row up: never hang the device
      __builtin_amdgcn_s_sleep(2);
    }
    if (e == 0) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int ch = ch0 + cg + 4 * j;
        if (ch < ncol) {
          double o[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = SCALE ? __builtin_fma(num[j].v[c], den, v[j][c]) : num[j].v[c] + v[j][c];
          const lap_v2d lo = {o[0], o[1]}, hi = {o[2], o[3]};
          lap_sfw_store32(x + (size_t)ch * cstride + (size_t)row * 4, lo, hi);
        }
      }
    }
  }
}

// OVF = false: no slot has more than 32 entries (B with m <= 32 neighbours): the overflow loads and gathers are compiled out
#define LAP_TRSV_NC(SCALE, OVF_, NC_, LS_, T_, SEG, RHS, RDW, X)                                                                     \
  hipLaunchKernelGGL((lap_sptrsv_kernel<SCALE, OVF_, NC_, LS_>), dim3(ncol * (LS_ / NC_) * (SEG).nsplit), dim3(kTriThreads), 0, st, T_, \
                     (T_).ptr, (T_).lsplit, n, (SEG).L0, (SEG).L1, (SEG).nsplit, (SEG).nrounds, RHS, RDW, X)
// block layout (nc == 4): a wide level (own launch, work-bound) handles a whole chunk per workgroup -- the 4 columns share every
// index / coefficient load and each 32-byte sector of x; a run of narrow levels (latency-bound) keeps one workgroup per column
#define LAP_TRSV_O(SCALE, OVF_, T_, SEG, RHS, RDW, X)                                                                                 \
  do {                                                                                                                                \
    if (nc == 4) { if ((SEG).nsplit > 1) LAP_TRSV_NC(SCALE, OVF_, 4, 4, T_, SEG, RHS, RDW, X); else LAP_TRSV_NC(SCALE, OVF_, 1, 4, T_, SEG, RHS, RDW, X); } \
    else LAP_TRSV_NC(SCALE, OVF_, 1, 1, T_, SEG, RHS, RDW, X);                                                                        \
  } while (0)
#define LAP_TRSV(SCALE, T_, SEG, RHS, RDW, X)                                                                                         \
  do { if ((T_).has_ovf) LAP_TRSV_O(SCALE, true, T_, SEG, RHS, RDW, X); else LAP_TRSV_O(SCALE, false, T_, SEG, RHS, RDW, X); } while (0)
hipError_t lap_dense_build(const LapDense& d, const double* A, hipStream_t st) {
  for (int l = 0; l < d.nblev; ++l) {
    const int k0 = d.blev[l], k1 = d.blev[l + 1];
    if (k1 > k0) hipLaunchKernelGGL(lap_dense_inv_kernel, dim3((k1 + 255) / 256, k1 - k0), dim3(256), 0, st, d, A, k0);
  }
  return hipGetLastError();
}
// the block of a solve: right-hand sides (with the part of the product that comes from outside the block), then the dense product;
// kDenseCols columns per pass
template <bool SCALE>
static void lap_dense_solve(const LapDense& d, const double* A, int n, const double* rhs, const double* rdw, double* x, int ncol, int nc, hipStream_t st) {
  if (d.K <= 0) return;
  const int per = nc == 4 ? 16 : kDenseCols;                     // chunks per pass (16 x 4 columns, or 64 plain columns)
  for (int c0 = 0; c0 < ncol; c0 += per) {
    const int cn = ncol - c0 < per ? ncol - c0 : per;
    if (nc == 4) {
      hipLaunchKernelGGL((lap_dense_rhs_kernel<4, SCALE>), dim3((d.K + 3) / 4, cn), dim3(256), 0, st, d, A, n, c0, cn, rhs, rdw, (const double*)x);
      double* part = d.tbuf + (size_t)d.K * kDenseCols;
      hipLaunchKernelGGL(lap_dense_gemm4_kernel, dim3((d.K + 31) / 32, kDenseSplit), dim3(256), 0, st, d, cn, part);
      hipLaunchKernelGGL(lap_dense_gemm4_sum_kernel, dim3((d.K * cn + 255) / 256), dim3(256), 0, st, d, n, c0, cn, (const double*)part, x);
    } else {
      hipLaunchKernelGGL((lap_dense_rhs_kernel<1, SCALE>), dim3((d.K + 3) / 4, cn), dim3(256), 0, st, d, A, n, c0, cn, rhs, rdw, (const double*)x);
      hipLaunchKernelGGL(lap_dense_matvec_kernel, dim3((d.K + 3) / 4, cn), dim3(256), 0, st, d, n, c0, cn, x);
    }
  }
}
// one launch (+ the sentinel prefill) for all levels of the segments [seg[0].L0, seg[nseg-1].L1); ncol column units (nc == 4: chunks)
template <bool SCALE>
static hipError_t lap_trsv_syncfree(const LapTri& T, const int* host_ptr, const LapSeg* seg, int nseg, int n, const double* rhs, const double* rdw, double* x,
                                    int ncol, int nc, int* err, hipStream_t st) {
  if (nseg <= 0) return hipSuccess;
  const int qa = host_ptr[seg[0].L0], qb = host_ptr[seg[nseg - 1].L1];
  if (qb <= qa) return hipSuccess;
  static int resident = 0;           // workgroups of 256 lanes the device holds at once (conservative: half of what the occupancy calculator admits)
  if (resident == 0) {
    int occ = 0, cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lap_sptrsv_sf_kernel<true, true, 1, 1>, kSfThreads, 0) != hipSuccess || occ < 1) occ = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 64;
    resident = std::max(1, occ * cus / 2);
    (void)hipGetLastError();
  }
  const int nslot = qb - qa;
  for (int c0 = 0; c0 < ncol; ) {       // column units per pass: all of them unless there are more than resident workgroups
    const int cn = std::min(ncol - c0, resident);
    const int per = std::max(1, std::min(resident / cn, (nslot + kSfThreads / 16 - 1) / (kSfThreads / 16)));
    const size_t coff = (size_t)c0 * (nc == 4 ? 4 : 1) * n;
    const dim3 pg((nslot + 255) / 256, cn), grid(per, cn);
#define LAP_SF(OVF_, NC_, LS_)                                                                                                          \
    do {                                                                                                                                \
      hipLaunchKernelGGL((lap_sf_prefill_kernel<NC_, LS_>), pg, dim3(256), 0, st, T, qa, qb, n, x + coff);                              \
      hipLaunchKernelGGL((lap_sptrsv_sf_kernel<SCALE, OVF_, NC_, LS_>), grid, dim3(kSfThreads), 0, st, T, n, qa, qb, rhs + coff, rdw, x + coff, err); \
    } while (0)
    if (T.has_ovf) LAP_SF(true, 1, 1); else LAP_SF(false, 1, 1);
#undef LAP_SF
    c0 += cn;
  }
  return hipGetLastError();
}
// probe block (nc == 4), barrier-free, one wavefront per (row, column unit of 4 J chunks): one launch (+ the sentinel prefill) per triangular solve.
// J = chunks per 16-lane group: 1 (J = 2 measured 5 % slower, J = 4 cannot keep its grid resident; profiles/r04_c_laplace_block_solve_v3_sweep.log).
template <bool SCALE, int J>
static hipError_t lap_trsv_syncfree_block_j(const LapTri& T, int n, int qa, int qb, const double* rhs, const double* rdw, double* x, int ncol, int* err,
                                            hipStream_t st) {
  const int nslot = qb - qa;
  const int units = (ncol + 4 * J - 1) / (4 * J);
  int occ = 0, cus = 0, dev = 0;
  (void)hipGetDevice(&dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lap_sptrsv_sfw_kernel<true, true, J>, kSfwThreads, 0) != hipSuccess || occ < 1) occ = 1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 64;
  (void)hipGetLastError();
  const int resident = std::max(1, occ * cus * 3 / 4);   // every workgroup of the launch must be resident (forward progress): stay below what the calculator admits
  int total = 2 * cus;                                   // (measured at config 4, profiles/r04_c_*: J = 1 with 512 workgroups 279 ms, 256: 295, 768 / 1024: 300)
  total = std::min(total, resident);
  const int per_unit = std::max(1, std::min(total / units, (nslot + kSfwThreads / 64 - 1) / (kSfwThreads / 64)));
  hipLaunchKernelGGL((lap_sf_prefill_kernel<4, 4>), dim3((nslot + 255) / 256, ncol), dim3(256), 0, st, T, qa, qb, n, x);
  if (T.has_ovf) hipLaunchKernelGGL((lap_sptrsv_sfw_kernel<SCALE, true, J>), dim3(per_unit, units), dim3(kSfwThreads), 0, st, T, n, qa, qb, ncol, rhs, rdw, x, err);
  else hipLaunchKernelGGL((lap_sptrsv_sfw_kernel<SCALE, false, J>), dim3(per_unit, units), dim3(kSfwThreads), 0, st, T, n, qa, qb, ncol, rhs, rdw, x, err);
  return hipGetLastError();
}
template <bool SCALE>
static hipError_t lap_trsv_syncfree_block(const LapTri& T, const int* host_ptr, const LapSeg* seg, int nseg, int n, const double* rhs, const double* rdw, double* x,
                                          int ncol, int* err, hipStream_t st) {
  if (nseg <= 0) return hipSuccess;
  const int qa = host_ptr[seg[0].L0], qb = host_ptr[seg[nseg - 1].L1];
  if (qb <= qa) return hipSuccess;
  return lap_trsv_syncfree_block_j<SCALE, 1>(T, n, qa, qb, rhs, rdw, x, ncol, err, st);   // J = 1: measured best (profiles/r04_c_laplace_block_solve_v3_sweep.log)
}
hipError_t lap_vadu(const LapLevels& lv, int n, const double* rdw, const double* r, double* z, double* t, int ncol, int nc, hipStream_t st) {
  if (nc == 4 && (lv.syncfree & 4)) {                          // bit 2: the probe block, one wavefront per row over all chunks (round 4)
    (void)lap_trsv_syncfree_block<false>(lv.bwd, lv.bwd_ptr_host, lv.bseg, lv.n_bseg, n, r, nullptr, t, ncol, lv.err, st);        // B^T t = r
    lap_dense_solve<false>(lv.bdense, lv.A, n, r, nullptr, t, ncol, nc, st);
    lap_dense_solve<true>(lv.fdense, lv.A, n, t, rdw, z, ncol, nc, st);
    (void)lap_trsv_syncfree_block<true>(lv.fwd, lv.fwd_ptr_host, lv.fseg, lv.n_fseg, n, t, rdw, z, ncol, lv.err, st);             // (D^-1 + W) B z = t
    return hipGetLastError();
  }
  if (nc == 1 && (lv.syncfree & 1)) {                          // bit 0: single vectors (mode finding), one 16-lane group per row
    (void)lap_trsv_syncfree<false>(lv.bwd, lv.bwd_ptr_host, lv.bseg, lv.n_bseg, n, r, nullptr, t, ncol, nc, lv.err, st); // B^T t = r
    lap_dense_solve<false>(lv.bdense, lv.A, n, r, nullptr, t, ncol, nc, st);
    lap_dense_solve<true>(lv.fdense, lv.A, n, t, rdw, z, ncol, nc, st);
    (void)lap_trsv_syncfree<true>(lv.fwd, lv.fwd_ptr_host, lv.fseg, lv.n_fseg, n, t, rdw, z, ncol, nc, lv.err, st);     // (D^-1 + W) B z = t
    return hipGetLastError();
  }
  for (int k = 0; k < lv.n_bseg; ++k) LAP_TRSV(false, lv.bwd, lv.bseg[k], r, (const double*)nullptr, t);     // B^T t = r
  lap_dense_solve<false>(lv.bdense, lv.A, n, r, nullptr, t, ncol, nc, st);                                    //   ... its last (narrow) levels as one dense block
  lap_dense_solve<true>(lv.fdense, lv.A, n, t, rdw, z, ncol, nc, st);                                         // (D^-1 + W) B z = t: the first (narrow) levels
  for (int k = 0; k < lv.n_fseg; ++k) LAP_TRSV(true, lv.fwd, lv.fseg[k], (const double*)t, rdw, z);
  return hipGetLastError();
}
// z = B^-1 (scale .* rhs): the forward half of lap_vadu on its own (probe vectors of the "vecchia_response" preconditioner, likelihoods.h:16446-16450)
hipError_t lap_fwd_solve(const LapLevels& lv, int n, const double* scale, const double* rhs, double* z, int ncol, int nc, hipStream_t st) {
  lap_dense_solve<true>(lv.fdense, lv.A, n, rhs, scale, z, ncol, nc, st);
  if (nc == 4 && (lv.syncfree & 4)) { (void)lap_trsv_syncfree_block<true>(lv.fwd, lv.fwd_ptr_host, lv.fseg, lv.n_fseg, n, rhs, scale, z, ncol, lv.err, st); return hipGetLastError(); }
  if (nc == 1 && (lv.syncfree & 1)) { (void)lap_trsv_syncfree<true>(lv.fwd, lv.fwd_ptr_host, lv.fseg, lv.n_fseg, n, rhs, scale, z, ncol, nc, lv.err, st); return hipGetLastError(); }
  for (int k = 0; k < lv.n_fseg; ++k) LAP_TRSV(true, lv.fwd, lv.fseg[k], rhs, scale, z);
  return hipGetLastError();
}
hipError_t lap_permute_factor(const double* A, const int* hpos, const int* opos, size_t nh, size_t novf, LapEnt* hent, LapEnt* oent, hipStream_t st) {
  const size_t cnt = nh > novf ? nh : novf;
  hipLaunchKernelGGL(lap_permute_factor_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, A, hpos, opos, nh, novf, hent, oent);
  return hipGetLastError();
}
int lap_cg_parts(int n) { const int p = (n + 4095) / 4096; return p < 1 ? 1 : (p > 64 ? 64 : p); }
hipError_t lap_cg_alpha(const double* r, const double* z, const double* h, const double* v, int n, int ncol, int nc, const CgScalars& sc, hipStream_t st) {
  const dim3 grid(lap_cg_parts(n), ncol);
  if (nc == 4) hipLaunchKernelGGL((cg_dots_kernel<4, true>), grid, dim3(1024), 0, st, r, z, h, v, n, sc);
  else hipLaunchKernelGGL((cg_dots_kernel<1, true>), grid, dim3(1024), 0, st, r, z, h, v, n, sc);
  return hipGetLastError();
}
hipError_t lap_cg_update(double* u, double* r, const double* h, const double* v, int n, int ncol, int nc, const CgScalars& sc, hipStream_t st) {
  const dim3 grid(lap_cg_parts(n), ncol);
  if (nc == 4) hipLaunchKernelGGL(cg_update_kernel<4>, grid, dim3(1024), 0, st, u, r, h, v, n, sc);
  else hipLaunchKernelGGL(cg_update_kernel<1>, grid, dim3(1024), 0, st, u, r, h, v, n, sc);
  hipLaunchKernelGGL(cg_rnorm_kernel, dim3((ncol * nc + 63) / 64), dim3(64), 0, st, sc, ncol * nc, lap_cg_parts(n));
  return hipGetLastError();
}
hipError_t lap_cg_beta(const double* r, const double* z, double* h, int n, int ncol, int nc, const CgScalars& sc, int j, int p_max, hipStream_t st) {
  const dim3 grid(lap_cg_parts(n), ncol);
  if (nc == 4) {
    hipLaunchKernelGGL((cg_dots_kernel<4, false>), grid, dim3(1024), 0, st, r, z, (const double*)nullptr, (const double*)nullptr, n, sc);
    hipLaunchKernelGGL(cg_hupdate_kernel<4>, grid, dim3(1024), 0, st, z, h, n, sc, j, p_max);
  } else {
    hipLaunchKernelGGL((cg_dots_kernel<1, false>), grid, dim3(1024), 0, st, r, z, (const double*)nullptr, (const double*)nullptr, n, sc);
    hipLaunchKernelGGL(cg_hupdate_kernel<1>, grid, dim3(1024), 0, st, z, h, n, sc, j, p_max);
  }
  return hipGetLastError();
}
hipError_t lap_lincomb(double* out, const double* x, const double* y, double cx, double cy, int n, hipStream_t st) {
  hipLaunchKernelGGL(lap_lincomb_kernel, GRID1(n), 0, st, out, x, y, cx, cy, n);
  return hipGetLastError();
}
hipError_t lap_scale_probes(const double* rv, const double* dw, int n, int ncol, int nc, double* out, hipStream_t st) {
  hipLaunchKernelGGL(lap_scale_probes_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, rv, dw, n, nc, out);
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

hipError_t lap_third_deriv(int link, const double* mode, const LikResp& y, const double* fe, int n, double* dW3, hipStream_t st, const int* dptr) {
  switch (link) {
    case 0: hipLaunchKernelGGL(lik_third_kernel<0>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 1: hipLaunchKernelGGL(lik_third_kernel<1>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 2: hipLaunchKernelGGL(lik_third_kernel<2>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 3: hipLaunchKernelGGL(lik_third_kernel<3>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 4: hipLaunchKernelGGL(lik_third_kernel<4>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 5: hipLaunchKernelGGL(lik_third_kernel<5>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 6: hipLaunchKernelGGL(lik_third_kernel<6>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 7: hipLaunchKernelGGL(lik_third_kernel<7>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    case 8: hipLaunchKernelGGL(lik_third_kernel<8>, GRID1(n), 0, st, mode, y, fe, n, dW3, dptr); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t lap_grad_F(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* sv, int n, double* out, hipStream_t st) {
  switch (link) {
    case 0: hipLaunchKernelGGL(lik_grad_F_kernel<0>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 1: hipLaunchKernelGGL(lik_grad_F_kernel<1>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 2: hipLaunchKernelGGL(lik_grad_F_kernel<2>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 3: hipLaunchKernelGGL(lik_grad_F_kernel<3>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 4: hipLaunchKernelGGL(lik_grad_F_kernel<4>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 5: hipLaunchKernelGGL(lik_grad_F_kernel<5>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 6: hipLaunchKernelGGL(lik_grad_F_kernel<6>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 7: hipLaunchKernelGGL(lik_grad_F_kernel<7>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    case 8: hipLaunchKernelGGL(lik_grad_F_kernel<8>, GRID1(n), 0, st, mode, y, fe, dld, sv, n, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t lap_grad_F_map(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* dW3, const double* sv, int n,
                          const int* dptr, double* out, hipStream_t st) {
  switch (link) {
    case 0: hipLaunchKernelGGL(lik_grad_F_map_kernel<0>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 1: hipLaunchKernelGGL(lik_grad_F_map_kernel<1>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 2: hipLaunchKernelGGL(lik_grad_F_map_kernel<2>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 3: hipLaunchKernelGGL(lik_grad_F_map_kernel<3>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 4: hipLaunchKernelGGL(lik_grad_F_map_kernel<4>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 5: hipLaunchKernelGGL(lik_grad_F_map_kernel<5>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 6: hipLaunchKernelGGL(lik_grad_F_map_kernel<6>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 7: hipLaunchKernelGGL(lik_grad_F_map_kernel<7>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    case 8: hipLaunchKernelGGL(lik_grad_F_map_kernel<8>, GRID1(n), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t lap_aux_grad(int link, const double* mode, const LikResp& y, const double* fe, const double* dld, const double* dW3, const double* sv, int n,
                        const int* dptr, double* out3, hipStream_t st) {
  if (link == 3) hipLaunchKernelGGL(lik_aux_grad_kernel<3>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else if (link == 4) hipLaunchKernelGGL(lik_aux_grad_kernel<4>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else if (link == 5) hipLaunchKernelGGL(lik_aux_grad_kernel<5>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else if (link == 6) hipLaunchKernelGGL(lik_aux_grad_kernel<6>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else if (link == 7) hipLaunchKernelGGL(lik_aux_grad_kernel<7>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else if (link == 8) hipLaunchKernelGGL(lik_aux_grad_kernel<8>, dim3(1), dim3(1024), 0, st, mode, y, fe, dld, dW3, sv, n, dptr, out3);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t lap_factor_deriv(const double4* pts, const int* nn, const double* A, int n, int m, int cov, int d3, double var, double a, double diag_nn, double nug,
                            int which, double* dA, double* dD, hipStream_t st) {
  if (m <= 62) {
    constexpr int T = 64;
    hipLaunchKernelGGL(lap_range_deriv_kernel<T>, dim3(n), dim3(T), sizeof(double) * ((T - 2) * (T - 1) + 7 * T), st, pts, nn, A, n, m, cov, d3, var, a, diag_nn, nug, which, dA, dD);
  } else {
    constexpr int T = 128;
    constexpr int lds = (int)sizeof(double) * ((T - 2) * (T - 1) + 7 * T);          // 135,184 B
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lap_range_deriv_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lap_range_deriv_kernel<T>, dim3(n), dim3(T), lds, st, pts, nn, A, n, m, cov, d3, var, a, diag_nn, nug, which, dA, dD);
  }
  return hipGetLastError();
}
hipError_t lap_range_deriv(const double4* pts, const int* nn, const double* A, int n, int m, int cov, int d3, double var, double a, double* dA, double* dD, hipStream_t st) {
  return lap_factor_deriv(pts, nn, A, n, m, cov, d3, var, a, var * (1.0 + 1e-10), 0.0, 0, dA, dD, st);
}
// Fisher information (CalcFisherInformation_Vecchia, re_model_template.h:10137-10230): H = (P + dD o T) / D on a block
__global__ void lap_fisher_mid_kernel(const double* __restrict__ P, const double* __restrict__ T, const double* __restrict__ D, const double* __restrict__ dD,
                                      int n, int nc, double* __restrict__ H) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * nc) return;
  const size_t o = (size_t)blockIdx.y * n * nc + g;
  const int i = (int)(g / nc);
  H[o] = (P[o] + dD[i] * T[o]) / D[i];
}
hipError_t lap_fisher_mid(const double* P, const double* T, const double* D, const double* dD, int n, int ncol, int nc, double* H, hipStream_t st) {
  hipLaunchKernelGGL(lap_fisher_mid_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, P, T, D, dD, n, nc, H);
  return hipGetLastError();
}
hipError_t lap_mul(const LapTri& T, int n, const double* x, double* out, int ncol, int nc, hipStream_t st) {
  LAP_SPMV(3, T, x, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, out);
  return hipGetLastError();
}
hipError_t lap_row_stats(const double* U, const double* PIZ, const double* BPIZ, const double* dW3, const double* rdw, int n, int t, int nc, double* dld, hipStream_t st) {
  hipLaunchKernelGGL(lap_row_stats_kernel, GRID1(n), 0, st, U, PIZ, BPIZ, dW3, rdw, n, t, nc, dld);
  return hipGetLastError();
}
hipError_t lap_coldots(const double* X, const double* Y, const double* T, int n, int ncol, int nc, double* out, hipStream_t st) {
  if (nc == 4) hipLaunchKernelGGL(lap_coldots_kernel<4>, dim3(ncol), dim3(1024), 0, st, X, Y, T, n, ncol * nc, out);
  else hipLaunchKernelGGL(lap_coldots_kernel<1>, dim3(ncol), dim3(1024), 0, st, X, Y, T, n, ncol * nc, out);
  return hipGetLastError();
}
hipError_t lap_deriv_mid(const double* R, const double* Z, const double* D, const double* dD, const double* W, int n, int ncol, int nc, int sel, double* H, double* V, hipStream_t st) {
  hipLaunchKernelGGL(lap_deriv_mid_kernel, dim3((unsigned)(((size_t)n * nc + 255) / 256), ncol), dim3(256), 0, st, R, Z, D, dD, W, n, nc, sel, H, V);
  return hipGetLastError();
}
hipError_t lap_sums3(const double* rdw, const double* D, const double* dD, int n, double* out3, hipStream_t st) {
  hipLaunchKernelGGL(lap_sums3_kernel, dim3(1), dim3(1024), 0, st, rdw, D, dD, n, out3);
  return hipGetLastError();
}

}  // namespace gpb
