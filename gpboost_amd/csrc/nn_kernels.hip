// gpboost_amd/csrc/nn_kernels.hip
//
// Ordered nearest-neighbour search for the Vecchia approximation on gfx950, bit-identical to
// the reference's CPU search (src/GPBoost/Vecchia_utils.cpp:1029-1093,
// find_nearest_neighbors_fast_internal): candidates are visited outward from the query in
// coordinate-sum order, alternating down/up; a direction stops at the first candidate whose
// squared sum-distance exceeds d * (current m-th smallest squared distance); a candidate replaces
// the current worst only if strictly closer, then bubbles up with strict '<'
// (include/GPBoost/utils.h:250-262).  Because ties and the pruning test make the result depend on
// visiting order and on every rounding, each query is scanned sequentially by ONE lane with the
// same fp64 operation order as the x86-64 (non-FMA) reference build: this file is compiled with
// floating-point contraction OFF.
//
// MI355X mapping: queries are assigned to lanes in coordinate-sum order, so the 64 lanes of a
// wavefront walk overlapping windows of the *sorted* record array {x0,x1,x2,sum} (32 B, one sector
// per candidate) that stay L1/L2 resident; the per-lane top-m lists live in LDS, laid out
// [slot][lane] so that the insertion shifts are bank-conflict free.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <math.h>
#include "nn_kernels.h"

namespace gpb {

template <int D>
__device__ __forceinline__ double sq_dist_seq(const double4& c, const double4& q) {
  // (coords(c, :) - coords(i, :)).squaredNorm(): sequential left-to-right sum (Vecchia_utils.cpp:1064)
  const double d0 = c.x - q.x;
  double s = d0 * d0;
  if constexpr (D >= 2) { const double d1 = c.y - q.y; s = s + d1 * d1; }
  if constexpr (D >= 3) { const double d2 = c.z - q.z; s = s + d2 * d2; }
  return s;
}

template <int D>
__global__ __launch_bounds__(64) void vecchia_nn_kernel(NNKernelArgs a) {
  extern __shared__ unsigned char smem[];
  const int m = a.m;
  double* s_sq = reinterpret_cast<double*>(smem);              // [m][64]
  int* s_id = reinterpret_cast<int*>(smem + (size_t)m * 64 * 8);  // [m][64]
  const int lane = threadIdx.x;
  const int qid = blockIdx.x * 64 + lane;
  if (qid >= a.nq) return;
  const int pos = a.qorder ? a.qorder[qid] : a.pos0 + qid;     // position in coordinate-sum order
  const int i = a.sorted_idx[pos];                             // original (Vecchia-order) index of the query
  if (i <= m || i < a.start_at) return;                        // first m+1 points: all predecessors (:788-813); rows below start_at: not asked for
  const double4 q = a.sorted_rec[pos];
  const int n = a.n;
  const int end_search_at = a.end_search_at;                   // :752-754 (n - 2 unless a prediction run restricts the candidates)
  const double dd = (double)D;
  for (int j = 0; j < m; ++j) s_sq[j * 64 + lane] = INFINITY;  // :1041-1043
  double worst = INFINITY;
  bool down = true, up = true;
  int up_i = pos, down_i = pos;
  // one candidate, exactly the reference's step: c = its index, r = its record (loaded only when c is eligible)
  auto visit = [&](int c, bool eligible, const double4& r, bool& dir) {
    if (eligible) {
      const double ds = r.w - q.w;                             // coords_sum[c] - coords_sum[i]
      const double smd = ds * ds;                              // std::pow(.,2)
      if (smd > dd * worst) {
        dir = false;
      } else {
        const double sed = sq_dist_seq<D>(r, q);
        if (sed < worst) {
          int k = m - 1;                                       // replace the worst, bubble up with strict '<'
          while (k > 0 && sed < s_sq[(k - 1) * 64 + lane]) {
            s_sq[k * 64 + lane] = s_sq[(k - 1) * 64 + lane];
            s_id[k * 64 + lane] = s_id[(k - 1) * 64 + lane];
            --k;
          }
          s_sq[k * 64 + lane] = sed;
          s_id[k * 64 + lane] = c;
          worst = s_sq[(m - 1) * 64 + lane];
        }
      }
    }
  };
  // The visiting ORDER is the reference's (down, up, down, up ...; :1049-1092), but the loads run ahead of it: the indices of the next
  // kAhead candidates of both directions are fetched together, then the records of the eligible ones together, and only then are the
  // candidates taken one by one.  (One candidate at a time meant two dependent L2 round trips per visit with 1.5 wavefronts per SIMD to
  // hide them: the kernel was latency-bound, SQ_WAIT_ANY >> busy cycles.)  Candidates fetched past the point where a direction stops
  // are simply not used.
  // (Also tried: unordered slots with an arg-max pass per insertion and one ranking pass at the end instead of the bubble -- bit-identical
  // too, but 8 % (m = 30) to 30 % (m = 40) slower: a new candidate usually lands near the end of the list, the bubble is short.)
  constexpr int kAhead = 4;
  while (up || down) {
    int cD[kAhead], cU[kAhead];
    bool eD[kAhead], eU[kAhead];
    double4 rD[kAhead], rU[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      cD[k] = a.sorted_idx[max(down_i - 1 - k, 0)];
      cU[k] = a.sorted_idx[min(up_i + 1 + k, n - 1)];
    }
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      eD[k] = cD[k] < i && cD[k] <= end_search_at;
      eU[k] = cU[k] < i && cU[k] <= end_search_at;
      rD[k] = q; rU[k] = q;
      if (eD[k]) rD[k] = a.sorted_rec[max(down_i - 1 - k, 0)];
      if (eU[k]) rU[k] = a.sorted_rec[min(up_i + 1 + k, n - 1)];
    }
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      if (down_i == 0) down = false;
      if (up_i == n - 1) up = false;
      if (down) { --down_i; visit(cD[k], eD[k], rD[k], down); }
      if (up) { ++up_i; visit(cU[k], eU[k], rU[k], up); }
    }
  }
  int* out = a.nn + (size_t)i * m;
  bool dup = false;
  for (int j = 0; j < m; ++j) {
    out[j] = s_id[j * 64 + lane];
    if (sqrt(s_sq[j * 64 + lane]) < 1e-10) dup = true;          // EPSILON_NUMBERS, :905-909
  }
  if (dup) atomicOr(a.has_duplicates, 1);
}

// ---- coordinate dimensions 4 .. 10 (generality path): the same search with records {x_0 .. x_{D-1}, sum} of D + 1 doubles ----------
template <int D>
__global__ __launch_bounds__(64) void vecchia_nn_nd_kernel(NNKernelArgs a) {
  extern __shared__ unsigned char smem[];
  const int m = a.m;
  double* s_sq = reinterpret_cast<double*>(smem);              // [m][64]
  int* s_id = reinterpret_cast<int*>(smem + (size_t)m * 64 * 8);  // [m][64]
  const int lane = threadIdx.x;
  const int qid = blockIdx.x * 64 + lane;
  if (qid >= a.nq) return;
  const int pos = a.qorder ? a.qorder[qid] : a.pos0 + qid;
  const int i = a.sorted_idx[pos];
  if (i <= m || i < a.start_at) return;
  constexpr int RS = D + 1;
  double q[RS];
#pragma unroll
  for (int c = 0; c < RS; ++c) q[c] = a.sorted_nd[(size_t)pos * RS + c];
  const int n = a.n;
  const int end_search_at = a.end_search_at;
  const double dd = (double)D;
  for (int j = 0; j < m; ++j) s_sq[j * 64 + lane] = INFINITY;
  double worst = INFINITY;
  bool down = true, up = true;
  int up_i = pos, down_i = pos;
  auto visit = [&](int p, bool& dir) {
    const int c = a.sorted_idx[p];
    if (c < i && c <= end_search_at) {
      const double* r = a.sorted_nd + (size_t)p * RS;
      const double ds = r[D] - q[D];
      const double smd = ds * ds;
      if (smd > dd * worst) {
        dir = false;
      } else {
        const double d0 = r[0] - q[0];
        double sed = d0 * d0;                                   // sequential left-to-right sum (Vecchia_utils.cpp:1064)
#pragma unroll
        for (int t = 1; t < D; ++t) { const double dt = r[t] - q[t]; sed = sed + dt * dt; }
        if (sed < worst) {
          int k = m - 1;
          while (k > 0 && sed < s_sq[(k - 1) * 64 + lane]) {
            s_sq[k * 64 + lane] = s_sq[(k - 1) * 64 + lane];
            s_id[k * 64 + lane] = s_id[(k - 1) * 64 + lane];
            --k;
          }
          s_sq[k * 64 + lane] = sed;
          s_id[k * 64 + lane] = c;
          worst = s_sq[(m - 1) * 64 + lane];
        }
      }
    }
  };
  while (up || down) {
    if (down_i == 0) down = false;
    if (up_i == n - 1) up = false;
    if (down) { --down_i; visit(down_i, down); }
    if (up) { ++up_i; visit(up_i, up); }
  }
  int* out = a.nn + (size_t)i * m;
  bool dup = false;
  for (int j = 0; j < m; ++j) {
    out[j] = s_id[j * 64 + lane];
    if (sqrt(s_sq[j * 64 + lane]) < 1e-10) dup = true;
  }
  if (dup) atomicOr(a.has_duplicates, 1);
}

// rows 0..m: neighbours are all predecessors in index order, -1 padded (:788-813)
__global__ void vecchia_nn_head_kernel(NNKernelArgs a, int d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n, m = a.m;
  const int rows = (n < m + 1) ? n : m + 1;
  if (t >= rows * m) return;
  const int i = t / m, j = t % m;
  if (i < a.start_at) return;
  a.nn[(size_t)i * m + j] = (j < i) ? j : -1;
  if (j < i) {                                                  // duplicate check of :799-811
    double s;
    if (d > 3) {
      s = 0.0;
      for (int c = 0; c < d; ++c) { const double t = a.coords_nd[(size_t)j * d + c] - a.coords_nd[(size_t)i * d + c]; s = s + t * t; }
    } else {
      const double4 p = a.pts[j], q = a.pts[i];
      s = (p.x - q.x) * (p.x - q.x);
      if (d >= 2) s = s + (p.y - q.y) * (p.y - q.y);
      if (d >= 3) s = s + (p.z - q.z) * (p.z - q.z);
    }
    if (sqrt(s) < 1e-10) atomicOr(a.has_duplicates, 1);
  }
}

void nn_query_order(const int* sorted_idx, int pos0, int pos1, int m, int start_at, int* out, int* nq) {
  constexpr int NB = 4 * 32;
  int cnt[NB + 1] = {0};
  auto key = [](int i) -> int { int k = (int)(4.0 * log2((double)i)); return k < 0 ? 0 : (k >= NB ? NB - 1 : k); };
  for (int p = pos0; p < pos1; ++p) { const int i = sorted_idx[p]; if (i > m && i >= start_at) ++cnt[key(i) + 1]; }
  for (int k = 0; k < NB; ++k) cnt[k + 1] += cnt[k];
  *nq = cnt[NB];
  for (int p = pos0; p < pos1; ++p) { const int i = sorted_idx[p]; if (i > m && i >= start_at) out[cnt[key(i)]++] = p; }
}

hipError_t launch_vecchia_nn(int d, const NNKernelArgs& a_in, hipStream_t st) {
  NNKernelArgs a = a_in;
  if (!a.qorder) a.nq = a.pos1 - a.pos0;
  const int m = a.m;
  hipLaunchKernelGGL(vecchia_nn_head_kernel, dim3(((m + 1) * m + 255) / 256), dim3(256), 0, st, a, d);
  if (a.n <= m + 1) return hipGetLastError();
  const size_t shmem = (size_t)m * 64 * 12;
  const int nblocks = ((a.qorder ? a.nq : a.pos1 - a.pos0) + 63) / 64;
  if (nblocks <= 0) return hipGetLastError();
  if (shmem > 64 * 1024 && d <= 3) {      // m > 85: the per-lane top-m lists need more than the default 64 KB of dynamic LDS
    const void* kf = d == 1 ? reinterpret_cast<const void*>(vecchia_nn_kernel<1>) : (d == 2 ? reinterpret_cast<const void*>(vecchia_nn_kernel<2>)
                                                                                            : reinterpret_cast<const void*>(vecchia_nn_kernel<3>));
    const hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
  }
#define GPB_NN_ND(D_)                                                                                                   \
  case D_: {                                                                                                          \
    if (a.sorted_nd == nullptr || a.coords_nd == nullptr) return hipErrorInvalidValue;                               \
    if (shmem > 64 * 1024) {                                                                                          \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vecchia_nn_nd_kernel<D_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      if (e != hipSuccess) return e;                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL(vecchia_nn_nd_kernel<D_>, dim3(nblocks), dim3(64), shmem, st, a);                             \
  } break
  switch (d) {
    case 1: hipLaunchKernelGGL(vecchia_nn_kernel<1>, dim3(nblocks), dim3(64), shmem, st, a); break;
    case 2: hipLaunchKernelGGL(vecchia_nn_kernel<2>, dim3(nblocks), dim3(64), shmem, st, a); break;
    case 3: hipLaunchKernelGGL(vecchia_nn_kernel<3>, dim3(nblocks), dim3(64), shmem, st, a); break;
    GPB_NN_ND(4); GPB_NN_ND(5); GPB_NN_ND(6); GPB_NN_ND(7); GPB_NN_ND(8); GPB_NN_ND(9); GPB_NN_ND(10);
    default: return hipErrorInvalidValue;
  }
#undef GPB_NN_ND
  return hipGetLastError();
}


// ---- k-means assignment step of the inducing-point selection (full-scale Vecchia: kmeans_plusplus -> calculate_means,
// src/GPBoost/GP_utils.cpp:237-280): cluster[r] = the FIRST mean at the smallest Euclidean distance sqrt(sum_c (x[c][r] - mean[j][c])^2),
// evaluated in the reference's operation order (this translation unit is compiled with fp contraction off, like the reference's baseline
// x86-64 build), so the assignments -- and with the host's ordered mean update the inducing points -- equal the reference's.
// x: column-major [d][n]; means: row-major [k][d]; one thread per point, the means (k <= 256, d <= 3) through LDS.
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const double* __restrict__ x, const double* __restrict__ means, int n, int d, int k, int* __restrict__ cl) {
  __shared__ double s_m[256 * 3];
  for (int e = threadIdx.x; e < k * d; e += 256) s_m[e] = means[e];
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double xr[3] = {0.0, 0.0, 0.0};
  for (int c = 0; c < d; ++c) xr[c] = x[(size_t)c * n + r];
  int best = 0;
  double bd = 0.0;
  for (int j = 0; j < k; ++j) {
    double s2 = 0.0;
    for (int c = 0; c < d; ++c) { const double t = xr[c] - s_m[j * d + c]; s2 += t * t; }
    const double dd = sqrt(s2);
    if (j == 0 || dd < bd) { bd = dd; best = j; }
  }
  cl[r] = best;
}
hipError_t launch_kmeans_assign(const double* x, const double* means, int n, int d, int k, int* cl, hipStream_t st) {
  if (k > 256 || d > 3 || d < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, means, n, d, k, cl);
  return hipGetLastError();
}

}  // namespace gpb
