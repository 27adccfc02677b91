// gpboost_amd/csrc/vecchia_aux_kernels.hip
//
// The small, HBM-bound companions of vecchia_point_kernel (kept in their own translation unit so that touching them
// does not recompile the 18 heavy instantiations): final reduction of the per-workgroup partial sums, packing of the
// response into the point records, sparse products with B = I - A and B^T, and the fp64-DPP self-test.
#include "dev_common.h"
#include "vecchia_kernels.h"

namespace gpb {

// Deterministic final reduction: one workgroup per term; each thread sums a strided subset of the block partials
// (layout [term][nblocks], contiguous per term) in a fixed order, then a fixed-shape tree.  out[t] = sum_b partials[t][b].
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
                                                               double* __restrict__ out, double* __restrict__ out2,
                                                               double* __restrict__ out_host) {
  __shared__ double s[1024];
  const int t = blockIdx.x;
  const double* p = partials + (size_t)t * nblocks;
  double acc = 0.0, comp = 0.0;   // Kahan on the per-thread chain
  for (int b = threadIdx.x; b < nblocks; b += 1024) {
    const double v = p[b] - comp;
    const double tmp = acc + v;
    comp = (tmp - acc) - v;
    acc = tmp;
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[t] = s[0];
    if (out_host) out_host[t] = s[0];      // pinned host memory: the host reads it after the stream's synchronisation, no copy on the stream
    // caller-facing layout {quad, logdet, bad, g1v, g2v, g1r, g2r}: terms 0 and 1 swapped w.r.t. GPB_P_*
    if (out2) out2[t == GPB_P_LOGDET ? 1 : (t == GPB_P_QUAD ? 0 : t)] = s[0];
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

// the same with the response read from the packed points (pts[i].w), rows [i0, i1): renews u after a new response arrived while A, D stay valid
__global__ void vecchia_By_pts_kernel(const double* __restrict__ A, const int* __restrict__ nn, const double4* __restrict__ pts, int m, int i0, int i1,
                                      double* __restrict__ u) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i1) return;
  double s = pts[i].w;
  for (int j = 0; j < m; ++j) {
    const int c = nn[(size_t)i * m + j];
    if (c >= 0) s = __builtin_fma(-A[(size_t)i * m + j], pts[c].w, s);
  }
  u[i] = s;
}

// w = B^T v via the transposed index (CSR over columns): w_j = v_j - sum_{e in T[j]} A_flat[e] v[e / m]
// With a shard [i0, i1) only the rows of B owned by this device contribute (the caller all-reduces the n-vector).
// 16 lanes per column: the first points of the ordering are neighbours of thousands of rows (at n = 1e5 the longest column has > 3000
// entries against a mean of 30), and with one lane per column that one lane set the kernel's time (243 us at n = 1e5; now the entries of
// a column are strided over its 16 lanes and summed by a fixed-shape butterfly -- a fixed order, so the result stays reproducible).
__global__ __launch_bounds__(256) void vecchia_Bt_kernel(const double* __restrict__ A, const int* __restrict__ t_ptr,
                                  const int* __restrict__ t_pos, int n, int m, int i0, int i1,
                                  const double* __restrict__ v, double* __restrict__ w) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
  const int jj = j < n ? j : n - 1;                  // whole groups stay active for the shuffles
  const int e0 = t_ptr[jj], e1 = t_ptr[jj + 1];
  double s = 0.0;
  for (int e = e0 + lane; e < e1; e += 16) {
    const int pos = t_pos[e];
    const int row = pos / m;
    if (row >= i0 && row < i1) s = __builtin_fma(-A[pos], v[row], s);
  }
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (lane == 0 && j < n) w[j] = ((j >= i0 && j < i1) ? v[j] : 0.0) + s;
}

// diag(B^T D^-1 B)_j = 1 / D_j + sum_{e in T[j]} A_flat[e]^2 / D[e / m]: the diagonal of Psi^-1 on the transformed scale, from which the
// predictive variances of the training-data random effects follow (PredictTrainingDataRandomEffects, re_model_template.h:4508-4514:
// var_i = sigma2 (1 - (B o (D^-1 B)) column sums)).  Same 16-lanes-per-column gather and fixed butterfly as vecchia_Bt_kernel.
__global__ __launch_bounds__(256) void vecchia_BtDinvB_diag_kernel(const double* __restrict__ A, const double* __restrict__ D,
                                  const int* __restrict__ t_ptr, const int* __restrict__ t_pos, int n, int m, double* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
  const int jj = j < n ? j : n - 1;
  const int e0 = t_ptr[jj], e1 = t_ptr[jj + 1];
  double s = 0.0;
  for (int e = e0 + lane; e < e1; e += 16) {
    const int pos = t_pos[e];
    const double a = A[pos];
    s = __builtin_fma(a, a / D[pos / m], s);
  }
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (lane == 0 && j < n) out[j] = 1.0 / D[j] + s;
}

// v = u / D elementwise
__global__ void scale_by_Dinv_kernel(const double* __restrict__ u, const double* __restrict__ D, int n, int i0, int i1,
                                     double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (i >= i0 && i < i1) ? u[i] / D[i] : 0.0;
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


// ---- linear-regression covariates, Gaussian likelihood (GPB_OptimLinRegrCoefCovPar: UpdateCoefGLS / CalcXTPsiInvX, re_model_template.h:10012-10019,
//      6624-6628): Gram matrix of U = B [x_1 .. x_p, y0] with weights 1/D, and the residual response y0 - X beta -----------------------------
// G[a][b] = sum_i U[a][i] U[b][i] / D[i], a >= b (lower triangle); one workgroup per pair, fixed order -> bit-reproducible
__global__ __launch_bounds__(1024) void gram_kernel(const double* __restrict__ U, const double* __restrict__ D, int n, int q, double* __restrict__ G) {
  __shared__ double s[1024];
  int t = blockIdx.x, a = 0;
  while (t >= a + 1) { t -= a + 1; ++a; }            // pair index -> (a, b = t), b <= a
  const int b = t;
  const double* ua = U + (size_t)a * n;
  const double* ub = U + (size_t)b * n;
  double acc = 0.0, comp = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double v = ua[i] * ub[i] / D[i] - comp;
    const double tmp = acc + v;
    comp = (tmp - acc) - v;
    acc = tmp;
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) { G[(size_t)a * q + b] = s[0]; G[(size_t)b * q + a] = s[0]; }
}
// pts[i].w = y0[i] - sum_j X[j][i] beta[j]   (response := residual; UpdateFixedEffects, re_model_template.h:2859-2871)
__global__ void resid_kernel(double4* __restrict__ pts, const double* __restrict__ y0, const double* __restrict__ X, const double* __restrict__ beta,
                             int n, int p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = y0[i];
  for (int j = 0; j < p; ++j) r -= X[(size_t)j * n + i] * beta[j];    // left-to-right like Eigen's X * beta row
  pts[i].w = r;
}
hipError_t launch_gram(const double* U, const double* D, int n, int q, double* G, hipStream_t st) {
  hipLaunchKernelGGL(gram_kernel, dim3(q * (q + 1) / 2), dim3(1024), 0, st, U, D, n, q, G);
  return hipGetLastError();
}
hipError_t launch_resid(double4* pts, const double* y0, const double* X, const double* beta, int n, int p, hipStream_t st) {
  hipLaunchKernelGGL(resid_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, y0, X, beta, n, p);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const double* partials, int nblocks, int nterms, double* out, double* out_user,
                                  hipStream_t st, double* out_host) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(nterms), dim3(1024), 0, st, partials, nblocks, out, out_user, out_host);
  return hipGetLastError();
}
// dst (pinned host memory) <- src (device), a handful of doubles: see vecchia_allreduce_terms
__global__ void publish_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}
hipError_t launch_publish(const double* src, double* dst_host, int n, hipStream_t st) {
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, src, dst_host, n);
  return hipGetLastError();
}
// Spatially sorted copy of the point records for the neighbour gathers of vecchia_point_kernel (round 5): pts[n + rank[k]] = pts[k], and the neighbour table
// rewritten to point into that copy (nn2 = n + rank[nn], -1 stays).  rank = position of point k in Morton order of the coordinates.
__global__ void scatter_pts_kernel(double4* __restrict__ pts, const int* __restrict__ rank, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) pts[(size_t)n + rank[k]] = pts[k];
}
__global__ void remap_nn_kernel(const int* __restrict__ nn, const int* __restrict__ rank, size_t cnt, int n, int* __restrict__ nn2) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < cnt) { const int c = nn[t]; nn2[t] = c >= 0 ? n + rank[c] : -1; }
}
hipError_t launch_scatter_pts(double4* pts, const int* rank, int n, hipStream_t st) {
  hipLaunchKernelGGL(scatter_pts_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, rank, n);
  return hipGetLastError();
}
hipError_t launch_remap_nn(const int* nn, const int* rank, size_t cnt, int n, int* nn2, hipStream_t st) {
  hipLaunchKernelGGL(remap_nn_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, nn, rank, cnt, n, nn2);
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
hipError_t launch_By_pts(const double* A, const int* nn, const double4* pts, int m, int i0, int i1, double* u, hipStream_t st) {
  if (i1 <= i0) return hipSuccess;
  hipLaunchKernelGGL(vecchia_By_pts_kernel, dim3((i1 - i0 + 255) / 256), dim3(256), 0, st, A, nn, pts, m, i0, i1, u);
  return hipGetLastError();
}
hipError_t launch_Bt(const double* A, const int* t_ptr, const int* t_pos, int n, int m, int i0, int i1, const double* v,
                     double* w, hipStream_t st) {
  hipLaunchKernelGGL(vecchia_Bt_kernel, dim3((n + 15) / 16), dim3(256), 0, st, A, t_ptr, t_pos, n, m, i0, i1, v, w);
  return hipGetLastError();
}
hipError_t launch_BtDinvB_diag(const double* A, const double* D, const int* t_ptr, const int* t_pos, int n, int m, double* out, hipStream_t st) {
  hipLaunchKernelGGL(vecchia_BtDinvB_diag_kernel, dim3(((size_t)n * 16 + 255) / 256), dim3(256), 0, st, A, D, t_ptr, t_pos, n, m, out);
  return hipGetLastError();
}
hipError_t launch_scale_by_Dinv(const double* u, const double* D, int n, int i0, int i1, double* v, hipStream_t st) {
  hipLaunchKernelGGL(scale_by_Dinv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, u, D, n, i0, i1, v);
  return hipGetLastError();
}
hipError_t launch_dpp_selftest(const double* in, double* out, hipStream_t st) {
  hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, st, in, out);
  return hipGetLastError();
}


}  // namespace gpb
