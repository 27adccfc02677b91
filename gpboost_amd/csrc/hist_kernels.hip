// gpboost_amd/csrc/hist_kernels.hip
//
// LightGBM feature-histogram build for gfx950.  Restates, for dense uint8 bins,
//   DenseBin<uint8_t,false>::ConstructHistogramInner   src/LightGBM/io/dense_bin.hpp:98-141
//   Dataset::ConstructHistogramsInner                  src/LightGBM/io/dataset.cpp:1143-1245
// i.e. for every row of the leaf: hist[f][bin(row,f)].grad += g[row]; and either .hess += h[row]
// or -- constant hessian -- ++count (converted to count * hess afterwards, dataset.cpp:1223-1226).
//
// MI355X mapping (HBM-bound byte work, no MFMA):
//   * bins are re-laid out once, at create time, from the reference's feature-major storage to
//     row-major [n][fpad]: a leaf's (gathered) rows then cost one 16-byte access per 16 features
//     instead of one byte per cache line;
//   * a workgroup owns 16 features x one chunk of rows; one lane = one row: a 16-byte load brings the
//     row's 16 bins, then the lane issues 16 LDS atomics -- at any instant all lanes of a wavefront update
//     the SAME feature's sub-histogram, so they only collide when two rows share a bin;
//   * sub-histograms are privatised in LDS (ds_add_f64 / ds_add_u32), written out per chunk and
//     summed over chunks in a fixed order by a second kernel: counts are exact (and therefore
//     reproducible); the fp64 sums depend on the LDS-atomic arrival order inside a chunk, i.e. they
//     are order-dependent exactly as the reference's per-thread block buffers are.
#include "hist_kernels.h"

namespace gpb {

template <bool HAS_HESS, bool HAS_IDX>
__global__ __launch_bounds__(256) void hist_build_kernel(HistKernelArgs a) {
  __shared__ double s_grad[GPB_HIST_FG][GPB_HIST_MAX_BIN + 1];
  __shared__ double s_hess[HAS_HESS ? GPB_HIST_FG : 1][HAS_HESS ? GPB_HIST_MAX_BIN + 1 : 1];
  __shared__ uint32_t s_cnt[GPB_HIST_FG][GPB_HIST_MAX_BIN + 1];
  const int tid = threadIdx.x;
  const int fg = blockIdx.x, chunk = blockIdx.y;   // the feature groups of one chunk are adjacent in dispatch order:
                                                   // they read the same 64-byte row segments while those are cache-hot
  for (int t = tid; t < GPB_HIST_FG * (GPB_HIST_MAX_BIN + 1); t += 256) {
    (&s_grad[0][0])[t] = 0.0;
    (&s_cnt[0][0])[t] = 0u;
    if constexpr (HAS_HESS) (&s_hess[0][0])[t] = 0.0;
  }
  __syncthreads();
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(r0 + a.rows_per_chunk, a.num_data);
  const uint8_t* base = a.bins_rm + (size_t)fg * GPB_HIST_FG;
  // one lane = one row: a 16-byte load brings the row's 16 bins of this feature group, the gradient load is
  // coalesced across the wavefront (or gathered through data_indices for a leaf)
  for (int r = r0 + tid; r < r1; r += 256) {
    const int row = HAS_IDX ? a.data_indices[r] : r;
    const uint4 bv = *reinterpret_cast<const uint4*>(base + (size_t)row * a.fpad);
    const double g = a.grad[row];
    double h = 0.0;
    if constexpr (HAS_HESS) h = a.hess[row];
    const unsigned long long lo = ((unsigned long long)bv.y << 32) | bv.x, hi = ((unsigned long long)bv.w << 32) | bv.z;
#pragma unroll
    for (int s = 0; s < GPB_HIST_FG; ++s) {
      // lane l handles feature (s + l) % 16 at step s: the 64 lanes of a wavefront spread over all 16 sub-histograms
      // instead of hammering one (4x fewer LDS-atomic collisions than a common feature per step)
      const int f = (s + tid) & 15;
      const int b = (int)(((f & 8) ? hi : lo) >> (8 * (f & 7))) & 0xff;
      atomicAdd(&s_grad[f][b], g);
      if constexpr (HAS_HESS) atomicAdd(&s_hess[f][b], h);
      else atomicAdd(&s_cnt[f][b], 1u);
      if constexpr (HAS_HESS) atomicAdd(&s_cnt[f][b], 1u);
    }
  }
  __syncthreads();
  const size_t pbase = ((size_t)chunk * a.fpad + (size_t)fg * GPB_HIST_FG) * GPB_HIST_MAX_BIN;
  for (int t = tid; t < GPB_HIST_FG * GPB_HIST_MAX_BIN; t += 256) {
    const int ff = t >> 8, b = t & 255;
    a.part_grad[pbase + t] = s_grad[ff][b];
    a.part_cnt[pbase + t] = s_cnt[ff][b];
    if constexpr (HAS_HESS) a.part_hess[pbase + t] = s_hess[ff][b];
  }
}

// one thread per (feature, bin): sum the chunk partials in chunk order
__global__ void hist_reduce_kernel(HistReduceArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = t >> 8, b = t & 255;
  if (f >= a.num_features) return;
  const int nb = a.bin_offsets[f + 1] - a.bin_offsets[f];
  if (b >= nb) return;
  double g = 0.0, h = 0.0;
  unsigned long long c = 0;
  for (int ch = 0; ch < a.nchunks; ++ch) {
    const size_t p = ((size_t)ch * a.fpad + f) * GPB_HIST_MAX_BIN + b;
    g += a.part_grad[p];
    c += a.part_cnt[p];
    if (a.has_hess) h += a.part_hess[p];
  }
  const size_t o = (size_t)a.bin_offsets[f] + b;
  a.hist_out[2 * o] = g;
  a.hist_out[2 * o + 1] = a.has_hess ? h : (double)c * a.const_hess;
  if (a.cnt_out) a.cnt_out[o] = c;
}

// feature-major [F][n] -> row-major [n][fpad] (padding features read as bin 0 and are never reduced)
__global__ void bins_transpose_kernel(const uint8_t* __restrict__ fm, uint8_t* __restrict__ rm, int n, int F, int fpad) {
  __shared__ uint8_t tile[16][64 + 4];
  const int f0 = blockIdx.y * 16, i0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 64 x 4
  for (int ff = ty; ff < 16; ff += 4) {
    const int f = f0 + ff, i = i0 + tx;
    tile[ff][tx] = (f < F && i < n) ? fm[(size_t)f * n + i] : 0;
  }
  __syncthreads();
  const int ff = threadIdx.x & 15, ii = threadIdx.x >> 4;    // 16 x 16
  for (int k = ii; k < 64; k += 16) {
    const int i = i0 + k;
    if (i < n) rm[(size_t)i * fpad + f0 + ff] = tile[ff][k];
  }
}

hipError_t launch_hist_build(const HistKernelArgs& a, hipStream_t st) {
  dim3 grid(a.fpad / GPB_HIST_FG, a.nchunks), block(256);
  const bool hh = a.hess != nullptr, hi = a.data_indices != nullptr;
  if (hh && hi) hipLaunchKernelGGL((hist_build_kernel<true, true>), grid, block, 0, st, a);
  else if (hh) hipLaunchKernelGGL((hist_build_kernel<true, false>), grid, block, 0, st, a);
  else if (hi) hipLaunchKernelGGL((hist_build_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((hist_build_kernel<false, false>), grid, block, 0, st, a);
  return hipGetLastError();
}
hipError_t launch_hist_reduce(const HistReduceArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(hist_reduce_kernel, dim3(a.num_features), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_bins_transpose(const uint8_t* bins_fm, uint8_t* bins_rm, int n, int F, int fpad, hipStream_t st) {
  hipLaunchKernelGGL(bins_transpose_kernel, dim3((n + 63) / 64, fpad / 16), dim3(256), 0, st, bins_fm, bins_rm, n, F, fpad);
  return hipGetLastError();
}

// ---- row a12: Dataset::FixHistogram (src/LightGBM/io/dataset.cpp:1272-1290) and FeatureHistogram::Subtract
// (src/LightGBM/treelearner/feature_histogram.hpp:79-83) on device-resident histograms -------------------------------
// One lane per feature walks its bins in ascending order: the same subtraction order as the reference's loop, so the
// reconstructed most-frequent-bin entry is bit-identical given the same histogram.
__global__ void hist_fix_kernel(double* __restrict__ hist, int num_features, const int* __restrict__ view_offset,
                                const int* __restrict__ num_bin, const int* __restrict__ most_freq_bin, double sum_gradient,
                                double sum_hessian) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num_features) return;
  const int mfb = most_freq_bin[f];
  if (mfb <= 0) return;
  double* v = hist + (size_t)view_offset[f] * 2;
  double g = sum_gradient, h = sum_hessian;
  const int nb = num_bin[f];
  for (int i = 0; i < nb; ++i) if (i != mfb) { g -= v[2 * i]; h -= v[2 * i + 1]; }
  v[2 * mfb] = g; v[2 * mfb + 1] = h;
}
__global__ void hist_subtract_kernel(const double* __restrict__ parent, const double* __restrict__ smaller, double* __restrict__ out, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) out[i] = parent[i] - smaller[i];
}
hipError_t launch_hist_fix(double* hist, int num_features, const int* view_offset, const int* num_bin, const int* most_freq_bin,
                           double sum_gradient, double sum_hessian, hipStream_t st) {
  hipLaunchKernelGGL(hist_fix_kernel, dim3((num_features + 63) / 64), dim3(64), 0, st, hist, num_features, view_offset, num_bin,
                     most_freq_bin, sum_gradient, sum_hessian);
  return hipGetLastError();
}
hipError_t launch_hist_subtract(const double* parent, const double* smaller, double* out, int len, hipStream_t st) {
  hipLaunchKernelGGL(hist_subtract_kernel, dim3((len + 255) / 256), dim3(256), 0, st, parent, smaller, out, len);
  return hipGetLastError();
}

}  // namespace gpb
