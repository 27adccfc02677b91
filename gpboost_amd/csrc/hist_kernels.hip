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
  // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin by their linear id and every XCD has its own L2, so the
  // feature groups of ONE chunk (which read the same 64-byte row segments) must sit 8 ids apart to meet in the same L2:
  // id = 8 * (groups * (chunk / 8) + fg) + chunk % 8.  (With fg fastest -- the first version -- the four groups of a chunk ran on
  // four XCDs and every row segment was fetched from HBM four times: FETCH_SIZE 1.44 GB per launch against 0.72 GB of rows + gradients.)
  const int id = blockIdx.x, groups = a.fpad / GPB_HIST_FG;
  int fg, chunk;
  if ((a.nchunks & 7) == 0) { chunk = (id / (8 * groups)) * 8 + (id & 7); fg = (id >> 3) % groups; }
  else { fg = id % groups; chunk = id / groups; }
  for (int t = tid; t < GPB_HIST_FG * (GPB_HIST_MAX_BIN + 1); t += 256) {
    (&s_grad[0][0])[t] = 0.0;
    (&s_cnt[0][0])[t] = 0u;
    if constexpr (HAS_HESS) (&s_hess[0][0])[t] = 0.0;
  }
  __syncthreads();
  const int nf = min(GPB_HIST_FG, a.num_features - fg * GPB_HIST_FG);   // real features of this group (the last group may be partial)
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(r0 + a.rows_per_chunk, a.num_data);
  const uint8_t* base = a.bins_rm + (size_t)fg * GPB_HIST_FG;
  // one lane = one row: a 16-byte load brings the row's 16 bins of this feature group, the gradient load is
  // coalesced across the wavefront (or gathered through data_indices for a leaf).  (An explicit software pipeline of the
  // next row's loads changes nothing: 12 resident wavefronts per CU already hide the HBM latency; the kernel is bound by
  // the LDS atomics, scripts/ubench/lds_atomics.hip.)
  for (int r = r0 + tid; r < r1; r += 256) {
    const int row = HAS_IDX ? a.data_indices[r] : r;
    const uint4 bv = *reinterpret_cast<const uint4*>(base + (size_t)row * a.fpad);
    const double g = a.grad[row];
    double h = 0.0;
    if constexpr (HAS_HESS) h = a.hess[row];
    const unsigned long long lo = ((unsigned long long)bv.y << 32) | bv.x, hi = ((unsigned long long)bv.w << 32) | bv.z;
    if (nf < GPB_HIST_FG) {
      // last, partial feature group: only its nf real features are accumulated (the padding features all sit in bin 0: 4 lanes of every
      // step would hit ONE address, the slowest case of the LDS atomic unit, for entries nobody reads -- 22 % of the atomics at F = 50)
      int f = tid % nf;
      for (int s = 0; s < nf; ++s) {
        const int b = (int)(((f & 8) ? hi : lo) >> (8 * (f & 7))) & 0xff;
        atomicAdd(&s_grad[f][b], g);
        if constexpr (HAS_HESS) atomicAdd(&s_hess[f][b], h);
        atomicAdd(&s_cnt[f][b], 1u);
        f = (f + 1 == nf) ? 0 : f + 1;
      }
      continue;
    }
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

// Sum of the chunk partials, in a FIXED order (reproducible for a given chunking): a workgroup owns 64 bins of one feature;
// its 16 slices of 64 lanes take the chunks ch = slice, slice + 16, ... (coalesced 512-byte segments, 4 loads in flight per lane),
// the slice totals are added in slice order.  200 workgroups x 1024 threads for F = 50 (the first version -- one thread per
// (feature, bin) walking all chunks, 50 workgroups -- took 207 us of a 760 us root pass at n = 1e7).
__global__ __launch_bounds__(1024) void hist_reduce_kernel(HistReduceArgs a) {
  __shared__ double s_g[16][64];
  __shared__ double s_h[16][64];
  __shared__ unsigned long long s_c[16][64];
  const int f = blockIdx.x, b = blockIdx.y * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int nb = a.bin_offsets[f + 1] - a.bin_offsets[f];
  double g = 0.0, h = 0.0;
  unsigned long long c = 0;
  const size_t stride = (size_t)a.fpad * GPB_HIST_MAX_BIN;
  const size_t p0 = (size_t)f * GPB_HIST_MAX_BIN + b;
  int ch = sl;
  for (; ch + 48 < a.nchunks; ch += 64) {
    const size_t p = p0 + (size_t)ch * stride;
    const double g0 = a.part_grad[p], g1 = a.part_grad[p + 16 * stride], g2 = a.part_grad[p + 32 * stride], g3 = a.part_grad[p + 48 * stride];
    const uint32_t c0 = a.part_cnt[p], c1 = a.part_cnt[p + 16 * stride], c2 = a.part_cnt[p + 32 * stride], c3 = a.part_cnt[p + 48 * stride];
    g += g0; g += g1; g += g2; g += g3;
    c += c0; c += c1; c += c2; c += c3;
    if (a.has_hess) {
      const double h0 = a.part_hess[p], h1 = a.part_hess[p + 16 * stride], h2 = a.part_hess[p + 32 * stride], h3 = a.part_hess[p + 48 * stride];
      h += h0; h += h1; h += h2; h += h3;
    }
  }
  for (; ch < a.nchunks; ch += 16) {
    const size_t p = p0 + (size_t)ch * stride;
    g += a.part_grad[p];
    c += a.part_cnt[p];
    if (a.has_hess) h += a.part_hess[p];
  }
  s_g[sl][threadIdx.x & 63] = g; s_h[sl][threadIdx.x & 63] = h; s_c[sl][threadIdx.x & 63] = c;
  __syncthreads();
  if (sl != 0 || b >= nb) return;
  g = 0.0; h = 0.0; c = 0;
  for (int k = 0; k < 16; ++k) { g += s_g[k][threadIdx.x]; h += s_h[k][threadIdx.x]; c += s_c[k][threadIdx.x]; }
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

// leaf id of every row from the resident row lists: position p of `rows` belongs to the segment with the largest begin <= p
__global__ void hist_label_rows_kernel(const int* __restrict__ rows, int n, const int* __restrict__ seg_begin, const int* __restrict__ seg_leaf,
                                       int nseg, int* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_begin[mid] <= p) lo = mid; else hi = mid - 1; }
  out[rows[p]] = seg_leaf[lo];
}
hipError_t launch_hist_label_rows(const int* rows, int n, const int* seg_begin, const int* seg_leaf, int nseg, int* out, hipStream_t st) {
  hipLaunchKernelGGL(hist_label_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rows, n, seg_begin, seg_leaf, nseg, out);
  return hipGetLastError();
}

hipError_t launch_hist_build(const HistKernelArgs& a, hipStream_t st) {
  dim3 grid((a.fpad / GPB_HIST_FG) * a.nchunks), block(256);
  const bool hh = a.hess != nullptr, hi = a.data_indices != nullptr;
  if (hh && hi) hipLaunchKernelGGL((hist_build_kernel<true, true>), grid, block, 0, st, a);
  else if (hh) hipLaunchKernelGGL((hist_build_kernel<true, false>), grid, block, 0, st, a);
  else if (hi) hipLaunchKernelGGL((hist_build_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((hist_build_kernel<false, false>), grid, block, 0, st, a);
  return hipGetLastError();
}
hipError_t launch_hist_reduce(const HistReduceArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(hist_reduce_kernel, dim3(a.num_features, GPB_HIST_MAX_BIN / 64), dim3(1024), 0, st, a);
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

// ---- split search on a device-resident leaf histogram -----------------------------------------------
// FeatureHistogram::FindBestThreshold for numerical features on the default regularisation path (lambda_l1 = 0, max_delta_step = 0,
// path_smooth = 0, no monotone constraints, no extra_trees): src/LightGBM/treelearner/feature_histogram.hpp:85-95, :97-114, :163-207,
// :857-1084, :797-836, :741-763; the winner among features as SerialTreeLearner::ComputeBestSplitForFeature +
// SplitInfo::operator> pick it (serial_tree_learner.cpp:725-756, split_info.hpp:126-153).
//
// One workgroup per feature.  The reference's scan is sequential only in its running sums; everything else of a step (rounded
// counts, continue / break tests, the gain with its two divisions) depends on those sums alone.  So: (1) the feature's entries and
// their rounded counts (contraction off: `hess * cnt_factor + 0.5f` must round twice) go to LDS; (2) ONE lane per scan direction (two
// wavefronts, concurrently) accumulates the sums in the reference's order, branch-free, and stores them per step; (3) all lanes apply
// the reference's continue / break tests -- the scan ends at the first break in scan order; (4) all lanes evaluate the gains; (5) a
// reduction picks the first maximal gain in scan order (the reference updates on `>` only).  Every field of the result is
// bit-identical to the sequential walk given the same histogram.  (The first version walked the bins with one lane per feature:
// 115 us per call, 2/3 of a tree at config 3.)
namespace {
struct SplitOut { double gain, left_output, right_output, lsg, lsh, rsg, rsh; unsigned threshold; int left_count, right_count, default_left; };
constexpr int kSplitSteps = GPB_HIST_MAX_BIN + 2;      // step index t + 1 (the forward scan may start at t = -1)
}  // namespace

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void hist_best_split_kernel(const double* __restrict__ hist, int num_features, const int* __restrict__ view_offset,
                                       const int* __restrict__ num_bin, const int* __restrict__ meta3 /* offset, default_bin, missing */,
                                       double sum_gradient, double sum_hessian_leaf, int num_data, double lambda_l2, int min_data_in_leaf,
                                       double min_sum_hessian, double min_gain_to_split, double* __restrict__ out10,
                                       int* __restrict__ out_default_left) {
#pragma clang fp contract(off)
  __shared__ double s_d[2 * (GPB_HIST_MAX_BIN + 1)];
  __shared__ double s_ag[2][kSplitSteps], s_ah[2][kSplitSteps];
  __shared__ int s_ac[2][kSplitSteps];
  __shared__ signed char s_eval[2][kSplitSteps];
  __shared__ double s_rg[256];
  __shared__ int s_rt[256], s_rs[256];
  __shared__ int s_cnt[GPB_HIST_MAX_BIN + 1];
  __shared__ int s_brk[2];                                 // first break of the reverse (max t) / forward (min t) scan
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= num_features) return;
  const double kEps = (double)1e-15f;                      // include/LightGBM/meta.h:54
  const double* data = hist + (size_t)view_offset[f] * 2;
  const int nb = num_bin[f], offset = meta3[3 * f], default_bin = meta3[3 * f + 1], missing = meta3[3 * f + 2];
  const double sum_hessian = sum_hessian_leaf + 2 * kEps;
  const double min_gain_shift = (sum_gradient * sum_gradient) / (sum_hessian + lambda_l2) + min_gain_to_split;
  const double l2 = lambda_l2;
  const bool two_scans = nb > 2 && missing != 0;
  const bool skip_default = two_scans && missing == 1;
  const int na_as_missing = (two_scans && missing != 1) ? 1 : 0;
  const int nent = nb - offset;                            // entries of the feature's view
  for (int i = tid; i < 2 * nent; i += 256) s_d[i] = data[i];
  for (int i = tid; i < 2 * kSplitSteps; i += 256) (&s_eval[0][0])[i] = 0;
  __syncthreads();
  const double cnt_factor = num_data / sum_hessian;
  for (int i = tid; i < nent; i += 256) s_cnt[i] = (int)(s_d[2 * i + 1] * cnt_factor + 0.5f);      // Common::RoundInt, utils/common.h:920-922
  if (tid < 2) s_brk[tid] = tid == 0 ? -2147483647 : 2147483647;
  __syncthreads();
  // (2) the running sums, one lane per direction, branch-free so that the LDS reads run ahead of the two add chains
  const int r_hi = nb - 1 - offset - na_as_missing, r_lo = 1 - offset;        // reverse scan: t = r_hi .. r_lo  (:880-960)
  const int f_hi = nb - 2 - offset;                                           // forward scan: t = f_lo .. f_hi  (:962-1050)
  const bool fwd_pre = na_as_missing && offset == 1;
  const int f_lo = fwd_pre ? -1 : 0;
  if (tid == 0) {
    double srg = 0.0, srh = kEps;
    int rc = 0;
#pragma unroll 4
    for (int t = r_hi; t >= r_lo; --t) {
      const bool skip = skip_default && (t + offset) == default_bin;
      const double g = s_d[2 * t], hh = s_d[2 * t + 1];
      const int c = s_cnt[t];
      const double srg2 = srg + g, srh2 = srh + hh;
      srg = skip ? srg : srg2; srh = skip ? srh : srh2; rc = skip ? rc : rc + c;
      s_ag[0][t + 1] = srg; s_ah[0][t + 1] = srh; s_ac[0][t + 1] = rc;
    }
  } else if (tid == 64 && two_scans) {
    double slg = 0.0, slh = kEps;
    int lc = 0;
    if (fwd_pre) {
      slg = sum_gradient; slh = sum_hessian - kEps; lc = num_data;
#pragma unroll 4
      for (int i = 0; i < nent; ++i) { slg -= s_d[2 * i]; slh -= s_d[2 * i + 1]; lc -= s_cnt[i]; }
    }
#pragma unroll 4
    for (int t = f_lo; t <= f_hi; ++t) {
      const bool skip = (skip_default && (t + offset) == default_bin) || t < 0;
      const int tt = t < 0 ? 0 : t;
      const double g = s_d[2 * tt], hh = s_d[2 * tt + 1];
      const int c = s_cnt[tt];
      const double slg2 = slg + g, slh2 = slh + hh;
      slg = skip ? slg : slg2; slh = skip ? slh : slh2; lc = skip ? lc : lc + c;
      s_ag[1][t + 1] = slg; s_ah[1][t + 1] = slh; s_ac[1][t + 1] = lc;
    }
  }
  __syncthreads();
  // (3) the reference's continue / break conditions of every step, in parallel; the scan stops at the FIRST break in scan order
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir) {
    for (int k = tid; k < kSplitSteps; k += 256) {
      const int t = k - 1;
      const bool in_range = dir == 0 ? (t >= r_lo && t <= r_hi) : (t >= f_lo && t <= f_hi);
      if (!in_range) continue;
      if (skip_default && (t + offset) == default_bin) continue;            // `continue` before anything is accumulated
      const double ah = s_ah[dir][k];
      const int ac = s_ac[dir][k];
      if (ac < min_data_in_leaf || ah < min_sum_hessian) continue;
      const int other_count = num_data - ac;
      const double other_h = sum_hessian - ah;
      if (other_count < min_data_in_leaf || other_h < min_sum_hessian) { if (dir == 0) atomicMax(&s_brk[0], t); else atomicMin(&s_brk[1], t); continue; }
      s_eval[dir][k] = 1;
    }
  }
  __syncthreads();
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir)
    for (int k = tid; k < kSplitSteps; k += 256) {
      const int t = k - 1;
      if (s_eval[dir][k] && (dir == 0 ? t < s_brk[0] : t > s_brk[1])) s_eval[dir][k] = 0;      // after the break in scan order
    }
  __syncthreads();
  SplitOut o;
  o.gain = -INFINITY; o.left_output = 0.0; o.right_output = 0.0; o.lsg = 0.0; o.lsh = 0.0; o.rsg = 0.0; o.rsh = 0.0;
  o.threshold = 0; o.left_count = 0; o.right_count = 0; o.default_left = 1;
  bool splittable = false;
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir) {
    // gains of the recorded steps, all lanes; position in scan order: reverse = descending t, forward = ascending t
    double gbest = -INFINITY; int tbest = -1; int any = 0;
    for (int k = tid; k < kSplitSteps; k += 256) {
      if (!s_eval[dir][k]) continue;
      double slg, slh, srg, srh;
      if (dir == 0) { srg = s_ag[0][k]; srh = s_ah[0][k]; slh = sum_hessian - srh; slg = sum_gradient - srg; }
      else { slg = s_ag[1][k]; slh = s_ah[1][k]; srh = sum_hessian - slh; srg = sum_gradient - slg; }
      const double current_gain = (slg * slg) / (slh + l2) + (srg * srg) / (srh + l2);
      if (current_gain <= min_gain_shift) continue;
      any = 1;
      const bool earlier = tbest < 0 || (dir == 0 ? k > tbest : k < tbest);
      if (current_gain > gbest || (current_gain == gbest && earlier)) { gbest = current_gain; tbest = k; }
    }
    s_rg[tid] = gbest; s_rt[tid] = tbest; s_rs[tid] = any;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if (tid < w) {
        const double g2 = s_rg[tid + w]; const int t2 = s_rt[tid + w];
        s_rs[tid] |= s_rs[tid + w];
        if (t2 >= 0) {
          const int t1 = s_rt[tid];
          const bool earlier = t1 < 0 || (dir == 0 ? t2 > t1 : t2 < t1);
          if (g2 > s_rg[tid] || (g2 == s_rg[tid] && earlier)) { s_rg[tid] = g2; s_rt[tid] = t2; }
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      if (s_rs[0]) splittable = true;
      const double best_gain = s_rg[0];
      const int k = s_rt[0];
      if (splittable && k >= 0 && best_gain > o.gain + min_gain_shift) {
        const int t = k - 1;
        double best_slg, best_slh; int best_left_count;
        if (dir == 0) { best_left_count = num_data - s_ac[0][k]; best_slg = sum_gradient - s_ag[0][k]; best_slh = sum_hessian - s_ah[0][k]; o.threshold = (unsigned)(t - 1 + offset); }
        else { best_left_count = s_ac[1][k]; best_slg = s_ag[1][k]; best_slh = s_ah[1][k]; o.threshold = (unsigned)(t + offset); }
        o.left_output = -best_slg / (best_slh + l2);
        o.left_count = best_left_count;
        o.lsg = best_slg; o.lsh = best_slh - kEps;
        o.right_output = -(sum_gradient - best_slg) / (sum_hessian - best_slh + l2);
        o.right_count = num_data - best_left_count;
        o.rsg = sum_gradient - best_slg; o.rsh = sum_hessian - best_slh - kEps;
        o.gain = best_gain - min_gain_shift;
        o.default_left = dir == 0 ? 1 : 0;
      }
    }
    __syncthreads();
  }
  if (tid != 0) return;
  if (!two_scans && missing == 2) o.default_left = 0;
  double* r = out10 + (size_t)f * 10;
  r[0] = o.gain; r[1] = (double)o.threshold; r[2] = o.left_count; r[3] = o.right_count; r[4] = o.left_output; r[5] = o.right_output;
  r[6] = o.lsg; r[7] = o.lsh; r[8] = o.rsg; r[9] = o.rsh;
  out_default_left[f] = o.default_left | (splittable ? 2 : 0);      // bit 1: FeatureHistogram::is_splittable() after the search
}

// the winner: larger gain, equal gains -> smaller feature index; features masked out by is_feature_used never win
__global__ void hist_pick_split_kernel(const double* __restrict__ out10, int num_features, const signed char* __restrict__ is_feature_used,
                                       int* __restrict__ best_feature) {
  // one wavefront: lanes stride over the features, then a shuffle reduction with SplitInfo::operator>'s order
  const int lane = threadIdx.x;
  int best_f = 2147483647;
  double best_gain = -INFINITY;
  for (int f = lane; f < num_features; f += 64) {
    if (is_feature_used && !is_feature_used[f]) continue;
    const double g = out10[(size_t)f * 10];
    if (g != best_gain ? g > best_gain : f < best_f) { best_gain = g; best_f = f; }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const double g2 = __shfl_xor(best_gain, off, 64);
    const int f2 = __shfl_xor(best_f, off, 64);
    if (g2 != best_gain ? g2 > best_gain : f2 < best_f) { best_gain = g2; best_f = f2; }
  }
  if (lane == 0) *best_feature = best_f == 2147483647 ? -1 : best_f;
}

hipError_t launch_hist_best_split(const double* hist, int num_features, const int* view_offset, const int* num_bin, const int* meta3,
                                  double sum_gradient, double sum_hessian, int num_data, double lambda_l2, int min_data_in_leaf,
                                  double min_sum_hessian, double min_gain_to_split, const signed char* is_feature_used, double* out10,
                                  int* out_default_left, int* best_feature, hipStream_t st) {
  hipLaunchKernelGGL(hist_best_split_kernel, dim3(num_features), dim3(256), 0, st, hist, num_features, view_offset, num_bin, meta3,
                     sum_gradient, sum_hessian, num_data, lambda_l2, min_data_in_leaf, min_sum_hessian, min_gain_to_split, out10,
                     out_default_left);
  if (best_feature)      // the tree grower picks on the host from the per-feature candidates it needs anyway
    hipLaunchKernelGGL(hist_pick_split_kernel, dim3(1), dim3(64), 0, st, (const double*)out10, num_features, is_feature_used, best_feature);
  return hipGetLastError();
}

// ---- partition of a leaf's rows by a numerical split (second half of SURVEY.md 8f rank 2) ---------------------------------
// DataPartition::Split -> Dataset::Split -> DenseBin::Split / SplitInner (src/LightGBM/io/dense_bin.hpp:176-307, single-feature
// group: min_bin = 1, USE_MIN_BIN = false).  A stable partition: both sides keep the order of data_indices, as the reference's
// per-thread blocks do when they are concatenated.  Three small kernels: per-block count of rows going left, exclusive scan of
// the block counts, classify again + scatter (1024 rows per block, 4 consecutive rows per lane).
namespace {
struct SplitRule { int max_bin, t_zero_bin, th, miss_zero, miss_na, mfb_zero, mfb_na, default_goes_left, missing_goes_left; };
__device__ __forceinline__ bool goes_left(const SplitRule& r, int bin) {
  if (1 < r.max_bin) {
    if ((r.miss_zero && !r.mfb_zero && bin == r.t_zero_bin) || (r.miss_na && !r.mfb_na && bin == r.max_bin)) return r.missing_goes_left;
    if (bin == 0) return ((r.miss_na && r.mfb_na) || (r.miss_zero && r.mfb_zero)) ? r.missing_goes_left : r.default_goes_left;
    return !(bin > r.th);
  }
  if (r.miss_zero && !r.mfb_zero && bin == r.t_zero_bin) return r.missing_goes_left;
  if (bin != r.max_bin) return ((r.miss_na && r.mfb_na) || (r.miss_zero && r.mfb_zero)) ? r.missing_goes_left : r.default_goes_left;
  return (r.miss_na && !r.mfb_na) ? r.missing_goes_left : (r.max_bin <= r.th);
}
}  // namespace

template <bool SCATTER>
__global__ __launch_bounds__(256) void hist_partition_kernel(const uint8_t* __restrict__ bins_rm, int fpad, int feature, SplitRule rule,
                                                             const int* __restrict__ data_indices, int cnt, int* __restrict__ blk_cnt,
                                                             const int* __restrict__ blk_off, int* __restrict__ lte, int* __restrict__ gt) {
  __shared__ int s_scan[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 1024 + tid * 4;
  int idx[4]; bool left[4];
  int nl = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = base + k;
    left[k] = false; idx[k] = 0;
    if (p < cnt) {
      idx[k] = data_indices ? data_indices[p] : p;
      left[k] = goes_left(rule, (int)bins_rm[(size_t)idx[k] * fpad + feature]);
      nl += left[k] ? 1 : 0;
    }
  }
  s_scan[tid] = nl;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {            // inclusive scan of the per-lane counts
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  if (!SCATTER) {
    if (tid == 255) blk_cnt[blockIdx.x] = s_scan[255];
    return;
  }
  int l = blk_off[blockIdx.x] + s_scan[tid] - nl;                      // rows going left before this lane's first row
  int g = (blockIdx.x * 1024 + tid * 4) - l;                           // rows going right before it = position - lefts
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < cnt) { if (left[k]) lte[l++] = idx[k]; else gt[g++] = idx[k]; }
  }
}

// exclusive scan of the block counts (one block; nblk is small: cnt / 1024), total written to off[nblk]
__global__ void hist_partition_scan_kernel(const int* __restrict__ blk_cnt, int nblk, int* __restrict__ off) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int run = 0;
  for (int b = 0; b < nblk; ++b) { off[b] = run; run += blk_cnt[b]; }
  off[nblk] = run;
}

hipError_t launch_hist_partition(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                 int missing_type, int default_left, unsigned threshold, const int* data_indices, int cnt, int* blk_cnt,
                                 int* blk_off, int* lte, int* gt, hipStream_t st) {
  SplitRule r;
  r.max_bin = max_bin;
  r.miss_zero = missing_type == 1; r.miss_na = missing_type == 2;
  r.mfb_zero = r.miss_zero && default_bin == most_freq_bin;                          // dense_bin.hpp:268-272
  r.mfb_na = r.miss_na && (max_bin == most_freq_bin + 1 && most_freq_bin > 0);       // :294-298 with min_bin = 1
  int th = (int)((threshold + 1u) & 0xffu), tz = (1 + default_bin) & 0xff;           // VAL_T = uint8_t arithmetic (:182-187)
  if (most_freq_bin == 0) { th = (th - 1) & 0xff; tz = (tz - 1) & 0xff; }
  r.th = th; r.t_zero_bin = tz;
  r.default_goes_left = (unsigned)most_freq_bin <= threshold;                        // :196-199
  r.missing_goes_left = (r.miss_zero || r.miss_na) && default_left;                  // :200-205
  const int nblk = (cnt + 1023) / 1024;
  if (nblk == 0) return hipSuccess;
  hipLaunchKernelGGL(hist_partition_kernel<false>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, data_indices, cnt, blk_cnt,
                     (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  hipLaunchKernelGGL(hist_partition_scan_kernel, dim3(1), dim3(64), 0, st, (const int*)blk_cnt, nblk, blk_off);
  hipLaunchKernelGGL(hist_partition_kernel<true>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, data_indices, cnt, blk_cnt,
                     (const int*)blk_off, lte, gt);
  return hipGetLastError();
}

}  // namespace gpb
