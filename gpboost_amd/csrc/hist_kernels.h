// gpboost_amd/csrc/hist_kernels.h -- launch interface of hist_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpb {

#define GPB_HIST_FG 16          // features per workgroup (one per lane of a 16-lane row)
#define GPB_HIST_MAX_BIN 256    // uint8 bins

struct HistKernelArgs {
  const uint8_t* bins_rm;   // [n][fpad] row-major bins, fpad = multiple of 16
  // round 6: a second, COMPACT copy [n][rstride] (rstride = F rounded up to 4 bytes; + 16 bytes of slack behind the last row) for the passes that STREAM every
  // row (no index list): the 64-byte padded rows cost 28 % more HBM bytes than the algorithmic count at F = 50 (counter traffic 1.24x).  Gathered
  // leaves keep the padded copy: a 64-byte-aligned row is ONE sector, a 52-byte row straddles two.  nullptr / 0: only the padded copy exists (F % 16 == 0).
  const uint8_t* bins_cm = nullptr; int rstride = 0;
  const int* data_indices;  // leaf rows or nullptr
  const double* grad;       // [n]
  const double* hess;       // [n] or nullptr (constant hessian)
  long long* part_grad;     // [nchunks][fpad / 16][256 bins][16 features]  fixed-point sums (integers, units of q)
  long long* part_hess;     // same layout  (non-constant hessian)
  uint32_t* part_cnt;       // same layout
  const unsigned long long* grad_max_bits;   // IEEE bits of max |grad| / max |hess| over the rows set by gpb_hip_hist_set_gradients:
  const unsigned long long* hess_max_bits;   // they fix the power-of-two scale q of the fixed-point sums (launch_hist_absmax)
  int fpad, num_data, rows_per_chunk, nchunks;
  int num_features;         // real features (<= fpad): the padding features of the last group are not accumulated
  // tree grower: when seg_counts != nullptr the rows are the SMALLER child of the split of the segment (seg_begin, seg_cnt) of
  // data_indices whose left counts {this rank, all ranks} sit in seg_counts (device memory); num_data / rows_per_chunk are ignored
  int quad0 = 0;            // hist_build_rows_kernel: first quad of feature groups of this launch
  int use_rows_kernel = 0;  // hist_build_rows_kernel (constant hessian, >= 4 feature groups): set by the host with a chunking of one workgroup per CU
  const int* seg_counts = nullptr;
  int seg_begin = 0, seg_cnt = 0, seg_gcnt = 0, seg_min_data_in_leaf = 0;
};

// regularisation of the split search beyond lambda_l2 (feature_histogram.hpp:137-161) + the parent_output argument of FindBestThreshold
struct SplitReg { double lambda_l1 = 0.0, max_delta_step = 0.0, path_smooth = 0.0, parent_output = 0.0; };

// categorical features (round 5): the configuration of FindBestThresholdCategoricalInner (include/LightGBM/config.h: max_cat_to_onehot 4,
// max_cat_threshold 32, cat_smooth 10, cat_l2 10, min_data_per_group 100)
struct CatCfg { int max_cat_to_onehot = 4, max_cat_threshold = 32, min_data_per_group = 100; double cat_smooth = 10.0, cat_l2 = 10.0; };

struct ChildrenSearchArgs {
  double* smaller;          // slot of the smaller child's histogram (fresh from the build, not yet fixed)
  double* parent;           // slot of the parent's histogram: becomes the larger child's
  const int* counts;        // {left rows of this rank, left rows of all ranks} of the split (device)
  const int* bin_offsets; const int* view_offset; const int* num_bin; const int* most_freq_bin; const int* meta3;
  int num_features, gcnt, min_data_in_leaf;
  double left_sum_gradient, left_sum_hessian, right_sum_gradient, right_sum_hessian;     // of the parent's split
  double lambda_l2, min_sum_hessian, min_gain_to_split;
  double lambda_l1 = 0.0, max_delta_step = 0.0, path_smooth = 0.0;
  double left_output = 0.0, right_output = 0.0;                                          // outputs of the parent's split = parent_output of the children
  double* out10;            // [2][F][10]: candidates of the smaller, then of the larger child
  int* out_flags;           // [2][F + 1]
  // chunk partials of the smaller child's build (hist_build_kernel's layout): when part_grad != nullptr the workgroups sum the chunks of THEIR feature
  // themselves (integer totals, converted as hist_reduce_kernel does: the same bits) and workgroup (f, 0) writes the entries to `smaller` -- no
  // reduce launch between build and search (round 4)
  const long long* part_grad = nullptr; const long long* part_hess = nullptr; const uint32_t* part_cnt = nullptr;
  const unsigned long long* grad_max_bits = nullptr; const unsigned long long* hess_max_bits = nullptr;
  int fpad = 0, nchunks = 0, has_hess = 0; double const_hess = 1.0;
  int own_f0 = 0, own_f1 = 2147483647;   // feature-block exchange of the data-parallel grower: the features whose bins this rank holds (default: all)
  const signed char* is_cat = nullptr; CatCfg cat; unsigned* out_cat = nullptr;   // categorical features: flags [F], configuration, [2][F][8] bitsets over bins
  unsigned* ticket = nullptr;   // device word, zero between launches: workgroups that have finished
  int* host_seq = nullptr;      // pinned host word: receives seq from the last workgroup (nullptr: the host synchronises the stream instead)
  int seq = 0;
};

struct HistReduceArgs {
  const long long* part_grad; const long long* part_hess; const uint32_t* part_cnt;
  const unsigned long long* grad_max_bits; const unsigned long long* hess_max_bits;
  const int* bin_offsets;   // [F+1]
  double* hist_out;         // [total_bins][2]  {grad, hess}
  unsigned long long* cnt_out;  // [total_bins]
  int fpad, nchunks, num_features;
  double const_hess; int has_hess;
  long long* limbs_out = nullptr;  // sharded handles: [5][limb_stride] integer totals {grad hi, grad lo, count, hess hi, hess lo} (word-major) INSTEAD of hist_out / cnt_out
  int limb_stride = 0;             // = total bins
};

hipError_t launch_hist_build(const HistKernelArgs& a, hipStream_t st);
hipError_t launch_hist_reduce(const HistReduceArgs& a, hipStream_t st);
hipError_t launch_hist_convert(const long long* limbs, int total_bins, const unsigned long long* grad_max_bits, const unsigned long long* hess_max_bits,
                               double const_hess, int has_hess, double* hist_out, unsigned long long* cnt_out, hipStream_t st);
hipError_t launch_hist_limbs_pack(const long long* limbs, int total_bins, int nwords, const int* blk_bin0, const int* blk_bins, int world, int maxblk, long long* send,
                                  hipStream_t st);
hipError_t launch_hist_convert_block(const long long* recv, int maxblk, int nbins, int bin0, const unsigned long long* grad_max_bits, const unsigned long long* hess_max_bits,
                                     double const_hess, int has_hess, double* hist_out, hipStream_t st);
hipError_t launch_hist_root_sums(const long long* limbs, int total_bins, const int* bin_offsets, const unsigned long long* grad_max_bits,
                                 const unsigned long long* hess_max_bits, double const_hess, int has_hess, double* out3, hipStream_t st);
hipError_t launch_hist_absmax(const double* v, int n, unsigned long long* out_bits, hipStream_t st);
hipError_t launch_hist_fix(double* hist, int num_features, const int* view_offset, const int* num_bin, const int* most_freq_bin,
                           double sum_gradient, double sum_hessian, hipStream_t st);
hipError_t launch_hist_best_split(const double* hist, int num_features, const int* view_offset, const int* num_bin, const int* meta3,
                                  double sum_gradient, double sum_hessian, int num_data, double lambda_l2, int min_data_in_leaf,
                                  double min_sum_hessian, double min_gain_to_split, SplitReg reg, const signed char* is_feature_used, double* out10,
                                  int* out_default_left, int* best_feature, hipStream_t st, const signed char* is_cat = nullptr, CatCfg cat = CatCfg(),
                                  unsigned* out_cat = nullptr);
hipError_t launch_hist_partition(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                 int missing_type, int default_left, unsigned threshold, const int* data_indices, int cnt, int* blk_cnt,
                                 int* blk_off, int* lte, int* gt, hipStream_t st, const unsigned* cat_bits8 = nullptr);
hipError_t launch_hist_partition_segment(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                         int missing_type, int default_left, unsigned threshold, const int* src, int cnt, int* blk_cnt,
                                         int* blk_off, int* dst, int* counts, int* host_counts, hipStream_t st,
                                         unsigned long long* tags = nullptr, unsigned epoch = 0, int host_seq = 0, int* err = nullptr,
                                         bool* host_seq_written = nullptr, const unsigned* cat_bits8 = nullptr);
// (tags: 256 device granules, zero at allocation, epoch > 0 and different for every launch -> segments of at most 262 144 rows go through
//  ONE launch, hist_partition_onepass_kernel; host_seq != 0: host_counts[2] receives it after host_counts[0..1], see *host_seq_written)
hipError_t launch_hist_children_search(const ChildrenSearchArgs& a, hipStream_t st);
hipError_t launch_hist_label_rows(const int* rows0, const int* rows1, int n, const int* seg_begin, const int* seg_leaf, const int* seg_buf, int nseg,
                                  int* out, hipStream_t st);
hipError_t launch_hist_subtract(const double* parent, const double* smaller, double* out, int len, hipStream_t st);
hipError_t launch_bins_transpose(const uint8_t* bins_fm, uint8_t* bins_rm, int n, int F, int fpad, hipStream_t st, int row_stride = 0);   // row_stride > 0: compact rows of that many bytes

}  // namespace gpb
