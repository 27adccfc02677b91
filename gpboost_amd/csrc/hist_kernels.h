// gpboost_amd/csrc/hist_kernels.h -- launch interface of hist_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpb {

#define GPB_HIST_FG 16          // features per workgroup (one per lane of a 16-lane row)
#define GPB_HIST_MAX_BIN 256    // uint8 bins

struct HistKernelArgs {
  const uint8_t* bins_rm;   // [n][fpad] row-major bins, fpad = multiple of 16
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
};

struct HistReduceArgs {
  const long long* part_grad; const long long* part_hess; const uint32_t* part_cnt;
  const unsigned long long* grad_max_bits; const unsigned long long* hess_max_bits;
  const int* bin_offsets;   // [F+1]
  double* hist_out;         // [total_bins][2]  {grad, hess}
  unsigned long long* cnt_out;  // [total_bins]
  int fpad, nchunks, num_features;
  double const_hess; int has_hess;
};

hipError_t launch_hist_build(const HistKernelArgs& a, hipStream_t st);
hipError_t launch_hist_reduce(const HistReduceArgs& a, hipStream_t st);
hipError_t launch_hist_absmax(const double* v, int n, unsigned long long* out_bits, hipStream_t st);
hipError_t launch_hist_fix(double* hist, int num_features, const int* view_offset, const int* num_bin, const int* most_freq_bin,
                           double sum_gradient, double sum_hessian, hipStream_t st);
hipError_t launch_hist_best_split(const double* hist, int num_features, const int* view_offset, const int* num_bin, const int* meta3,
                                  double sum_gradient, double sum_hessian, int num_data, double lambda_l2, int min_data_in_leaf,
                                  double min_sum_hessian, double min_gain_to_split, const signed char* is_feature_used, double* out10,
                                  int* out_default_left, int* best_feature, hipStream_t st);
hipError_t launch_hist_partition(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                 int missing_type, int default_left, unsigned threshold, const int* data_indices, int cnt, int* blk_cnt,
                                 int* blk_off, int* lte, int* gt, hipStream_t st);
hipError_t launch_hist_label_rows(const int* rows, int n, const int* seg_begin, const int* seg_leaf, int nseg, int* out, hipStream_t st);
hipError_t launch_hist_subtract(const double* parent, const double* smaller, double* out, int len, hipStream_t st);
hipError_t launch_bins_transpose(const uint8_t* bins_fm, uint8_t* bins_rm, int n, int F, int fpad, hipStream_t st);

}  // namespace gpb
