/*!
 * integration/hip_tree_learner.h -- reference-side glue (route B of INTEGRATION.md), OUR file: it is copied next to
 * src/LightGBM/treelearner/tree_learner.cpp of a scratch copy of the reference by oracle/Makefile.routeB and compiled there.
 *
 * A SerialTreeLearner whose histogram construction (SerialTreeLearner::ConstructHistograms, serial_tree_learner.cpp:351-373 ->
 * Dataset::ConstructHistograms -> DenseBin::ConstructHistogramInner, dense_bin.hpp:98-141) runs on the MI355X through the
 * extern "C" shim of lib_gpboost_amd.so (include/gpb_hip.h), the way CUDATreeLearner overrides it for CUDA
 * (cuda_tree_learner.cpp:767).  Registered for device_type = "gpu" in TreeLearner::CreateTreeLearner (tree_learner.cpp:15-52) when the
 * reference is built with -DUSE_HIP_GP; Config::CheckParamConflict already forces col-wise (dense) bins for that device type
 * (config.cpp:349-355).  Feature groups are dense uint8 columns here (one feature or an EFB bundle, <= 256 bins); a data set with a
 * multi-value / sparse group keeps the reference's CPU histograms (never a silent wrong answer).
 */
#ifndef LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_
#define LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_

#ifdef USE_HIP_GP

#include <LightGBM/dataset.h>
#include <LightGBM/utils/log.h>

#include <gpb_hip.h>

#include <memory>
#include <vector>

#include "serial_tree_learner.h"

namespace LightGBM {

class HIPTreeLearner : public SerialTreeLearner {
 public:
  explicit HIPTreeLearner(const Config* tree_config) : SerialTreeLearner(tree_config) {}
  ~HIPTreeLearner() { if (hist_) gpb_hip_hist_free(hist_); }

  void Init(const Dataset* train_data, bool is_constant_hessian) override {
    SerialTreeLearner::Init(train_data, is_constant_hessian);
    CreateDeviceBins();
  }

  void ResetTrainingDataInner(const Dataset* train_data, bool is_constant_hessian, bool reset_multi_val_bin) override {
    SerialTreeLearner::ResetTrainingDataInner(train_data, is_constant_hessian, reset_multi_val_bin);
    CreateDeviceBins();
  }

 protected:
  void BeforeTrain() override {
    SerialTreeLearner::BeforeTrain();
    // one upload per tree: gradients (and hessians unless constant) in data order, as CUDATreeLearner::BeforeTrain does
    if (hist_ && gpb_hip_hist_set_gradients(hist_, gradients_, share_state_->is_constant_hessian ? nullptr : hessians_)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
  }

  void ConstructHistograms(const std::vector<int8_t>& is_feature_used, bool use_subtract) override {
    if (!hist_) {
      SerialTreeLearner::ConstructHistograms(is_feature_used, use_subtract);
      return;
    }
    BuildLeaf(smaller_leaf_splits_.get(), smaller_leaf_histogram_array_[0].RawData() - kHistOffset);
    if (larger_leaf_histogram_array_ != nullptr && !use_subtract) {
      BuildLeaf(larger_leaf_splits_.get(), larger_leaf_histogram_array_[0].RawData() - kHistOffset);
    }
  }

 private:
  void BuildLeaf(const LeafSplits* leaf, hist_t* out) {
    const data_size_t cnt = leaf->num_data_in_leaf();
    const data_size_t* idx = (cnt == num_data_) ? nullptr : leaf->data_indices();
    // (gradient sum, hessian sum) pairs per bin in group_bin_boundaries_ order: exactly the buffer Dataset::ConstructHistogramsInner fills
    if (gpb_hip_hist_build(hist_, idx, cnt, share_state_->is_constant_hessian ? static_cast<double>(hessians_[0]) : 1.0, out, nullptr)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
  }

  void CreateDeviceBins() {
    if (hist_) { gpb_hip_hist_free(hist_); hist_ = nullptr; }
    const int num_groups = train_data_->num_feature_groups();
    std::vector<int32_t> offsets(num_groups + 1);
    for (int g = 0; g < num_groups; ++g) {
      if (train_data_->IsMultiGroup(g) || train_data_->FeatureGroupNumBin(g) > 256) {
        Log::Warning("HIPTreeLearner: feature group %d is multi-valued or has more than 256 bins; histograms stay on the CPU.", g);
        return;
      }
      offsets[g] = static_cast<int32_t>(train_data_->GroupBinBoundary(g));
    }
    offsets[num_groups] = static_cast<int32_t>(train_data_->NumTotalBin());
    std::vector<uint8_t> bins(static_cast<size_t>(num_groups) * num_data_);   // feature(-group)-major, the layout of DenseBin storage
#pragma omp parallel for schedule(static)
    for (int g = 0; g < num_groups; ++g) {
      std::unique_ptr<BinIterator> it(train_data_->FeatureGroupIterator(g));
      it->Reset(0);
      uint8_t* col = bins.data() + static_cast<size_t>(g) * num_data_;
      for (data_size_t i = 0; i < num_data_; ++i) col[i] = static_cast<uint8_t>(it->RawGet(i));
    }
    if (gpb_hip_hist_create(num_data_, num_groups, bins.data(), offsets.data(), &hist_)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    Log::Info("HIPTreeLearner: %d feature groups x %d rows resident on the GPU (%d bins in total)", num_groups, num_data_, offsets[num_groups]);
  }

  gpb_hip_hist_t* hist_ = nullptr;
};

}  // namespace LightGBM

#endif  // USE_HIP_GP
#endif  // LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_
