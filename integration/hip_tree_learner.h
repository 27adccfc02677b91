/*!
 * integration/hip_tree_learner.h -- reference-side glue (route B of INTEGRATION.md), OUR file: it is copied next to
 * src/LightGBM/treelearner/tree_learner.cpp of a scratch copy of the reference by oracle/Makefile.routeB and compiled there.
 *
 * A SerialTreeLearner whose histogram construction (SerialTreeLearner::ConstructHistograms, serial_tree_learner.cpp:351-373 ->
 * Dataset::ConstructHistograms -> DenseBin::ConstructHistogramInner, dense_bin.hpp:98-141) runs on the MI355X through the
 * extern "C" shim of lib_gpboost_amd.so (include/gpb_hip.h), the way CUDATreeLearner overrides it for CUDA
 * (cuda_tree_learner.cpp:767).  Registered for device_type = "gpu" in TreeLearner::CreateTreeLearner (tree_learner.cpp:15-52) when the
 * reference is built with -DUSE_HIP_GP; Config::CheckParamConflict already forces col-wise (dense) bins for that device type
 * (config.cpp:349-355).  The resident columns are either the reference's feature groups (dense uint8 columns: one feature or an EFB bundle,
 * <= 256 bins) or -- round 5, whenever whole trees can be grown -- one column per FEATURE, so that bundles of any width, multi-value groups and
 * sparse bins are served as well (CreateDeviceBins).  A data set that fits neither keeps the reference's CPU histograms (never a silent wrong answer).
 *
 * Whole trees: when the configuration is the one gpb_hip_hist_grow_tree restates (numerical AND categorical features, every regularisation path of
 * the two searches, depth limit, column sampling by tree, bagging; no forced splits / constraints / extra_trees), Train() hands the whole leaf-wise
 * growth to the device (row lists resident, one synchronisation per split) and rebuilds the reference's own Tree object and
 * DataPartition from the returned arrays -- GBDT (shrinkage, score update, the GPBoost leaf update) carries on unchanged.  Anything
 * else: SerialTreeLearner::Train with the device histograms above.
 */
#ifndef LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_
#define LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_

#ifdef USE_HIP_GP

#include <LightGBM/dataset.h>
#include <LightGBM/utils/log.h>

#include <gpb_hip.h>

#include <algorithm>
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
    if (!keep_device_bins_) { CreateDeviceBins(); subset_bins_ = false; bag_is_subset_ = false; bag_rows_ = nullptr; bag_cnt_ = 0; }
  }

  void SetBaggingData(const Dataset* subset, const data_size_t* used_indices, data_size_t num_data) override {
    // Two forms of bagging (gbdt.cpp:925-945): on the full Dataset (subset == nullptr; bags above half of the rows per iteration) and on a copied
    // subset Dataset (small bags), whose rows are renumbered 0 .. num_data - 1 and whose gradients arrive compacted.  The device keeps the bins
    // of the FULL data in both: the grower starts from the bag's rows (gpb_hip_hist_set_root_rows); for a subset the compact gradients are
    // spread to full row positions (BeforeTrain) and the returned row labels are read at the bag's rows.  If whole trees cannot be grown on
    // the device for this configuration, a subset gets its own device bins as before (the reference's Train over device histograms).
    const bool unified = subset != nullptr && hist_ != nullptr && whole_tree_ok_ && !subset_bins_ && WholeTreeConfigOnly() && used_indices != nullptr &&
                         Ascending(used_indices, num_data) && used_indices[num_data - 1] < hist_rows_;
    keep_device_bins_ = unified;
    SerialTreeLearner::SetBaggingData(subset, used_indices, num_data);
    keep_device_bins_ = false;
    bag_rows_ = nullptr; bag_cnt_ = 0; bag_is_subset_ = false;
    bagging_ = false;
    if (subset != nullptr) {
      if (unified) { bag_rows_ = used_indices; bag_cnt_ = num_data; bag_is_subset_ = true; }
      else { bagging_ = true; subset_bins_ = true; }
    } else if (used_indices != nullptr && hist_ != nullptr && num_data < hist_rows_) {
      if (Ascending(used_indices, num_data)) { bag_rows_ = used_indices; bag_cnt_ = num_data; } else { bagging_ = true; }
    }
    bag_dirty_ = true;
  }

  Tree* Train(const score_t* gradients, const score_t* hessians, bool is_first_tree) override {
    if (bag_is_subset_ && !WholeTreeConfig()) {      // (the configuration changed under a subset bag: give the subset its own device bins)
      bag_is_subset_ = false; bag_rows_ = nullptr; bag_cnt_ = 0; bagging_ = true; subset_bins_ = true;
      CreateDeviceBins();
    }
    if (!hist_ || !whole_tree_ok_ || bagging_ || !WholeTreeConfig()) {
      // a device-grown tree left the partition with that tree's leaf count (ResetByLeafPred -> ResetLeaves(nl)); the reference's Train
      // indexes leaf_begin_ / leaf_count_ up to num_leaves (DataPartition::Split)
      data_partition_->ResetLeaves(config_->num_leaves);
      labels_tree_leaves_ = 0;                          // the partition below is the host learner's: leaf_of_row_ no longer describes it
      return SerialTreeLearner::Train(gradients, hessians, is_first_tree);
    }
    if (!announced_) { Log::Info("HIPTreeLearner: whole trees are grown on the GPU (gpb_hip_hist_grow_tree)"); announced_ = true; }
    gradients_ = gradients;
    hessians_ = hessians;
    BeforeTrain();                                   // root sums (LeafSplits::Init), data partition reset, gradient upload (override below)
    if (bag_dirty_) {                                // the bag changes every bagging_freq iterations: one upload then
      if (gpb_hip_hist_set_root_rows(hist_, bag_rows_, bag_cnt_)) Log::Fatal("%s", gpb_hip_get_last_error());
      bag_dirty_ = false;
    }
    const int L = config_->num_leaves;
    const bool const_hess = share_state_->is_constant_hessian;
    int32_t nl = 0;
    std::vector<int32_t> sf(L), dl(L), lc(L), rc(L), icnt(L), lcnt(L);
    // the row labels come back into ONE buffer that lives as long as the learner: the library page-locks a caller buffer it sees
    // repeatedly (hipHostRegister), so the per-tree download runs at the PCIe rate
    if (static_cast<data_size_t>(leaf_of_row_.size()) != hist_rows_) leaf_of_row_.assign(hist_rows_, 0);
    std::vector<int32_t>& leaf_of_row = leaf_of_row_;
    std::vector<uint32_t> thr(L);
    std::vector<double> gain(L), lval(L);
    if (gpb_hip_hist_set_regularisation(hist_, config_->lambda_l1, config_->max_delta_step > 0.0 ? config_->max_delta_step : 0.0,
                                        config_->path_smooth > 0.0 ? config_->path_smooth : 0.0, 0.0)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    if (gpb_hip_hist_set_max_depth(hist_, config_->max_depth)) Log::Fatal("%s", gpb_hip_get_last_error());
    if (has_categorical_) {                          // the Config fields FindBestThresholdCategoricalInner reads (feature_histogram.hpp:278-519)
      std::vector<int8_t> is_cat(train_data_->num_features());
      for (int f = 0; f < train_data_->num_features(); ++f) is_cat[f] = train_data_->FeatureBinMapper(f)->bin_type() == BinType::CategoricalBin ? 1 : 0;
      if (gpb_hip_hist_set_categorical(hist_, is_cat.data(), config_->max_cat_to_onehot, config_->max_cat_threshold, config_->cat_smooth, config_->cat_l2,
                                       config_->min_data_per_group)) {
        Log::Fatal("%s", gpb_hip_get_last_error());
      }
    }
    // feature_fraction: the columns col_sampler_ drew for this tree in BeforeTrain (serial_tree_learner.cpp:258)
    if (gpb_hip_hist_set_feature_mask(hist_, config_->feature_fraction < 1.0 ? col_sampler_.is_feature_used_bytree().data() : nullptr)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    if (gpb_hip_hist_grow_tree(hist_, L, smaller_leaf_splits_->sum_gradients(), smaller_leaf_splits_->sum_hessians(), config_->lambda_l2,
                               config_->min_data_in_leaf, config_->min_sum_hessian_in_leaf, config_->min_gain_to_split,
                               const_hess ? static_cast<double>(hessians_[0]) : 1.0, &nl, sf.data(), thr.data(), dl.data(), lc.data(), rc.data(),
                               gain.data(), icnt.data(), lval.data(), lcnt.data(), leaf_of_row.data())) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    std::vector<double> info(static_cast<size_t>(std::max(nl - 1, 1)) * 6);
    if (nl > 1 && gpb_hip_hist_last_tree_node_info(hist_, nl - 1, info.data())) Log::Fatal("%s", gpb_hip_get_last_error());
    std::vector<int32_t> node_cat(static_cast<size_t>(std::max(nl - 1, 1)), 0);
    std::vector<uint32_t> node_bits(static_cast<size_t>(std::max(nl - 1, 1)) * 8, 0u);
    if (nl > 1 && has_categorical_ && gpb_hip_hist_last_tree_cat_nodes(hist_, nl - 1, node_cat.data(), node_bits.data())) Log::Fatal("%s", gpb_hip_get_last_error());
    auto tree = std::unique_ptr<Tree>(new Tree(L, false, false));
    for (int k = 0; k + 1 < nl; ++k) {
      int leaf = lc[k];                              // the split leaf kept its id on the left: the leftmost leaf below node k
      while (leaf >= 0) leaf = lc[leaf];
      leaf = ~leaf;
      const int inner = sf[k];
      if (node_cat[k]) {
        // SerialTreeLearner::SplitInner, categorical branch (serial_tree_learner.cpp:612-646): the set over BINS as it came from the search, and the
        // same set over the real category values (BinMapper::BinToValue through Dataset::RealThreshold)
        std::vector<uint32_t> cat_bins;
        for (int b = 0; b < 256; ++b) if ((node_bits[8 * static_cast<size_t>(k) + (b >> 5)] >> (b & 31)) & 1u) cat_bins.push_back(static_cast<uint32_t>(b));
        std::vector<uint32_t> cat_bitset_inner = Common::ConstructBitset(cat_bins.data(), static_cast<int>(cat_bins.size()));
        std::vector<int> threshold_int(cat_bins.size());
        for (size_t c = 0; c < cat_bins.size(); ++c) threshold_int[c] = static_cast<int>(train_data_->RealThreshold(inner, cat_bins[c]));
        std::vector<uint32_t> cat_bitset = Common::ConstructBitset(threshold_int.data(), static_cast<int>(threshold_int.size()));
        tree->SplitCategorical(leaf, inner, train_data_->RealFeatureIndex(inner), cat_bitset_inner.data(), static_cast<int>(cat_bitset_inner.size()),
                               cat_bitset.data(), static_cast<int>(cat_bitset.size()), info[6 * k], info[6 * k + 1], static_cast<int>(info[6 * k + 2]),
                               static_cast<int>(info[6 * k + 3]), info[6 * k + 4], info[6 * k + 5], static_cast<float>(gain[k]),
                               train_data_->FeatureBinMapper(inner)->missing_type());
        continue;
      }
      tree->Split(leaf, inner, train_data_->RealFeatureIndex(inner), thr[k], train_data_->RealThreshold(inner, thr[k]), info[6 * k], info[6 * k + 1],
                  static_cast<int>(info[6 * k + 2]), static_cast<int>(info[6 * k + 3]), info[6 * k + 4], info[6 * k + 5], static_cast<float>(gain[k]),
                  train_data_->FeatureBinMapper(inner)->missing_type(), dl[k] != 0);
    }
    labels_tree_leaves_ = 0;
    if (bag_rows_ == nullptr) {                      // (ResetByLeafLabels: ResetByLeafPred as a parallel counting sort, data_partition.hpp seam)
      data_partition_->ResetByLeafLabels(leaf_of_row.data(), hist_rows_, nl);
      labels_tree_leaves_ = nl;                      // every row is in the tree and leaf_of_row_ IS the partition: the two O(n) walks below read it directly
    } else if (bag_is_subset_) {                     // the subset Dataset numbers its rows by position in the bag
      std::vector<int> pred(bag_cnt_);
      for (data_size_t k = 0; k < bag_cnt_; ++k) pred[k] = leaf_of_row[bag_rows_[k]];
      data_partition_->ResetByLeafLabels(pred.data(), bag_cnt_, nl);
    } else {
      // the partition of the BAG (what AddPredictionToScore and the leaf refits walk): ResetByLeafPred numbers positions, here positions in
      // the bag -> mapped to row indices in place
      std::vector<int> pred(bag_cnt_);
      for (data_size_t k = 0; k < bag_cnt_; ++k) pred[k] = leaf_of_row[bag_rows_[k]];
      data_partition_->ResetByLeafLabels(pred.data(), bag_cnt_, nl);
      data_size_t* idx = const_cast<data_size_t*>(data_partition_->indices());
      for (data_size_t k = 0; k < bag_cnt_; ++k) idx[k] = bag_rows_[idx[k]];
    }
    return tree.release();
  }

  // GBDT's two O(n) walks over the partition of the tree just grown (gbdt.cpp:470-476 Newton leaf values of the GPBoost algorithm, :606-611 UpdateScore): the base class
  // runs one thread per LEAF over that leaf's row list (serial_tree_learner.cpp:818-828, serial_tree_learner.h:98-113) -- a 31-leaf tree over 1e6 rows keeps most threads
  // idle behind the biggest leaf.  After a device-grown tree without bagging the row -> leaf labels of gpb_hip_hist_grow_tree are that partition, so both walks run
  // over ROWS instead: same values (every row receives its leaf's index / one addition of its leaf's output), sequential memory, all threads.
  void GetDataLeafIndices(Tree* tree, int* data_leaf_index) const override {
    if (labels_tree_leaves_ > 0 && tree->num_leaves() == labels_tree_leaves_ && static_cast<data_size_t>(leaf_of_row_.size()) == num_data_) {
#pragma omp parallel for schedule(static)
      for (data_size_t i = 0; i < num_data_; ++i) data_leaf_index[i] = leaf_of_row_[i];
      return;
    }
    SerialTreeLearner::GetDataLeafIndices(tree, data_leaf_index);
  }
  void AddPredictionToScore(const Tree* tree, double* out_score) const override {
    if (labels_tree_leaves_ > 1 && tree->num_leaves() == labels_tree_leaves_ && static_cast<data_size_t>(leaf_of_row_.size()) == num_data_) {
      std::vector<double> out(tree->num_leaves());
      for (int l = 0; l < tree->num_leaves(); ++l) out[l] = static_cast<double>(tree->LeafOutput(l));
#pragma omp parallel for schedule(static)
      for (data_size_t i = 0; i < num_data_; ++i) out_score[i] += out[leaf_of_row_[i]];
      return;
    }
    SerialTreeLearner::AddPredictionToScore(tree, out_score);
  }

 protected:
  void BeforeTrain() override {
    SerialTreeLearner::BeforeTrain();
    // one upload per tree: gradients (and hessians unless constant) in data order, as CUDATreeLearner::BeforeTrain does
    const score_t* g = gradients_;
    const score_t* hs = share_state_->is_constant_hessian ? nullptr : hessians_;
    if (hist_ && bag_is_subset_) {                   // compact gradients of a subset bag -> the rows of the full data the device bins belong to
      full_grad_.assign(hist_rows_, 0.0f);
      for (data_size_t k = 0; k < bag_cnt_; ++k) full_grad_[bag_rows_[k]] = gradients_[k];
      g = full_grad_.data();
      if (hs != nullptr) {
        full_hess_.assign(hist_rows_, 0.0f);
        for (data_size_t k = 0; k < bag_cnt_; ++k) full_hess_[bag_rows_[k]] = hessians_[k];
        hs = full_hess_.data();
      }
    }
    else if (hist_ && hist_rows_ >= (1 << 17)) {
      // staging buffers OWNED by this learner, page-locked once (gpb_hip_hist_register_host_buffers): the Booster's gradient vectors are destroyed
      // BEFORE its tree learner (gbdt.h member order), so they must not stay registered; a parallel copy into the staging buffer + an upload at the
      // PCIe rate beats the pageable copy of the Booster's own buffer (8 MB at n = 1e6: ~0.6 ms against ~2.5 ms)
      full_grad_.resize(hist_rows_);
#pragma omp parallel for schedule(static)
      for (data_size_t k = 0; k < hist_rows_; ++k) full_grad_[k] = gradients_[k];
      g = full_grad_.data();
      if (hs != nullptr) {
        full_hess_.resize(hist_rows_);
#pragma omp parallel for schedule(static)
        for (data_size_t k = 0; k < hist_rows_; ++k) full_hess_[k] = hessians_[k];
        hs = full_hess_.data();
      }
    }
    if (hist_ && (g == full_grad_.data()) && (g != registered_grad_ || hs != registered_hess_)) {
      if (gpb_hip_hist_register_host_buffers(hist_, g, hs, nullptr)) Log::Fatal("%s", gpb_hip_get_last_error());
      registered_grad_ = g; registered_hess_ = hs;
    }
    if (hist_ && gpb_hip_hist_set_gradients(hist_, g, hs)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
  }

  void ConstructHistograms(const std::vector<int8_t>& is_feature_used, bool use_subtract) override {
    if (!hist_ || !host_layout_) {                      // (per-feature columns serve whole trees only: their histogram is not laid out as the host's)
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
    if (hist_) { gpb_hip_hist_free(hist_); hist_ = nullptr; registered_grad_ = registered_hess_ = nullptr; }   // (the free unregisters the staging buffers)
    whole_tree_ok_ = false; host_layout_ = false;
    const int num_groups = train_data_->num_feature_groups();
    const int F = train_data_->num_features();
    // Two layouts of the resident columns.
    //  * GROUP columns (the reference's own: one uint8 column per feature group, its histogram in group_bin_boundaries_ order): possible when no group is
    //    multi-valued or wider than 256 bins; the per-leaf histograms can then be handed back to the reference's host search (ConstructHistograms below).
    //  * FEATURE columns (round 5): one column per feature in the layout of a single-feature group (0 = most frequent bin, else the bin, shifted by one
    //    when the most frequent bin is not bin 0 -- FeatureGroup::PushData, feature_group.h:199-213), whatever the reference keeps the feature in: an EFB
    //    bundle of any width, a multi-value group, a sparse bin (BinIterator::Get returns the feature's own bin for all of them).  Whole trees are grown
    //    from these; the histogram never leaves the device, so its layout need not be the host's.
    bool groups_ok = true, all_single = num_groups == F;
    for (int g = 0; g < num_groups; ++g) {
      if (train_data_->IsMultiGroup(g) || train_data_->FeatureGroupNumBin(g) > 256) groups_ok = false;
    }
    for (int f = 0; f < F && all_single; ++f) if (train_data_->Feature2Group(f) != f) all_single = false;
    bool features_ok = true;
    for (int f = 0; f < F; ++f) {
      const BinMapper* bm = train_data_->FeatureBinMapper(f);
      if (bm->num_bin() + (bm->GetMostFreqBin() == 0 ? 0 : 1) > 256) features_ok = false;
    }
    const bool feature_columns = features_ok && !(groups_ok && all_single) && WholeTreeConfigOnly();
    if (!feature_columns && !groups_ok) {
      Log::Warning("HIPTreeLearner: a feature group is multi-valued or has more than 256 bins and the configuration is not one whole trees are grown for on the "
                   "GPU%s; histograms stay on the CPU.", features_ok ? "" : " (a feature has more than 255 bins)");
      return;
    }
    const int ncol = feature_columns ? F : num_groups;
    std::vector<int32_t> offsets(ncol + 1);
    std::vector<uint8_t> bins(static_cast<size_t>(ncol) * num_data_);   // column-major, the layout of DenseBin storage
    if (feature_columns) {
      offsets[0] = 0;
      for (int f = 0; f < F; ++f) {
        const BinMapper* bm = train_data_->FeatureBinMapper(f);
        offsets[f + 1] = offsets[f] + bm->num_bin() + (bm->GetMostFreqBin() == 0 ? 0 : 1);
      }
#pragma omp parallel for schedule(static)
      for (int f = 0; f < F; ++f) {
        const int mfb = static_cast<int>(train_data_->FeatureBinMapper(f)->GetMostFreqBin());
        std::unique_ptr<BinIterator> it(train_data_->FeatureIterator(f));
        it->Reset(0);
        uint8_t* col = bins.data() + static_cast<size_t>(f) * num_data_;
        for (data_size_t i = 0; i < num_data_; ++i) {
          const int b = static_cast<int>(it->Get(i));
          col[i] = static_cast<uint8_t>(b == mfb ? 0 : (mfb == 0 ? b : b + 1));
        }
      }
    } else {
      for (int g = 0; g < num_groups; ++g) offsets[g] = static_cast<int32_t>(train_data_->GroupBinBoundary(g));
      offsets[num_groups] = static_cast<int32_t>(train_data_->NumTotalBin());
#pragma omp parallel for schedule(static)
      for (int g = 0; g < num_groups; ++g) {
        std::unique_ptr<BinIterator> it(train_data_->FeatureGroupIterator(g));
        it->Reset(0);
        uint8_t* col = bins.data() + static_cast<size_t>(g) * num_data_;
        for (data_size_t i = 0; i < num_data_; ++i) col[i] = static_cast<uint8_t>(it->RawGet(i));
      }
    }
    if (gpb_hip_hist_create(num_data_, ncol, bins.data(), offsets.data(), &hist_)) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    hist_rows_ = num_data_;
    bag_dirty_ = true;
    host_layout_ = !feature_columns;
    // whole-tree growth on the device: one column per feature (numerical or categorical); the features' histogram views and FeatureMetainfo
    // (HistogramPool::SetFeatureInfo, feature_histogram.hpp:1146-1182; view = one bin past the start of the column, train_share_states.cpp:296-300)
    whole_tree_ok_ = ncol == F;
    std::vector<int32_t> voff(F), nbin(F), mfb(F), off(F), dbin(F), miss(F);
    std::vector<int8_t> is_cat(F, 0);
    bool any_cat = false;
    for (int f = 0; f < F && whole_tree_ok_; ++f) {
      const BinMapper* bm = train_data_->FeatureBinMapper(f);
      if (!feature_columns && train_data_->Feature2Group(f) != f) { whole_tree_ok_ = false; break; }
      voff[f] = offsets[f] + 1;
      nbin[f] = bm->num_bin(); mfb[f] = static_cast<int32_t>(bm->GetMostFreqBin());
      off[f] = mfb[f] == 0 ? 1 : 0; dbin[f] = static_cast<int32_t>(bm->GetDefaultBin()); miss[f] = static_cast<int32_t>(bm->missing_type());
      is_cat[f] = bm->bin_type() == BinType::CategoricalBin ? 1 : 0;
      any_cat = any_cat || is_cat[f];
    }
    if (whole_tree_ok_ && (gpb_hip_hist_set_fix_info(hist_, voff.data(), nbin.data(), mfb.data()) ||
                           gpb_hip_hist_set_split_info(hist_, off.data(), dbin.data(), miss.data()) ||
                           gpb_hip_hist_pool_resize(hist_, config_->num_leaves + 1))) {
      Log::Fatal("%s", gpb_hip_get_last_error());
    }
    has_categorical_ = whole_tree_ok_ && any_cat;
    Log::Info("HIPTreeLearner: %d %s columns x %d rows resident on the GPU (%d bins in total%s)", ncol, feature_columns ? "per-feature (unbundled)" : "feature-group",
              num_data_, offsets[ncol], has_categorical_ ? "; categorical features searched on the GPU" : "");
  }

  // the configuration gpb_hip_hist_grow_tree restates: every regularisation path of the numerical threshold search (lambda_l1, lambda_l2,
  // max_delta_step, path_smooth), max_depth and the per-tree column sample; nothing that changes the candidate set per node
  static bool Ascending(const data_size_t* v, data_size_t cnt) {
    for (data_size_t i = 1; i < cnt; ++i) if (v[i] <= v[i - 1]) return false;
    return cnt > 0;
  }
  bool WholeTreeConfig() const { return WholeTreeConfigOnly() && num_data_ == train_data_->num_data(); }
  bool WholeTreeConfigOnly() const {
    return config_->num_leaves >= 2 && config_->lambda_l1 >= 0.0 &&
           !config_->extra_trees && !config_->linear_tree &&
           config_->feature_fraction_bynode >= 1.0 && config_->monotone_constraints.empty() && config_->interaction_constraints_vector.empty() &&
           config_->feature_contri.empty() &&         // per-feature gain penalty (feature_histogram.hpp:94: gain *= meta_->penalty) is not restated
           (forced_split_json_ == nullptr || forced_split_json_->is_null()) && cegb_ == nullptr;
  }

  gpb_hip_hist_t* hist_ = nullptr;
  bool whole_tree_ok_ = false, bagging_ = false, announced_ = false, bag_dirty_ = false;
  bool host_layout_ = false;                           // the device columns are the reference's feature groups: per-leaf histograms can go back to the host search
  bool has_categorical_ = false;                       // some feature is categorical (searched by gpb_hip_hist_set_categorical's configuration)
  bool keep_device_bins_ = false;                      // inside SetBaggingData: the base class switches to the subset Dataset, the device bins stay
  bool bag_is_subset_ = false;                         // the bag is a copied subset Dataset (rows renumbered) over full-data device bins
  bool subset_bins_ = false;                           // the device bins were built from a subset Dataset (older path)
  data_size_t hist_rows_ = 0;                          // rows of the Dataset the device bins were built from
  std::vector<score_t> full_grad_, full_hess_;         // learner-owned staging buffers (page-locked once: gpb_hip_hist_register_host_buffers)
  const score_t* registered_grad_ = nullptr; const score_t* registered_hess_ = nullptr;
  std::vector<int32_t> leaf_of_row_;                   // row -> leaf of the last device-grown tree
  int labels_tree_leaves_ = 0;                         // > 0: leaf_of_row_ equals the data partition of a tree with this many leaves (device-grown, no bagging)
  const data_size_t* bag_rows_ = nullptr;              // the current bag (GBDT's bag_data_indices_, alive until the next SetBaggingData)
  data_size_t bag_cnt_ = 0;
};

}  // namespace LightGBM

#endif  // USE_HIP_GP
#endif  // LIGHTGBM_TREELEARNER_HIP_TREE_LEARNER_H_
