/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" window onto the *unmodified reference* (headers included from
 * /root/reference, linked against oracle/_ref/lib_gpboost_ref.so) that exposes the
 * intermediate quantities the reference's public C API does not: the Vecchia
 * ordering and neighbour lists, B = I - A and D^-1, y_aux, and the covariance
 * parameter gradient of CalcGradPars.  oracle/make_golden.py uses it to produce the
 * fixtures in tests/golden/; the parity tests use it (when oracle/_ref is present)
 * to check gpb_oracle.c and the HIP path against the reference itself.
 *
 * Private members are reached by re-declaring the access specifiers for this
 * translation unit only; no reference source is copied or modified.
 */
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <memory>
#define private public
#define protected public
#include <GPBoost/re_model.h>
#include <LightGBM/c_api.h>
#include <LightGBM/dataset.h>
#include <LightGBM/feature_group.h>
#include <LightGBM/train_share_states.h>
#include <LightGBM/config.h>
#include <LightGBM/tree.h>
#include <LightGBM/tree_learner.h>
#include <omp.h>
#include <LightGBM/treelearner/feature_histogram.hpp>   /* src/LightGBM/treelearner (Makefile.ref adds -I$(REF)/src) */
#undef private
#undef protected

using namespace GPBoost;

extern "C" {

__attribute__((visibility("default")))
void* refdrv_create(int n, const double* coords_colmajor, int d, const char* cov_fct, double shape, int m,
                    const char* ordering, int seed, const char* likelihood, int num_threads) {
  try {
    auto* mdl = new REModel(n, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 1, coords_colmajor, d, nullptr, 0,
                            cov_fct, shape, m > 0 ? "vecchia" : "none", 1., 0., m > 0 ? m : 20, ordering, 500, 1., "kmeans++",
                            likelihood, 1., "cholesky", seed, num_threads, false, false, nullptr, 1.);
    return mdl;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_create: %s\n", e.what());
    return nullptr;
  }
}

__attribute__((visibility("default")))
void refdrv_free(void* h) { delete reinterpret_cast<REModel*>(h); }

/* Vecchia order: perm[k] = original index of the k-th point of the ordering */
__attribute__((visibility("default")))
int refdrv_get_perm(void* h, int* perm) {
  auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
  const auto& idx = t->data_indices_per_cluster_[t->unique_clusters_[0]];
  for (size_t i = 0; i < idx.size(); ++i) perm[i] = idx[i];
  return (int)idx.size();
}

__attribute__((visibility("default")))
int refdrv_get_neighbors(void* h, int m, int* nn) {
  auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
  const auto& v = t->nearest_neighbors_[t->unique_clusters_[0]][0];
  for (size_t i = 0; i < v.size(); ++i)
    for (int j = 0; j < m; ++j) nn[i * m + j] = (j < (int)v[i].size()) ? v[i][j] : -1;
  return (int)v.size();
}

/* nll + gradient wrt log of the transformed parameters (sigma2, sigma1_2/sigma2, a), exactly the
 * sequence of the L-BFGS functor (include/GPBoost/optim_utils.h:299-338). */
__attribute__((visibility("default")))
int refdrv_nll_grad(void* h, const double* y, const double* cov_pars_orig, double* nll, double* grad3,
                    double* cov_pars_trans_out) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    t->SetY(y);
    vec_t orig = Eigen::Map<const vec_t>(cov_pars_orig, 3), trafo(3);
    t->TransformCovPars(orig, trafo);
    for (int k = 0; k < 3; ++k) cov_pars_trans_out[k] = trafo[k];
    if ((int)t->estimate_cov_par_index_.size() != 3) t->estimate_cov_par_index_ = std::vector<int>(3, 1);
    t->CalcCovFactorOrModeAndNegLL(trafo, nullptr);
    *nll = t->GetNegLogLikelihood();
    vec_t grad_cov, grad_beta;
    t->CalcGradPars(trafo, trafo[0], true, false, grad_cov, grad_beta, true, false, nullptr, false);
    for (int k = 0; k < 3; ++k) grad3[k] = grad_cov[k];
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_nll_grad: %s\n", e.what());
    return -1;
  }
}

/* after refdrv_nll_grad: A (n x m, aligned with the neighbour table), D^-1 diagonal, y_aux in Vecchia order */
__attribute__((visibility("default")))
int refdrv_get_factor(void* h, int m, double* A, double* Dinv, double* yaux) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    const int c0 = t->unique_clusters_[0];
    const sp_mat_t& B = t->B_[c0][0];
    const sp_mat_t& Di = t->D_inv_[c0][0];
    const auto& nn = t->nearest_neighbors_[c0][0];
    const int n = (int)nn.size();
    for (int i = 0; i < n; ++i) {
      Dinv[i] = Di.coeff(i, i);
      for (int j = 0; j < m; ++j) A[(size_t)i * m + j] = (j < (int)nn[i].size()) ? -B.coeff(i, nn[i][j]) : 0.;
    }
    if (yaux) {
      t->CalcYAux(1., false);
      const vec_t& ya = t->y_aux_[c0];
      for (int i = 0; i < n; ++i) yaux[i] = ya[i];
    }
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_get_factor: %s\n", e.what());
    return -1;
  }
}

/* after refdrv_nll_grad (factor at the given parameters, y = F - y set): the reference's own
 * REModelTemplate::NewtonUpdateLeafValues (include/GPBoost/re_model_template.h:4982-5063, Vecchia branch :5002-5008),
 * preceded by CalcYAux exactly as CalcGradientF does it (:3313-3316).  leaf index per DATA point. */
__attribute__((visibility("default")))
int refdrv_newton_leaf(void* h, const int* data_leaf_index, int num_leaves, double marg_variance, double* leaf_values) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    t->CalcYAux(marg_variance, false);
    t->y_aux_has_been_calculated_ = true;
    t->NewtonUpdateLeafValues(data_leaf_index, num_leaves, leaf_values, marg_variance);
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_newton_leaf: %s\n", e.what());
    return -1;
  }
}

/* Histogram of one leaf built by the reference itself: LGBM_DatasetCreateFromMat (the reference's own binning, no
 * feature bundling, col-wise) followed by Dataset::ConstructHistograms (include/LightGBM/dataset.h:471-500 ->
 * src/LightGBM/io/dataset.cpp:1143-1245).  Returns the reference's STORED group bins (what DenseBin holds, bin 0 =
 * most-frequent-bin placeholder) and the raw histogram (grad, hess) pairs laid out by group_bin_boundaries_. */
__attribute__((visibility("default")))
int refdrv_hist(int n, int F, const double* X_rowmajor, int max_bin, const int* data_indices, int num_data,
                const double* grad, const double* hess, double const_hess, int* num_groups_out, int* group_num_bin,
                unsigned char* bins_out, double* hist_out, int* feat_view_offset, int* feat_num_bin, int* feat_most_freq_bin,
                double* sums2, double* hist_fixed_out, const char* extra_params, const double* split_cfg4, int* feat_meta3,
                double* split_out10, int* split_default_left, int num_part, const int* part_ftd3, int* part_lte_out,
                int* part_lte_count, int* feat_is_cat, unsigned* split_cat_bits8, const unsigned* part_cat_bits8) {
  try {
    using namespace LightGBM;
    char params[256];
    snprintf(params, sizeof(params), "max_bin=%d min_data_in_bin=1 enable_bundle=false force_col_wise=true verbosity=-1 num_threads=4 %s", max_bin,
             extra_params ? extra_params : "");
    DatasetHandle dh = nullptr;
    if (LGBM_DatasetCreateFromMat(X_rowmajor, C_API_DTYPE_FLOAT64, n, F, 1, params, nullptr, &dh) != 0) {
      fprintf(stderr, "refdrv_hist: %s\n", LGBM_GetLastError());
      return -1;
    }
    Dataset* ds = reinterpret_cast<Dataset*>(dh);
    const int ng = ds->num_groups_;
    *num_groups_out = ng;
    for (int g = 0; g < ng; ++g) {
      group_num_bin[g] = ds->feature_groups_[g]->num_total_bin_;
      std::unique_ptr<BinIterator> it(ds->feature_groups_[g]->bin_data_->GetIterator(0, group_num_bin[g] - 1, 0));
      it->Reset(0);
      for (int i = 0; i < n; ++i) bins_out[(size_t)g * n + i] = (unsigned char)it->RawGet(i);
    }
    std::vector<score_t> g_all(grad, grad + n), h_all(n, const_hess), og(n), oh(n);
    if (hess) std::copy(hess, hess + n, h_all.begin());
    std::vector<int8_t> used(ds->num_features(), 1);
    std::unique_ptr<TrainingShareStates> share(ds->GetShareStates(og.data(), oh.data(), used, hess == nullptr, true, false));
    std::vector<hist_t> hist((size_t)ds->NumTotalBin() * 2 + 16, 0.);
    ds->ConstructHistograms(used, data_indices, num_data, g_all.data(), h_all.data(), og.data(), oh.data(), share.get(), hist.data());
    std::copy(hist.begin(), hist.begin() + (size_t)ds->NumTotalBin() * 2, hist_out);
    /* Dataset::FixHistogram (src/LightGBM/io/dataset.cpp:1272-1290) on every feature, called the way
     * SerialTreeLearner::FindBestSplitsFromHistograms does (serial_tree_learner.cpp:400-403): data = the feature's view of the
     * leaf histogram (one bin past the start of its group: the group's bin 0 is the most-frequent-bin placeholder), sums = the
     * leaf's gradient / hessian totals */
    if (hist_fixed_out) {
      double sg = 0., sh = 0.;
      for (int k = 0; k < num_data; ++k) { const int i = data_indices ? data_indices[k] : k; sg += g_all[i]; sh += h_all[i]; }
      sums2[0] = sg; sums2[1] = sh;
      for (int f = 0; f < ds->num_features(); ++f) {
        const int g = ds->feature2group_[f];
        const BinMapper* bm = ds->FeatureBinMapper(f);
        const int off = (int)ds->group_bin_boundaries_[g] + 1;   /* FeatureHistogram::data_ (train_share_states.cpp:296-300, feature_group.h bin_offsets_[0] = 1) */
        feat_view_offset[f] = off; feat_num_bin[f] = bm->num_bin(); feat_most_freq_bin[f] = (int)bm->GetMostFreqBin();
        ds->FixHistogram(f, sg, sh, hist.data() + (size_t)off * 2);
      }
      std::copy(hist.begin(), hist.begin() + (size_t)ds->NumTotalBin() * 2, hist_fixed_out);
      /* FeatureHistogram::FindBestThreshold of every feature on the fixed histogram, exactly as
       * SerialTreeLearner::ComputeBestSplitForFeature calls it (serial_tree_learner.cpp:736-740); feature metas from the
       * reference's own HistogramPool::SetFeatureInfo (feature_histogram.hpp:1146-1182).
       * split_cfg4 = { lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split, lambda_l1, max_delta_step, path_smooth,
       *                parent_output } (8 doubles; parent_output NaN: the root's, as GetParentOutput computes it) */
      if (split_out10) {
        Config config;
        config.lambda_l2 = split_cfg4[0]; config.min_data_in_leaf = (int)split_cfg4[1];
        config.min_sum_hessian_in_leaf = split_cfg4[2]; config.min_gain_to_split = split_cfg4[3];
        config.lambda_l1 = split_cfg4[4]; config.max_delta_step = split_cfg4[5]; config.path_smooth = split_cfg4[6];
        /* round 5: [8..12] = max_cat_to_onehot, max_cat_threshold, cat_smooth, cat_l2, min_data_per_group (NaN: the reference's defaults) */
        if (!std::isnan(split_cfg4[8])) config.max_cat_to_onehot = (int)split_cfg4[8];
        if (!std::isnan(split_cfg4[9])) config.max_cat_threshold = (int)split_cfg4[9];
        if (!std::isnan(split_cfg4[10])) config.cat_smooth = split_cfg4[10];
        if (!std::isnan(split_cfg4[11])) config.cat_l2 = split_cfg4[11];
        if (!std::isnan(split_cfg4[12])) config.min_data_per_group = (int)split_cfg4[12];
        std::vector<FeatureMetainfo> metas;
        HistogramPool::SetFeatureInfo<true, true>(ds, &config, &metas);
        /* root leaf: parent_output as GetParentOutput computes it (serial_tree_learner.cpp:758-770); used by path smoothing only */
        const double parent_output = std::isnan(split_cfg4[7])
            ? FeatureHistogram::CalculateSplittedLeafOutput<true, true, true, false>(sg, sh, config.lambda_l1, config.lambda_l2, config.max_delta_step,
                                                                                     BasicConstraint(), config.path_smooth, (data_size_t)num_data, 0)
            : split_cfg4[7];
        for (int f = 0; f < ds->num_features(); ++f) {
          feat_meta3[3 * f] = metas[f].offset; feat_meta3[3 * f + 1] = (int)metas[f].default_bin; feat_meta3[3 * f + 2] = (int)metas[f].missing_type;
          FeatureHistogram fh;
          fh.Init(hist.data() + (size_t)feat_view_offset[f] * 2, &metas[f]);
          SplitInfo si;
          BasicConstraintEntry no_constraint;            /* (the categorical search reads constraints->LeftToBasicConstraint() even without monotone constraints) */
          fh.FindBestThreshold(sg, sh, num_data, &no_constraint, parent_output, &si);
          double* r = split_out10 + (size_t)f * 10;
          r[0] = si.gain; r[1] = (double)si.threshold; r[2] = si.left_count; r[3] = si.right_count; r[4] = si.left_output;
          r[5] = si.right_output; r[6] = si.left_sum_gradient; r[7] = si.left_sum_hessian; r[8] = si.right_sum_gradient; r[9] = si.right_sum_hessian;
          split_default_left[f] = si.default_left ? 1 : 0;
          /* categorical feature (round 5): cat_threshold = the BINS going left (feature_histogram.hpp:495-514); threshold column = their number */
          if (feat_is_cat) {
            feat_is_cat[f] = ds->FeatureBinMapper(f)->bin_type() == BinType::CategoricalBin ? 1 : 0;
            for (int w = 0; w < 8; ++w) split_cat_bits8[8 * f + w] = 0u;
            if (feat_is_cat[f] && si.gain > kMinScore) {
              r[1] = (double)si.num_cat_threshold;
              for (int c = 0; c < si.num_cat_threshold; ++c) split_cat_bits8[8 * f + (si.cat_threshold[c] >> 5)] |= 1u << (si.cat_threshold[c] & 31);
            }
          }
        }
      }
    }
    /* Dataset::Split (include/LightGBM/dataset.h:506-516 -> FeatureGroup::Split, feature_group.h:345-376 -> DenseBin::Split /
     * SplitInner, src/LightGBM/io/dense_bin.hpp:176-282) of the leaf's rows for num_part (feature, threshold, default_left)
     * triples; the lte list of request p goes to part_lte_out + p * num_data (gt = the remaining rows in their original order) */
    for (int p = 0; p < num_part; ++p) {
      std::vector<data_size_t> idx(num_data), lte(num_data), gt(num_data);
      for (int k = 0; k < num_data; ++k) idx[k] = data_indices ? data_indices[k] : k;
      const uint32_t th = (uint32_t)part_ftd3[3 * p + 1];
      const bool cat = part_cat_bits8 && ds->FeatureBinMapper(part_ftd3[3 * p])->bin_type() == BinType::CategoricalBin;
      /* categorical request: the bitset over bins (SerialTreeLearner::SplitInner hands Common::ConstructBitset(cat_threshold), serial_tree_learner.cpp:617-640) */
      const data_size_t nl = cat ? ds->Split(part_ftd3[3 * p], part_cat_bits8 + 8 * (size_t)p, 8, part_ftd3[3 * p + 2] != 0, idx.data(), num_data, lte.data(), gt.data())
                                 : ds->Split(part_ftd3[3 * p], &th, 1, part_ftd3[3 * p + 2] != 0, idx.data(), num_data, lte.data(), gt.data());
      part_lte_count[p] = nl;
      std::copy(lte.begin(), lte.begin() + nl, part_lte_out + (size_t)p * num_data);
    }
    LGBM_DatasetFree(dh);
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_hist: %s\n", e.what());
    return -1;
  }
}

/* One tree grown by the reference's own SerialTreeLearner (src/LightGBM/treelearner/serial_tree_learner.cpp:159-210) on its own
 * Dataset, for given gradients / hessians (hess == NULL: constant hessian 1, the GPBoost Gaussian case).  One OpenMP thread, so
 * that the root sums (leaf_splits.hpp:73-86) are a plain left-to-right summation.  params: LightGBM parameter string (num_leaves,
 * lambda_l2, min_data_in_leaf, ...).  Outputs sized num_leaves: per internal node k < num_leaves - 1: split_feature_inner,
 * threshold_in_bin, default_left, left_child, right_child, split_gain, internal_count; per leaf: leaf_value, leaf_count.
 * Also returns the stored group bins and the feature metas the device side needs (as refdrv_hist does). */
__attribute__((visibility("default")))
int refdrv_train_tree(int n, int F, const double* X_rowmajor, const char* params, const double* grad, const double* hess,
                      int* num_groups_out, int* group_num_bin, unsigned char* bins_out, int* feat_view_offset, int* feat_num_bin,
                      int* feat_most_freq_bin, int* feat_meta3, int* num_leaves_out, int* split_feature_inner, int* threshold_in_bin,
                      int* default_left, int* left_child, int* right_child, double* split_gain, int* internal_count,
                      double* leaf_value, int* leaf_count, int* feat_layout4, int* node_is_cat, unsigned* node_cat_bits8, int max_columns) {
  try {
    using namespace LightGBM;
    omp_set_num_threads(1);
    DatasetHandle dh = nullptr;
    if (LGBM_DatasetCreateFromMat(X_rowmajor, C_API_DTYPE_FLOAT64, n, F, 1, params, nullptr, &dh) != 0) {
      fprintf(stderr, "refdrv_train_tree: %s\n", LGBM_GetLastError());
      return -1;
    }
    Dataset* ds = reinterpret_cast<Dataset*>(dh);
    const int ng = ds->num_groups_;
    /* COLUMNS of stored bins (what a device keeps resident): a group whose features share ONE Bin (single-feature groups, EFB bundles:
     * feature_group.h:50-76, bin_offsets_) is one column; a multi-value group keeps one Bin per feature (multi_bin_data_, :484-496, stored
     * like a single-feature group: 0 = most frequent bin, PushData :199-213) and gives one column per feature.
     * feat_layout4 (round 5) per feature: {column, min_bin, max_bin of the feature inside the column (FeatureGroup::Split :353-354, :382-383),
     * bin type (0 numerical, 1 categorical)}; feat_view_offset = first histogram entry of FeatureHistogram::data_ in a histogram laid out
     * column after column (= group_bin_boundaries_ + bin_offsets_[sub] when no group is multi-valued, train_share_states.cpp:296-300). */
    int ncol = 0;
    std::vector<int> col_base;               /* first histogram entry of every column */
    std::vector<std::vector<int>> col_of(ng);
    int base = 0;
    /* max_columns < 0: UNBUNDLED columns -- one column per feature in the layout of a single-feature group (0 = most frequent bin, else the bin,
     * shifted by one when the most frequent bin is not bin 0: PushData :199-213 with bin_offsets_[0] = 1), whatever Bin the reference keeps the
     * feature in (dense, sparse, a bundle of any width, a multi-value group): BinIterator::Get returns the feature's own bin for all of them. */
    const bool unbundle = max_columns < 0;
    if (unbundle) {
      int nmv = 0, widest = 0;
      for (int g = 0; g < ng; ++g) { nmv += ds->feature_groups_[g]->is_multi_val_ ? 1 : 0; widest = std::max(widest, ds->feature_groups_[g]->num_total_bin_); }
      fprintf(stderr, "refdrv_train_tree: %d features in %d feature groups (%d multi-value), widest group %d bins\n", ds->num_features(), ng, nmv, widest);
      for (int f = 0; f < ds->num_features(); ++f) {
        const BinMapper* bm = ds->FeatureBinMapper(f);
        const int mfb = (int)bm->GetMostFreqBin();
        const int nb = bm->num_bin() + (mfb == 0 ? 0 : 1);
        if (nb > 256) { fprintf(stderr, "refdrv_train_tree: a feature with %d stored bins\n", nb); return -1; }
        std::unique_ptr<BinIterator> it(ds->FeatureIterator(f));
        it->Reset(0);
        for (int i = 0; i < n; ++i) {
          const int b = (int)it->Get(i);
          bins_out[(size_t)f * n + i] = (unsigned char)(b == mfb ? 0 : (mfb == 0 ? b : b + 1));
        }
        group_num_bin[f] = nb;
        col_base.push_back(base);
        base += nb;
      }
      ncol = ds->num_features();
    }
    for (int g = 0; g < ng && !unbundle; ++g) {
      FeatureGroup* fg = ds->feature_groups_[g].get();
      const int nsub = fg->is_multi_val_ ? fg->num_feature_ : 1;
      for (int sidx = 0; sidx < nsub; ++sidx) {
        if (max_columns > 0 && ncol >= max_columns) { fprintf(stderr, "refdrv_train_tree: more than %d columns\n", max_columns); return -1; }
        int nb;
        std::unique_ptr<BinIterator> it;
        if (fg->is_multi_val_) {
          const int addi = fg->bin_mappers_[sidx]->GetMostFreqBin() == 0 ? 0 : 1;
          nb = fg->bin_mappers_[sidx]->num_bin() + addi;
          it.reset(fg->multi_bin_data_[sidx]->GetIterator(0, nb - 1, 0));
        } else {
          nb = fg->num_total_bin_;
          it.reset(fg->bin_data_->GetIterator(0, nb - 1, 0));
        }
        if (nb > 256) { fprintf(stderr, "refdrv_train_tree: a column with %d bins\n", nb); return -1; }
        group_num_bin[ncol] = nb;
        it->Reset(0);
        for (int i = 0; i < n; ++i) bins_out[(size_t)ncol * n + i] = (unsigned char)it->RawGet(i);
        col_of[g].push_back(ncol);
        col_base.push_back(base);
        base += nb;
        ++ncol;
      }
    }
    *num_groups_out = ncol;
    Config config;
    config.Set(Config::Str2Map(params));
    std::vector<FeatureMetainfo> metas;
    HistogramPool::SetFeatureInfo<true, true>(ds, &config, &metas);
    for (int f = 0; f < ds->num_features(); ++f) {
      const BinMapper* bm = ds->FeatureBinMapper(f);
      const int g = ds->feature2group_[f], sub = ds->feature2subfeature_[f];
      FeatureGroup* fg = ds->feature_groups_[g].get();
      int col, min_bin, max_bin;
      if (unbundle) {
        col = f; min_bin = 1; max_bin = group_num_bin[f] - 1;
      } else if (fg->is_multi_val_) {
        col = col_of[g][sub]; min_bin = 1; max_bin = group_num_bin[col] - 1;
      } else {
        col = col_of[g][0]; min_bin = (int)fg->bin_offsets_[sub]; max_bin = (int)fg->bin_offsets_[sub + 1] - 1;
      }
      feat_view_offset[f] = col_base[col] + min_bin;
      feat_num_bin[f] = bm->num_bin(); feat_most_freq_bin[f] = (int)bm->GetMostFreqBin();
      feat_meta3[3 * f] = metas[f].offset; feat_meta3[3 * f + 1] = (int)metas[f].default_bin; feat_meta3[3 * f + 2] = (int)metas[f].missing_type;
      if (feat_layout4) {
        feat_layout4[4 * f] = col; feat_layout4[4 * f + 1] = min_bin; feat_layout4[4 * f + 2] = max_bin;
        feat_layout4[4 * f + 3] = bm->bin_type() == BinType::CategoricalBin ? 1 : 0;
      }
    }
    std::vector<score_t> g_all(grad, grad + n), h_all(n, 1.0);
    if (hess) std::copy(hess, hess + n, h_all.begin());
    std::unique_ptr<TreeLearner> tl(TreeLearner::CreateTreeLearner("serial", "cpu", &config));
    tl->Init(ds, hess == nullptr);
    const json11::Json no_forced_splits;                 /* GBDT::Init hands the learner a null Json (gbdt.cpp:115): the member is otherwise uninitialised */
    tl->SetForcedSplit(&no_forced_splits);
    std::unique_ptr<Tree> tree(tl->Train(g_all.data(), h_all.data(), true));

    const int nl = tree->num_leaves_;
    *num_leaves_out = nl;
    for (int k = 0; k < nl - 1; ++k) {
      split_feature_inner[k] = tree->split_feature_inner_[k]; threshold_in_bin[k] = (int)tree->threshold_in_bin_[k];
      default_left[k] = Tree::GetDecisionType(tree->decision_type_[k], kDefaultLeftMask) ? 1 : 0;
      left_child[k] = tree->left_child_[k]; right_child[k] = tree->right_child_[k];
      split_gain[k] = tree->split_gain_[k]; internal_count[k] = tree->internal_count_[k];
      /* categorical node (Tree::SplitCategorical, src/LightGBM/io/tree.cpp:76-108): threshold_in_bin_ indexes cat_boundaries_inner_; the bitset
       * over the feature's BINS of the categories going left */
      if (node_is_cat) {
        const bool is_cat = Tree::GetDecisionType(tree->decision_type_[k], kCategoricalMask);
        node_is_cat[k] = is_cat ? 1 : 0;
        for (int w = 0; w < 8; ++w) node_cat_bits8[8 * k + w] = 0u;
        if (is_cat) {
          const int ci = (int)tree->threshold_in_bin_[k];
          const int b0 = tree->cat_boundaries_inner_[ci], b1 = tree->cat_boundaries_inner_[ci + 1];
          for (int w = b0; w < b1 && w - b0 < 8; ++w) node_cat_bits8[8 * k + (w - b0)] = tree->cat_threshold_inner_[w];
        }
      }
    }
    for (int k = 0; k < nl; ++k) { leaf_value[k] = tree->leaf_value_[k]; leaf_count[k] = tree->leaf_count_[k]; }
    tl.reset();
    LGBM_DatasetFree(dh);
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_train_tree: %s\n", e.what());
    return -1;
  }
}

/* Boosting gradient for non-Gaussian data: d(-approximate marginal log-likelihood) / dF at the fixed effects F (REModel::CalcGradient ->
 * CalcGradientF -> CalcGradFLaplace, re_model_template.h:3298-3321), data order.  h is a handle of the reference's own C API
 * (GPB_CreateREModel of lib_gpboost_ref.so = REModel*) whose covariance parameters have been set (GPB_SetOptimConfig(init_cov_pars)). */
__attribute__((visibility("default")))
int refdrv_laplace_grad_F(void* h, const double* y, const double* fixed_effects, double* grad_out) {
  try {
    auto* m = reinterpret_cast<REModel*>(h);
    m->SetY(y);
    m->CalcGradient(grad_out, fixed_effects, true);
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_laplace_grad_F: %s\n", e.what());
    return -1;
  }
}

/* Covariance-parameter gradient of the Laplace-approximated negative marginal log-likelihood, read off the reference's OWN gradient routine
 * (REModelTemplate::CalcGradPars -> Likelihood::CalcGradNegMargLikelihoodLaplaceApproxVecchia, include/GPBoost/re_model_template.h:2050-2098,
 * include/GPBoost/likelihoods.h:6521-6700) -- not off an optimiser step.  Sequence = the L-BFGS functor's (include/GPBoost/optim_utils.h:299-338):
 * the public evaluation first (sets y, initialises the mode: what GPB_EvalNegLogLikelihood does, c_api.cpp:2849-2858), then
 * CalcCovFactorOrModeAndNegLL at the transformed parameters (the mode finding restarts from the mode just found) and CalcGradPars.
 * h: handle of the reference's own C API (REModel*); solver thresholds (cg_delta_conv, delta_conv_mode_finding) as set by GPB_SetOptimConfig before.
 * cov_pars_orig = (sigma1^2, rho) [+ nothing: no auxiliary parameters here]; grad_out[k] = d(-mll) / d log(transformed parameter k) = wrt (log sigma1^2, log a).
 * naux > 0 (likelihoods with auxiliary parameters, estimate_aux_pars): grad_out holds num_cov_par + naux entries, the last naux wrt log(aux). */
__attribute__((visibility("default")))
int refdrv_laplace_nll_grad(void* h, const double* y, const double* cov_pars_orig, const double* fixed_effects, int estimate_aux,
                            double* nll_public, double* nll_functor, double* grad_out, int* ngrad_out) {
  try {
    auto* m = reinterpret_cast<REModel*>(h);
    std::vector<double> cp(cov_pars_orig, cov_pars_orig + m->num_cov_pars_);
    double negll = 0.;
    m->EvalNegLogLikelihood(y, cp.data(), negll, fixed_effects, true, false);
    *nll_public = negll;
    if (m->matrix_format_ != "den_mat_t") { fprintf(stderr, "refdrv_laplace_nll_grad: matrix format %s\n", m->matrix_format_.c_str()); return -1; }
    auto* t = m->re_model_den_.get();
    vec_t orig = Eigen::Map<const vec_t>(cov_pars_orig, m->num_cov_pars_), trafo(m->num_cov_pars_);
    t->TransformCovPars(orig, trafo);
    if ((int)t->estimate_cov_par_index_.size() != m->num_cov_pars_) t->estimate_cov_par_index_ = std::vector<int>(m->num_cov_pars_, 1);
    t->estimate_aux_pars_ = estimate_aux != 0 && t->NumAuxPars() > 0;
    t->CalcCovFactorOrModeAndNegLL(trafo, fixed_effects);
    *nll_functor = t->GetNegLogLikelihood();
    vec_t grad_cov, grad_beta;
    t->CalcGradPars(trafo, 1., true, false, grad_cov, grad_beta, false, false, fixed_effects, false);
    for (int k = 0; k < (int)grad_cov.size(); ++k) grad_out[k] = grad_cov[k];
    *ngrad_out = (int)grad_cov.size();
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_laplace_nll_grad: %s\n", e.what());
    return -1;
  }
}

/* Newton iterations of the LAST mode finding of a non-Gaussian model (Likelihood::num_it_mode_finding_, likelihoods.h:1298, :3984): the reference's own count,
 * to set beside the device path's (round 6, VERDICT r05 #7: are the 16 Newton steps of the t likelihood at config 4's size the reference's?) */
__attribute__((visibility("default")))
int refdrv_num_it_mode_finding(void* h) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    if (!t) return -1;
    return t->likelihood_[t->unique_clusters_[0]]->num_it_mode_finding_;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_num_it_mode_finding: %s\n", e.what());
    return -1;
  }
}

/* ---- full-scale Vecchia ("VIF"), Gaussian likelihood: the reference's own REModel with gp_approx = "full_scale_vecchia" ----
 * refdrv_nll_grad works on such a handle unchanged (CalcGradPars dispatches to CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i,
 * include/GPBoost/re_model_template.h:2205-2330); the two functions below expose what its public API does not: the inducing points
 * and the derivative factors B_grad = -dA / D_grad of the residual process (src/GPBoost/Vecchia_utils.cpp:1503-1524, 1640-1656). */
__attribute__((visibility("default")))
void* refdrv_create_vif(int n, const double* coords_colmajor, int d, const char* cov_fct, double shape, int m,
                        const char* ordering, int seed, int num_ind_points, int num_threads) {
  try {
    return new REModel(n, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 1, coords_colmajor, d, nullptr, 0,
                       cov_fct, shape, "full_scale_vecchia", 1., 0., m, ordering, num_ind_points, 1., "kmeans++",
                       "gaussian", 1., "cholesky", seed, num_threads, false, false, nullptr, 1.);
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_create_vif: %s\n", e.what());
    return nullptr;
  }
}

/* after refdrv_nll_grad on a VIF handle: dA (n x m, aligned with the neighbour table) and dD of parameter ipar (0: variance, 1: range),
 * derivatives wrt the log of the transformed parameters; optionally the factor itself and y_aux (Vecchia order) */
__attribute__((visibility("default")))
int refdrv_get_grad_factor(void* h, int m, int ipar, double* dA, double* dD) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    const int c0 = t->unique_clusters_[0];
    const sp_mat_t& Bg = t->B_grad_[c0][0][ipar];
    const sp_mat_t& Dg = t->D_grad_[c0][0][ipar];
    const auto& nn = t->nearest_neighbors_[c0][0];
    const int n = (int)nn.size();
    for (int i = 0; i < n; ++i) {
      dD[i] = Dg.coeff(i, i);
      for (int j = 0; j < m; ++j) dA[(size_t)i * m + j] = (j < (int)nn[i].size()) ? -Bg.coeff(i, nn[i][j]) : 0.;
    }
    return 0;
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_get_grad_factor: %s\n", e.what());
    return -1;
  }
}

/* y_aux = Psi^-1 y of the last evaluation (Vecchia order), any approximation */
__attribute__((visibility("default")))
int refdrv_get_yaux(void* h, double* yaux) {
  try {
    auto* t = reinterpret_cast<REModel*>(h)->re_model_den_.get();
    const int c0 = t->unique_clusters_[0];
    t->CalcYAux(1., false);
    const vec_t& ya = t->y_aux_[c0];
    for (int i = 0; i < (int)ya.size(); ++i) yaux[i] = ya[i];
    return (int)ya.size();
  } catch (std::exception& e) {
    fprintf(stderr, "refdrv_get_yaux: %s\n", e.what());
    return -1;
  }
}

}  // extern "C"
