"""Test harness: the control flow of SerialTreeLearner::Train (src/LightGBM/treelearner/serial_tree_learner.cpp:159-210, :283-323,
:325-449, :565-690) on top of the five hot-path primitives -- leaf histogram, FixHistogram, parent - smaller, split search, leaf
partition -- with either the CPU oracle or the MI355X library as the back-end.  TEST INFRASTRUCTURE (the tree learner's control plane
stays the reference's own host code); what it shows is that the primitives compose to the reference's own tree.
"""
import numpy as np

K_EPS = float(np.float32(1e-15))


class OracleBackend(object):
    """bins (G, n) uint8 + metas; histograms live in a dict of numpy arrays."""

    def __init__(self, orc, bins, gnb, view_offset, num_bin, most_freq_bin, meta3, grad, hess, is_cat=None, cat_cfg=None):
        self.orc, self.bins = orc, bins
        self.is_cat = np.zeros(bins.shape[0], dtype=np.int32) if is_cat is None else np.asarray(is_cat, dtype=np.int32)
        self.cat_cfg = orc.CAT_DEFAULTS if cat_cfg is None else cat_cfg
        self.last_cat_bits = np.zeros((bins.shape[0], 8), dtype=np.uint32)
        self.bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
        self.gnb, self.vo, self.nb, self.mfb, self.meta3 = gnb, view_offset, num_bin, most_freq_bin, meta3
        self.grad, self.hess = grad, hess
        self.slots = {}
        self.F = bins.shape[0]

    def build_fix(self, slot, idx, sg, sh):
        hg, hc, hh = self.orc.hist_build(self.bins, self.bo, idx, self.grad, self.hess)
        hist = np.stack([hg, hh], axis=1)
        self.slots[slot] = self.orc.hist_fix(hist, self.vo, self.nb, self.mfb, sg, sh)

    def subtract(self, parent, smaller, out):
        self.slots[out] = self.orc.hist_subtract(self.slots[parent], self.slots[smaller])

    def search(self, slot, sg, sh, cnt, cfg, used, parent_output=0.0):
        best, out, dl = self.orc.find_best_split(self.slots[slot], self.vo, self.nb, self.meta3[:, 0], self.meta3[:, 1], self.meta3[:, 2],
                                                 sg, sh, cnt, *cfg, parent_output=parent_output)
        spl = self.orc.find_best_split.last_splittable.copy()
        self.last_cat_bits = np.zeros((self.F, 8), dtype=np.uint32)
        for f in np.nonzero(self.is_cat)[0]:          # categorical features: FindBestThresholdCategoricalInner instead of the threshold scans
            row, fl, bits = self.orc.find_best_split_cat(self.slots[slot], self.vo[f], self.nb[f], self.meta3[f, 0], sg, sh, cnt, *cfg,
                                                         parent_output=parent_output, cat_cfg=self.cat_cfg)
            out[f] = row; dl[f] = fl & 1; spl[f] = (fl >> 1) & 1; self.last_cat_bits[f] = bits
        return out, dl, spl

    def partition(self, idx, f, thr, dl, cat_bits=None):
        if self.is_cat[f]:
            return self.orc.split_leaf_layout(self.bins[f], 1, self.gnb[f] - 1, False, self.meta3[f, 1], self.mfb[f], self.meta3[f, 2], dl, thr, True, cat_bits, idx)
        return self.orc.split_leaf(self.bins[f], self.gnb[f] - 1, self.meta3[f, 1], self.mfb[f], self.meta3[f, 2], dl, thr, idx)


class GpuBackend(object):
    def __init__(self, shim, bins, gnb, view_offset, num_bin, most_freq_bin, meta3, grad, hess, num_leaves, is_cat=None, cat_cfg=None):
        bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
        self.hb = shim.HistBuilder(bins, bo)
        self.hb.pool_resize(num_leaves + 1)
        self.hb.set_fix_info(view_offset, num_bin, most_freq_bin)
        self.hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
        self.is_cat = np.zeros(bins.shape[0], dtype=np.int32) if is_cat is None else np.asarray(is_cat, dtype=np.int32)
        if self.is_cat.any():
            self.hb.set_categorical(self.is_cat, *(cat_cfg if cat_cfg is not None else (4, 32, 10.0, 10.0, 100)))
        self.last_cat_bits = np.zeros((bins.shape[0], 8), dtype=np.uint32)
        self.hb.set_gradients(grad, hess)
        self.F = bins.shape[0]

    def build_fix(self, slot, idx, sg, sh):
        self.hb.build_slot(slot, idx)
        self.hb.fix_slot(slot, sg, sh)

    def subtract(self, parent, smaller, out):
        self.hb.subtract_slots(parent, smaller, out)

    def search(self, slot, sg, sh, cnt, cfg, used, parent_output=0.0):
        if len(cfg) > 4:
            self.hb.set_regularisation(cfg[4], cfg[5], cfg[6], parent_output)
        best, out, dl = self.hb.find_best_split(slot, sg, sh, cnt, *cfg[:4])
        self.last_cat_bits = self.hb.last_cat_bits.copy()
        return out, dl, self.hb.last_splittable.copy()

    def partition(self, idx, f, thr, dl, cat_bits=None):
        return self.hb.split_leaf(idx, f, thr, dl, cat_bits=cat_bits if self.is_cat[f] else None)

    def close(self):
        self.hb.close()


def _better(a, b):
    """SplitInfo::operator> (split_info.hpp:126-153) on (gain, feature) pairs; feature -1 compares as INT32_MAX."""
    fa = a[1] if a[1] >= 0 else 2 ** 31 - 1
    fb = b[1] if b[1] >= 0 else 2 ** 31 - 1
    return a[0] > b[0] if a[0] != b[0] else fa < fb


def root_parent_output(sg, sh, l1, l2, max_delta_step):
    """SerialTreeLearner::GetParentOutput for the root (serial_tree_learner.cpp:758-770): its own output, L1 and max_delta_step applied,
    no smoothing (feature_histogram.hpp:743-765)."""
    if l1 > 0:
        ret = -float(np.sign(sg)) * max(0.0, abs(sg) - l1) / (sh + l2)
    else:
        ret = -sg / (sh + l2)
    if max_delta_step > 0 and abs(ret) > max_delta_step:
        ret = float(np.sign(ret)) * max_delta_step
    return ret


def grow_tree(be, grad, hess, n, num_leaves, cfg, max_depth=0, feature_mask=None, root_rows=None):
    """cfg = (lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split[, lambda_l1, max_delta_step, path_smooth]).
    Returns the same arrays ref_train_tree does."""
    l2, min_data, min_hess, min_gain = cfg[:4]
    l1, mds, smooth = (cfg[4:7] if len(cfg) > 4 else (0.0, 0.0, 0.0))
    F = be.F
    hs = np.ones(n) if hess is None else hess
    # root sums: left-to-right summation (leaf_splits.hpp:73-86 with one thread)
    sg = float(np.cumsum(grad)[-1]); sh = float(np.cumsum(hs)[-1])
    idx = {0: None}
    cnt = {0: n}
    if root_rows is not None:                    # bagging: the root holds these rows (DataPartition::Init with used_data_indices_); sums over them
        rr = np.ascontiguousarray(root_rows, dtype=np.int32)
        sg = float(np.cumsum(grad[rr])[-1]); sh = float(np.cumsum(hs[rr])[-1])
        idx = {0: rr}; cnt = {0: rr.size}
    sums = {0: (sg, sh)}
    best = {}                                    # leaf -> dict(gain, feature, row) of its best split
    splittable = {}
    slot_of = {0: 0}                             # leaf -> histogram slot; every split takes one new slot for the smaller child
    next_free = [1]
    leaf_value = {0: 0.0}
    nleaves = 1

    def search_leaf(leaf, used):
        g_, h_ = sums[leaf]
        # parent_output of FindBestThreshold: the leaf's own output (LeafSplits::weight), the root's unsmoothed output for the root
        po = root_parent_output(g_, h_, l1, l2, mds) if nleaves == 1 else leaf_value[leaf]
        out, dl, spl = be.search(slot_of[leaf], g_, h_, cnt[leaf], cfg, used, po)
        spl = np.where(used > 0, spl, 0)
        top = (-np.inf, -1); row = None
        for f in range(F):
            if not used[f]:
                continue
            cand = (out[f, 0], f)
            if _better(cand, top):
                top = cand; row = (out[f].copy(), int(dl[f]), be.last_cat_bits[f].copy())
        best[leaf] = dict(gain=top[0], feature=top[1], row=row)
        splittable[leaf] = spl

    be.build_fix(0, idx[0], sg, sh)
    # feature_fraction: the tree's sampled columns (ColSampler::is_feature_used_bytree, serial_tree_learner.cpp:329); children inherit the mask
    # through the parent's is_splittable flags
    search_leaf(0, np.ones(F, dtype=np.int8) if feature_mask is None else np.asarray(feature_mask, dtype=np.int8))
    nodes = dict(split_feature_inner=[], threshold_in_bin=[], default_left=[], left_child=[], right_child=[], split_gain=[], internal_count=[],
                 node_is_cat=[], node_cat_bits=[])
    is_cat = getattr(be, "is_cat", np.zeros(F, dtype=np.int32))
    ncat_nodes = 0
    node_rows = []                               # rows of the leaf each node split (for the tests' tie analysis)
    leaf_parent_node = {0: -1}
    leaf_is_left = {0: True}
    left = right = None
    leaf_depth = {0: 0}
    for split in range(num_leaves - 1):
        if split > 0:
            nl_, nr_ = cnt[left], cnt[right]
            if max_depth > 0 and leaf_depth[left] >= max_depth:                # BeforeFindBestSplit :286-295
                best[left] = dict(gain=-np.inf, feature=best.get(left, {}).get("feature", -1), row=None)
                best[right] = dict(gain=-np.inf, feature=-1, row=None)
            elif nr_ < 2 * min_data and nl_ < 2 * min_data:                    # BeforeFindBestSplit :296-306
                best[left] = dict(gain=-np.inf, feature=best.get(left, {}).get("feature", -1), row=None)
                best[right] = dict(gain=-np.inf, feature=-1, row=None)
            else:
                smaller, larger = (left, right) if nl_ < nr_ else (right, left)
                used = splittable[left].astype(np.int8)        # the parent's is_splittable flags (:330-334); `left` kept the parent's id
                p_slot = slot_of[left]                          # ... and its histogram
                s_slot = next_free[0]; next_free[0] += 1
                be.build_fix(s_slot, idx[smaller], *sums[smaller])   # smaller leaf: construct + FixHistogram (:400-403)
                be.subtract(p_slot, s_slot, p_slot)                  # larger leaf: parent - smaller, in place (:419-421)
                slot_of[smaller], slot_of[larger] = s_slot, p_slot
                search_leaf(smaller, used)
                search_leaf(larger, used)
        # ArgMax over the leaves (first maximal element under operator>)
        top_leaf, top = -1, (-np.inf, -1)
        for leaf in range(nleaves):
            b = best[leaf]
            cand = (b["gain"], b["feature"])
            if top_leaf < 0 or _better(cand, top):
                top_leaf, top = leaf, cand
        b = best[top_leaf]
        if not (b["gain"] > 0.0):
            break
        row, dl, cbits = b["row"]
        f, thr = b["feature"], int(row[1])
        rows = idx[top_leaf] if idx[top_leaf] is not None else np.arange(n, dtype=np.int32)
        node_rows.append(rows)
        if is_cat[f]:
            lte, gt = be.partition(rows, f, thr, dl, cat_bits=cbits)
            thr = ncat_nodes; ncat_nodes += 1          # Tree::SplitCategorical: threshold_in_bin_ = index of the node's bitset (tree.cpp:76-108)
        else:
            lte, gt = be.partition(rows, f, thr, dl)
        left, right = top_leaf, nleaves
        node = len(nodes["split_feature_inner"])
        # Tree::Split bookkeeping (include/LightGBM/tree.h): children are ~leaf; the parent's pointer to this leaf becomes the new node
        pn = leaf_parent_node[top_leaf]
        if pn >= 0:
            if leaf_is_left[top_leaf]:
                nodes["left_child"][pn] = node
            else:
                nodes["right_child"][pn] = node
        nodes["split_feature_inner"].append(f); nodes["threshold_in_bin"].append(thr); nodes["default_left"].append(dl)
        nodes["node_is_cat"].append(int(is_cat[f])); nodes["node_cat_bits"].append(cbits if is_cat[f] else np.zeros(8, dtype=np.uint32))
        nodes["left_child"].append(~left); nodes["right_child"].append(~right)
        nodes["split_gain"].append(float(np.float32(row[0] + min_gain))); nodes["internal_count"].append(len(lte) + len(gt))
        leaf_parent_node[left] = node; leaf_parent_node[right] = node
        leaf_is_left[left] = True; leaf_is_left[right] = False
        idx[left], idx[right] = lte, gt
        cnt[left], cnt[right] = len(lte), len(gt)                                # update_cnt (:591-595)
        sums[left], sums[right] = (row[6], row[7]), (row[8], row[9])
        leaf_value[left], leaf_value[right] = row[4], row[5]
        leaf_depth[left] = leaf_depth[right] = leaf_depth[top_leaf] + 1
        nleaves += 1
    out = {k: np.asarray(v) for k, v in nodes.items()}
    out["num_leaves"] = nleaves
    out["node_rows"] = node_rows
    out["leaf_value"] = np.asarray([leaf_value[k] for k in range(nleaves)])
    out["leaf_count"] = np.asarray([cnt[k] for k in range(nleaves)])
    return out


