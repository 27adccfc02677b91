"""GPU (MI355X): LightGBM feature-histogram build through the C ABI against the oracle.
Bin counts are bit-exact.  The reference's fp64 gradient / hessian sums are order-dependent (per-thread block buffers,
train_share_states.h:46-109), so against the ORACLE the sums are compared to 1e-10 relative; the device's own sums are fixed-point
totals and bit-reproducible (test_histogram_is_reproducible_and_subtractable; across rank layouts: tests/test_multirank_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n, F, seed, max_bin=255):
    rng = np.random.default_rng(seed)
    nb = rng.integers(2, max_bin + 2, size=F)
    nb[0] = 256; nb[-1] = 2
    bo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    bins = np.stack([rng.integers(0, nb[f], size=n) for f in range(F)]).astype(np.uint8)
    return bins, bo, rng.standard_normal(n), rng.uniform(0.5, 2.0, size=n), rng


@pytest.mark.parametrize("n,F", [(100000, 50), (4097, 3), (1000, 17), (50, 64), (1200000, 50), (600000, 70)])   # the last two: the whole-row kernel (one full quad of feature groups; a full and a partial one)
def test_histogram_counts_exact_and_sums_close(lib_built, orc, n, F):
    from gpboost_amd import shim
    bins, bo, grad, hess, rng = _case(n, F, seed=n + F)
    hb = shim.HistBuilder(bins, bo)
    leaf = np.sort(rng.choice(n, size=max(1, n // 3), replace=False)).astype(np.int32)
    shuffled = rng.permutation(leaf).astype(np.int32)     # data_indices need not be sorted
    for hs in (None, hess):
        hb.set_gradients(grad, hs)
        for di in (None, leaf, shuffled, leaf[:1]):
            hist, cnt = hb.build(di, const_hess=1.0)
            hg, hc, hh = orc.hist_build(bins, bo, di, grad, hs, const_hess=1.0)
            assert np.array_equal(cnt, hc), "bin counts must be bit-exact"
            scale = np.abs(hg).max() + 1.0
            np.testing.assert_allclose(hist[:, 0], hg, rtol=0, atol=1e-10 * scale)
            if hs is None:
                assert np.array_equal(hist[:, 1], hh), "constant hessian: count * hess is exact"
            else:
                np.testing.assert_allclose(hist[:, 1], hh, rtol=0, atol=1e-10 * (np.abs(hh).max() + 1.0))


def test_histogram_is_reproducible_and_subtractable(lib_built, orc):
    """larger = parent - smaller (serial_tree_learner.cpp:419-421) holds exactly for counts."""
    from gpboost_amd import shim
    n, F = 200000, 50
    bins, bo, grad, hess, rng = _case(n, F, seed=5)
    hb = shim.HistBuilder(bins, bo); hb.set_gradients(grad, None)
    parent, pc = hb.build(None)
    again, pc2 = hb.build(None)
    # counts are integers; the sums are integer totals of once-rounded gradients (fixed-point words), converted once: BOTH
    # are independent of the order of the LDS atomics, i.e. repeated builds are bit-identical (the reference's per-thread block buffers
    # are not fixed across thread counts)
    assert np.array_equal(pc, pc2) and np.array_equal(parent, again)
    # ... and independent of the chunking / the kernel variant: the same rows handed over as an index list (other launch shape), in
    # another order, and with per-row hessians present (hist_build_kernel instead of the whole-row kernel) give the same gradient sums
    perm = rng.permutation(n).astype(np.int32)
    h2, c2 = hb.build(perm)
    assert np.array_equal(c2, pc) and np.array_equal(h2, parent)
    hb.set_gradients(grad, hess)
    h3, c3 = hb.build(None)
    assert np.array_equal(c3, pc) and np.array_equal(h3[:, 0], parent[:, 0])
    h4, _ = hb.build(perm)
    assert np.array_equal(h4, h3)
    hb.set_gradients(grad, None)
    mask = rng.uniform(size=n) < 0.37
    left = np.nonzero(mask)[0].astype(np.int32); right = np.nonzero(~mask)[0].astype(np.int32)
    hl, cl = hb.build(left); hr, cr = hb.build(right)
    assert np.array_equal(cl + cr, pc)
    np.testing.assert_allclose(hl[:, 0] + hr[:, 0], parent[:, 0], rtol=0, atol=1e-9)
    assert int(pc[bo[0]:bo[1]].sum()) == n


def test_full_size_properties_n1e7(lib_built):
    """SURVEY.md 8d's histogram size (n = 1e7 rows, F = 50, 255 bins): properties that need no oracle pass -- every feature's counts
    sum to the rows of the leaf, two features equal numpy's bincount exactly (counts) / to 1e-9 (sums), every feature's gradient sums
    add up to the total gradient, and parent = left + right holds exactly for counts and to rounding for the sums."""
    from gpboost_amd import shim
    n, F, nb = 10000000, 50, 255
    rng = np.random.default_rng(9)
    bins = rng.integers(0, nb, size=(F, n), dtype=np.uint8)
    bo = (np.arange(F + 1) * nb).astype(np.int32)
    grad = rng.standard_normal(n)
    hb = shim.HistBuilder(bins, bo); hb.set_gradients(grad, None)
    hist, cnt = hb.build(None)
    assert np.array_equal(cnt.reshape(F, nb).sum(axis=1), np.full(F, n, dtype=np.uint64))
    tot = grad.sum()
    np.testing.assert_allclose(hist[:, 0].reshape(F, nb).sum(axis=1), tot, rtol=0, atol=1e-8 * np.abs(grad).sum())
    for f in (0, 37):
        assert np.array_equal(cnt[bo[f]:bo[f + 1]], np.bincount(bins[f], minlength=nb).astype(np.uint64))
        ref = np.bincount(bins[f], weights=grad, minlength=nb)
        np.testing.assert_allclose(hist[bo[f]:bo[f + 1], 0], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    left = np.flatnonzero(bins[3] < 100).astype(np.int32)
    right = np.flatnonzero(bins[3] >= 100).astype(np.int32)
    hl, cl = hb.build(left); hr, cr = hb.build(right)
    assert np.array_equal(cl + cr, cnt)
    assert cl[bo[3] + 100:bo[4]].sum() == 0 and cr[bo[3]:bo[3] + 100].sum() == 0        # the split feature separates cleanly
    np.testing.assert_allclose(hl[:, 0] + hr[:, 0], hist[:, 0], rtol=0, atol=1e-9 * np.abs(hist[:, 0]).max())
    hb.close()


def test_histogram_against_reference_fixture(lib_built):
    """The reference's own stored bins and its Dataset::ConstructHistograms output (tests/golden/hist_ref.npz)."""
    import os
    from gpboost_amd import shim
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hist_ref.npz"))
    X, grad, hess, leaf = cases.make_hist_data()
    bins, gnb = g["bins"], g["group_num_bin"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    for hi, hs in enumerate((None, hess)):
        hb.set_gradients(grad, hs)
        for li, di in enumerate((None, leaf)):
            ref = g["hist_leaf%d_hess%d" % (li, hi)]
            hist, cnt = hb.build(di, const_hess=1.0)
            np.testing.assert_allclose(hist[:, 0], ref[:, 0], rtol=0, atol=1e-10 * (np.abs(ref[:, 0]).max() + 1))
            if hs is None:
                assert np.array_equal(hist[:, 1], ref[:, 1]), "count * hess must be bit-exact"
                assert np.array_equal(cnt.astype(np.float64), ref[:, 1])
            else:
                np.testing.assert_allclose(hist[:, 1], ref[:, 1], rtol=0, atol=1e-10 * (np.abs(ref[:, 1]).max() + 1))


def test_fix_histogram_and_subtraction_against_reference_fixture(lib_built, orc):
    """Row a12 on device-resident histograms: Dataset::FixHistogram for every feature (two of the six features have a most
    frequent bin > 0) and FeatureHistogram::Subtract, against the reference's own output (tests/golden/hist_ref.npz)."""
    import os
    import gpboost_amd
    from gpboost_amd import shim
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hist_ref.npz"))
    X, grad, hess, leaf = cases.make_hist_data()
    bins, gnb = g["bins"], g["group_num_bin"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    with pytest.raises(gpboost_amd.GPBoostError):
        hb.build_slot(0, None)                       # no pool yet
    hb.pool_resize(3)
    hb.set_gradients(grad, hess)
    with pytest.raises(gpboost_amd.GPBoostError):
        hb.fix_slot(0, 0.0, 0.0)                     # feature views not set
    hb.set_fix_info(g["fix_view_offset"], g["fix_num_bin"], g["fix_most_freq_bin"])
    for slot, (li, di) in enumerate(((0, None), (1, leaf))):
        key = "leaf%d_hess1" % li
        hb.build_slot(slot, di)
        raw = hb.get_slot(slot)
        sums = g["fix_sums_" + key]
        hb.fix_slot(slot, sums[0], sums[1])
        fixed = hb.get_slot(slot)
        # bit-identical to the reference's FixHistogram applied to THIS histogram (same subtraction order) ...
        assert np.array_equal(fixed, orc.hist_fix(raw, g["fix_view_offset"], g["fix_num_bin"], g["fix_most_freq_bin"], sums[0], sums[1]))
        # ... and equal to the reference's fixed histogram up to the summation order of the build (fp64 atomics)
        ref = g["hist_fixed_" + key]
        np.testing.assert_allclose(fixed, ref, rtol=0, atol=1e-9 * (np.abs(ref).max() + 1))
        assert not np.array_equal(fixed, raw)
    hb.subtract_slots(0, 1, 2)                       # larger = parent - smaller
    assert np.array_equal(hb.get_slot(2), hb.get_slot(0) - hb.get_slot(1))
    hb.subtract_slots(0, 1, 0)                       # in place, as FeatureHistogram::Subtract does
    assert np.array_equal(hb.get_slot(0), hb.get_slot(2))
    with pytest.raises(gpboost_amd.GPBoostError):
        hb.get_slot(3)
    hb.close()


def test_histogram_rccl_allreduce_single_rank(lib_built):
    """Data-parallel histogram path with a 1-rank RCCL communicator: the scale goes through ncclAllReduce(max), the integer totals
    through ncclAllReduce(sum, int64), the conversion runs afterwards -- and must reproduce the plain build bit for bit (several ranks:
    tests/test_multirank_gpu.py on the device, tests/test_distributed_cpu.py on CPU / gloo)."""
    import gpboost_amd
    from gpboost_amd import shim
    rng = np.random.default_rng(5)
    n, F = 20000, 7
    nb = rng.integers(2, 257, size=F); bo = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    bins = np.stack([rng.integers(0, nb[f], size=n) for f in range(F)]).astype(np.uint8)
    grad, hess = rng.standard_normal(n), rng.uniform(0.5, 2.0, size=n)
    hb = shim.HistBuilder(bins, bo); hb.set_gradients(grad, None)
    leaf = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    with pytest.raises(gpboost_amd.GPBoostError):
        hb.build_allreduce(leaf)                           # no communicator
    h0, c0 = hb.build(leaf)
    hb.set_gradients(grad, hess)
    h0h, _ = hb.build(leaf)
    hb.comm_init(shim.comm_unique_id(), 0, 1)
    with pytest.raises(gpboost_amd.GPBoostError):
        hb.build_allreduce(leaf)                           # a new communicator invalidates the scale: gradients must be set again
    hb.set_gradients(grad, None)
    h1, c1 = hb.build_allreduce(leaf)
    assert np.array_equal(c1, c0) and np.array_equal(h1, h0)
    hb.set_gradients(grad, hess)
    h1h, c1h = hb.build_allreduce(leaf)
    assert np.array_equal(c1h, c0) and np.array_equal(h1h, h0h)
    hb.close()


@pytest.mark.parametrize("name", ["plain", "zero_missing", "nan"])
def test_split_search_against_reference_fixture(lib_built, orc, name):
    """FeatureHistogram::FindBestThreshold per feature + the winner, on device-resident histograms, against the
    reference's own output (tests/golden/split_ref.npz): (i) on the reference's fixed histogram uploaded as is -> every SplitInfo field
    bit-identical; (ii) end to end from the bins (build -> fix -> search) -> same winner and threshold, sums to the build's 1e-10."""
    import os
    from gpboost_amd import shim
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_ref.npz"))
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3 = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(2)
    hb.set_fix_info(g[name + "_view_offset"], g[name + "_num_bin"], g[name + "_most_freq_bin"])
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    for ci, cfg in enumerate(cases.SPLIT_CFGS):
        for li, di in enumerate((None, leaf)):
            for hi, hs in enumerate((None, hess)):
                key = "%s_cfg%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums, ref, ref_dl = g[key + "_sums"], g[key + "_split"], g[key + "_default_left"]
                nd = bins.shape[1] if di is None else di.size
                # end to end on the device
                hb.set_gradients(grad, hs)
                hb.build_slot(0, di)
                hb.fix_slot(0, sums[0], sums[1])
                best, out, dl = hb.find_best_split(0, sums[0], sums[1], nd, *cfg)
                raw = hb.get_slot(0)
                obest, oout, odl = orc.find_best_split(raw, g[name + "_view_offset"], g[name + "_num_bin"], meta3[:, 0], meta3[:, 1],
                                                       meta3[:, 2], sums[0], sums[1], nd, *cfg)
                assert best == obest and np.array_equal(out, oout) and np.array_equal(dl, odl)      # bit-identical given the histogram
                assert best == int(np.argmax(ref[:, 0]))
                assert out[best, 1] == ref[best, 1]                                                 # same threshold
                if dl[best] != ref_dl[best]:
                    # the two scan directions found the SAME partition (no row of the leaf sits in the missing / default bin): their gains are
                    # mathematically equal and the reference's `>` picks a direction by the last bit of its order-dependent fp64 sums
                    # (oracle check: 1e-12 relative noise on the reference's own histogram flips this fixture's default side in half of the
                    # trials).  The counts of a SplitInfo are estimates from the hessian sums (RoundInt(hess * cnt_factor) per bin and
                    # direction, feature_histogram.hpp:857-1084), so with per-row hessians the two directions may differ by a row or two.
                    assert abs(out[best, 2] - ref[best, 2]) <= (0 if hs is None else 2) and abs(out[best, 3] - ref[best, 3]) <= (0 if hs is None else 2)
                np.testing.assert_allclose(out[best, [0, 4, 5, 6, 7, 8, 9]], ref[best, [0, 4, 5, 6, 7, 8, 9]], rtol=1e-9, atol=1e-9)
    # a masked-out winner never wins; an empty mask yields -1
    used = np.ones(hb.F, dtype=np.int8); used[best] = 0
    b2, _, _ = hb.find_best_split(0, sums[0], sums[1], nd, *cfg, is_feature_used=used)
    assert b2 != best and b2 >= 0
    b3, _, _ = hb.find_best_split(0, sums[0], sums[1], nd, *cfg, is_feature_used=np.zeros(hb.F, dtype=np.int8))
    assert b3 == -1
    hb.close()


@pytest.mark.parametrize("name", ["plain", "zero_missing", "nan"])
def test_leaf_partition_against_reference_fixture(lib_built, name):
    """DataPartition::Split on the device against the reference's own Dataset::Split: identical lists, order included."""
    import os
    from gpboost_amd import shim
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_ref.npz"))
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3 = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.set_fix_info(g[name + "_view_offset"], g[name + "_num_bin"], g[name + "_most_freq_bin"])
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    req, cnts, flat = g[name + "_part_req"], g[name + "_part_lte_count"], g[name + "_part_lte"]
    pos = 0
    for (f, th, dl), nl in zip(req, cnts):
        lte, gt = hb.split_leaf(leaf, f, th, dl)
        assert np.array_equal(lte, flat[pos:pos + nl]), (name, f, th, dl)
        assert np.array_equal(np.sort(np.concatenate([lte, gt])), leaf) and np.all(np.diff(gt) > 0)
        pos += nl
    # all rows (data_indices = NULL) and an unsorted index list: stable in the given order
    lte_all, gt_all = hb.split_leaf(None, 0, 20, 1)
    assert lte_all.size + gt_all.size == bins.shape[1] and np.all(np.diff(lte_all) > 0)
    perm = np.random.default_rng(1).permutation(leaf).astype(np.int32)
    lte_p, gt_p = hb.split_leaf(perm, 1, 30, 0)
    l0, g0 = hb.split_leaf(leaf, 1, 30, 0)
    inl = np.isin(perm, l0)
    assert np.array_equal(lte_p, perm[inl]) and np.array_equal(gt_p, perm[~inl])
    hb.close()


def test_categorical_split_search_and_partition_against_reference_fixture(lib_built, orc):
    """Round 5: FindBestThresholdCategoricalInner on device-resident histograms and DenseBin::SplitCategorical on the device, against the reference's
    own output (tests/golden/split_cat_ref.npz): end to end from the bins (build -> fix -> search) the device's SplitInfo is bit-identical to the
    oracle's on the device's own histogram, and has the reference's winner sets of bins (its sums to the build's 1e-9); the partitions are identical,
    order included."""
    import os
    from gpboost_amd import shim
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_cat_ref.npz"))
    name = "cat"
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3, is_cat = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"], g[name + "_is_categorical"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(2)
    hb.set_fix_info(g[name + "_view_offset"], g[name + "_num_bin"], g[name + "_most_freq_bin"])
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    same_sets = total = 0
    for ci, (cfg, cc) in enumerate(cases.SPLIT_CAT_CFGS):
        hb.set_categorical(is_cat, *cc)
        hb.set_regularisation(*(cfg[4:8] if len(cfg) > 4 else (0.0, 0.0, 0.0, 0.0)))
        for li, di in enumerate((None, leaf)):
            for hi, hs in enumerate((None, hess)):
                key = "%s_cfg%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums, ref = g[key + "_sums"], g[key + "_split"]
                nd = bins.shape[1] if di is None else di.size
                hb.set_gradients(grad, hs)
                hb.build_slot(0, di)
                hb.fix_slot(0, sums[0], sums[1])
                best, out, dl = hb.find_best_split(0, sums[0], sums[1], nd, *cfg[:4])
                raw = hb.get_slot(0)
                for f in np.flatnonzero(is_cat):
                    row, fl, bits = orc.find_best_split_cat(raw, g[name + "_view_offset"][f], g[name + "_num_bin"][f], meta3[f, 0], sums[0], sums[1], nd,
                                                            *cfg, cat_cfg=cc)
                    assert np.array_equal(out[f], row), (key, f, out[f], row)                 # bit-identical given the histogram
                    assert dl[f] == (fl & 1) and hb.last_splittable[f] == ((fl >> 1) & 1)
                    assert np.array_equal(hb.last_cat_bits[f], bits)
                    if np.isfinite(ref[f, 0]):
                        total += 1
                        rb = g[key + "_cat_bits"][f]
                        if np.array_equal(hb.last_cat_bits[f], rb):
                            same_sets += 1
                            np.testing.assert_allclose(out[f, [0, 4, 5, 6, 7, 8, 9]], ref[f, [0, 4, 5, 6, 7, 8, 9]], rtol=1e-9, atol=1e-9)
                            assert out[f, 1] == ref[f, 1]
                        else:
                            # the one void tie of the sorted search: both scan directions end at their cap with COMPLEMENTARY halves of the used bins --
                            # the same partition with left and right exchanged, mathematically equal gains, the winner decided by the last bits of the
                            # histogram sums (tests/cases.py, make_split_data("cat")).  Anything else is a defect.
                            assert not (hb.last_cat_bits[f] & rb).any() and out[f, 1] == ref[f, 1]
                            np.testing.assert_allclose(out[f, 0], ref[f, 0], rtol=1e-12)
                            np.testing.assert_allclose(out[f, [2, 3]], ref[f, [3, 2]], atol=4)
                num = np.flatnonzero(is_cat == 0)
                assert not hb.last_cat_bits[num].any()
                assert best == int(np.argmax(ref[:, 0]))
    # (the sorted search orders bins by sum_grad / (sum_hess + cat_smooth): two bins whose keys differ in the last bits of the histogram sums may
    #  swap; none does on this fixture)
    assert total >= 30 and same_sets >= total - 1
    hb.set_regularisation(0.0, 0.0, 0.0, 0.0)
    pos = 0
    for (f, th, dl_), w, nl in zip(g[name + "_part_req"], g[name + "_part_bits"], g[name + "_part_lte_count"]):
        lte, gt = hb.split_leaf(leaf, f, th, dl_, cat_bits=w)
        assert np.array_equal(lte, g[name + "_part_lte"][pos:pos + nl]), (f, w)
        assert np.array_equal(np.sort(np.concatenate([lte, gt])), leaf) and np.all(np.diff(gt) > 0)
        pos += nl
    assert pos == g[name + "_part_lte"].size
    hb.close()


@pytest.mark.parametrize("name", ["plain_l31", "plain_l15_reg", "nan_l20", "zero_missing_l12", "plain_l1", "plain_mds", "nan_smooth", "plain_all_reg", "plain_depth4",
                                  "cat_l15", "cat_defaults", "cat_onehot_l1", "cat_smooth", "efb_l15", "efb_rowwise_l15"])
@pytest.mark.parametrize("hi", [0, 1])
def test_device_primitives_grow_the_reference_tree(lib_built, name, hi):
    """The five device primitives (leaf histogram, FixHistogram, parent - smaller, split search, leaf partition), driven by the control
    flow of SerialTreeLearner::Train (tests/tree_harness.py), grow the tree the reference's own SerialTreeLearner grows on its own
    Dataset (tests/golden/tree_ref.npz): same splits, thresholds, default directions and counts; leaf values / gains to the summation
    order of the histogram build."""
    import os
    from gpboost_amd import shim
    from tests import cases
    from tests import tree_harness as th
    r5 = name in cases.TREE_CASES_R5       # round 5: categorical columns / bundled groups (the fixture's bins are the unbundled per-feature columns)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref_r5.npz" if r5 else "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    k = "%s_hess%d_" % (name, hi)
    hs = hess if hi else None
    be = th.GpuBackend(shim, g[k + "bins"], g[k + "group_num_bin"], g[k + "view_offset"], g[k + "num_bin"], g[k + "most_freq_bin"],
                       g[k + "meta3"], grad, hs, L, is_cat=g[k + "layout"][:, 3] if r5 else None, cat_cfg=cases.tree_cat_cfg(name))
    t = th.grow_tree(be, grad, hs, X.shape[0], L, cfg, max_depth=cases.tree_max_depth(name))
    be.close()
    assert t["num_leaves"] == int(g[k + "num_leaves"])
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count"):
        assert np.array_equal(t[key], g[k + key]), key
    if r5:
        assert np.array_equal(t["node_is_cat"], g[k + "node_is_cat"])
        assert np.array_equal(np.asarray(t["node_cat_bits"]).reshape(-1, 8), g[k + "node_cat_bits"])       # the same sets of bins go left
    np.testing.assert_allclose(t["leaf_value"], g[k + "leaf_value"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t["split_gain"], g[k + "split_gain"], rtol=1e-6)
    # default_left: for features with a missing-value type the reverse and the forward scan find the SAME split whenever the leaf holds
    # no missing rows; which of the two (mathematically equal) gains is larger is decided in the last bit of the histogram sums, so the
    # flag may differ from the reference's there.  Every such node must be one where the direction is void: both directions give the
    # identical partition of the node's rows (checked with the oracle's DenseBin::SplitInner).  Without missing-value types there is
    # one scan and the flag is exact.
    flipped = np.flatnonzero(t["default_left"] != g[k + "default_left"])
    if data in ("plain", "cat", "efb"):
        assert flipped.size == 0
    from oracle import orc
    bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
    for nd in flipped:
        f, thr, rows = int(t["split_feature_inner"][nd]), int(t["threshold_in_bin"][nd]), t["node_rows"][nd]
        parts = [orc.split_leaf(bins[f], gnb[f] - 1, meta3[f, 1], mfb[f], meta3[f, 2], dl, thr, rows) for dl in (0, 1)]
        assert np.array_equal(parts[0][0], parts[1][0]) and np.array_equal(parts[0][1], parts[1][1]), nd


@pytest.mark.parametrize("name", ["plain_l31", "plain_l15_reg", "nan_l20", "zero_missing_l12", "plain_l1", "plain_mds", "nan_smooth", "plain_all_reg", "plain_depth4",
                                  "cat_l15", "cat_defaults", "cat_onehot_l1", "cat_smooth", "efb_l15", "efb_rowwise_l15"])
@pytest.mark.parametrize("hi", [0, 1])
def test_resident_tree_grower_grows_the_reference_tree(lib_built, name, hi):
    """gpb_hip_hist_grow_tree (row lists of the leaves resident on the device, control flow in C++) against the reference's own
    SerialTreeLearner tree (tests/golden/tree_ref.npz) and against the single-step harness."""
    import os
    from gpboost_amd import shim
    from oracle import orc
    from tests import cases
    r5 = name in cases.TREE_CASES_R5
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref_r5.npz" if r5 else "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    n = X.shape[0]
    k = "%s_hess%d_" % (name, hi)
    hs = hess if hi else None
    bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
    is_cat = g[k + "layout"][:, 3] if r5 else np.zeros(bins.shape[0], dtype=np.int32)
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(L + 1)
    hb.set_fix_info(g[k + "view_offset"], g[k + "num_bin"], mfb)
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    if is_cat.any():
        hb.set_categorical(is_cat, *cases.tree_cat_cfg(name))
    hb.set_gradients(grad, hs)
    sg = float(np.cumsum(grad)[-1]); sh = float(np.cumsum(np.ones(n) if hs is None else hs)[-1])
    if len(cfg) > 4:                      # lambda_l1, max_delta_step, path_smooth (the grower tracks parent_output itself)
        hb.set_regularisation(cfg[4], cfg[5], cfg[6])
    hb.set_max_depth(cases.tree_max_depth(name))
    t = hb.grow_tree(L, sg, sh, *cfg[:4])
    assert t["num_leaves"] == int(g[k + "num_leaves"])
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count"):
        assert np.array_equal(t[key], g[k + key]), key
    np.testing.assert_allclose(t["leaf_value"], g[k + "leaf_value"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t["split_gain"], g[k + "split_gain"], rtol=1e-6)
    if data in ("plain", "cat", "efb"):
        assert np.array_equal(t["default_left"], g[k + "default_left"])
    if r5:       # categorical nodes: the reference's nodes and its sets of bins (Tree::SplitCategorical's cat_threshold_inner_)
        assert np.array_equal(t["node_is_cat"], g[k + "node_is_cat"]) and np.array_equal(t["node_cat_bits"], g[k + "node_cat_bits"])
        assert (data != "cat") or int(t["node_is_cat"].sum()) >= 5
    # leaf of every row: sizes equal the leaf counts, and replaying the tree's splits on the bins reproduces the labels
    dli = t["data_leaf_index"]
    assert np.array_equal(np.bincount(dli, minlength=t["num_leaves"]), t["leaf_count"])
    lab = np.zeros(n, dtype=np.int32)
    leaf_of_node = {0: 0}
    nleaves = 1
    for nd in range(t["num_leaves"] - 1):
        lf = leaf_of_node[nd]
        rows = np.flatnonzero(lab == lf).astype(np.int32)
        f = int(t["split_feature_inner"][nd])
        if is_cat[f]:
            lte, gt = orc.split_leaf_layout(bins[f], 1, gnb[f] - 1, False, meta3[f, 1], mfb[f], meta3[f, 2], 0, 0, True, t["node_cat_bits"][nd], rows)
        else:
            lte, gt = orc.split_leaf(bins[f], gnb[f] - 1, meta3[f, 1], mfb[f], meta3[f, 2], int(t["default_left"][nd]), int(t["threshold_in_bin"][nd]), rows)
        lab[gt] = nleaves
        if t["left_child"][nd] >= 0:
            leaf_of_node[int(t["left_child"][nd])] = lf
        if t["right_child"][nd] >= 0:
            leaf_of_node[int(t["right_child"][nd])] = nleaves
        nleaves += 1
    assert np.array_equal(lab, dli)
    # a second tree on the same handle (workspaces are reused) gives the same result
    t2 = hb.grow_tree(L, sg, sh, *cfg[:4])
    assert np.array_equal(t2["data_leaf_index"], dli) and np.array_equal(t2["threshold_in_bin"], t["threshold_in_bin"])
    # data-parallel form with a 1-rank communicator: the scale, every new histogram's integer totals and every left count go through ncclAllReduce
    hb.comm_init(shim.comm_unique_id(), 0, 1)
    hb.set_gradients(grad, hs)            # collective: the ranks agree the fixed-point scale
    t3 = hb.grow_tree(L, float("nan"), float("nan"), *cfg[:4])      # sharded form: the root sums come from the all-reduced integer totals
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "data_leaf_index"):
        assert np.array_equal(t3[key], t[key]), key                     # (default_left: void ties, see the harness test above)
    np.testing.assert_allclose(t3["leaf_value"], t["leaf_value"], rtol=1e-10, atol=1e-13)
    hb.close()


def test_tree_grower_with_a_column_sample_and_a_depth_limit(lib_built):
    """feature_fraction's per-tree column sample (gpb_hip_hist_set_feature_mask) and max_depth in the resident grower against the control flow of
    SerialTreeLearner::Train over the ORACLE's primitives (tests/tree_harness.py) with the same mask: masked columns never split, the rest of the
    tree is the reference's choice among the remaining ones."""
    import os
    from gpboost_amd import shim
    from oracle import orc
    from tests import cases
    from tests import tree_harness as th
    name, hi = "plain_l31", 0
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    n = X.shape[0]
    k = "%s_hess%d_" % (name, hi)
    bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
    F = bins.shape[0]
    mask = np.ones(F, dtype=np.int8)
    mask[int(g[k + "split_feature_inner"][0])] = 0              # the root's own choice is not available
    mask[F - 1] = 0
    ob = th.OracleBackend(orc, bins, gnb, g[k + "view_offset"], g[k + "num_bin"], mfb, meta3, grad, None)
    ref = th.grow_tree(ob, grad, None, n, L, cfg, max_depth=5, feature_mask=mask)
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(L + 1)
    hb.set_fix_info(g[k + "view_offset"], g[k + "num_bin"], mfb)
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    hb.set_gradients(grad, None)
    hb.set_feature_mask(mask)
    hb.set_max_depth(5)
    sg = float(np.cumsum(grad)[-1]); sh = float(n)
    t = hb.grow_tree(L, sg, sh, *cfg[:4])
    assert t["num_leaves"] == ref["num_leaves"] and t["num_leaves"] > 8
    assert not np.any(mask[t["split_feature_inner"]] == 0)
    for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "default_left"):
        assert np.array_equal(t[key], ref[key]), key
    np.testing.assert_allclose(t["leaf_value"], ref["leaf_value"], rtol=1e-9, atol=1e-12)
    hb.set_feature_mask(None); hb.set_max_depth(0)              # back to the unrestricted tree of the fixture
    t0 = hb.grow_tree(L, sg, sh, *cfg[:4])
    assert np.array_equal(t0["split_feature_inner"], g[k + "split_feature_inner"])
    hb.close()


def test_tree_grower_on_a_bag_of_rows(lib_built):
    """Bagging: the root of the tree holds a subset of the rows (gpb_hip_hist_set_root_rows) -- against SerialTreeLearner::Train's control flow over
    the ORACLE's primitives started from the same subset; rows outside the bag are labelled -1."""
    import os
    from gpboost_amd import shim
    from oracle import orc
    from tests import cases
    from tests import tree_harness as th
    name = "plain_l15_reg"
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree_ref.npz"))
    data, params, L, cfg = cases.tree_params(name)
    X, grad, hess, leaf = cases.make_split_data(data)
    n = X.shape[0]
    for hi in (0, 1):
        k = "%s_hess%d_" % (name, hi)
        hs = hess if hi else None
        bins, gnb, mfb, meta3 = g[k + "bins"], g[k + "group_num_bin"], g[k + "most_freq_bin"], g[k + "meta3"]
        bag = np.sort(np.random.default_rng(77).choice(n, size=int(0.7 * n), replace=False)).astype(np.int32)
        ob = th.OracleBackend(orc, bins, gnb, g[k + "view_offset"], g[k + "num_bin"], mfb, meta3, grad, hs)
        ref = th.grow_tree(ob, grad, hs, n, L, cfg, root_rows=bag)
        bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
        hb = shim.HistBuilder(bins, bo)
        hb.pool_resize(L + 1)
        hb.set_fix_info(g[k + "view_offset"], g[k + "num_bin"], mfb)
        hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
        hb.set_gradients(grad, hs)
        hb.set_root_rows(bag)
        sg = float(np.cumsum(grad[bag])[-1]); sh = float(np.cumsum((np.ones(n) if hs is None else hs)[bag])[-1])
        t = hb.grow_tree(L, sg, sh, *cfg[:4])
        assert t["num_leaves"] == ref["num_leaves"] and t["num_leaves"] > 4
        for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "default_left"):
            assert np.array_equal(t[key], ref[key]), key
        np.testing.assert_allclose(t["leaf_value"], ref["leaf_value"], rtol=1e-9, atol=1e-12)
        dli = t["data_leaf_index"]
        out = np.ones(n, dtype=bool); out[bag] = False
        assert np.all(dli[out] == -1) and np.all(dli[bag] >= 0)
        assert np.array_equal(np.bincount(dli[bag], minlength=t["num_leaves"]), t["leaf_count"])
        hb.set_root_rows(None)                                       # back to all rows: the fixture's tree
        t0 = hb.grow_tree(L, float(np.cumsum(grad)[-1]), float(np.cumsum(np.ones(n) if hs is None else hs)[-1]), *cfg[:4])
        assert np.array_equal(t0["split_feature_inner"], g[k + "split_feature_inner"])
        hb.close()


@pytest.mark.parametrize("name", ["plain", "zero_missing", "nan"])
def test_split_search_regularisation_paths_against_reference_fixture(lib_built, name):
    """lambda_l1 / max_delta_step / path_smooth with a given parent_output (gpb_hip_hist_set_regularisation): the device search is bit-identical
    to the oracle's on the device-built histogram, and agrees with the reference's own FindBestThreshold (tests/golden/split_ref.npz, the
    USE_L1 / USE_MAX_OUTPUT / USE_SMOOTHING instances) in feature, threshold and -- to the summation order of the histogram -- values."""
    import os
    from gpboost_amd import shim
    from oracle import orc
    from tests import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "split_ref.npz"))
    X, grad, hess, leaf = cases.make_split_data(name)
    bins, gnb, meta3 = g[name + "_bins"], g[name + "_group_num_bin"], g[name + "_meta3"]
    bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
    hb = shim.HistBuilder(bins, bo)
    hb.pool_resize(2)
    hb.set_fix_info(g[name + "_view_offset"], g[name + "_num_bin"], g[name + "_most_freq_bin"])
    hb.set_split_info(meta3[:, 0], meta3[:, 1], meta3[:, 2])
    for ci, cfg in enumerate(cases.SPLIT_CFGS_REG):
        hb.set_regularisation(cfg[4], cfg[5], cfg[6], cfg[7])
        for li, di in enumerate((None, leaf)):
            for hi, hs in enumerate((None, hess)):
                key = "%s_cfgr%d_leaf%d_hess%d" % (name, ci, li, hi)
                sums, ref, ref_dl = g[key + "_sums"], g[key + "_split"], g[key + "_default_left"]
                nd = bins.shape[1] if di is None else di.size
                hb.set_gradients(grad, hs)
                hb.build_slot(0, di)
                hb.fix_slot(0, sums[0], sums[1])
                best, out, dl = hb.find_best_split(0, sums[0], sums[1], nd, *cfg[:4])
                raw = hb.get_slot(0)
                obest, oout, odl = orc.find_best_split(raw, g[name + "_view_offset"], g[name + "_num_bin"], meta3[:, 0], meta3[:, 1],
                                                       meta3[:, 2], sums[0], sums[1], nd, *cfg)
                assert best == obest and np.array_equal(out, oout) and np.array_equal(dl, odl), key      # bit-identical given the histogram
                assert best == int(np.argmax(ref[:, 0])), key
                assert out[best, 1] == ref[best, 1], key
                np.testing.assert_allclose(out[best, [0, 4, 5, 6, 7, 8, 9]], ref[best, [0, 4, 5, 6, 7, 8, 9]], rtol=1e-9, atol=1e-9)
    hb.set_regularisation(0.0, 0.0, 0.0, 0.0)
    hb.close()
