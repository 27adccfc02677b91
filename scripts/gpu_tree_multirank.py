"""Data-parallel tree grower with several ranks on ONE device (in-process group: the ranks are threads, the transport is the library's local one): the two
exchanges of gpb_hip_hist_grow_tree below the root -- 'feature_blocks' (reduce-scatter of the integer totals by feature block, every rank searches its own
features, best-split exchange; DataParallelTreeLearner's scheme) against 'allreduce' (every rank all-reduces every histogram and searches everything) -- on
config 3's shape (n = 1e5, F = 50, 255 bins, 31 leaves).  Identical trees are asserted; the times are those of W ranks time-sharing one GPU through a transport
with host barriers, i.e. a statement about the PROTOCOL's work (messages, launches, synchronisations), not a scaling number."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd                      # noqa: E402
from gpboost_amd import shim            # noqa: E402

gpboost_amd.set_device(0)
n, F, NB, L = 100000, 50, 255, 31
rng = np.random.default_rng(3)
X = rng.uniform(size=(n, F))
bins = np.minimum((X * (NB - 1)).astype(np.int64) + 1, NB - 1).astype(np.uint8).T.copy()
gnb = np.full(F, NB, dtype=np.int32)
bo = np.concatenate([[0], np.cumsum(gnb)]).astype(np.int32)
grad = np.sin(4 * X[:, 0]) + X[:, 1] ** 2 + 0.5 * rng.standard_normal(n)
voff = (bo[:-1] + 1).astype(np.int32); nbin = np.full(F, NB, dtype=np.int32); mfb = np.zeros(F, dtype=np.int32)
REPS = 20
out = {}
ref = None
for W in (1, 2, 4, 8):
    owner = rng.integers(0, W, size=n)
    parts = [np.flatnonzero(owner == r) for r in range(W)]
    for exch in ("feature_blocks", "allreduce"):
        grp = shim.LocalGroup(W)

        def rank(r):
            rows = parts[r]
            hb = shim.HistBuilder(np.ascontiguousarray(bins[:, rows]), bo)
            hb.pool_resize(L + 1)
            hb.set_fix_info(voff, nbin, mfb)
            hb.set_split_info(np.ones(F, dtype=np.int32), np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32))
            hb.comm_init_local(grp, r)
            hb.set_feature_block_exchange(exch == "feature_blocks")
            hb.set_gradients(grad[rows], None)
            t = hb.grow_tree(L, float("nan"), float("nan"), 0.0, 20, 1e-3, 0.0, want_leaf_index=False)
            t0 = time.perf_counter()
            for _ in range(REPS):
                hb.grow_tree(L, float("nan"), float("nan"), 0.0, 20, 1e-3, 0.0, want_leaf_index=False)
            dt = (time.perf_counter() - t0) / REPS
            hb.close()
            return t, dt
        res = grp.run(rank)
        grp.close()
        t = res[0][0]
        if ref is None:
            ref = t
        for key in ("split_feature_inner", "threshold_in_bin", "left_child", "right_child", "internal_count", "leaf_count", "leaf_value", "split_gain"):
            for rr in res:
                assert np.array_equal(rr[0][key], ref[key]), (W, exch, key)
        ms = 1e3 * max(rr[1] for rr in res)
        out["W%d_%s_ms_per_tree" % (W, exch)] = round(ms, 3)
        print("W = %d, %-14s: %.2f ms per 31-leaf tree (max over ranks); tree identical to the one-rank tree" % (W, exch, ms), file=sys.stderr, flush=True)
bins_total = int(bo[-1])
out["message_per_leaf_bytes"] = {"allreduce_3_words": 24 * bins_total, "reduce_scatter_3_words_per_rank_W8": 24 * bins_total // 8, "best_split_exchange": 2 * 16 * 24 * 8}
print(json.dumps(out))
