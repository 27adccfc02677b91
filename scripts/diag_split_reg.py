"""Diagnostic: the split search of the device against the oracle's at every search of a harness-grown tree (a TREE_CASES entry)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpboost_amd import shim
from oracle import orc
from tests import cases
from tests import tree_harness as th

name = sys.argv[1] if len(sys.argv) > 1 else "plain_mds"
g = np.load(os.path.join(ROOT, "tests", "golden", "tree_ref.npz"))
data, params, L, cfg = cases.tree_params(name)
X, grad, hess, leaf = cases.make_split_data(data)
k = "%s_hess0_" % name
args = (g[k + "bins"], g[k + "group_num_bin"], g[k + "view_offset"], g[k + "num_bin"], g[k + "most_freq_bin"], g[k + "meta3"], grad, None)
gb = th.GpuBackend(shim, *args, L)
ob = th.OracleBackend(orc, *args)


class Both(object):
    F = gb.F

    def build_fix(self, *a): gb.build_fix(*a); ob.build_fix(*a)
    def subtract(self, *a): gb.subtract(*a); ob.subtract(*a)
    def partition(self, *a): return ob.partition(*a)

    def search(self, slot, sg, sh, cnt, cfg, used, po=0.0):
        o1, d1, s1 = gb.search(slot, sg, sh, cnt, cfg, used, po)
        o2, d2, s2 = ob.search(slot, sg, sh, cnt, cfg, used, po)
        hd = gb.hb.get_slot(slot) if hasattr(gb.hb, "get_slot") else None
        bad = np.flatnonzero(~np.all((o1 == o2) | (np.isinf(o1) & np.isinf(o2)), axis=1))
        print("search slot %d cnt %d: %d features differ; splittable equal %s" % (slot, cnt, bad.size, np.array_equal(s1, s2)))
        for f in bad[:3]:
            print("  f", f, "gpu", o1[f], "\n      orc", o2[f])
        if hd is not None:
            print("  hist max abs diff", np.abs(hd - ob.slots[slot]).max())
        return o2, d2, s2


t = th.grow_tree(Both(), grad, None, X.shape[0], L, cfg)
