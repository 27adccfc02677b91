"""Histogram build timing on the MI355X (kernels only, HIP events): root pass and a gathered leaf, constant / per-row hessians."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd          # noqa: E402
from gpboost_amd import shim   # noqa: E402

gpboost_amd.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
F, nb = 50, 255
rng = np.random.default_rng(2)
bins = rng.integers(0, nb, size=(F, n), dtype=np.uint8)
bo = (np.arange(F + 1) * nb).astype(np.int32)
hb = shim.HistBuilder(bins, bo)
g = rng.standard_normal(n)
hb.set_gradients(g, None)
hb.bench(None, 1.0, 3)
ms = hb.bench(None, 1.0, 10)
byt = n * (F + 8 + 4) + F * nb * 16
print("root, const hess: %.3f ms  %.0f GB/s algorithmic" % (ms, byt / ms / 1e6))
leaf = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
hb.bench(leaf, 1.0, 2)
ms = hb.bench(leaf, 1.0, 10)
print("leaf n/2 gathered: %.3f ms" % ms)
hb.set_gradients(g, np.abs(g) + 0.1)
hb.bench(None, 1.0, 2)
ms = hb.bench(None, 1.0, 10)
print("root, per-row hess: %.3f ms" % ms)
# correctness spot check against numpy
hist, cnt = hb.build(None)
f = 7
ref = np.bincount(bins[f], weights=g, minlength=nb)
print("max abs err feature 7:", np.abs(hist[bo[f]:bo[f + 1], 0] - ref).max(), "counts ok:", np.array_equal(cnt[bo[f]:bo[f + 1]], np.bincount(bins[f], minlength=nb)))
