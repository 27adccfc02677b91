"""Where does the per-evaluation host overhead go? (n = 1e6 and an 8-GPU-sized shard of 125k points)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpboost_amd import shim
from tests import cases
n, m = 1000000, 30
coords, y = cases.synthetic(n, 2, seed=1)
st = shim.VecchiaState(coords, m); st.find_neighbors(); st.set_y(y)
for (i0, i1) in ((0, n), (0, 125008)):
    st.set_shard(i0, i1)
    for _ in range(3):
        st.nll_terms(0, 10.0, 10.0)
    K = 200
    t0 = time.perf_counter()
    for k in range(K):
        st.nll_terms(0, 10.0 * (1 + 1e-3 * (k % 5)), 10.0)
    dt = (time.perf_counter() - t0) / K
    ms_tot, ms_k, _ = st.bench(0, 0, 10.0, 10.0, 3, 50)
    print("shard %d points: python loop %.1f us/eval | back-to-back on stream %.1f us/eval | point kernel %.1f us" % (
        i1 - i0, dt * 1e6, ms_tot / 50 * 1e3, ms_k * 1e3), flush=True)
