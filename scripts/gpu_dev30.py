"""Dev iteration on the m=30 kernels (GPBOOST_AMD_LIB selects the library): parity vs oracle + timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
from gpboost_amd import shim
from oracle import orc
from tests import cases
print("lib:", os.environ.get("GPBOOST_AMD_LIB"), flush=True)
for (n, d, m, ct) in [(3000, 2, 30, 0), (3000, 3, 25, 2), (2500, 2, 22, 1), (40, 2, 30, 0)]:
    coords, y = cases.synthetic(n, d, seed=n)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 1)
    st = shim.VecchiaState(co, m); st.set_neighbors(nn); st.set_y(y[perm])
    var = 10.0; a = 10.0 * [1, 3 ** .5, 5 ** .5][ct]
    out, g = orc.vecchia_nll_grad(co, nn, ct, np.array([0.1, var, a]), y[perm])
    t3 = st.nll_terms(ct, var, a); t7 = st.grad_terms(ct, var, a)
    gg = shim.grad_from_terms(n, t7, 0.1)
    st.factor(ct, var, a); A, D, u = st.get_factor()
    Ao, Do, _ = orc.vecchia_factor(co, nn, ct, var, a)
    print("n=%d d=%d m=%d cov=%d: yPy rel %.1e logdet rel %.1e grad rel %s | A max %.1e D rel %.1e" % (
        n, d, m, ct, abs(t3[0] - out[0]) / abs(out[0]), abs(t3[1] - out[1]) / abs(out[1]),
        np.abs(gg - g) / np.abs(g).max(), np.abs(A - Ao).max(), np.abs(D / Do - 1).max()), flush=True)
n, m = 1000000, 30
coords, y = cases.synthetic(n, 2, seed=1)
st = shim.VecchiaState(coords, m); st.find_neighbors(); st.set_y(y)
st.bench(0, 0, 10.0, 10.0, 1, 200)     # clocks up (bench.py does the same before its warm-up)
for ct in (0, 2):
    ms_tot, ms_k, out = st.bench(0, ct, 10.0, 10.0, 3, 40)
    ms_tot_g, ms_k_g, out_g = st.bench(2, ct, 10.0, 10.0, 3, 20)
    print("cov=%d nll: %.3f ms/eval total, kernel %.3f ms | grad: kernel %.3f ms | terms %s" % (ct, ms_tot / 40, ms_k, ms_k_g, out[:3]), flush=True)
