"""Secondary kernels on the MI355X: exact-GP path timings (assembly GB/s, factorisation, solves), histogram
kernel timings (HIP events), PCIe-inclusive likelihood rate through GPB_EvalNegLogLikelihood, C5 configuration."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
from gpboost_amd import shim
from tests import cases

SECTIONS = set(sys.argv[1:]) or {"exact", "hist", "pcie", "c5"}
print("== exact GP (dense) ==", flush=True)
for n in ((2000, 8192, 16384) if "exact" in SECTIONS else ()):
    coords, y = cases.synthetic(n, 2, seed=1)
    st = shim.ExactState(coords); st.set_y(y)
    st.nll_terms(1, 10.0, 17.3)
    out, _, ms = st.nll_terms(1, 10.0, 17.3)
    np_ = ((n + 63) // 64) * 64
    nt = (np_ + 127) // 128
    wbytes = nt * (nt + 1) // 2 * 128 * 128 * 8
    print("n=%d: assembly %.3f ms (%.1f GB/s written, lower tiles) | cholesky %.3f ms (%.2f TFLOP/s) | solves %.3f ms | terms %s" % (
        n, ms[0], wbytes / ms[0] / 1e6, ms[1], n ** 3 / 3.0 / ms[1] / 1e9, ms[2], out), flush=True)
    st.close()

print("== histogram ==", flush=True)
for n, F in (((100000, 50), (10000000, 50)) if "hist" in SECTIONS else ()):
    rng = np.random.default_rng(0)
    bins = rng.integers(0, 255, size=(F, n), dtype=np.uint8)
    bo = (np.arange(F + 1) * 255).astype(np.int32)
    g = rng.standard_normal(n); hs = rng.uniform(0.5, 2, size=n)
    hb = shim.HistBuilder(bins, bo)
    leaf = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    for name, hess, di in (("all rows, const hess", None, None), ("half of rows (leaf), const hess", None, leaf), ("all rows, hessians", hs, None)):
        hb.set_gradients(g, hess)
        hb.build(di)
        ms = hb.bench(di, 1.0, 10)
        r = n if di is None else di.size
        byts = r * (F + 8 + (4 if di is not None else 0) + (8 if hess is not None else 0)) + F * 255 * 16
        print("n=%d F=%d %-32s %.4f ms/build -> %.1f GB/s algorithmic" % (n, F, name, ms, byts / ms / 1e6), flush=True)
    hb.close()

print("== PCIe-inclusive likelihood rate (host y uploaded every call) ==", flush=True)
if "pcie" not in SECTIONS and "c5" not in SECTIONS:
    sys.exit(0)
n, m = 1000000, 30
coords, y = cases.synthetic(n, 2, seed=1)
mdl = gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=m, vecchia_ordering="random", seed=1)
cp = np.array([0.1, 1.0, 0.1])
mdl.neg_log_likelihood(cp, y)
t0 = time.perf_counter()
for k in range(10):
    mdl.neg_log_likelihood(cp * (1 + 0.001 * k), y)
dt = (time.perf_counter() - t0) / 10
print("GPB_EvalNegLogLikelihood incl. host permutation + H2D of y: %.3f ms/eval -> %.1f evals/s" % (dt * 1e3, 1 / dt), flush=True)
t0 = time.perf_counter()
for k in range(5):
    mdl.neg_log_likelihood_and_gradient(cp * (1 + 0.001 * k), y)
dt = (time.perf_counter() - t0) / 5
print("nll + gradient incl. host permutation + H2D of y: %.3f ms/eval" % (dt * 1e3), flush=True)

print("== config 5 shape on one GPU: n=1e6, d=3, Matern-2.5, m=40 ==", flush=True)
coords, y = cases.synthetic(n, 3, seed=1)
st = shim.VecchiaState(coords, 40)
t0 = time.perf_counter(); st.find_neighbors(); print("neighbour search %.2f s" % (time.perf_counter() - t0), flush=True)
st.set_y(y)
ms_tot, ms_k, out = st.bench(0, 2, 10.0, 22.36, 2, 10)
ms_tot_g, ms_k_g, _ = st.bench(2, 2, 10.0, 22.36, 1, 3)
B = n * (4 * 40 + 8 * 3 * 41 + 8 * 41)
print("nll %.3f ms/eval (kernel %.3f ms, %.1f GB/s algorithmic) | grad kernel %.3f ms" % (ms_tot / 10, ms_k, B / ms_k / 1e6, ms_k_g), flush=True)
