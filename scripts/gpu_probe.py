"""First-contact diagnostics on the MI355X box: DPP self-test, small parity, timings.  Prints, never asserts."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpboost_amd
from gpboost_amd import shim
from oracle import orc
from tests import cases


def step(name, fn):
    t = time.time()
    try:
        r = fn()
        print("[ok  ] %-40s %.3fs %s" % (name, time.time() - t, "" if r is None else r), flush=True)
        return r
    except Exception as e:
        print("[FAIL] %-40s %s: %s" % (name, type(e).__name__, e), flush=True)
        traceback.print_exc()


print("devices:", gpboost_amd.device_count(), flush=True)
step("dpp selftest", gpboost_amd.selftest)


def small(n, d, m, ct, ordering="random"):
    coords, y = cases.synthetic(n, d, seed=n)
    perm, co, nn = orc.vecchia_setup(coords, m, ordering, 1)
    st = shim.VecchiaState(co, m)
    st.find_neighbors()
    nn_g = st.get_neighbors()
    st.set_y(y[perm])
    var, a = 10.0, 10.0 * (1.0 if ct == 0 else np.sqrt(3.) if ct == 1 else np.sqrt(5.))
    t = st.grad_terms(ct, var, a)
    t3 = st.nll_terms(ct, var, a)
    out, g = orc.vecchia_nll_grad(co, nn, ct, np.array([0.1, var, a]), y[perm])
    A, D, Ag, Dg, bad = orc.vecchia_factor(co, nn, ct, var, a, grad=True)
    gg = shim.grad_from_terms(n, t, 0.1)
    return "nn_equal=%s yPy rel %.2e logdet rel %.2e | nllterms rel %.2e %.2e | grad %s vs %s" % (
        np.array_equal(nn, nn_g), abs(t[0] - out[0]) / abs(out[0]), abs(t[1] - out[1]) / abs(out[1]),
        abs(t3[0] - out[0]) / abs(out[0]), abs(t3[1] - out[1]) / abs(out[1]), gg, g)


for args in [(500, 2, 10, 0), (2000, 2, 30, 0), (2000, 3, 40, 2), (1000, 1, 5, 1), (3000, 2, 20, 1)]:
    step("parity n=%d d=%d m=%d cov=%d" % args, lambda a=args: small(*a))


def timing(n, d, m, ct):
    coords, y = cases.synthetic(n, d, seed=1)
    st = shim.VecchiaState(coords, m)
    t0 = time.time(); st.find_neighbors(); t_nn = time.time() - t0
    st.set_y(y)
    ms_tot, ms_k, out = st.bench(0, ct, 10.0, 10.0, 2, 10)
    ms_tot_g, ms_k_g, out_g = st.bench(2, ct, 10.0, 10.0, 1, 5)
    B = n * (4 * m + 8 * d * (m + 1) + 8 * (m + 1))
    return "nn search %.2fs | nll: %.3f ms/eval (kernel %.3f ms, %.1f GB/s algorithmic) | grad: %.3f ms/eval (kernel %.3f) | terms %s" % (
        t_nn, ms_tot / 10, ms_k, B / ms_k / 1e6, ms_tot_g / 5, ms_k_g, out[:3])


for args in [(100000, 2, 30, 0), (1000000, 2, 30, 0), (1000000, 3, 40, 2)]:
    step("timing n=%d d=%d m=%d cov=%d" % args, lambda a=args: timing(*a))


def hist_timing(n, F):
    rng = np.random.default_rng(0)
    bins = rng.integers(0, 255, size=(F, n)).astype(np.uint8)
    bo = (np.arange(F + 1) * 255).astype(np.int32)
    g = rng.standard_normal(n)
    hb = shim.HistBuilder(bins, bo); hb.set_gradients(g, None)
    hb.build(None)
    t0 = time.time()
    for _ in range(5):
        hist, cnt = hb.build(None)
    dt = (time.time() - t0) / 5
    hg, hc, hh = orc.hist_build(bins, bo, None, g, None)
    return "%.3f ms/build (host wall incl. D2H) counts_equal=%s grad maxdiff %.2e" % (dt * 1e3, np.array_equal(cnt, hc), np.abs(hist[:, 0] - hg).max())


step("hist n=1e5 F=50", lambda: hist_timing(100000, 50))
step("hist n=1e7 F=50", lambda: hist_timing(10000000, 50))
