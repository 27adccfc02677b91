"""Config 5's shape on one GPU (n = 1e6, d = 3, Matern-2.5, m = 40): kernel times of the likelihood and the gradient launch (HIP events inside
gpb_hip_vecchia_bench) and the gradient's values -- used for the A/B of the MT = 40 gradient instance (GPBOOST_AMD_LIB selects the library)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpboost_amd
from gpboost_amd import shim
gpboost_amd.set_device(0)
n, d, m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 3, 40
rng = np.random.default_rng(1)
co = rng.uniform(size=(n, d)); y = rng.standard_normal(n)
st = shim.VecchiaState(co, m)
st.find_neighbors()
st.set_y(y)
out = {"n": n, "d": d, "m": m, "lib": os.environ.get("GPBOOST_AMD_LIB", "default")}
for name, mode in (("nll", 0), ("grad", 2)):
    st.bench(mode, 2, 1.5, 9.0, 3, 3)
    t, k, terms = st.bench(mode, 2, 1.5, 9.0, 3, 10)
    out[name + "_kernel_ms"] = round(k, 4); out[name + "_terms"] = terms[: (3 if mode == 0 else 7)].tolist()
out["grad_over_nll"] = round(out["grad_kernel_ms"] / out["nll_kernel_ms"], 3)
print(json.dumps(out))
