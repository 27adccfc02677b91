"""are the Newton iteration counts of the auxiliary-parameter likelihoods at config 4's size (t: 16-18 steps, gamma: 4, lognormal: 1 -- profiles/r05_ad_*,
profiles/r06_a_trace_aux_*) the REFERENCE's own on the same data?  Runs the unmodified reference (oracle/_ref, CPU, iterative / vadu = what the device path implements) on the
data of scripts/gpu_r6_targets.py aux:<lik>, one FRESH evaluation at (1.0, 0.1) with the default and one with tightened thresholds (default auxiliary parameters), prints Likelihood::num_it_mode_finding_
(oracle/ref_driver.cpp: refdrv_num_it_mode_finding) and the value, and keeps both in tests/golden/config4_size_aux_ref.json (tests/test_atsize_gpu.py compares the device with them).  Build container only (minutes per likelihood on 8 cores)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refdrv  # noqa: E402

n, m = 100000, 30
OUT = {}
for lik in (sys.argv[1:] or ["lognormal", "gamma", "t"]):
    rng = np.random.default_rng(7)
    cc = rng.uniform(size=(n, 2))
    eta = np.sin(4 * cc[:, 0]) + np.cos(3 * cc[:, 1])
    if lik == "t":
        y = 0.8 * eta + 0.35 * rng.standard_t(4.0, size=n)
    elif lik == "gamma":
        y = rng.gamma(2.0, np.exp(0.5 * eta) / 2.0)
    else:
        y = np.exp(0.5 * eta + np.sqrt(0.2) * rng.standard_normal(n))
    fn = refdrv._lib().refdrv_num_it_mode_finding
    fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
    for key, cfg in (("default", {}), ("tight", dict(cg_delta_conv=1e-6, delta_conv_mode_finding=1e-13))):
        mdl = refdrv.RefCAPIModel(cc, "exponential", 0.5, m, "random", 1, threads=8, likelihood=lik, matrix_inversion_method="iterative")
        if cfg:
            mdl.set_optim_config(**cfg)
        t0 = time.perf_counter()
        v = mdl.neg_log_likelihood(np.array([1.0, 0.1]), y)
        dt = time.perf_counter() - t0
        nit = fn(mdl.h)
        print("reference, %s at config 4's size (n = %d, m = %d), fresh evaluation at (1.0, 0.1), %s thresholds %s: negll %.8f, Newton iterations of the mode finding (num_it_mode_finding_) = %d, %.1f s on 8 threads"
              % (lik, n, m, key, cfg, v, nit, dt), flush=True)
        OUT[lik + "_" + key] = dict(negll=v, newton_it=int(nit), seconds_8_threads=dt, thresholds=cfg)
        del mdl
import json  # noqa: E402
path = os.path.join(ROOT, "tests", "golden", "config4_size_aux_ref.json")
old = json.load(open(path)) if os.path.exists(path) else {}
old.update(OUT)
json.dump(old, open(path, "w"), indent=1, sort_keys=True)
