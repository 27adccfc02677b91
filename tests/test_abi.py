"""CPU: the C-ABI library loads, exports every symbol the headers declare, and refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, macro):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)      # drop preprocessor lines (the macro's own definition)
    return sorted(set(re.findall(macro + r"\s+[\w\s\*]*?\b(\w+)\s*\(", txt)))


def test_headers_declare_something():
    assert len(_declared("gpb_hip.h", "GPB_HIP_EXPORT")) >= 20
    assert "GPB_EvalNegLogLikelihood" in _declared("gpboost_c_api_subset.h", "GPBOOST_C_EXPORT")


def test_library_exports_every_declared_symbol(lib_built):
    lib = ctypes.CDLL(lib_built)
    names = _declared("gpb_hip.h", "GPB_HIP_EXPORT") + _declared("gpboost_c_api_subset.h", "GPBOOST_C_EXPORT")
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "symbols declared in include/*.h but not exported: %s" % missing


def test_reference_signature_is_kept_for_on_path_functions():
    """The subset header must restate the reference prototypes verbatim (argument order and types)."""
    ref = "/root/reference/include/LightGBM/c_api.h"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")

    def protos(path, macro, names):
        txt = open(path).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", "", txt)
        out = {}
        for n in names:
            mm = re.search(macro + r"\s+int\s+" + n + r"\s*\((.*?)\)\s*;", txt, flags=re.S)
            assert mm, n
            args = [re.sub(r"\s+", " ", a.strip()) for a in mm.group(1).split(",")]
            out[n] = [a if "(*" in a else re.sub(r"\s*\b\w+$", "", a) for a in args]   # drop parameter names (not inside a function-pointer type)
        return out
    # every GPB_* function of the reference header (32) and the log hook: all are re-exported with the reference's own prototypes
    ref_txt = re.sub(r"/\*.*?\*/", "", open(ref).read(), flags=re.S)
    names = sorted(set(re.findall(r"GPBOOST_C_EXPORT\s+int\s+(GPB_\w+)\s*\(", ref_txt))) + ["LGBM_RegisterLogCallback"]
    assert len(names) == 33, names
    a = protos(ref, "GPBOOST_C_EXPORT", names)
    b = protos(os.path.join(ROOT, "include", "gpboost_c_api_subset.h"), "GPBOOST_C_EXPORT", names)
    for n in names:
        assert a[n] == b[n], n


def test_no_gpu_means_loud_failure_not_fallback(lib_built):
    import gpboost_amd
    if gpboost_amd.device_count() > 0:
        pytest.skip("a GPU is visible here")
    coords = np.random.default_rng(0).uniform(size=(50, 2))
    with pytest.raises(gpboost_amd.GPBoostError) as ei:
        gpboost_amd.GPModel(gp_coords=coords, cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    assert "no HIP device" in str(ei.value) or "no CPU fallback" in str(ei.value)
    with pytest.raises(gpboost_amd.GPBoostError):
        gpboost_amd.shim.VecchiaState(coords, 10)
    with pytest.raises(gpboost_amd.GPBoostError):
        gpboost_amd.shim.HistBuilder(np.zeros((3, 50), dtype=np.uint8), [0, 4, 8, 12])


def test_out_of_scope_models_are_rejected_with_a_message(lib_built):
    import gpboost_amd
    coords = np.random.default_rng(0).uniform(size=(50, 2))
    for kw in (dict(gp_approx="fitc"), dict(gp_approx="vecchia", cov_function="gaussian"),
               dict(gp_approx="vecchia", likelihood="bernoulli_logit"), dict(gp_approx="vecchia", cov_fct_shape=0.7),
               dict(gp_approx="vecchia", vecchia_ordering="time")):
        args = dict(gp_coords=coords, cov_function="matern", cov_fct_shape=1.5, num_neighbors=10)
        args.update(kw)
        with pytest.raises(gpboost_amd.GPBoostError) as ei:
            gpboost_amd.GPModel(**args)
        assert "hot path" in str(ei.value)


def test_product_never_imports_the_oracle():
    """The shipped package must not reference oracle/ (it is test infrastructure)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "gpboost_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libgpb_oracle" not in txt, f


def test_api_timeline_and_marks(lib_built):
    """GPB_HIP_API_TIMING=2 (round 6): the call timeline next to the per-entry-point table, with the caller's own marks (gpb_hip_api_mark) -- the tool that found the reference's
    per-leaf walks in route B's boosting iteration.  No device needed: marks and the report do not touch it.  Without the variable the report is an error
    and a mark a no-op."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "L = ctypes.CDLL(sys.argv[1])\n"
            "L.gpb_hip_get_last_error.restype = ctypes.c_char_p\n"
            "a = ctypes.c_char_p(b'MARK first'); b = ctypes.c_char_p(b'MARK second')\n"
            "assert L.gpb_hip_api_mark(a) == 0 and L.gpb_hip_api_mark(b) == 0\n"
            "rc = L.gpb_hip_api_timing_report(1)\n"
            "print('rc', rc, L.gpb_hip_get_last_error().decode() if rc else '')\n")
    on = subprocess.run([sys.executable, "-c", code, lib_built], env=dict(os.environ, GPB_HIP_API_TIMING="2"), capture_output=True, text=True)
    assert on.returncode == 0 and "rc 0" in on.stdout, on.stdout + on.stderr
    lines = [l for l in on.stderr.splitlines() if l.startswith("[gpb_hip api timeline]")]
    assert len(lines) >= 3 and "MARK first" in lines[1] and "MARK second" in lines[2], on.stderr
    off = subprocess.run([sys.executable, "-c", code, lib_built], env={k: v for k, v in os.environ.items() if k != "GPB_HIP_API_TIMING"}, capture_output=True, text=True)
    assert off.returncode == 0 and "rc -1" in off.stdout and "GPB_HIP_API_TIMING is not set" in off.stdout and "timeline" not in off.stderr, off.stdout + off.stderr
