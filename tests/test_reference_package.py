"""CPU: the reference's OWN Python package (python-package/gpboost, unmodified, read from /root/reference) loads lib_gpboost_amd.so.

Drop-in route A of INTEGRATION.md: the only thing that changes for the package is where `find_lib_path()` points.  The package registers
its logger at import (basic.py:117-129 -> LGBM_RegisterLogCallback) and binds GPB_* functions lazily by name (basic.py:5206-7118); every
one of them must resolve.  Without a GPU a model cannot be created -- that has to fail loudly through the package's own error path
(GPBoostError from LGBM_GetLastError), never fall back.  What the package computes THROUGH this library on the MI355X is recorded by
scripts/gpu_reference_package.py (profiles/r02_*_reference_package_on_mi355x.log).  Skipped where /root/reference does not exist."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PKG = "/root/reference/python-package"

DRIVER = r'''
import json, re, sys, types
sys.modules.setdefault("optuna", types.ModuleType("optuna"))       # optional dependency of the package, absent here
fake = types.ModuleType("gpboost.libpath")                          # route A: libpath points at this library
fake.find_lib_path = lambda: [sys.argv[1]]
sys.modules["gpboost.libpath"] = fake
sys.path.insert(0, sys.argv[2])
import numpy as np
import gpboost as gpb
from gpboost import basic
src = open(basic.__file__).read()
used = sorted(set(re.findall(r"_LIB\.(GPB_\w+|LGBM_GetLastError|LGBM_RegisterLogCallback)", src)))
missing = [n for n in used if not hasattr(basic._LIB, n)]
out = {"used": used, "missing": missing}
rng = np.random.default_rng(0)
try:
    gpb.GPModel(gp_coords=rng.uniform(size=(50, 2)), cov_function="exponential", gp_approx="vecchia", num_neighbors=10)
    out["create"] = "ok"
except gpb.basic.GPBoostError as e:
    out["create"] = "GPBoostError: " + str(e)
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(REF_PKG), reason="reference tree not present on this machine")
def test_reference_python_package_binds_every_symbol(lib_built):
    import json
    r = subprocess.run([sys.executable, "-c", DRIVER, lib_built, REF_PKG], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out["used"]) >= 32 and "GPB_CreateREModel" in out["used"]
    assert out["missing"] == [], "the reference's GPModel binds symbols this library does not export: %s" % out["missing"]
    import gpboost_amd
    if gpboost_amd.device_count() == 0:
        assert out["create"].startswith("GPBoostError:") and "no CPU fallback" in out["create"], out["create"]
    else:
        assert out["create"] == "ok"
