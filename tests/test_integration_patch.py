"""integration/reference_hip_seams.patch (INTEGRATION.md route B) applies cleanly (no fuzz) to the reference tree it was made against and touches only the
files INTEGRATION.md names; the seams carry the USE_HIP_GP build flag (integration/Makefile.routeB compiles the patched translation units, tests/test_routeB_seams_cpu.py
and tests/test_routes_gpu.py run them)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "reference_hip_seams.patch")
FILES = ["include/GPBoost/likelihoods.h", "include/GPBoost/re_model_template.h", "src/GPBoost/Vecchia_utils.cpp",
         "src/LightGBM/treelearner/data_partition.hpp", "src/LightGBM/treelearner/tree_learner.cpp"]


def _patched_files():
    return [l.split()[1][2:] for l in open(PATCH) if l.startswith("+++ b/")]


def test_patch_touches_only_the_files_named_in_integration_md():
    assert _patched_files() == FILES
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in FILES:
        assert os.path.basename(f) in doc, f


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_patch_applies_cleanly_to_the_reference(tmp_path):
    for f in FILES:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
    # every hunk is new text behind the build flag or names it: the flag appears in each patched file, and the reference's own CUDA switch is untouched
    for f in FILES:
        text = open(tmp_path / f, encoding="utf-8", errors="replace").read()
        assert "USE_HIP_GP" in text or f.endswith("data_partition.hpp"), f      # (data_partition.hpp gains one member function, used by the HIP learner only)
        assert text.count("USE_CUDA_GP") == open(os.path.join(REF, f), encoding="utf-8", errors="replace").read().count("USE_CUDA_GP"), f
