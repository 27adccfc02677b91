"""CPU: the route-B integration files stay applicable and bind only what the library provides.

 * integration/reference_hip_seams.patch applies cleanly to the reference's own files (dry run on copies in a temp directory; skipped when
   /root/reference is absent, e.g. on the GPU box);
 * every gpb_hip_* function the patch and integration/hip_tree_learner.h call is declared in include/gpb_hip.h and exported by the library.
"""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "reference_hip_seams.patch")
LEARNER = os.path.join(ROOT, "integration", "hip_tree_learner.h")
REF = "/root/reference"


def _patched_files():
    return sorted(set(re.findall(r"^\+\+\+ b/(\S+)", open(PATCH).read(), flags=re.M)))


def test_patch_applies_to_the_reference(tmp_path):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this machine")
    if shutil.which("patch") is None:
        pytest.skip("patch(1) not installed")
    files = _patched_files()
    assert files, "the patch names no files"
    for f in files:
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(REF, f), dst)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout       # clean hunks, no fuzz


def _called(text):
    text = re.sub(r"//[^\n]*", "", text)
    return set(re.findall(r"\b(gpb_hip_\w+)\s*\(", text))


def _declared():
    txt = open(os.path.join(ROOT, "include", "gpb_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"GPB_HIP_EXPORT\s+[\w\s\*]*?\b(gpb_hip_\w+)\s*\(", txt))


def test_integration_files_call_only_declared_and_exported_functions(lib_built):
    added = "\n".join(l[1:] for l in open(PATCH).read().splitlines() if l.startswith("+") and not l.startswith("+++"))
    used = _called(added) | _called(open(LEARNER).read())
    assert len(used) >= 15, sorted(used)
    declared = _declared()
    assert not (used - declared), "called by the integration files but not declared in include/gpb_hip.h: %s" % sorted(used - declared)
    lib = ctypes.CDLL(lib_built)
    missing = [n for n in sorted(used) if not hasattr(lib, n)]
    assert not missing, missing


def test_learner_header_covers_what_integration_md_promises():
    """The whole-tree path of HIPTreeLearner hands over regularisation, depth limit, column sample and bag."""
    src = open(LEARNER).read()
    for fn in ("gpb_hip_hist_grow_tree", "gpb_hip_hist_last_tree_node_info", "gpb_hip_hist_set_regularisation", "gpb_hip_hist_set_max_depth",
               "gpb_hip_hist_set_feature_mask", "gpb_hip_hist_set_root_rows", "gpb_hip_hist_set_gradients"):
        assert fn in src, fn
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for fn in ("gpb_hip_hist_set_regularisation", "gpb_hip_hist_set_max_depth", "gpb_hip_hist_set_feature_mask", "gpb_hip_hist_set_root_rows"):
        assert fn in doc, fn
