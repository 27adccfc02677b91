"""GPU (MI355X): the two drop-in routes of INTEGRATION.md, run where the driver can see them.

Route A -- the reference's OWN, unmodified Python package (a scratch copy under oracle/_ref/refpkg, made by `make -C oracle ref` from
/root/reference/python-package; git-ignored, it travels to the GPU box with the snapshot) with `find_lib_path()` pointing at
gpboost_amd/lib_gpboost_amd.so: its GPModel creates, evaluates, fits, predicts through this library and reproduces the R suite's goldens
(124.2252524; the 378-iteration fit; the prediction goldens; a fit with a linear regression term; a Bernoulli-logit evaluation).
Binding: python-package/gpboost/basic.py:117-129, 5206-7118.

Route B -- the reference's own HOST code (REModel, its optimiser, Booster / GBDT / SerialTreeLearner) compiled with
integration/reference_hip_seams.patch + integration/hip_tree_learner.h against this library (integration/Makefile.routeB ->
integration/_build/lib_gpboost_hip.so): GPU_use = true reproduces GPU_use = false (likelihood 1e-8, same optimiser iterations, predictions),
device_type = gpu reproduces device_type = cpu after 20 boosting iterations in 9 configurations (seam: tree_learner.cpp:15-52).

Both run as subprocesses of scripts/ (they replace modules / load a second library); skipped only where the prebuilt files are absent.
Nothing here reads /root/reference at run time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFPKG = os.path.join(ROOT, "oracle", "_ref", "refpkg", "gpboost", "basic.py")
HIPLIB = os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so")


_LAST_STDERR = [""]


def _run(args, ok_line, timeout):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    _LAST_STDERR[0] = r.stderr
    tail = (r.stdout[-4000:] + "\n--- stderr ---\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert ok_line in r.stdout, tail
    return r.stdout


@pytest.mark.skipif(not os.path.isfile(REFPKG), reason="oracle/_ref/refpkg (scratch copy of the reference's Python package) not built")
def test_route_a_reference_python_package_reproduces_the_r_goldens_on_the_device(lib_built):
    out = _run(["scripts/gpu_reference_package.py"], "REFERENCE PACKAGE ON MI355X: OK", 900)
    assert "iterations 378" in out and "124.2252524" in out


@pytest.mark.skipif(not os.path.isfile(HIPLIB), reason="integration/_build/lib_gpboost_hip.so (route-B build of the reference) not built")
def test_route_b_reference_remodel_with_gpu_use_reproduces_its_cpu_path(lib_built):
    out = _run(["scripts/gpu_routeB.py", "--test", "--gp-only"], "ROUTE B ON MI355X: OK", 1500)
    assert out.count("GPU_use=true reproduces the CPU path of the same build") == 2
    # round 4: the Laplace seams (Bernoulli-logit and Poisson, n = 5000: evaluations and lbfgs fits) against the CPU path's stored values, and the GPBoost loop
    # (round 5: + the same seams with cg_preconditioner_type = "pivoted_cholesky")
    # (round 5, second widening: + gamma, negative_binomial, beta and t with their auxiliary parameters estimated, and gamma with pivoted_cholesky -- 8 legs in all)
    assert out.count("(mode finding, stochastic log-determinant and gradient on the device) reproduces the CPU path of the same build") == 8
    assert "Laplace bernoulli_logit:pivoted_cholesky n=5000: GPU_use=true" in out
    for lik in ("gamma", "negative_binomial", "beta", "t", "gamma:pivoted_cholesky"):
        assert "Laplace %s n=5000: GPU_use=true" % lik in out
    assert "y_aux and Newton leaf values from the resident factor) reproduces the CPU path" in out


@pytest.mark.skipif(not os.path.isfile(HIPLIB), reason="integration/_build/lib_gpboost_hip.so (route-B build of the reference) not built")
def test_route_b_reference_booster_with_device_type_gpu_reproduces_device_type_cpu(lib_built):
    out = _run(["scripts/gpu_routeB.py", "--test", "--trees-only"], "ROUTE B ON MI355X: OK", 1500)
    assert "device_type=gpu (HIPTreeLearner, whole trees) reproduces device_type=cpu" in out
    # round 5: categorical columns and columns the reference bundles (EFB): whole trees on the device from per-feature columns, no CPU-histogram fallback
    assert "device_type=gpu (HIPTreeLearner, whole trees, per-feature columns) reproduces device_type=cpu" in out
    both = out + "\n" + _LAST_STDERR[0]
    assert "per-feature (unbundled) columns" in both and "categorical features searched on the GPU" in both
    assert "histograms stay on the CPU" not in both


CONFIG3_REF = os.path.join(ROOT, "tests", "golden", "config3_loop_ref.npz")


@pytest.mark.skipif(not (os.path.isfile(HIPLIB) and os.path.isfile(CONFIG3_REF)), reason="route-B build or tests/golden/config3_loop_ref.npz absent")
def test_config3_whole_boosting_loop_reproduces_the_cpu_path(lib_built):
    """BASELINE config 3 as a whole loop: 10 boosting iterations at n = 1e5 x 50 features x 255 bins with the Vecchia GP trained inside the loop, through
    the reference's own Booster with (i) GPU_use = true and (ii) GPU_use = true + device_type = gpu, against the predictions and covariance parameters the same
    build's CPU path left in tests/golden/config3_loop_ref.npz (scripts/gpu_config3_loop.py --make-ref): predictions 1e-8 of their scale, parameters 1e-6."""
    out = _run(["scripts/gpu_config3_loop.py"], "CONFIG 3 WHOLE LOOP ON MI355X: OK", 900)
    assert out.count("max |prediction - CPU path|") == 2
