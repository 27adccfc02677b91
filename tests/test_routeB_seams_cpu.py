"""INTEGRATION.md route B without a GPU: the route-B build of the reference (integration/_build/lib_gpboost_hip.so = the reference's own host code with
integration/reference_hip_seams.patch) with tests/mock_shim's CPU restatement of gpb_hip_* PRELOADED in place of lib_gpboost_amd.so.  What runs is
the patched host code -- HipCreateVecchiaStates, HipCalcCovFactorVecchia (B refilled through the cached pattern), the fused evaluations inside the
optimiser, CalcYAux after a new response at unchanged parameters, NewtonUpdateLeafValues, and the Laplace seams of a Bernoulli-logit model
(Likelihood::FindModePostRandEffCalcMLLVecchia, its gradient, ResetModeToPreviousValue; evaluations and an lbfgs fit) -- and GPU_use = true must
reproduce GPU_use = false of the same build.  The MI355X run of the same script is scripts/gpu_routeB.py (profiles/r04_*_routeB.log)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPLIB = os.path.join(ROOT, "integration", "_build", "lib_gpboost_hip.so")
MOCK = os.path.join(ROOT, "tests", "mock_shim", "libgpb_c_api_on_oracle_TEST_ONLY.so")


@pytest.mark.skipif(not os.path.isfile(HIPLIB), reason="integration/_build/lib_gpboost_hip.so (route-B build of the reference) not built")
def test_patched_reference_host_code_on_the_cpu_restatement_of_the_shim():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "mock_shim")], check=True)
    env = dict(os.environ, LD_PRELOAD=MOCK, OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_routeB.py"), "--cpu-mock"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "ROUTE B SEAMS ON THE CPU RESTATEMENT: OK" in out.stdout
    assert "HIP device detected" in out.stdout + out.stderr      # the seams were active (the mock reports one device)
    assert "Laplace bernoulli_logit" in out.stdout and "reproduces the CPU path of the same build" in out.stdout
    # round 5: the same seams with cg_preconditioner_type = "pivoted_cholesky" (HipEligible admits it; the host's PivotedCholsekyFactorizationSigma is skipped)
    assert "Laplace bernoulli_logit:pivoted_cholesky n=800: GPU_use=true (mode finding, stochastic log-determinant and gradient on the device) reproduces the CPU path" in out.stdout
    # round 5, second widening: likelihoods with auxiliary parameters (gamma: real-valued response + shape; t: scale and df under Fisher-Laplace) -- the response by
    # gpb_hip_vecchia_laplace_set_response_real, the parameters by _set_aux_pars at every evaluation, their gradient by _grad_aux_current; the fit estimates them
    for lik in ("gamma", "t"):
        assert "Laplace %s n=800: GPU_use=true (mode finding, stochastic log-determinant and gradient on the device) reproduces the CPU path" % lik in out.stdout
