"""CPU: the HOST half of the C API (gpboost_amd/csrc/gpb_c_api.cpp + gpb_optim.cpp, unmodified) end to end, with a CPU restatement of the shim
(tests/mock_shim/mock_gpb_hip.cpp, built on the oracle) standing in for the device library.

What this covers: the C API's orchestration -- cluster handling, repeated locations, covariates (scaling, initial coefficients, the lbfgs over
covariance parameters and coefficients), fixed effects, prediction bookkeeping (unique prediction locations, response transforms), the numerical
Hessians of the standard errors -- by running the SAME test functions the MI355X runs (`-m gpu`) in a child process whose GPBOOST_AMD_LIB points at
tests/mock_shim/libgpb_c_api_on_oracle_TEST_ONLY.so.  What it does NOT cover: the HIP kernels and the real shim (gpb_hip.cpp); those are what the
-m gpu run on the device tests.  The mock library is test infrastructure: nothing in the package finds it unless a test sets GPBOOST_AMD_LIB."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_shim", "libgpb_c_api_on_oracle_TEST_ONLY.so")


@pytest.fixture(scope="module")
def mock_lib(orc):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "mock_shim"), "--no-print-directory"], check=True, capture_output=True)
    assert os.path.isfile(MOCK)
    return MOCK


def _run_gpu_tests_on_the_mock(mock_lib, files, extra=()):
    # four worker processes (pytest-xdist) with two OpenMP threads each: the oracle's restatements are serial loops, the test functions independent
    env = dict(os.environ, GPBOOST_AMD_LIB=mock_lib, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-rx", "-n", "4"] + list(extra) + [os.path.join(ROOT, "tests", f) for f in files]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert " failed" not in tail and " error" not in tail and "xfailed" not in tail, tail
    return tail


def test_device_tests_written_without_a_gpu_pass_on_the_cpu_restatement_of_the_shim(mock_lib):
    """The tests that sort last (tests/test_zz_*): prediction with cluster ids (the R golden), training-data random effects / standard errors / fits
    with covariates of the non-Gaussian models, the R suite's logit / probit prediction goldens through the C API.  They pass on the MI355X since
    round 4 (profiles/r04_*); here every one of them must pass against the oracle-backed shim as well."""
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_zz_cluster_prediction_gpu.py", "test_zz_laplace_train_re_gpu.py"])
    assert "31 passed" in tail, tail      # (round 6: + 4 fits with covariates AND sample weights)


def test_validated_device_tests_still_pass_on_the_cpu_restatement_of_the_shim(mock_lib):
    """A regression net for the C API's host code under the tests that HAVE run on the MI355X: non-Gaussian predictive variances / response predictions
    and repeated locations, the five Gaussian prediction types incl. the R goldens, Gaussian fits with covariates (33 tests)."""
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_laplace_predvar.py", "test_laplace_dup.py", "test_predtypes.py", "test_coef.py"])
    assert "33 passed" in tail, tail


def test_auxiliary_parameter_likelihoods_through_the_c_api_on_the_cpu_restatement_of_the_shim(mock_lib):
    """Round 5: gamma / negative_binomial through the model surface (GPB_SetOptimConfig(init_aux_pars, estimate_aux_pars), GPB_EvalNegLogLikelihood,
    GPB_OptimCovPar with the shape in the lbfgs vector and FindInitialAuxPars' start, GPB_GetAuxPars, response predictions): the host code under
    tests/test_zz_laplace_aux_gpu.py's model-API tests, against the reference's fixtures, with the oracle-backed shim."""
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_zz_laplace_aux_gpu.py"], extra=["-k", "model_api"])
    assert "7 passed" in tail, tail      # (4 gamma / negative_binomial cases + 2 beta cases + the error paths)


def test_student_t_likelihood_through_the_c_api_on_the_cpu_restatement_of_the_shim(mock_lib):
    """Round 5: the t likelihood's two auxiliary parameters through the model surface (GPB_SetOptimConfig(init_aux_pars[2]), the lbfgs vector (log sigma1^2, log a, log scale,
    log df), the MAD start of the scale, GPB_GetAuxPars with two values, response predictions): tests/test_zz_laplace_t_gpu.py's model-API tests on the oracle-backed shim."""
    # (round 6: the standard-deviation cases of gamma and t run on the device only -- 2 x 10 gradient evaluations of the C restatement each; the t_fix_df one runs here)
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_zz_laplace_t_gpu.py"], extra=["-k", "model_api and not (standard_deviations and (gamma_n1500 or t_n1500)) and not (gradient_descent and (gd_gamma or gd_t))"])
    assert "11 passed" in tail, tail      # (round 6, last: + 2 gaussian_latent cases -- the Gaussian likelihood through the Laplace machinery, "error_variance", init variance = half the sample variance; round 6, later: + gradient_descent with estimated auxiliary parameters -- the two short fits here, the 25-40-iteration ones on the device (all five pass on this shim: 6 min); round 6: + "t_fix_df" and likelihood_additional_param, standard deviations of auxiliary parameters, nelder_mead with the shape in the simplex; 2 t cases + 2 lognormal cases: the log-variance's moment start, "log_variance", the response mean exp(m + v / 2) and its variance)


def test_pivoted_cholesky_preconditioner_through_the_c_api_on_the_cpu_restatement_of_the_shim(mock_lib):
    """Round 5: cg_preconditioner_type = "pivoted_cholesky" through the model surface (GPB_SetOptimConfig(cg_preconditioner_type, piv_chol_rank) incl.
    ParsePreconditionerAlias and the rank checks, GPB_GetCGPreconditionerType, GPB_EvalNegLogLikelihood, GPB_OptimCovPar): the host code under
    tests/test_zz_laplace_pivchol_gpu.py's model-API tests, against the reference's fixtures, with the oracle-backed shim."""
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_zz_laplace_pivchol_gpu.py"], extra=["-k", "model_api"])
    assert "13 passed" in tail, tail      # (5 pivoted_cholesky cases + 3 fitc cases + 4 with weights / repeated locations + the error paths)


def test_vecchia_response_preconditioner_on_the_cpu_restatement_of_the_shim(mock_lib):
    """Round 6: cg_preconditioner_type = "vecchia_response" -- tests/test_zz_laplace_vresp_gpu.py as a whole (shim-level values against the reference's fixtures and the
    oracle, the refused gradient, the model surface with the alias, Nelder-Mead fits, weights / repeated locations, m > 62 and d = 4) on the oracle-backed shim."""
    tail = _run_gpu_tests_on_the_mock(mock_lib, ["test_zz_laplace_vresp_gpu.py"])
    assert "16 passed" in tail, tail      # (5 values + 3 oracle-step cases + 5 model-surface cases + 2 with weights / repeated locations + the generality kernel's shapes)


ROUTE_A_DRIVER = r'''
import json, sys, types
sys.modules.setdefault("optuna", types.ModuleType("optuna"))       # optional dependency of the reference's package, absent here
fake = types.ModuleType("gpboost.libpath")                          # route A of INTEGRATION.md: only find_lib_path() changes
fake.find_lib_path = lambda: [sys.argv[1]]
sys.modules["gpboost.libpath"] = fake
sys.path.insert(0, "/root/reference/python-package")
sys.path.insert(0, sys.argv[3])
import numpy as np
import gpboost as gpb
from tests import cases
lik = sys.argv[2]
coords, y, X = cases.laplace_coef_data(lik, 2)
n = 700
coords, y, X = coords[:n], y[:n], X[:n]
cpred = np.random.default_rng(3).uniform(size=(12, 2)); Xp = np.c_[np.ones(12), np.sin(3 * cpred[:, 0] + cpred[:, 1])]
m = gpb.GPModel(gp_coords=coords, cov_function="matern", cov_fct_shape=1.5, likelihood=lik, gp_approx="vecchia", num_neighbors=20,
                vecchia_ordering="random", seed=2)
m.fit(y=y, X=X)                                                     # the package's defaults: lbfgs, initial coefficients from the model without the GP
out = {"cov_pars": np.asarray(m.get_cov_pars()).ravel().tolist(), "coef": np.asarray(m.get_coef()).ravel().tolist(),
       "num_it": int(m._get_num_optim_iter()), "nll": float(m.get_current_neg_log_likelihood())}
pl = m.predict(gp_coords_pred=cpred, X_pred=Xp, predict_var=True, predict_response=False)
pr = m.predict(gp_coords_pred=cpred, X_pred=Xp, predict_var=True)   # predict_response = True is the package's default
out["latent_mu"] = np.asarray(pl["mu"]).tolist(); out["latent_var"] = np.asarray(pl["var"]).tolist()
out["resp_mu"] = np.asarray(pr["mu"]).tolist(); out["resp_var"] = np.asarray(pr["var"]).tolist()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/python-package") or not os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "lib_gpboost_ref.so")),
                    reason="needs /root/reference and oracle/_ref (the build container)")
@pytest.mark.parametrize("lik", ["bernoulli_logit", "poisson"])
def test_the_references_own_package_gets_the_same_answers_from_this_host_code(mock_lib, lik):
    """Drop-in route A on the CPU: the reference's UNMODIFIED Python package fits a non-Gaussian Vecchia model with covariates and predicts -- once
    through the reference's own library (oracle/_ref/lib_gpboost_ref.so), once through this repository's C API host code on the oracle-backed shim.
    Same script, only find_lib_path() differs.  Iterations agree; estimates, likelihood and latent predictive means agree to 1e-8 for the logit model (seen 1e-12) and to the noise of the default solver
    tolerances for the Poisson model (1e-4); the latent
    variances are EXACT here and a 1000-sample estimate in the reference (within 15 %), so the response predictions that integrate over them agree
    to 1e-3."""
    import json
    res = {}
    for tag, lib in (("ref", os.path.join(ROOT, "oracle", "_ref", "lib_gpboost_ref.so")), ("ours", mock_lib)):
        out = subprocess.run([sys.executable, "-c", ROUTE_A_DRIVER, lib, lik, ROOT], capture_output=True, text=True, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        assert lines, out.stdout[-2000:] + out.stderr[-3000:]
        res[tag] = json.loads(lines[-1][7:])
    a, b = res["ours"], res["ref"]
    import numpy as np
    assert a["num_it"] == b["num_it"]
    tight = lik == "bernoulli_logit"        # logit: 1e-12 seen; Poisson: the CG solves stopped at |r| < 1e-2 leave 1e-4 of noise in either implementation's path
    np.testing.assert_allclose(a["cov_pars"], b["cov_pars"], rtol=1e-8 if tight else 2e-3)
    np.testing.assert_allclose(a["coef"], b["coef"], rtol=1e-8 if tight else 2e-3)
    assert abs(a["nll"] - b["nll"]) <= (1e-9 if tight else 1e-6) * abs(b["nll"])
    np.testing.assert_allclose(a["latent_mu"], b["latent_mu"], rtol=1e-8 if tight else 1e-2, atol=1e-10 if tight else 2e-3)
    np.testing.assert_allclose(a["latent_var"], b["latent_var"], rtol=0.15)
    np.testing.assert_allclose(a["resp_mu"], b["resp_mu"], rtol=2e-3)
    np.testing.assert_allclose(a["resp_var"], b["resp_var"], rtol=5e-3)


@pytest.mark.skipif(not os.path.isdir("/root/reference/python-package") or not os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "lib_gpboost_ref.so")),
                    reason="needs /root/reference and oracle/_ref (the build container)")
@pytest.mark.parametrize("scenario", ["gauss_clusters", "logit_plain", "probit_offset", "poisson_dups_cov", "gauss_pred_types", "logit_more", "gauss_misc", "poisson_misc", "gauss_covariates", "gauss_covariates_gd", "gauss_edges", "round5_widening", "round5_lognormal", "round6_widening"])
def test_route_a_scenarios_agree_with_the_references_own_library(mock_lib, scenario):
    """tests/route_a_driver.py: the reference's unmodified package, once on the reference's library, once on this host code (oracle-backed shim).
      gauss_clusters    Gaussian Vecchia model with cluster ids: fit, prediction with cluster ids of observed and unobserved clusters, two prediction types
      logit_plain       Bernoulli logit: fit, latent / response prediction, training-data random effects, likelihood evaluation
      probit_offset     Bernoulli probit with an offset at fit and prediction time (the fit's offset must be remembered, re_model_template.h:1185-1188:
                        found by this test)
      poisson_dups_cov  Poisson with repeated locations AND covariates (coefficients in the lbfgs vector, initial values from the model without the GP)
      gauss_pred_types  one Gaussian model fitted three times (gradient descent, simplex search, lbfgs with a parameter held fixed: the optimiser-dependent
                        defaults are resolved at the FIRST fit and inherited by the later ones, re_model_template.h:8318-8347: found by this test), all
                        five prediction types, saved prediction data
      logit_more        logit with covariates AND an offset, 'latent_order_obs_first_cond_all' prediction, then gradient descent / simplex search / lbfgs
                        with the variance held fixed on one model
      gauss_misc        three clusters with a random ordering, an offset at fit time and at prediction time WITHOUT y (the response the model keeps is y as
                        passed in, not y - offset: found by this test), prediction points that are training points, 3-d coordinates; repeated locations
                        in a Gaussian model
      gauss_covariates  Gaussian fits with covariates (lbfgs, coefficients by generalised least squares at every evaluation): with an offset, with initial
                        values, with the GP variance held fixed -- the run whose last line search fails, after which the coefficients are those of the
                        last ACCEPTED iterate (ResetProfiledOutVariablesToLag1, optim_utils.h:383-390: found by this test); coefficient standard
                        deviations, predictions with X_pred, covariance matrix
      gauss_covariates_gd  the same with optimizer_cov 'gradient_descent': ONE least-squares update of the coefficients per iteration, fixed during the
                        step-size search (re_model_template.h:1478-1481); Nesterov acceleration on / off, offset + init_coef, the range held fixed,
                        the intercept not in the first column
      gauss_edges       one coordinate dimension; convergence by relative change in the parameters, momentum offset / acceleration rate, two lbfgs
                        corrections, fits stopped after one iteration; a simplex search that runs a parameter to zero ends with the reference's
                        'Check failed: pars[i] > 0.' on both sides (it used to return an infinite range here: found by this test)
      round5_widening   a Student-t model (two auxiliary parameters, Fisher-Laplace) fitted and evaluated with the fitc preconditioner, a beta regression fitted with
                        the pivoted_cholesky preconditioner (rank 40): estimates, auxiliary parameters, iteration counts, predictions
      round5_lognormal  the lognormal likelihood (one auxiliary parameter, "log_variance"): fit, response / latent predictions, an evaluation with pivoted_cholesky
      round6_widening   the vecchia_response preconditioner (evaluation, a 20-iteration Nelder-Mead fit), a gaussian_latent fit with response predictions, gradient descent with an
                        estimated gamma shape, t_fix_df with standard deviations
      poisson_misc      Poisson with an offset, random ordering, Matern 2.5: fit, standard errors, latent variances / covariance, training random effects
    Everything deterministic agrees to 1e-6 (seen 1e-7 .. 1e-15, iteration counts equal); the reference's random-vector estimates of predictive variances
    scatter around this library's exact values."""
    import json
    import numpy as np
    res = {}
    for tag, lib in (("ref", os.path.join(ROOT, "oracle", "_ref", "lib_gpboost_ref.so")), ("ours", mock_lib)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "route_a_driver.py"), lib, scenario, ROOT], capture_output=True, text=True, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        assert lines, out.stdout[-2000:] + out.stderr[-3000:]
        res[tag] = json.loads(lines[-1][7:])
    a, b = res["ours"], res["ref"]
    assert sorted(a) == sorted(b)
    for k in b:
        x, y = np.asarray(a[k], dtype=float), np.asarray(b[k], dtype=float)
        if k == "num_it":
            assert a[k] == b[k], (a[k], b[k])
        elif k.startswith("stoch_"):
            np.testing.assert_allclose(x, y, rtol=0.2, err_msg=k)
        elif k.startswith("stochse_"):
            np.testing.assert_allclose(x, y, rtol=2e-2, err_msg=k)
        elif k.startswith("stochm_"):
            np.testing.assert_allclose(x, y, rtol=5e-3, err_msg=k)
        elif k.startswith("flat_"):      # estimates of a fit whose likelihood is flat in the range (beta regression, pivoted_cholesky rank 40): seen 2.6e-6 at the tight thresholds
            np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-8, err_msg=k)
        else:      # (absolute part relative to the vector's scale: entries of a mode near zero were seen 1.2e-8 apart -- 4e-8 of the largest entry -- on a loaded machine, where the
                   #  reference's OpenMP reductions take another order)
            np.testing.assert_allclose(x, y, rtol=1e-6, atol=max(1e-8, 1e-7 * float(np.abs(y).max())) if y.size else 1e-8, err_msg=k)
