"""Seeded synthetic inputs shared by the golden-fixture generator, the CPU tests and the GPU parity tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> case.  "data": "r_fixture" = the reference's R test fixture
# (R-package/tests/testthat/test_GPModel_gaussian_process.R:36-60); otherwise uniform coords from default_rng(seed_data).
GOLDEN_CASES = {
    "r_exp_m30_none": dict(data="r_fixture", cov_function="exponential", shape=0.5, m=30, ordering="none", seed=0,
                           cov_pars=[(0.1, 1.6, 0.2), (0.5, 0.9, 0.05)]),
    "r_mat15_m30_random": dict(data="r_fixture", cov_function="matern", shape=1.5, m=30, ordering="random", seed=0,
                               cov_pars=[(0.1, 1.6, 0.2)]),
    "r_mat25_m99_none": dict(data="r_fixture", cov_function="matern", shape=2.5, m=99, ordering="none", seed=0,
                             cov_pars=[(0.1, 1.6, 0.2)]),
    "u2d_n3000_exp_m30": dict(data="uniform", n=3000, d=2, seed_data=11, cov_function="exponential", shape=0.5, m=30,
                              ordering="random", seed=1, cov_pars=[(0.1, 1.0, 0.1)]),
    "u3d_n3000_mat25_m40": dict(data="uniform", n=3000, d=3, seed_data=12, cov_function="matern", shape=2.5, m=40,
                                ordering="random", seed=1, cov_pars=[(0.1, 1.0, 0.1)]),
    "u1d_n1000_mat15_m10": dict(data="uniform", n=1000, d=1, seed_data=13, cov_function="matern", shape=1.5, m=10,
                                ordering="random", seed=3, cov_pars=[(0.2, 1.3, 0.05)]),
    "grid2d_n1024_exp_m20": dict(data="grid", n=1024, d=2, seed_data=14, cov_function="exponential", shape=0.5, m=20,
                                 ordering="random", seed=2, cov_pars=[(0.1, 1.0, 0.15)]),
    "dup2d_n600_exp_m15": dict(data="duplicates", n=600, d=2, seed_data=15, cov_function="exponential", shape=0.5, m=15,
                               ordering="random", seed=5, cov_pars=[(0.3, 1.0, 0.2)]),
    "tiny_n8_m30": dict(data="uniform", n=8, d=2, seed_data=16, cov_function="exponential", shape=0.5, m=30,
                        ordering="none", seed=0, cov_pars=[(0.1, 1.0, 0.3)]),
    # coordinate dimensions beyond 3 (the generality path: d-dimensional neighbour search + the LDS-resident point kernel)
    "u5d_n700_mat15_m20": dict(data="uniform", n=700, d=5, seed_data=31, cov_function="matern", shape=1.5, m=20,
                               ordering="random", seed=4, cov_pars=[(0.2, 1.1, 0.4)]),
    "u4d_n1500_exp_m30": dict(data="uniform", n=1500, d=4, seed_data=32, cov_function="exponential", shape=0.5, m=30,
                              ordering="random", seed=2, cov_pars=[(0.1, 1.0, 0.3)]),
    "u10d_n400_mat25_m70": dict(data="uniform", n=400, d=10, seed_data=33, cov_function="matern", shape=2.5, m=70,
                                ordering="none", seed=0, cov_pars=[(0.3, 0.8, 1.5)]),
}


def make_data(c):
    """-> (coords (n, d), y (n,)) in DATA order."""
    if c["data"] == "r_fixture":
        from oracle import orc
        return orc.r_fixture()
    rng = np.random.default_rng(c["seed_data"])
    n, d = c["n"], c["d"]
    if c["data"] == "uniform":
        coords = rng.uniform(size=(n, d))
    elif c["data"] == "grid":            # exact ties in distances and coordinate sums
        k = int(round(n ** (1. / d)))
        assert k ** d == n
        ax = np.arange(k) / k
        coords = np.stack(np.meshgrid(*([ax] * d), indexing="ij"), axis=-1).reshape(n, d)
        coords = coords[rng.permutation(n)]
    elif c["data"] == "duplicates":      # every location observed three times
        base = rng.uniform(size=(n // 3, d))
        coords = np.concatenate([base, base, base])[rng.permutation(n)]
    else:
        raise ValueError(c["data"])
    y = np.sin(4 * coords[:, 0]) + 0.5 * rng.standard_normal(n)
    return coords, y


# Vecchia-Laplace (Bernoulli-logit, iterative methods) cases -- BASELINE config 4.  Outputs of the reference's
# GPB_EvalNegLogLikelihood are stored in tests/golden/laplace_ref.npz (oracle/make_golden.py).
LAPLACE_CASES = {
    "lap_u2d_n2000_exp_m20": dict(n=2000, d=2, seed_data=21, cov_function="exponential", shape=0.5, m=20, ordering="random",
                                  seed=1, cov_pars=[(1.0, 0.1), (2.5, 0.05)]),
    "lap_u2d_n1500_mat15_m30": dict(n=1500, d=2, seed_data=22, cov_function="matern", shape=1.5, m=30, ordering="random",
                                    seed=2, cov_pars=[(1.0, 0.15)]),
    "lap_u3d_n1200_mat25_m15": dict(n=1200, d=3, seed_data=23, cov_function="matern", shape=2.5, m=15, ordering="none",
                                    seed=0, cov_pars=[(0.7, 0.3)]),
}


# Solver thresholds at which the Laplace gradient is PINNED to north_star's 1e-8 (tests/golden/laplace_grad_ref.npz: *_grad_direct, from the reference's own
# CalcGradPars): at the reference's defaults (cg_delta_conv 1e-2, delta_conv_mode_finding 1e-8) two correct implementations differ by which CG iteration
# crosses the threshold (~1e-5 on the gradient); at these the reference's gradient no longer moves (1e-10 thresholds: < 1.1e-9 relative).
LAPLACE_TIGHT = dict(cg_delta_conv=1e-8, delta_conv_mode_finding=1e-13)
# Predictions against the reference's CHOLESKY-based fixtures (exact solves) at north_star's 1e-8 (round 6): the
# prediction legs run the mode finding to convergence (relative change of the objective 1e-15, Newton systems to a residual norm of 1e-11)
LAPLACE_PRED_TIGHT = dict(cg_delta_conv=1e-11, delta_conv_mode_finding=1e-16)
# ... and the fixtures they are compared with come from the reference's Cholesky-based mode finding run to convergence (oracle/make_golden.py laplace_pred_refresh): at the
# 1e-13 of LAPLACE_TIGHT the reference's own mode is up to 1.2e-7 from the converged one (negbin_n1500), its values at 1e-16 (Cholesky) and 1e-15 (iterative) agree to 3e-13
LAPLACE_PRED_REF = dict(cg_delta_conv=1e-8, delta_conv_mode_finding=1e-16)


def make_binary_data(c):
    """-> (coords (n, d), y01 (n,) float 0/1) in DATA order: Bernoulli draws around a smooth latent surface."""
    rng = np.random.default_rng(c["seed_data"])
    n, d = c["n"], c["d"]
    coords = rng.uniform(size=(n, d))
    latent = 1.5 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.3
    y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-latent))).astype(np.float64)
    return coords, y


def make_count_data(c):
    """-> (coords, y counts as float) in DATA order: Poisson draws with log-mean = a smooth surface (same coords as make_binary_data)."""
    rng = np.random.default_rng(c["seed_data"])
    n, d = c["n"], c["d"]
    coords = rng.uniform(size=(n, d))
    latent = 1.2 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.4
    y = rng.poisson(np.exp(latent)).astype(np.float64)
    return coords, y


# Likelihoods with an auxiliary parameter (first slice of round 5): gamma (shape) and negative_binomial (shape), estimated
# jointly with the covariance parameters (likelihoods.h:298-322, CalcGradNegLogLikAuxPars :14185-14215).  model: a LAPLACE_CASES entry (coordinates,
# covariance function, neighbours, ordering); aux: the parameter the likelihood / gradient fixtures are evaluated at; true_aux: what the data were drawn with.
LAPLACE_AUX_CASES = {
    "gamma_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="gamma", aux=2.0, true_aux=2.5),
    "gamma_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="gamma", aux=0.8, true_aux=1.2),
    "negbin_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="negative_binomial", aux=1.5, true_aux=1.7),
    # (its fit ends in a line search whose second trial point the reference accepts by 2e-8 of the likelihood: at the DEFAULT solver thresholds two correct
    #  implementations part there -- estimates 1.2e-2 apart at likelihoods 2e-8 apart; at cases.LAPLACE_TIGHT they agree to 1e-6.  flat_default marks it.)
    "negbin_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="negative_binomial", aux=3.0, true_aux=4.0, flat_default=True),
    # round 5, second slice: beta regression (mean = sigmoid(location), precision = aux; likelihoods.h:378-383, :11903-11913): responses in (0, 1)
    # (fe_scale: with the full offset of laplace_fixed_effects the reference's own mode finding ends in NaN on these data -- the beta likelihood is not log-concave in
    #  the location parameter and the information turns negative far from the data -- so the fixed-effects fixtures of the beta cases use 0.3 of it)
    "beta_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="beta", aux=6.0, true_aux=8.0, fe_scale=0.3),
    # (its fit at the DEFAULT thresholds: equal iteration counts, estimates 4e-4 apart -- stopping-rule noise of the cg_delta_conv = 1e-2 solves along 14 iterations;
    #  at cases.LAPLACE_TIGHT 1e-6.  flat_default widens the default-threshold comparison as for negbin_n2000)
    "beta_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="beta", aux=15.0, true_aux=12.0, fe_scale=0.3, flat_default=True),
}


# Student-t likelihood (round 5, third slice; likelihoods.h:384-423: location = the latent value, auxiliary parameters (scale, df), both estimated; the reference's default
# approximation "fisher_laplace": the information is the constant Fisher information).  aux: where the value / gradient fixtures are evaluated; true_*: what the data were drawn with.
LAPLACE_T_CASES = {
    "t_n1500": dict(model="lap_u2d_n1500_mat15_m30", aux=(0.5, 3.0), true_scale=0.4, true_df=4.0),
    "t_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", aux=(0.3, 6.0), true_scale=0.3, true_df=5.0),
    # lognormal (round 5, fourth slice; the same constant-information structure as t under Fisher-Laplace, ONE auxiliary parameter: the variance of log y,
    # likelihoods.h:30-34, :505-513): mean of y = exp(location)
    "lognormal_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="lognormal", aux=(0.3,), true_s2=0.2),
    "lognormal_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="lognormal", aux=(0.15,), true_s2=0.1),
    # gaussian_latent (round 6: the Gaussian likelihood through the Laplace machinery, ONE auxiliary parameter "error_variance", constant information 1 / aux, one Newton step
    # with one trial point -- likelihoods.h:458-475): y = latent + sqrt(s2) * normal
    "gaussian_latent_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="gaussian_latent", aux=(0.3,), true_s2=0.2),
    "gaussian_latent_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="gaussian_latent", aux=(0.15,), true_s2=0.1),
}


# optimizer_cov = "gradient_descent" with estimated auxiliary parameters (round 6): two learning rates (covariance block / auxiliary block), both Armijo conditions, joint
# halving -- re_model_template.h:1514-1660, :8354-8375, :8690-8850.  name -> (case table, case, likelihood, number of auxiliary parameters, GPB_SetOptimConfig arguments).
# Fixture: tests/golden/laplace_aux_gd_ref.npz (oracle/make_golden.py laplace_aux_gd), fits at LAPLACE_TIGHT.
LAPLACE_AUX_GD_CASES = {
    "gd_gamma_n1500": ("aux", "gamma_n1500", "gamma", 1, dict(max_iter=40)),
    "gd_gamma_n1500_plain": ("aux", "gamma_n1500", "gamma", 1, dict(max_iter=25, use_nesterov_acc=False, lr_cov=0.05)),
    "gd_negbin_n1500": ("aux", "negbin_n1500", "negative_binomial", 1, dict(max_iter=40)),
    "gd_t_n1500": ("t", "t_n1500", "t", 2, dict(max_iter=40)),
    "gd_lognormal_u3d_n1200_parchange": ("t", "lognormal_u3d_n1200", "lognormal", 1, dict(max_iter=60, convergence_criterion="relative_change_in_parameters", delta_rel_conv=1e-3)),
}


def aux_gd_case(name):
    """-> (coords, y, LAPLACE_CASES model entry, likelihood, naux, config) of a LAPLACE_AUX_GD_CASES entry."""
    table, case, lik, naux, cfg = LAPLACE_AUX_GD_CASES[name]
    cs = (LAPLACE_AUX_CASES if table == "aux" else LAPLACE_T_CASES)[case]
    coords, y = (make_aux_data if table == "aux" else make_t_data)(cs)
    return coords, y, LAPLACE_CASES[cs["model"]], lik, naux, cfg


def make_t_data(tc):
    """-> (coords, y) in DATA order for a LAPLACE_T_CASES entry: a smooth surface + scale * Student-t(df) noise."""
    c = LAPLACE_CASES[tc["model"]]
    rng = np.random.default_rng(c["seed_data"])
    n, d = c["n"], c["d"]
    coords = rng.uniform(size=(n, d))
    latent = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.3
    if tc.get("lik", "t") == "lognormal":      # log y ~ N(latent - s2 / 2, s2)
        s2 = tc["true_s2"]
        return coords, np.exp(latent - 0.5 * s2 + np.sqrt(s2) * np.random.default_rng(c["seed_data"] + 4100).standard_normal(n))
    if tc.get("lik", "t") == "gaussian_latent":
        return coords, latent + np.sqrt(tc["true_s2"]) * np.random.default_rng(c["seed_data"] + 4200).standard_normal(n)
    y = latent + tc["true_scale"] * np.random.default_rng(c["seed_data"] + 4000).standard_t(tc["true_df"], size=n)
    return coords, y


def aux_fixed_effects(ac, coords):
    """Offset of the fixed-effects fixtures of a LAPLACE_AUX_CASES entry (data order): laplace_fixed_effects, scaled by the entry's fe_scale."""
    return ac.get("fe_scale", 1.0) * laplace_fixed_effects(coords)


# Sample weights for non-Gaussian models (round 5; Likelihood::weights_): name -> (likelihood, data source).  Weights: uniform(0.3, 2.5), every 17th 0.05, every
# 11th exactly 1.  (Weights that are exactly ZERO are outside this path: d information / d loc is then zero at those data and the reference leaves its closed
#  form diag((Sigma^-1 + W)^-1) = (d logdet / d mode) / (d information / d loc) for a stochastic trace estimate, likelihoods.h:6754-6768 -- the library refuses them.)
LAPLACE_WEIGHT_CASES = {
    "w_logit_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit"),
    "w_poisson_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="poisson"),
    "w_gamma_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="gamma", aux=2.0, true_aux=2.5),
    "w_negbin_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="negative_binomial", aux=3.0, true_aux=4.0),
}
# Proportions under the logit / probit links (round 5): binomial_* -- y = successes / trials, the trials (1..20) are the sample weights; quasi_bernoulli_* -- any
# y in [0, 1] (here: a noisy proportion, some exact zeros and ones), with and without weights.  Same fixture layout as the weighted cases.
LAPLACE_PROP_CASES = {
    "binomial_logit_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="binomial_logit"),
    "binomial_probit_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="binomial_probit"),
    "quasi_logit_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="quasi_bernoulli_logit", weights=False),
    "quasi_probit_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="quasi_bernoulli_probit"),
}
LAPLACE_WEIGHT_CASES.update(LAPLACE_PROP_CASES)
LAPLACE_PROP_CASES_LIKS = ("binomial_logit", "binomial_probit", "quasi_bernoulli_logit", "quasi_bernoulli_probit")


# Preconditioner "pivoted_cholesky" of the Vecchia-Laplace iterative methods (round 5; re_model_template.h:5906, the (W^-1 + Sigma) form of the solves,
# CG_utils.cpp:231-499): name -> LAPLACE_CASES model, likelihood, rank of the pivoted Cholesky factor (None: the reference's default 50), optional auxiliary
# parameter (gamma / negative_binomial draw their responses as LAPLACE_AUX_CASES does).  Fixture: tests/golden/laplace_pivchol_ref.npz.
LAPLACE_PIVCHOL_CASES = {
    "pc_logit_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="bernoulli_logit", rank=None),
    "pc_poisson_n1500_r20": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", rank=20),
    "pc_probit_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="bernoulli_probit", rank=None),
    "pc_gamma_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="gamma", rank=None, aux=2.0, true_aux=2.5),
    "pc_negbin_n2000_r30": dict(model="lap_u2d_n2000_exp_m20", lik="negative_binomial", rank=30, aux=3.0, true_aux=4.0),
    # cg_preconditioner_type = "fitc" (pc="fitc"; rank = number of inducing points, None: the reference's default 200): the same solves with
    # P = diag(W^-1 + Sigma_m[0][0] - ||V_i||^2) + C Sigma_m^-1 C', inducing points by kmeans++ from the model's generator (re_model_template.h:9502-9593)
    "fitc_logit_n1500_r100": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", rank=100, pc="fitc"),
    "fitc_poisson_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="poisson", rank=None, pc="fitc"),
    "fitc_gamma_u3d_n1200_r60": dict(model="lap_u3d_n1200_mat25_m15", lik="gamma", rank=60, aux=0.8, true_aux=1.2, pc="fitc"),
}


# cg_preconditioner_type = "vecchia_response" (round 6; the fifth entry of SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, re_model_template.h:5906): the (W^-1 + Sigma) form of
# the solves preconditioned with the Vecchia approximation of W^-1 + Sigma itself (likelihoods.h:16315-16323, :16439-16450, :16471-16473).  EVALUATION ONLY -- the reference
# refuses the gradient with it (likelihoods.h:6570-6572), so a fit is a Nelder-Mead fit.  Same case layout as LAPLACE_PIVCHOL_CASES; "extra": a LAPLACE_PC_EXTRA-style
# entry (sample weights / repeated locations) through the model surface.  Fixture: tests/golden/laplace_vresp_ref.npz (oracle/make_golden.py laplace_vresp).
LAPLACE_VRESP_CASES = {
    "vr_logit_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="bernoulli_logit", rank=None),
    "vr_poisson_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", rank=None),
    "vr_probit_u3d_n1200": dict(model="lap_u3d_n1200_mat25_m15", lik="bernoulli_probit", rank=None),
    "vr_gamma_n1500": dict(model="lap_u2d_n1500_mat15_m30", lik="gamma", rank=None, aux=2.0, true_aux=2.5),
    "vr_negbin_n2000": dict(model="lap_u2d_n2000_exp_m20", lik="negative_binomial", rank=None, aux=3.0, true_aux=4.0),
}
LAPLACE_VRESP_SECOND_PARS = (0.6, 0.22)          # a second evaluation point (variance, range) on the same model (the preconditioner's factor is renewed, the mode starts at 0 again)
LAPLACE_VRESP_EXTRA_CASES = {
    "vrw_poisson_n2000": dict(weights_case="w_poisson_n2000", pc="vecchia_response", rank=None),
    "vrdup_logit": dict(dup=("dup_mat15_m20_random", "bernoulli_logit"), pc="vecchia_response", rank=None),
}
LAPLACE_VRESP_NM = dict(optimizer_cov="nelder_mead", maxit=30)      # the fit of the fixture: 30 Nelder-Mead iterations at LAPLACE_TIGHT


# The low-rank preconditioners together with sample weights / repeated locations (the information W is then a weighted sum / a sum over a location's data; the factor L_k /
# the inducing points live on the unique locations): model-surface checks against the reference library -- tests/golden/laplace_pc_extra_ref.npz.
LAPLACE_PC_EXTRA_CASES = {
    "pcw_poisson_n2000": dict(weights_case="w_poisson_n2000", pc="pivoted_cholesky", rank=40),
    "fitcw_gamma_n1500": dict(weights_case="w_gamma_n1500", pc="fitc", rank=70),
    "pcdup_logit": dict(dup=("dup_mat15_m20_random", "bernoulli_logit"), pc="pivoted_cholesky", rank=30),
    "fitcdup_poisson": dict(dup=("dup_exp_m15_none", "poisson"), pc="fitc", rank=50),
}


def pc_extra_model(ec):
    """-> (GPModel keyword arguments incl. coordinates [and weights], y, cov_pars, aux or None) of a LAPLACE_PC_EXTRA_CASES entry."""
    if "weights_case" in ec:
        wc = LAPLACE_WEIGHT_CASES[ec["weights_case"]]
        c = LAPLACE_CASES[wc["model"]]
        coords, y, w = make_weight_data(wc)
        kw = dict(likelihood=wc["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia", num_neighbors=c["m"],
                  vecchia_ordering=c["ordering"], seed=c["seed"], weights=w)
        return kw, y, np.asarray(c["cov_pars"][0], dtype=np.float64), wc.get("aux")
    name, lik = ec["dup"]
    cf, sh, m, ordering, seed = LAPLACE_DUP_CASES[name]
    coords, y, _, _ = laplace_dup_data(lik)
    kw = dict(likelihood=lik, gp_coords=coords, cov_function=cf, cov_fct_shape=sh, gp_approx="vecchia", num_neighbors=m, vecchia_ordering=ordering, seed=seed)
    return kw, y, np.asarray(LAPLACE_DUP_COV_PARS[0], dtype=np.float64), None


def pivchol_rank(pc):
    """Columns of the low-rank part a LAPLACE_PIVCHOL_CASES entry asks for (None: the reference's defaults 50 / 200)."""
    return pc["rank"] if pc["rank"] is not None else (200 if pc.get("pc") == "fitc" else 50)


def make_pivchol_data(pc):
    """-> (coords, y) in DATA order for a LAPLACE_PIVCHOL_CASES entry."""
    c = LAPLACE_CASES[pc["model"]]
    if "aux" in pc:
        return make_aux_data(pc)
    return make_count_data(c) if pc["lik"] == "poisson" else make_binary_data(c)


def make_weight_data(wc):
    """-> (coords, y, weights) in DATA order for a LAPLACE_WEIGHT_CASES entry."""
    c = LAPLACE_CASES[wc["model"]]
    if wc["lik"].startswith("binomial") or wc["lik"].startswith("quasi"):
        rng = np.random.default_rng(c["seed_data"])
        coords = rng.uniform(size=(c["n"], c["d"]))
        lat = 1.2 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.2
        p = 1.0 / (1.0 + np.exp(-lat))
        rng2 = np.random.default_rng(c["seed_data"] + 4000)
        if wc["lik"].startswith("binomial"):
            trials = rng2.integers(1, 21, size=c["n"]).astype(np.float64)
            return coords, rng2.binomial(trials.astype(np.int64), p) / trials, trials
        y = np.clip(p + 0.25 * rng2.standard_normal(c["n"]), 0.0, 1.0)          # (a fifth of them exactly 0 or 1)
        w = rng2.uniform(0.3, 2.5, size=c["n"]) if wc.get("weights", True) else None
        return coords, y, w
    if wc["lik"] in ("gamma", "negative_binomial"):
        coords, y = make_aux_data(wc)
    else:
        coords, yb = make_binary_data(c)
        if wc["lik"] == "poisson":
            lat = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.3
            y = np.random.default_rng(c["seed_data"] + 2000).poisson(np.exp(lat)).astype(np.float64)
        else:
            y = yb.astype(np.float64)
    w = np.random.default_rng(c["seed_data"] + 3000).uniform(0.3, 2.5, size=coords.shape[0])
    w[::17] = 0.05; w[5::11] = 1.0
    return coords, y, w


def make_aux_data(ac):
    """-> (coords, y) in DATA order for a LAPLACE_AUX_CASES entry: the coordinates of its model (as make_binary_data draws them), responses with
    log-mean = a smooth surface: gamma(shape true_aux, mean mu) / negative binomial(size true_aux, mean mu)."""
    c = LAPLACE_CASES[ac["model"]]
    rng = np.random.default_rng(c["seed_data"])
    n, d = c["n"], c["d"]
    coords = rng.uniform(size=(n, d))
    latent = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.3
    mu = np.exp(latent)
    r = ac["true_aux"]
    rng2 = np.random.default_rng(c["seed_data"] + 1000)
    if ac["lik"] == "beta":
        pm = 1.0 / (1.0 + np.exp(-(1.4 * latent - 0.2)))
        y = np.clip(rng2.beta(pm * r, (1.0 - pm) * r), 1e-6, 1.0 - 1e-6)
    elif ac["lik"] == "gamma":
        y = rng2.gamma(r, mu / r)
    else:
        y = rng2.negative_binomial(r, r / (r + mu)).astype(np.float64)
    return coords, y


def laplace_fixed_effects(coords):
    """Offset of the location parameter used by the fixed-effects Laplace fixtures (data order)."""
    return 0.8 * np.cos(6 * coords[:, -1]) - 0.3


def synthetic_binary(n, d, seed=1):
    """BASELINE config 4 inputs: coords U[0,1]^d, labels Bernoulli(sigmoid(smooth surface)), default_rng(seed)."""
    return make_binary_data(dict(n=n, d=d, seed_data=seed))


# Split search (FeatureHistogram::FindBestThreshold) fixtures: three data sets x two configurations, tests/golden/split_ref.npz
SPLIT_DATA = {"plain": dict(seed=51, params=""),                       # no missing values: MissingType::None, one reverse scan
              "zero_missing": dict(seed=52, params="zero_as_missing=true"),   # MissingType::Zero, both scans, default bin skipped
              "nan": dict(seed=53, params=""),                           # NaNs in two features: MissingType::NaN, both scans
              # round 5 (tree cases only): categorical columns; exclusive sparse columns the reference bundles
              "cat": dict(seed=54, params="categorical_feature=2,5"),
              "efb": dict(seed=55, params="")}
SPLIT_DATA_UNIT = ("plain", "zero_missing", "nan")                        # the data sets of the split / partition unit fixtures (split_ref.npz)
SPLIT_CFGS = [(0.0, 20, 1e-3, 0.0), (1.5, 5, 1e-3, 0.1)]              # lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split
# the other regularisation paths: ... + lambda_l1, max_delta_step, path_smooth, parent_output (min_gain_to_split > 0 wherever max_delta_step clips:
# candidates whose children are clipped to the same output have gain 0 up to rounding noise)
SPLIT_CFGS_REG = [(0.5, 20, 1e-3, 0.0, 3.0, 0.0, 0.0, 0.0), (0.0, 20, 1e-3, 0.05, 0.0, 0.2, 0.0, 0.0), (0.1, 10, 1e-3, 0.0, 0.0, 0.0, 25.0, 0.07),
                  (1.0, 20, 1e-3, 0.01, 1.5, 0.3, 10.0, -0.11)]


# round 5, categorical search: (split cfg as SPLIT_CFGS / SPLIT_CFGS_REG, (max_cat_to_onehot, max_cat_threshold, cat_smooth, cat_l2, min_data_per_group))
SPLIT_CAT_CFGS = [((0.0, 20, 1e-3, 0.0), (4, 32, 10.0, 10.0, 100)),
                  ((1.5, 5, 1e-3, 0.1), (4, 32, 5.0, 2.0, 30)),
                  ((0.5, 20, 1e-3, 0.0, 3.0, 0.0, 0.0, 0.0), (64, 32, 10.0, 10.0, 100)),           # every categorical column one-hot, L1
                  ((0.1, 10, 1e-3, 0.0, 0.0, 0.3, 25.0, 0.07), (4, 6, 1.0, 0.0, 20)),                 # max_delta_step + smoothing, short scans
                  ((1.0, 20, 1e-3, 0.01, 1.5, 0.3, 10.0, -0.11), (2, 32, 20.0, 10.0, 200))]


def split_partition_requests(num_bin):
    """(feature, threshold, default_left) triples exercised by the partition fixture: a low, a middle and the highest admissible
    threshold of every feature, both default directions."""
    req = []
    for f, nb in enumerate(num_bin):
        for th in sorted({0, max(0, nb // 2 - 1), max(0, nb - 2)}):
            for dl in (0, 1):
                req.append((f, th, dl))
    return req


def make_split_data(name):
    c = SPLIT_DATA[name]
    rng = np.random.default_rng(c["seed"])
    n, F = 6000, 6
    if name == "cat":
        # two categorical columns (round 5): 3 categories (one-hot search, num_bin <= max_cat_to_onehot) and 40 categories with 10 % of the rows in
        # category 0 on top (sorted many-vs-many search), next to four numerical columns
        X = rng.uniform(size=(n, F))
        X[:, 2] = rng.integers(0, 3, size=n)
        X[:, 5] = rng.integers(0, 40, size=n) * (rng.uniform(size=n) < 0.9)
        # category effects in three clear groups, the rest zero: the best many-vs-many split takes a few categories from one END of the sorted order.
        # (With 40 generic effects the best split is the half / half one, which BOTH scan directions reach at their cap max_num_cat = (used + 1) / 2 --
        #  complementary sets, mathematically equal gains, and the winner is decided by the last bits of the histogram sums: a void tie like the
        #  default_left of a numerical split without missing rows, not something a fixture can pin.)
        eff = np.zeros(40)
        idx = rng.permutation(40)
        eff[idx[:7]] = 1.5 + 0.2 * rng.standard_normal(7); eff[idx[7:12]] = -1.2 + 0.2 * rng.standard_normal(5); eff[idx[12:16]] = 0.6
        g = np.sin(6 * X[:, 0]) + 0.8 * (X[:, 2] == 1) + eff[X[:, 5].astype(int)] + 0.3 * rng.standard_normal(n)
        h = rng.uniform(0.5, 2.0, size=n)
        leaf = np.sort(rng.choice(n, size=2500, replace=False)).astype(np.int32)
        return X, g, h, leaf
    if name == "efb":
        # mutually exclusive sparse columns (round 5): the reference bundles columns 2..9 into ONE feature group (exclusive feature bundling,
        # dataset.cpp FastFeatureBundling) of 376 bins -- wider than a byte
        X = np.zeros((n, 10))
        X[:, 0] = rng.uniform(size=n); X[:, 1] = rng.uniform(size=n)
        c12 = rng.integers(0, 12, size=n)
        for k in range(6):
            X[:, 2 + k] = (c12 == k) * rng.uniform(0.5, 2.0, size=n)
        X[:, 8] = (c12 == 8) * rng.integers(1, 4, size=n)
        X[:, 9] = (c12 == 9) * rng.uniform(size=n)
        g = np.sin(6 * X[:, 0]) + X[:, 3] - 0.5 * X[:, 5] + 0.4 * X[:, 8] + 0.3 * rng.standard_normal(n)
        h = rng.uniform(0.5, 2.0, size=n)
        leaf = np.sort(rng.choice(n, size=2500, replace=False)).astype(np.int32)
        return X, g, h, leaf
    X = rng.uniform(size=(n, F))
    X[:, 1] = 2.0 * X[:, 1] - 1.0
    X[:, 2] = np.round(X[:, 2] * 12) / 12
    X[:, 3] = np.where(rng.uniform(size=n) < 0.75, 0.4, X[:, 3])
    X[:, 4] = (rng.uniform(size=n) < 0.6) * rng.uniform(size=n)
    if name == "nan":
        X[rng.uniform(size=n) < 0.07, 0] = np.nan
        X[rng.uniform(size=n) < 0.3, 5] = np.nan
    g = np.sin(6 * np.nan_to_num(X[:, 0])) + 0.7 * (np.nan_to_num(X[:, 1]) > 0.2) - 0.5 * np.nan_to_num(X[:, 4]) + 0.3 * rng.standard_normal(n)
    h = rng.uniform(0.5, 2.0, size=n)
    leaf = np.sort(rng.choice(n, size=2500, replace=False)).astype(np.int32)
    return X, g, h, leaf


# One whole tree (SerialTreeLearner::Train) -- tests/golden/tree_ref.npz.  name -> (split data set, LightGBM parameter string)
TREE_CASES = {
    "plain_l31": ("plain", "max_bin=255 num_leaves=31 min_data_in_leaf=20 lambda_l2=0"),
    "plain_l15_reg": ("plain", "max_bin=63 num_leaves=15 min_data_in_leaf=40 lambda_l2=2.5 min_gain_to_split=0.05"),
    "nan_l20": ("nan", "max_bin=63 num_leaves=20 min_data_in_leaf=10 lambda_l2=0.1"),
    "zero_missing_l12": ("zero_missing", "max_bin=63 num_leaves=12 min_data_in_leaf=25 lambda_l2=0 zero_as_missing=true"),
    # the other regularisation paths of FindBestThreshold (feature_histogram.hpp:137-161): L1, max_delta_step, path smoothing
    "plain_l1": ("plain", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.5 lambda_l1=3.0"),
    # (max_delta_step clips both children of many candidates to the same output: their gain is exactly 0 up to rounding noise, and a tree
    #  that may split on a gain of 1e-13 depends on the last bits of the histogram sums -- min_gain_to_split keeps the case well-posed)
    "plain_mds": ("plain", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0 max_delta_step=0.2 min_gain_to_split=0.05"),
    "nan_smooth": ("nan", "max_bin=63 num_leaves=16 min_data_in_leaf=10 lambda_l2=0.1 path_smooth=25"),
    "plain_depth4": ("plain", "max_bin=63 num_leaves=31 min_data_in_leaf=20 lambda_l2=0 max_depth=4"),
    "plain_all_reg": ("plain", "max_bin=255 num_leaves=20 min_data_in_leaf=20 lambda_l2=1.0 lambda_l1=1.5 max_delta_step=0.3 path_smooth=10 min_gain_to_split=0.01"),
    # round 5 -- categorical features (FindBestThresholdCategoricalInner, feature_histogram.hpp:278-519): the sorted many-vs-many search with
    # explicit and with default (cat_smooth = 10, cat_l2 = 10, min_data_per_group = 100, max_cat_threshold = 32) settings, the one-hot search for
    # every categorical column (max_cat_to_onehot = 64) with L1, path smoothing with a small max_cat_threshold
    "cat_l15": ("cat", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.1 categorical_feature=2,5 min_data_per_group=50 cat_smooth=5"),
    "cat_defaults": ("cat", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.1 categorical_feature=2,5"),
    "cat_onehot_l1": ("cat", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.1 categorical_feature=2,5 max_cat_to_onehot=64 lambda_l1=0.5"),
    "cat_smooth": ("cat", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.1 categorical_feature=2,5 path_smooth=20 max_cat_threshold=4 cat_l2=1"),
    # round 5 -- exclusive feature bundling left ON (the reference's default): eight sparse columns in one group of 376 bins; the device keeps
    # one column per FEATURE (the fixture's bins are the unbundled columns, oracle/ref_driver.cpp)
    "efb_l15": ("efb", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0 enable_bundle=true"),
    "efb_rowwise_l15": ("efb", "max_bin=63 num_leaves=15 min_data_in_leaf=20 lambda_l2=0.5 enable_bundle=true force_row_wise=true"),
}
TREE_COMMON = " min_data_in_bin=1 enable_bundle=false force_col_wise=true verbosity=-1 num_threads=1 min_sum_hessian_in_leaf=0.001"
TREE_CASES_R5 = ("cat_l15", "cat_defaults", "cat_onehot_l1", "cat_smooth", "efb_l15", "efb_rowwise_l15")


def tree_cat_cfg(name):
    """(max_cat_to_onehot, max_cat_threshold, cat_smooth, cat_l2, min_data_per_group) of a tree case (the reference's defaults where not given)."""
    kv = dict(t.split("=") for t in TREE_CASES[name][1].split())
    return (int(kv.get("max_cat_to_onehot", 4)), int(kv.get("max_cat_threshold", 32)), float(kv.get("cat_smooth", 10.0)), float(kv.get("cat_l2", 10.0)),
            int(kv.get("min_data_per_group", 100)))


def tree_params(name):
    data, p = TREE_CASES[name]
    kv = dict(t.split("=") for t in p.split())
    cfg = (float(kv.get("lambda_l2", 0.0)), int(kv.get("min_data_in_leaf", 20)), 1e-3, float(kv.get("min_gain_to_split", 0.0)))
    reg = (float(kv.get("lambda_l1", 0.0)), float(kv.get("max_delta_step", 0.0)), float(kv.get("path_smooth", 0.0)))
    if any(v != 0.0 for v in reg):
        cfg = cfg + reg
    common = TREE_COMMON
    if "enable_bundle" in kv:              # (Config::Str2Map keeps the first value of a repeated key -- still, say it once)
        common = common.replace(" enable_bundle=false", "")
    if "force_row_wise" in kv:
        common = common.replace(" force_col_wise=true", "")
    return data, p + common, int(kv["num_leaves"]), cfg


def tree_max_depth(name):
    kv = dict(t.split("=") for t in TREE_CASES[name][1].split())
    return int(kv.get("max_depth", 0))


# Newton leaf update (row a9): leaf assignment per DATA point for the golden cases that carry a "leaf_values_*" entry
LEAF_CASES = {"r_exp_m30_none": 7, "u2d_n3000_exp_m30": 31, "u1d_n1000_mat15_m10": 16}   # case -> number of leaves


def make_leaf_index(name, n):
    """Half of the points get a spatially coherent leaf (as a tree split would), half a random one."""
    L = LEAF_CASES[name]
    rng = np.random.default_rng(1000 + L)
    coords, _ = make_data(GOLDEN_CASES[name])
    leaf = (np.floor(coords[:, 0] * L).astype(np.int64) % L)
    rnd = rng.uniform(size=n) < 0.5
    leaf[rnd] = rng.integers(0, L, size=int(rnd.sum()))
    return leaf.astype(np.int32), L


# Several independent realisations of the GP (cluster_ids): reference GPB_EvalNegLogLikelihood stored in tests/golden/clusters_ref.npz
CLUSTER_CASE = dict(n=1500, d=2, seed_data=41, num_clusters=3, cov_function="exponential", shape=0.5, m=12, ordering="random", seed=7,
                    cov_pars=(0.2, 1.3, 0.1))


def make_cluster_data(c=CLUSTER_CASE):
    rng = np.random.default_rng(c["seed_data"])
    coords = rng.uniform(size=(c["n"], c["d"]))
    y = np.cos(3 * coords[:, 0]) + 0.4 * rng.standard_normal(c["n"])
    ids = rng.integers(10, 10 + c["num_clusters"], size=c["n"]).astype(np.int32)      # interleaved, labels 10, 11, 12
    return coords, y, ids


# Parameter estimation (GPB_OptimCovPar): model / data of a GOLDEN_CASES entry (or "clusters" = CLUSTER_CASE) + optimiser settings.
# init = "r" -> (var(y)/2, var(y)/2, mean(dist)/3) as in R-package/tests/testthat/test_GPModel_gaussian_process.R:1096; None ->
# the reference's own FindInitCovPar.  Outputs of the reference's GPB_OptimCovPar are stored in tests/golden/optim_ref.npz.
R_GD = dict(optimizer_cov="gradient_descent", lr_cov=0.1, acc_rate_cov=0.5, delta_rel_conv=1e-6, use_nesterov_acc=True)
OPTIM_CASES = {
    # the R suite's own fit: 378 iterations, (0.03297349, 1.07691542, 0.11378505), nll 122.7680889 (test_GPModel_gaussian_process.R:1316-1324)
    "r_gd_nesterov_parcrit": dict(model="r_exp_m30_none", init="r", cpu=True, cfg=dict(R_GD, convergence_criterion="relative_change_in_parameters")),
    "r_gd_nesterov_llcrit": dict(model="r_exp_m30_none", init=None, cpu=True, cfg=dict(R_GD)),
    "r_gd_plain": dict(model="r_exp_m30_none", init=None, cpu=True, cfg=dict(optimizer_cov="gradient_descent", use_nesterov_acc=False)),
    "r_gd_offset0_lr1": dict(model="r_exp_m30_none", init="r", cpu=True,
                             cfg=dict(optimizer_cov="gradient_descent", lr_cov=1.0, acc_rate_cov=0.7, momentum_offset=0)),   # step halving
    "r_gd_maxiter7": dict(model="r_exp_m30_none", init=None, cpu=True, cfg=dict(R_GD, max_iter=7)),
    "r_lbfgs_default": dict(model="r_exp_m30_none", init=None, cpu=True, cfg=dict()),
    "r_lbfgs_init_m3": dict(model="r_exp_m30_none", init="r", cpu=True, cfg=dict(optimizer_cov="lbfgs", m_lbfgs=3, delta_rel_conv=1e-9)),
    "r_mat15_lbfgs": dict(model="r_mat15_m30_random", init=None, cpu=True, cfg=dict()),
    "r_mat25_gd": dict(model="r_mat25_m99_none", init=None, cpu=True, cfg=dict(R_GD)),
    "u1d_n1000_mat15_lbfgs": dict(model="u1d_n1000_mat15_m10", init=None, cpu=True, cfg=dict()),
    "dup2d_n600_gd": dict(model="dup2d_n600_exp_m15", init=None, cpu=False, cfg=dict(R_GD)),
    "u2d_n3000_lbfgs": dict(model="u2d_n3000_exp_m30", init=None, cpu=True, cfg=dict()),          # n > 1000: sub-sampled FindInitCovPar
    "u3d_n3000_mat25_gd": dict(model="u3d_n3000_mat25_m40", init=None, cpu=False, cfg=dict(R_GD)),
    "clusters_lbfgs": dict(model="clusters", init=None, cpu=False, cfg=dict()),
    "u5d_n700_mat15_lbfgs": dict(model="u5d_n700_mat15_m20", init=None, cpu=True, cfg=dict()),       # d = 5: the generality path end to end
    # derivative-free simplex search (OptimLib's nm as GPBoost ships it): likelihood evaluations only
    "r_nelder_mead": dict(model="r_exp_m30_none", init=None, cpu=True, cfg=dict(optimizer_cov="nelder_mead")),
    "r_nelder_mead_init_parcrit": dict(model="r_exp_m30_none", init="r", cpu=True,
                                       cfg=dict(optimizer_cov="nelder_mead", convergence_criterion="relative_change_in_parameters", delta_rel_conv=1e-6)),
    "u1d_n1000_mat15_nelder_mead_maxit25": dict(model="u1d_n1000_mat15_m10", init=None, cpu=True, cfg=dict(optimizer_cov="nelder_mead", max_iter=25)),
}


# Parameter estimation for the non-Gaussian likelihoods (Vecchia-Laplace): LAPLACE_CASES entry, likelihood, optimiser settings.
# Outputs of the reference's GPB_OptimCovPar: tests/golden/optim_laplace_ref.npz.  exact_it: the iteration count must match exactly
# (otherwise +-2: a preconditioned CG that stops at |r| < 1e-2 sits inside every evaluation).
OPTIM_LAPLACE_CASES = {
    "logit_n1500_lbfgs": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", cfg=dict(), exact_it=True),
    "logit_n1500_gd_nesterov": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", cfg=dict(optimizer_cov="gradient_descent"), exact_it=True),
    "logit_n1500_gd_plain": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit",
                                 cfg=dict(optimizer_cov="gradient_descent", use_nesterov_acc=False, lr_cov=0.05), exact_it=True),
    "probit_n1500_lbfgs": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_probit", cfg=dict(), exact_it=True),
    "poisson_n1500_lbfgs": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", cfg=dict(), exact_it=True),
    "poisson_n1500_gd_nesterov": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", cfg=dict(optimizer_cov="gradient_descent"), exact_it=False),
    "logit_u3d_n1200_lbfgs": dict(model="lap_u3d_n1200_mat25_m15", lik="bernoulli_logit", cfg=dict(), exact_it=True),
    # with fixed effects (offset of the location parameter: how the boosting loop passes the ensemble's scores)
    "logit_n1500_lbfgs_fe": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", cfg=dict(), exact_it=True, fe=True),
    "poisson_n1500_lbfgs_fe": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", cfg=dict(), exact_it=True, fe=True),
    # the same fits with the iterative solvers' thresholds tightened (cases.LAPLACE_TIGHT): no stopping-rule noise in any evaluation of the trajectory ->
    # the fit is reproducible to the accuracy of the evaluations (estimates 1e-6, likelihood 1e-8; the default-threshold fits above: 1e-4 / 1e-7)
    "logit_n1500_lbfgs_tight": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", cfg=dict(LAPLACE_TIGHT), exact_it=True, tight=True),
    "probit_n1500_lbfgs_tight": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_probit", cfg=dict(LAPLACE_TIGHT), exact_it=True, tight=True),
    "poisson_n1500_lbfgs_tight": dict(model="lap_u2d_n1500_mat15_m30", lik="poisson", cfg=dict(LAPLACE_TIGHT), exact_it=True, tight=True),
    "logit_u3d_n1200_lbfgs_fe_tight": dict(model="lap_u3d_n1200_mat25_m15", lik="bernoulli_logit", cfg=dict(LAPLACE_TIGHT), exact_it=True, tight=True, fe=True),
    # simplex search (likelihood evaluations only; starts from marginal variance 0.1, re_model_template.h:4904-4909); 12 iterations keep the CPU test short
    "logit_n1500_nelder_mead_maxit12": dict(model="lap_u2d_n1500_mat15_m30", lik="bernoulli_logit", cfg=dict(optimizer_cov="nelder_mead", max_iter=12),
                                            exact_it=True),
}


def optim_case(name):
    """-> (coords, y, cluster_ids | None, model dict, init_cov_pars | None, cfg dict)"""
    c = OPTIM_CASES[name]
    if c["model"] == "clusters":
        mc = CLUSTER_CASE
        coords, y, ids = make_cluster_data(mc)
    else:
        mc = GOLDEN_CASES[c["model"]]
        coords, y = make_data(mc)
        ids = None
    init = None
    if c["init"] == "r":
        from scipy.spatial.distance import pdist
        init = np.array([np.var(y, ddof=1) / 2, np.var(y, ddof=1) / 2, pdist(coords).mean() / 3])
    return coords, y, ids, mc, init, c["cfg"]


# Linear regression term X beta + GP (GPB_OptimLinRegrCoefCovPar): GOLDEN_CASES model, how X is made, optimiser settings.  The coefficients
# are profiled out by GLS (the reference's default optimizer_coef = "wls" for Gaussian data).  Outputs of the reference: tests/golden/optim_coef_ref.npz.
COEF_CASES = {
    # data of the R suite's "Vecchia approximation for Gaussian process model with linear regression term"
    # (test_GPModel_gaussian_process.R:1556-1583: X = cbind(1, sin((1:n - n/2)^2 * 2 pi / n)), beta = (2, 2), num_neighbors = n + 2)
    "r_m99_none_wls": dict(model="r_mat25_m99_none", cov_function="exponential", shape=0.5, X="r", init="r", cfg=dict(optimizer_cov="lbfgs")),
    "r_m30_none_wls_default": dict(model="r_exp_m30_none", X="r", init=None, cfg=dict()),
    "u2d_n3000_p4_wls": dict(model="u2d_n3000_exp_m30", X="random4", init=None, cfg=dict()),
}
COEF_PRED_COORDS = np.array([[0.1, 0.9], [0.2, 0.4], [0.7, 0.55]])      # coord_test / X_test of the R suite (:1575-1576)
COEF_PRED_X_R = np.array([[1., -0.5], [1., 0.2], [1., 0.4]])


def coef_case(name):
    """-> (coords, y, X, model dict, init_cov_pars | None, cfg dict, X_pred)"""
    c = COEF_CASES[name]
    mc = dict(GOLDEN_CASES[c["model"]])
    for k in ("cov_function", "shape"):
        if k in c:
            mc[k] = c[k]
    coords, y0 = make_data(mc)
    n = coords.shape[0]
    if c["X"] == "r":
        i = np.arange(1, n + 1)
        X = np.column_stack([np.ones(n), np.sin((i - n / 2) ** 2 * 2 * np.pi / n)])
        beta = np.array([2., 2.])
        Xp = COEF_PRED_X_R
    else:
        rng = np.random.default_rng(77)
        X = np.column_stack([np.ones(n), rng.standard_normal((n, 3))])
        X[:, 3] = 0.5 * X[:, 2] + X[:, 3]                               # correlated columns
        beta = np.array([1.5, -0.7, 0.3, 2.0])
        Xp = np.column_stack([np.ones(3), rng.standard_normal((3, 3))])
    y = y0 + X @ beta
    init = None
    if c["init"] == "r":
        from scipy.spatial.distance import pdist
        init = np.array([np.var(y, ddof=1) / 2, np.var(y, ddof=1) / 2, pdist(coords).mean() / 3])
    return coords, y, X, mc, init, c["cfg"], Xp


def synthetic(n, d, seed=1):
    """BASELINE.md's synthetic inputs: coords U[0,1]^d, y ~ N(0,1), default_rng(seed)."""
    rng = np.random.default_rng(seed)
    coords = rng.uniform(size=(n, d))
    y = rng.standard_normal(n)
    return coords, y


# Histogram fixture: inputs for the reference's own binning + Dataset::ConstructHistograms (oracle/ref_driver.cpp:refdrv_hist)
HIST_CASE = dict(n=20000, F=6, seed=3, max_bin=255, leaf_size=7000)


def make_hist_data(c=HIST_CASE):
    rng = np.random.default_rng(c["seed"])
    n, F = c["n"], c["F"]
    X = rng.uniform(size=(n, F))
    X[:, 2] = np.round(X[:, 2] * 10) / 10                             # 11 distinct values
    X[:, 4] = (rng.uniform(size=n) < 0.7) * rng.uniform(size=n)        # 30 % exact zeros (most-frequent bin)
    X[:, 1] = 2.0 * X[:, 1] - 1.0                                     # zero (default) bin in the middle: most_freq_bin > 0
    X[:, 3] = np.where(rng.uniform(size=n) < 0.8, 0.3, X[:, 3])       # 80 % one non-zero value: most_freq_bin > 0, dominant
    g = rng.standard_normal(n)
    h = rng.uniform(0.5, 2, size=n)
    leaf = np.sort(rng.choice(n, size=c["leaf_size"], replace=False)).astype(np.int32)
    return X, g, h, leaf


# Full-scale Vecchia ("VIF", gp_approx = "full_scale_vecchia"), Gaussian likelihood, Euclidean neighbours, kmeans++ inducing points:
# the reference's own GPB_EvalNegLogLikelihood (tests/golden/vif_ref.npz; oracle/make_golden.py vif).  name -> (n, d, cov_function,
# shape, m, num_ind_points, ordering, seed, list of cov_pars)
VIF_CASES = {
    "vif_u2d_n1500_exp_m15_k40_none": (1500, 2, "exponential", 0.5, 15, 40, "none", 1, [(0.2, 0.8, 0.15), (0.05, 1.5, 0.3)]),
    "vif_u2d_n1500_exp_m15_k40_random": (1500, 2, "exponential", 0.5, 15, 40, "random", 1, [(0.2, 0.8, 0.15)]),
    "vif_u2d_n3000_mat15_m30_k100_random": (3000, 2, "matern", 1.5, 30, 100, "random", 3, [(0.1, 1.0, 0.1), (0.3, 0.6, 0.25)]),
    "vif_u3d_n2000_mat25_m20_k64_random": (2000, 3, "matern", 2.5, 20, 64, "random", 2, [(0.1, 1.0, 0.2)]),
    "vif_u2d_n20000_exp_m30_k200_random": (20000, 2, "exponential", 0.5, 30, 200, "random", 1, [(0.1, 1.0, 0.1)]),
    "vif_u2d_n100000_exp_m30_k200_random": (100000, 2, "exponential", 0.5, 30, 200, "random", 1, [(0.1, 1.0, 0.1)]),
    # round 4: the per-point kernels' other tile counts -- 32 <= m <= 47 (three 16 x 16 MFMA tile rows), 48 <= m <= 62 (four), m > 62 (two wavefronts, scalar Gram)
    "vif_u2d_n1500_exp_m40_k50_random": (1500, 2, "exponential", 0.5, 40, 50, "random", 2, [(0.2, 0.8, 0.15)]),
    "vif_u2d_n1500_mat15_m55_k60_random": (1500, 2, "matern", 1.5, 55, 60, "random", 3, [(0.1, 1.0, 0.2)]),
    "vif_u2d_n1200_exp_m70_k40_random": (1200, 2, "exponential", 0.5, 70, 40, "random", 1, [(0.2, 0.8, 0.15)]),
}


def vif_data(name):
    n, d, cf, sh, m, k, ordering, seed, cps = VIF_CASES[name]
    coords, _ = synthetic(n, d, seed=11 + d)
    rng = np.random.default_rng(17)
    y = np.sin(4 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.5 * rng.standard_normal(n)
    return coords, y


# Full-scale Vecchia (VIF) with NON-GAUSSIAN likelihoods (round 6; FindModePostRandEffCalcMLLFSVA, likelihoods.h:3379-3750; iterative methods, "fitc" preconditioner with its
# own kmeans++ inducing points): the unmodified reference's values / gradients / fits (tests/golden/vif_laplace_ref.npz; oracle/make_golden.py vif_laplace).
# name -> dict(n, d, cov_function, shape, m, k = num_ind_points, ordering, seed, lik, aux (None: the likelihood's default), rank = fitc_piv_chol_preconditioner_rank
# (None: the reference's default, 200), cov_pars = [(sigma1^2, rho)])
# Solver thresholds of the "fitc" fixtures and of the device legs.  The Poisson case shows how far "identical inputs" must go for a gradient at 1e-8: the objective of the mode
# finding is flat to 1e-13 around modes 8e-7 apart, and the gradient moves by 4e-8 over that distance.  (i) With delta_conv_mode_finding = 1e-13 (LAPLACE_TIGHT) the fourth Newton
# increment of the objective, 2.6e-10, sits ON the threshold 1e-13 x 2793 -- round-off decides whether a fifth step is taken; (ii) with 1e-16 the fifth step IS entered, but its
# increment is below the objective's round-off and the sign of that noise decides whether Armijo halves the step away; (iii) the CG residual bound enters the gradient ~100-fold.
# 1e-12 / 1e-12 stops the implementations after the same Newton step: the oracle and the reference then agree to 1.5e-9 on all three cases (measured), the device on the logit and
# gamma cases.  The Poisson case keeps a floor: W = exp(location) reaches ~20 and (W^-1 + Sigma) has condition ~1e5 with right-hand sides of norm ~1e3, so the TRUE residual of
# its CG solves stalls near 1e-8 whatever the recursive residual says; the mode is then determined to ~8e-7 only and every implementation lands on one of two gradients 4e-8 apart
# (reference: 22.918063903 / ...356 / ...344 / ...298 over four threshold pairs; device and oracle: the same two values at other pairs).  grad_rtol of that case says so.
# The "vifdu" / "none" fixtures of the oracle test stay at LAPLACE_TIGHT (values only).
VIF_LAPLACE_TIGHT = dict(cg_delta_conv=1e-12, delta_conv_mode_finding=1e-12)
VIF_LAPLACE_CASES = {
    "vifl_u2d_n1500_exp_m15_k40_logit": dict(n=1500, d=2, cov_function="exponential", shape=0.5, m=15, k=40, ordering="random", seed=1, lik="bernoulli_logit", aux=None, rank=50,
                                             cov_pars=[(0.8, 0.25), (1.6, 0.1)]),
    "vifl_u2d_n2000_mat15_m20_k64_poisson": dict(n=2000, d=2, cov_function="matern", shape=1.5, m=20, k=64, ordering="random", seed=2, lik="poisson", aux=None, rank=None,
                                                 cov_pars=[(0.6, 0.2)], grad_rtol=1e-7),
    # (likelihoods whose information does not depend on the location parameter -- Fisher-Laplace t, lognormal: the auxiliary parameters' traces take another branch, likelihoods.h:5700-5760)
    "vifl_u2d_n1200_exp_m15_k40_t": dict(n=1200, d=2, cov_function="exponential", shape=0.5, m=15, k=40, ordering="random", seed=3, lik="t", aux=[0.4, 5.0], rank=50,
                                         cov_pars=[(0.7, 0.2)]),
    "vifl_u2d_n1200_mat15_m15_k40_lognormal": dict(n=1200, d=2, cov_function="matern", shape=1.5, m=15, k=40, ordering="random", seed=4, lik="lognormal", aux=0.2, rank=50,
                                                   cov_pars=[(0.7, 0.2)]),
    "vifl_u2d_n1200_exp_m15_k40_negbin": dict(n=1200, d=2, cov_function="exponential", shape=0.5, m=15, k=40, ordering="random", seed=5, lik="negative_binomial", aux=2.0, rank=50,
                                              cov_pars=[(0.7, 0.2)]),
    # sample weights (Likelihood::weights_: every per-datum term of the likelihood and of its derivatives x w_d): cases.vif_laplace_weights
    "vifl_u2d_n1200_exp_m15_k40_poisson_weighted": dict(n=1200, d=2, cov_function="exponential", shape=0.5, m=15, k=40, ordering="random", seed=6, lik="poisson", aux=None, rank=50,
                                                        cov_pars=[(0.7, 0.2)], weights=True),
    "vifl_u3d_n1500_mat25_m15_k40_gamma": dict(n=1500, d=3, cov_function="matern", shape=2.5, m=15, k=40, ordering="none", seed=1, lik="gamma", aux=2.0, rank=64,
                                               cov_pars=[(0.5, 0.3)]),
}


# Fits of VIF x non-Gaussian models by the reference's own GPB_OptimCovPar (tests/golden/vif_laplace_ref.npz, keys <fit>_*; oracle/make_golden.py vif_laplace_fit):
# fit name -> (case, GPB_SetOptimConfig settings on top of LAPLACE_TIGHT; init_cov_pars given so that neither side draws the sub-sample of FindInitCovPar)
VIF_LAPLACE_FITS = {
    "vifl_fit_logit_lbfgs": ("vifl_u2d_n1500_exp_m15_k40_logit", dict(optimizer_cov="lbfgs", init_cov_pars=[1.0, 0.2], max_iter=30)),
    "vifl_fit_gamma_lbfgs_aux": ("vifl_u3d_n1500_mat25_m15_k40_gamma", dict(optimizer_cov="lbfgs", init_cov_pars=[0.6, 0.25], max_iter=30, estimate_aux_pars=True)),
    "vifl_fit_logit_nelder_mead": ("vifl_u2d_n1500_exp_m15_k40_logit", dict(optimizer_cov="nelder_mead", init_cov_pars=[1.0, 0.2], max_iter=25)),
    # with a linear predictor X beta, X = (1, cos(4 x_0)): the coefficients ride in the lbfgs vector (GPB_OptimLinRegrCoefCovPar)
    "vifl_fit_logit_lbfgs_covariates": ("vifl_u2d_n1500_exp_m15_k40_logit", dict(optimizer_cov="lbfgs", init_cov_pars=[1.0, 0.2], max_iter=30, covariates=True)),
}


def vif_laplace_weights(name):
    """Sample weights of a VIF_LAPLACE_CASES entry with weights = True (data order), None otherwise."""
    c = VIF_LAPLACE_CASES[name]
    if not c.get("weights"):
        return None
    return np.random.default_rng(4242 + c["n"]).uniform(0.5, 2.0, size=c["n"])


def vif_laplace_covariates(coords):
    return np.c_[np.ones(coords.shape[0]), np.cos(4 * coords[:, 0])]


def vif_laplace_data(name):
    """-> (coords, y): a smooth latent surface, the response drawn from the case's likelihood."""
    c = VIF_LAPLACE_CASES[name]
    rng = np.random.default_rng(100 + c["n"] + c["d"])
    coords = rng.uniform(size=(c["n"], c["d"]))
    lat = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.2
    if c["lik"] == "bernoulli_logit":
        y = (rng.uniform(size=c["n"]) < 1.0 / (1.0 + np.exp(-1.5 * lat))).astype(np.float64)
    elif c["lik"] == "poisson":
        y = rng.poisson(np.exp(lat)).astype(np.float64)
    elif c["lik"] == "gamma":
        y = rng.gamma(2.0, np.exp(0.5 * lat) / 2.0)
    elif c["lik"] == "t":
        y = lat + 0.35 * rng.standard_t(4.0, size=c["n"])
    elif c["lik"] == "lognormal":
        y = np.exp(lat + 0.4 * rng.standard_normal(c["n"]))
    elif c["lik"] == "negative_binomial":
        mu = np.exp(0.8 * lat)
        y = rng.negative_binomial(3.0, 3.0 / (3.0 + mu)).astype(np.float64)
    else:
        raise ValueError(c["lik"])
    return coords, y


# Sample weights, Gaussian Vecchia model (re_model_template.h:403-431: error variance sigma^2 / w_i): the reference's own evaluations, fits and
# predictions (tests/golden/weights_ref.npz; oracle/make_golden.py weights).  name -> (n, d, cov_function, shape, m, ordering, seed)
WEIGHT_CASES = {
    "w_u2d_n2000_exp_m15_random": (2000, 2, "exponential", 0.5, 15, "random", 1),
    "w_u2d_n3000_mat15_m30_none": (3000, 2, "matern", 1.5, 30, "none", 1),
    "w_u3d_n2500_mat25_m40_random": (2500, 3, "matern", 2.5, 40, "random", 2),
    "w_u2d_n1200_exp_m70_random": (1200, 2, "exponential", 0.5, 70, "random", 3),       # m > 62: the LDS-resident kernel
}
WEIGHT_COV_PARS = [(0.2, 0.8, 0.15), (0.05, 1.5, 0.3)]


def weight_data(name):
    n, d, cf, sh, m, ordering, seed = WEIGHT_CASES[name]
    coords, _ = synthetic(n, d, seed=21 + d)
    rng = np.random.default_rng(23)
    w = rng.uniform(0.3, 3.0, size=n)
    y = np.sin(4 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.5 * rng.standard_normal(n) / np.sqrt(w)
    cpred = np.random.default_rng(29).uniform(size=(40, d))
    return coords, y, w, cpred


# The Gaussian Vecchia prediction types beyond 'order_obs_first_*' (Vecchia_utils.cpp:2203-2666): the reference's own predictive means and
# covariance matrices at given parameters (tests/golden/predtypes_ref.npz; oracle/make_golden.py predtypes).
# name -> (n, d, cov_function, shape, m, ordering, seed, n_pred, num_neighbors_pred, cov_pars)
PREDTYPE_CASES = {
    "pt_r100_exp_m30_none": (100, 2, "exponential", 0.5, 30, "none", 1, 3, 30, (0.02, 1.2, 0.9)),      # the R suite's data and prediction points
    "pt_u2d_n800_mat15_m20_random": (800, 2, "matern", 1.5, 20, "random", 3, 25, 15, (0.1, 1.0, 0.15)),
    "pt_u3d_n600_mat25_m15_random": (600, 3, "matern", 2.5, 15, "random", 2, 12, 40, (0.3, 0.7, 0.25)),  # num_neighbors_pred > num_neighbors, 32 < m <= 62
    "pt_u2d_n500_exp_m10_none": (500, 2, "exponential", 0.5, 10, "none", 1, 30, 70, (0.05, 1.5, 0.1)),   # m_pred > 62: the LDS-resident kernel
}
PRED_TYPES = ("order_pred_first", "latent_order_obs_first_cond_obs_only", "latent_order_obs_first_cond_all")


def predtype_data(name):
    n, d, cf, sh, m, ordering, seed, npred, mpred, cp = PREDTYPE_CASES[name]
    if name.startswith("pt_r100"):
        from oracle import orc
        coords, y = orc.r_fixture()
        return coords, y, np.array([[0.1, 0.9], [0.10001, 0.90001], [0.7, 0.55]])
    coords, _ = synthetic(n, d, seed=31 + d)
    rng = np.random.default_rng(37)
    y = np.sin(4 * coords[:, 0]) * np.cos(3 * coords[:, -1]) + 0.3 * rng.standard_normal(n)
    cpred = np.random.default_rng(41).uniform(0.3, 0.6, size=(npred, d))      # close together: prediction points neighbour each other
    return coords, y, cpred


# Repeated locations under a non-Gaussian likelihood (the reference's unique-location mapping, Vecchia_utils.cpp:1156-1168): 400 distinct 2D
# locations, 1200 data; tests/golden/laplace_dup_ref.npz (oracle/make_golden.py laplace_dup).  name -> (cov_function, shape, m, ordering, seed)
def laplace_coef_data(lik, n_cov=3):
    """Non-Gaussian data with a linear predictor for the fits with covariates (coefficients inside the lbfgs vector): the coordinates of
    LAPLACE_CASES['lap_u2d_n1500_mat15_m30'], X = (1, sin(3 c0 + c1), c1^2 - 0.3)[:, :n_cov], response drawn around surface + X beta0.
    -> (coords, y, X)"""
    c = LAPLACE_CASES["lap_u2d_n1500_mat15_m30"]
    rng = np.random.default_rng(c["seed_data"] + 100)
    n = c["n"]
    coords = rng.uniform(size=(n, 2))
    X = np.c_[np.ones(n), np.sin(3 * coords[:, 0] + coords[:, 1]), coords[:, 1] ** 2 - 0.3][:, :n_cov]
    beta0 = np.array([0.4, -0.8, 1.1])[:n_cov]
    surf = 0.9 * np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, 1])
    eta = surf + X @ beta0
    if lik == "poisson":
        y = rng.poisson(np.exp(0.6 * eta)).astype(np.float64)
    else:
        y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    return coords, y, X


LAPLACE_DUP_CASES = {"dup_exp_m15_none": ("exponential", 0.5, 15, "none", 1), "dup_mat15_m20_random": ("matern", 1.5, 20, "random", 3)}
LAPLACE_DUP_COV_PARS = [(1.0, 0.2), (0.5, 0.1)]


def laplace_dup_data(lik):
    rng = np.random.default_rng(3)
    nu, nd = 400, 1200
    cu = rng.uniform(size=(nu, 2))
    idx = np.concatenate([np.arange(nu), rng.integers(0, nu, size=nd - nu)]); rng.shuffle(idx)
    lat = np.sin(4 * cu[:, 0]) * np.cos(3 * cu[:, 1])
    if lik == "poisson":
        y = rng.poisson(np.exp(0.5 * lat[idx])).astype(np.float64)
    else:
        y = (rng.uniform(size=nd) < 1.0 / (1.0 + np.exp(-2.0 * lat[idx]))).astype(np.float64)
    fe = 0.3 * np.cos(5 * np.arange(nd) / nd)                         # differs between the data of one location
    cpred = np.random.default_rng(5).uniform(size=(30, 2))
    return cu[idx], y, fe, cpred


# What the reference's OWN predictions reproduce to (relative to the scale of the vector): its mode finding ends on the rounding noise of the objective
# (CheckConvergenceModeFinding, likelihoods.h:16078-16128: a change below the threshold OR any decrease ends it; the Armijo test :3929-3966 rejects a last Newton step
# whose gain, ~ curvature x error^2, is below that noise ~ 1e-16 |objective|), so the mode is defined to ~ sqrt(1e-16 |objective| / curvature) ~ 1e-7 ... 1e-6 only:
# between delta_conv_mode_finding = 1e-13 and 1e-16 its Cholesky-based predictions move by up to 9.8e-7 (t_u3d_n1200; *_pred_spread in the fixtures), and the C
# restatement of the same algorithm ends 2.1e-7 from it on negbin_n1500 whatever the thresholds.  Round 5 compared at 1e-5.
PRED_REPRODUCIBILITY = 1e-6


def assert_pred_close(got, ref, spread=0.0, rtol=1e-8):
    """Predictions of the non-Gaussian models against the reference's Cholesky-based fixtures (converged mode, oracle/make_golden.py laplace_pred_refresh): 1e-8 per
    entry, plus -- relative to the largest entry of the vector -- what the reference itself reproduces to (PRED_REPRODUCIBILITY above, or three times the case's own
    measured spread if that is larger)."""
    import numpy as _np
    ref = _np.asarray(ref, dtype=_np.float64)
    scale = float(_np.abs(ref).max())
    _np.testing.assert_allclose(_np.asarray(got, dtype=_np.float64), ref, rtol=rtol, atol=max(PRED_REPRODUCIBILITY * scale, 3.0 * float(spread)))


def check_predictions_against_reference(gpb, kw, g, name, y, cp, aux=None, fixed_effects=None):
    """Latent and response predictions of a non-Gaussian model the way the fixture's generator made them: a FRESH model per prediction (no earlier evaluation whose mode
    the Newton iteration would continue from), auxiliary parameters through init_aux_pars, thresholds LAPLACE_PRED_TIGHT."""
    import numpy as _np
    sp = g[name + "_pred_spread"] if (name + "_pred_spread") in g else _np.zeros(4)
    for resp in ((False, True) if (name + "_resp_mu") in g else (False,)):
        mdl = gpb.GPModel(**kw)
        params = dict(LAPLACE_PRED_TIGHT)
        if aux is not None:
            params["init_aux_pars"] = _np.atleast_1d(_np.asarray(aux, dtype=_np.float64))
        mdl.set_optim_params(params)
        pr = mdl.predict(y=y, gp_coords_pred=g[name + "_coords_pred"], cov_pars=cp, predict_var=True, predict_response=resp)
        key = "_resp" if resp else "_latent"
        assert_pred_close(pr["mu"], g[name + key + "_mu"], sp[2 if resp else 0])
        assert_pred_close(pr["var"], g[name + key + "_var"], sp[3 if resp else 1])


def laplace_coef_weights(n):
    """Sample weights of the weighted fits with covariates (round 6): uniform(0.3, 2.5), every 17th 0.05, every 11th exactly 1 (data order)."""
    w = np.random.default_rng(7311).uniform(0.3, 2.5, size=n)
    w[::17] = 0.05
    w[::11] = 1.0
    return w
