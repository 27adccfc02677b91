"""Debug aid: VIF x logit with covariates -- value at the reference's estimates, and this library's fit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
import gpboost_amd as gpb
name = "vifl_u2d_n1500_exp_m15_k40_logit"
c = cases.VIF_LAPLACE_CASES[name]
g = np.load(os.path.join("tests", "golden", "vif_laplace_ref.npz"))
coords, y = cases.vif_laplace_data(name)
X = cases.vif_laplace_covariates(coords)
fit = "vifl_fit_logit_lbfgs_covariates"
def mk():
    m = gpb.GPModel(gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], likelihood=c["lik"], gp_approx="full_scale_vecchia", num_neighbors=c["m"], num_ind_points=c["k"], vecchia_ordering=c["ordering"], seed=c["seed"])
    m.set_optim_params(dict(cases.LAPLACE_TIGHT, fitc_piv_chol_preconditioner_rank=c["rank"]))
    return m
m = mk()
v = m.neg_log_likelihood(cov_pars=g[fit + "_cov_pars"], y=y, fixed_effects=X @ g[fit + "_coef"])
print("value at the reference's estimates: %.10f (reference's final %.10f)" % (v, float(g[fit + "_negll"])))
m = mk()
m.fit(y, X=X, params=dict(cases.LAPLACE_TIGHT, optimizer_cov="lbfgs", init_cov_pars=[1.0, 0.2], maxit=30, init_coef_aux_pars_from_iid_model=False, trace=True))
print("fit here:", m.get_cov_pars(), m.get_coef(), m.get_num_optim_iter(), m.get_current_neg_log_likelihood(), "reference:", g[fit + "_cov_pars"], g[fit + "_coef"], int(g[fit + "_num_it"]))
