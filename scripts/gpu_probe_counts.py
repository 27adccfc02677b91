import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tests import cases
from oracle import orc
import gpboost_amd
from gpboost_amd import shim
RC = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}
TIGHT_ORC = dict(cg_delta_conv=cases.LAPLACE_TIGHT["cg_delta_conv"], delta_conv_mode=cases.LAPLACE_TIGHT["delta_conv_mode_finding"])
pc = dict(cases.LAPLACE_PIVCHOL_CASES["pc_logit_n2000"], rank=50)
c = cases.LAPLACE_CASES[pc["model"]]
coords, y = cases.make_pivchol_data(pc)
perm, co, nn = orc.vecchia_setup(coords, c["m"], c["ordering"], c["seed"])
ct = orc.cov_type_id(c["cov_function"], c["shape"])
var, rho = c["cov_pars"][0]; a = RC[ct] / rho
for pcn in ("vadu", "pivoted_cholesky"):
    for t in (48, 52, 56, 60, 64):
        st = shim.VecchiaState(co, c["m"]); st.set_neighbors(nn); st.laplace_set_likelihood(pc["lik"]); st.laplace_set_labels(y[perm].astype(np.int32))
        if pcn == "vadu": st.laplace_set_preconditioner("vadu")
        else: st.laplace_set_preconditioner("pivoted_cholesky", 50)
        nll, grad = st.laplace_eval_grad(ct, var, a, num_rand_vec=t, **cases.LAPLACE_TIGHT)
        if pcn == "vadu":
            on, og = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=pc["lik"], num_rand_vec=t, **TIGHT_ORC)
        else:
            with orc.pivoted_cholesky_preconditioner(co, ct, var, a, rank=50):
                on, og = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=pc["lik"], num_rand_vec=t, **TIGHT_ORC)
        print(pcn, t, "rel diff value %.2e" % (abs(nll - on) / abs(on)), "grad %.2e" % (np.abs(grad - og).max() / np.abs(og).max()), flush=True)
        st.close()
