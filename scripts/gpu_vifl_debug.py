"""Debug aid: device vs oracle parts of the VIF x non-Gaussian gradient for one case (scripts/gpu_run.sh py:...)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from oracle import orc
from gpboost_amd import shim
name = sys.argv[1] if len(sys.argv) > 1 else "vifl_u2d_n2000_mat15_m20_k64_poisson"
tight = dict(cg_delta_conv=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8, delta_conv_mode_finding=float(sys.argv[3]) if len(sys.argv) > 3 else 1e-13)
RC = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}
c = cases.VIF_LAPLACE_CASES[name]
g = np.load(os.path.join("tests", "golden", "vif_laplace_ref.npz"))
coords, y = cases.vif_laplace_data(name)
rank = 200 if c["rank"] is None else c["rank"]
perm, co, nn, ip, ip2 = orc.vif_setup(coords, c["m"], c["k"], c["ordering"], c["seed"], num_ind_points_preconditioner=rank)
ct = orc.cov_type_id(c["cov_function"], c["shape"])
var, rho = c["cov_pars"][0]
a = RC[ct] / rho
st = shim.VecchiaState(co, c["m"])
st.set_neighbors(nn); st.vif_set_inducing_points(ip); st.laplace_set_likelihood(c["lik"])
if c["lik"] == "gamma":
    st.laplace_set_response_real(y[perm]); st.laplace_set_aux(c["aux"])
else:
    st.laplace_set_labels(y[perm].astype(np.int32))
st.laplace_set_preconditioner("fitc", rank); st.laplace_set_inducing_points(ip2)
nll, grad, parts = st.laplace_eval_grad(ct, var, a, want_parts=True, **tight)
on, og, op = orc.vif_laplace_grad(co, nn, ip, ip2, ct, var, a, y[perm], likelihood=c["lik"], aux=c["aux"], want_parts=True, cg_delta_conv=tight["cg_delta_conv"], delta_conv_mode=tight["delta_conv_mode_finding"])
np.set_printoptions(precision=12, linewidth=200)
print("nll dev/orc/ref", repr(nll), repr(on), float(g[name + "_fitc_negll_direct_0"]))
print("grad dev", grad); print("grad orc", og); print("grad ref", g[name + "_fitc_grad_0"])
print("per_par dev\n", parts["per_par"]); print("per_par orc\n", op["per_par"])
print("dld max diff", np.abs(parts["dlogdet_dmode"] - op["dlogdet_dmode"]).max(), "sv max diff", np.abs(parts["implicit_solve"] - op["implicit_solve"]).max(), "sv max", np.abs(op["implicit_solve"]).max())
nll2, info = st.laplace_logit(ct, var, a, want_mode=True, **tight)
with orc.vif_laplace(co, nn, ip, ct, var, a, "fitc", ip2) as ctx:
    f = ctx.factor
    on2, oinfo = orc.vecchia_laplace_logit(co, nn, ct, var, a, y[perm], likelihood=c["lik"], factor=(f["A"], f["D"]), aux=c["aux"], cg_delta_conv=tight["cg_delta_conv"], delta_conv_mode=tight["delta_conv_mode_finding"])
print("dev  it", info["newton_it"], info["cg_it"], info["lanczos_it"], "logdet", repr(info["log_det"]), "mll_no_det", repr(info["mll_no_det"]))
print("orc  it", oinfo["newton_it"], oinfo["cg_it"], oinfo["lanczos_it"], "logdet", repr(oinfo["log_det"]), "mll_no_det", repr(oinfo["mll_no_det"]))
print("mode max diff", np.abs(info["mode"] - oinfo["mode"]).max())
