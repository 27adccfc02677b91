"""Stage-by-stage comparison of the device Laplace gradient with the oracle (development aid; prints, never asserts)."""
import sys, time, traceback
import numpy as np
sys.path.insert(0, ".")
from oracle import orc
from tests import cases
from gpboost_amd import shim

RC = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}
orc.build()


def stage(name, fn):
    t0 = time.time()
    try:
        fn()
    except Exception:
        print("STAGE %s FAILED" % name); traceback.print_exc(file=sys.stdout)
    print("  [%s %.1fs]" % (name, time.time() - t0), flush=True)


def run(n, d, m, ct, lik):
    print("=== n=%d d=%d m=%d ct=%d %s" % (n, d, m, ct, lik), flush=True)
    coords, y = cases.synthetic_binary(n, d, seed=600 + n)
    if lik == "poisson":
        y = np.random.default_rng(7).poisson(1.0 + y).astype(np.float64)
    perm, co, nn = orc.vecchia_setup(coords, m, "random", 4)
    var, a = 0.9, RC[ct] / 0.15
    st = shim.VecchiaState(co, m)
    st.set_neighbors(nn)

    def s1():
        dA, dD = st.laplace_range_deriv(ct, var, a)
        A, D, Ag, Dg, bad = orc.vecchia_factor(co, nn, ct, var, a, gauss=False, grad=True)
        print("  dA max abs err %.3e (scale %.3e)  dD max abs err %.3e (scale %.3e)" % (
            np.abs(dA - Ag[1]).max(), np.abs(Ag[1]).max(), np.abs(dD - Dg[1]).max(), np.abs(Dg[1]).max()))
    stage("range_deriv", s1)

    def s2():
        st.laplace_set_likelihood(lik)
        st.laplace_set_labels(y[perm].astype(np.int32))
        negll, g, parts = st.laplace_eval_grad(ct, var, a, want_parts=True)
        ref, gref, op = orc.vecchia_laplace_grad(co, nn, ct, var, a, y[perm], likelihood=lik, want_parts=True)
        print("  negll %.12g ref %.12g rel %.2e" % (negll, ref, abs(negll - ref) / abs(ref)))
        print("  dlogdet_dmode max err %.3e (scale %.3e)" % (np.abs(parts["dlogdet_dmode"] - op["dlogdet_dmode"]).max(), np.abs(op["dlogdet_dmode"]).max()))
        print("  implicit_solve max err %.3e (scale %.3e)" % (np.abs(parts["implicit_solve"] - op["implicit_solve"]).max(), np.abs(op["implicit_solve"]).max()))
        print("  per_par dev\n", parts["per_par"], "\n  per_par orc\n", op["per_par"])
        print("  grad dev", g, "orc", gref)
        _, g2 = st.laplace_eval_grad(ct, var, a)
        print("  repeat identical:", np.array_equal(g, g2))
    stage("gradient", s2)
    st.close()


def fit(name):
    import os, gpboost_amd as gpb
    g = np.load(os.path.join("tests", "golden", "optim_laplace_ref.npz"))
    oc = cases.OPTIM_LAPLACE_CASES[name]
    c = cases.LAPLACE_CASES[oc["model"]]
    coords, y = cases.make_count_data(c) if oc["lik"] == "poisson" else cases.make_binary_data(c)
    mdl = gpb.GPModel(likelihood=oc["lik"], gp_coords=coords, cov_function=c["cov_function"], cov_fct_shape=c["shape"], gp_approx="vecchia",
                      num_neighbors=c["m"], vecchia_ordering=c["ordering"], seed=c["seed"])
    params = dict(oc["cfg"]); params["init_cov_pars"] = g[name + "_init_cov_pars"]
    def s():
        mdl.fit(y, params=params)
        print("  %s: it %d (ref %d) cov %s (ref %s) nll %.10g (ref %.10g)" % (name, mdl.get_num_optim_iter(), int(g[name + "_num_it"]), mdl.get_cov_pars(),
              g[name + "_cov_pars"], mdl.get_current_neg_log_likelihood(), float(g[name + "_negll"])))
    stage("fit " + name, s)


if __name__ == "__main__":
    run(2000, 2, 10, 1, "bernoulli_logit")
    run(1500, 3, 20, 2, "poisson")
    fit("logit_n1500_lbfgs")
    fit("logit_n1500_gd_nesterov")
