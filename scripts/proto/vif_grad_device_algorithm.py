"""numpy prototype of the DEVICE formulation of the full-scale Vecchia (VIF) gradient (gpboost_amd/csrc/vif_kernels.hip):
row-major n x k matrices, the low-rank derivative parts of the per-point systems folded into three k-vectors per point, all traces as
row-wise dot products.  Checked against oracle/orc.py:vif_grad_terms (the restatement of the reference's expressions).  Development aid."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from oracle import orc
from scipy.spatial.distance import cdist
from scipy.linalg import cholesky, solve_triangular, cho_solve


def device_algorithm(co, nn, ip, ct, var, a, y):
    n, m = nn.shape
    k = ip.shape[0]
    # host, k x k
    dip = cdist(ip, ip)
    Sm0 = orc._matern(ct, dip, var, a); Sm = Sm0.copy(); Sm[np.diag_indices(k)] *= 1 + 1e-6
    Lm = cholesky(Sm, lower=True); Linv = np.linalg.inv(Lm); Si = Linv.T @ Linv
    dSm = [Sm0, orc._matern_grad_log_range(ct, dip, var, a)]
    Mp = [Si @ dSm[p] @ Si for p in range(2)]
    # D1, D2
    dnm = cdist(co, ip)
    C = orc._matern(ct, dnm, var, a); dC = orc._matern_grad_log_range(ct, dnm, var, a)
    V = C @ Linv.T
    # D3 factor
    A = np.zeros((n, m)); D = np.empty(n); u = np.empty(n)
    chol = [None] * n
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = var + 1 - V[i] @ V[i]; u[i] = y[i]
        if idx.size:
            Cnn = orc._matern(ct, cdist(co[idx], co[idx]), var, a) - V[idx] @ V[idx].T + np.eye(idx.size)
            c = orc._matern(ct, cdist(co[idx], co[i:i + 1])[:, 0], var, a) - V[idx] @ V[i]
            chol[i] = (cholesky(Cnn, lower=True), True)
            A[i, :idx.size] = cho_solve(chol[i], c)
            D[i] -= A[i, :idx.size] @ c; u[i] -= A[i, :idx.size] @ y[idx]
    # D4 SpMM
    def Bmul(X):
        out = X.copy()
        for i in range(n):
            idx = nn[i][nn[i] >= 0]
            out[i] -= A[i, :idx.size] @ X[idx]
        return out
    Q = Bmul(C); QdC = Bmul(dC)
    # D5 gram
    Wg = Q.T @ (Q / D[:, None]); r = Q.T @ (u / D)
    W = Sm + Wg; Lw = cholesky(W, lower=True); w = cho_solve((Lw, True), r); Winv = cho_solve((Lw, True), np.eye(k))
    quad = u @ (u / D) - r @ w
    logdet = np.log(D).sum() - 2 * np.log(np.diag(Lm)).sum() + 2 * np.log(np.diag(Lw)).sum()
    # D6 GEMMs
    Hm = Q @ Winv; X1 = Q @ Si; V1 = Q @ (2 * Si - Mp[0]); X2r = QdC @ Si - Q @ Mp[1]
    # D7
    v = (u - Q @ w) / D; z = y - C @ w
    # D8 per point
    s = np.zeros((2, 7))
    for i in range(n):
        idx = nn[i][nn[i] >= 0]; kk = idx.size
        all_ = np.concatenate([idx, [i]]).astype(int)
        At = np.concatenate([A[i, :kk], [-1.0]])
        dist = cdist(co[all_], co[all_])
        K = orc._matern(ct, dist, var, a); dK = orc._matern_grad_log_range(ct, dist, var, a)
        t0 = C[all_] @ V1[i]; t1 = dC[all_] @ X1[i] + C[all_] @ X2r[i]
        g = C[idx] @ Hm[i]
        h = [K @ At + t0, dK @ At + t1]
        kappa = Q[i] @ Hm[i]; di = 1.0 / D[i]
        for p in range(2):
            x = cho_solve(chol[i], h[p][:kk]) if kk else np.zeros(0)
            dD = A[i, :kk] @ h[p][:kk] - h[p][kk]
            s[p, 1] += dD / D[i]
            s[p, 2] += 2 * (x @ z[idx]) * v[i] - v[i] ** 2 * dD
            s[p, 3] += di * (x @ g)
            s[p, 4] += dD * di * di * kappa
        s[0, 5] += di * kappa; s[1, 5] += di * (QdC[i] @ Hm[i])
        s[0, 6] += v[i] * (Q[i] @ w); s[1, 6] += v[i] * (QdC[i] @ w)
    gout = np.zeros((2, 2))
    for p in range(2):
        dquad = s[p, 2] - 2 * s[p, 6] + w @ dSm[p] @ w
        dlogdet = s[p, 1] - np.trace(Si @ dSm[p]) + np.trace(Winv @ dSm[p]) + 2 * s[p, 5] + 2 * s[p, 3] - s[p, 4]
        gout[p] = [0.5 * dquad, 0.5 * dlogdet]
    return quad, logdet, gout


if __name__ == "__main__":
    import cases
    for name in ["vif_u2d_n1500_exp_m15_k40_random", "vif_u3d_n2000_mat25_m20_k64_random"]:
        n, d, cf, sh, m, k, ordering, seed, cps = cases.VIF_CASES[name]
        coords, y = cases.vif_data(name)
        ct = orc.cov_type_id(cf, sh)
        perm, co, nn, ip = orc.vif_setup(coords, m, k, ordering, seed)
        pt = orc.transform_cov_pars(ct, np.array([0.2, 0.8, 0.15]))
        q0, l0, g0, *_ = orc.vif_grad_terms(co, nn, ip, ct, pt[1], pt[2], y[perm])
        q1, l1, g1 = device_algorithm(co, nn, ip, ct, pt[1], pt[2], y[perm])
        print(name, abs(q0 - q1) / abs(q0), abs(l0 - l1) / abs(l0), np.abs(g0 - g1) / np.abs(g0).max())
