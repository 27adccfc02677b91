"""ctypes front-end of the CPU oracle (oracle/libgpb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by gpboost_amd/.  See gpb_oracle.c for the
reference file:line each routine restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COV_TYPES = {"exponential": 0, "matern_0.5": 0, "matern_1.5": 1, "matern_2.5": 2}


def cov_type_id(cov_function, shape=0.5):
    """'exponential' == Matern(0.5) (include/GPBoost/cov_fcts.h:113-203)."""
    if cov_function == "exponential":
        return 0
    if cov_function == "matern":
        return {0.5: 0, 1.5: 1, 2.5: 2}[float(shape)]
    raise ValueError(cov_function)


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgpb_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def transform_cov_pars(cov_type, pars_orig):
    out = np.empty(3)
    lib().orc_transform_cov_pars(C.c_int(cov_type), _p(_f64(pars_orig), C.c_double), _p(out, C.c_double))
    return out


def shuffle(n, seed):
    idx = np.empty(n, dtype=np.int32)
    lib().orc_shuffle(C.c_int(n), C.c_int(seed), _p(idx, C.c_int))
    return idx


def sort_indices(v):
    v = _f64(v)
    idx = np.empty(v.size, dtype=np.int32)
    lib().orc_sort_indices(_p(v, C.c_double), C.c_int(v.size), _p(idx, C.c_int))
    return idx


def coords_sum(coords):
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    out = np.empty(n)
    lib().orc_coords_sum(_p(cm, C.c_double), C.c_int(n), C.c_int(d), _p(out, C.c_double))
    return out


def neighbors(coords, m, want_sqd=False):
    """coords: (n, d) in Vecchia order.  Returns int32 (n, m'), -1 padded (m' = min(m, n-1))."""
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    m = min(m, n - 1)
    ss = sort_indices(coords_sum(cm))
    nn = np.empty((n, m), dtype=np.int32)
    sqd = np.full((n, m), np.inf) if want_sqd else None
    lib().orc_vecchia_neighbors(_p(cm, C.c_double), C.c_int(n), C.c_int(d), C.c_int(m),
                                _p(ss, C.c_int), _p(nn, C.c_int), _p(sqd, C.c_double))
    return (nn, sqd) if want_sqd else nn


def neighbors_range(coords, m, start_at, end_search_at):
    """find_nearest_neighbors_Vecchia_fast with its start_at / end_search_at arguments; returns all n rows (rows < start_at = -1)."""
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    m = min(m, (n - 2 if end_search_at < 0 else end_search_at) + 1)
    ss = sort_indices(coords_sum(cm))
    nn = np.empty((n, m), dtype=np.int32)
    lib().orc_vecchia_neighbors_range(_p(cm, C.c_double), C.c_int(n), C.c_int(d), C.c_int(m), _p(ss, C.c_int), C.c_int(start_at),
                                      C.c_int(end_search_at), _p(nn, C.c_int), None)
    return nn


def predict_obs_only(coords_obs, y_obs, coords_pred, cov_type, pars_trans, m_pred, predict_response=False):
    """Vecchia prediction 'order_obs_first_cond_obs_only', Gaussian likelihood (CalcPredVecchiaObservedFirstOrder with
    CondObsOnly = true, src/GPBoost/Vecchia_utils.cpp:1701-2060): every prediction point conditions on its m_pred nearest OBSERVED
    points.  coords_obs / y_obs in Vecchia order; pars_trans = (sigma2, sigma1_2 / sigma2, a).  Returns (mean, var): var includes the
    error variance iff predict_response (re_model_template.h: the nugget is added in the factor and subtracted again otherwise)."""
    co = np.asarray(coords_obs, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n_obs, n_pred = co.shape[0], cp.shape[0]
    call = np.vstack([co, cp])
    nn = neighbors_range(call, m_pred, n_obs, n_obs - 1)
    A, D, bad = vecchia_factor(call, nn, cov_type, pars_trans[1], pars_trans[2], gauss=True)
    yall = np.concatenate([np.asarray(y_obs, dtype=np.float64), np.zeros(n_pred)])
    rows = slice(n_obs, n_obs + n_pred)
    mean = np.einsum("ij,ij->i", A[rows], np.where(nn[rows] >= 0, yall[np.maximum(nn[rows], 0)], 0.))
    var = pars_trans[0] * (D[rows] if predict_response else D[rows] - 1.0)
    return mean, var


def predict_cond_all(coords_obs, y_obs, coords_pred, cov_type, pars_trans, m_pred, predict_response=False):
    """Vecchia prediction 'order_obs_first_cond_all', Gaussian likelihood (CalcPredVecchiaObservedFirstOrder with CondObsOnly = false,
    src/GPBoost/Vecchia_utils.cpp:1701-2093): a prediction point conditions on its m_pred nearest points among the observed AND the
    preceding prediction points (neighbour search with end_search_at = -1, :1806-1822); mean = Bp^-1 (-Bpo y) (:2061-2064), covariance =
    Bp^-1 Dp Bp^-T (:2077-2090), nugget removed from its diagonal unless predict_response.  Dense algebra in the number of prediction
    points: a checker for small cases (the device path for this prediction type does not exist yet).  Returns (mean, cov)."""
    co = np.asarray(coords_obs, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n_obs, n_pred = co.shape[0], cp.shape[0]
    call = np.vstack([co, cp])
    nn = neighbors_range(call, m_pred, n_obs, -1)
    A, D, bad = vecchia_factor(call, nn, cov_type, pars_trans[1], pars_trans[2], gauss=True)
    Bpo = np.zeros((n_pred, n_obs)); Bp = np.eye(n_pred)
    for i in range(n_pred):
        for j, c in enumerate(nn[n_obs + i]):
            if c < 0:
                continue
            if c < n_obs:
                Bpo[i, c] -= A[n_obs + i, j]
            else:
                Bp[i, c - n_obs] -= A[n_obs + i, j]
    mean = np.linalg.solve(Bp, -Bpo @ np.asarray(y_obs, dtype=np.float64))
    Bpi = np.linalg.inv(Bp)
    cov = pars_trans[0] * (Bpi @ np.diag(D[n_obs:]) @ Bpi.T)
    if not predict_response:
        cov = cov - pars_trans[0] * np.eye(n_pred)
    return mean, cov


def _dense_B(nn, A):
    """I - A as a dense matrix from the factor rows (neighbour indices nn, -1 padded)."""
    n = nn.shape[0]
    B = np.eye(n)
    for i in range(n):
        for j, c in enumerate(nn[i]):
            if c >= 0:
                B[i, c] -= A[i, j]
    return B


def predict_pred_first(coords_obs, y_obs, coords_pred, cov_type, pars_trans, m_pred, predict_response=False):
    """Vecchia prediction 'order_pred_first', Gaussian likelihood (CalcPredVecchiaPredictedFirstOrder, src/GPBoost/Vecchia_utils.cpp:2203-2444):
    the prediction points come FIRST in the ordering, then the observed ones (Vecchia order); neighbours of every point among all preceding
    points (:2228-2255), factor rows with the nugget on every diagonal (:2388-2396), B = [[Bp, 0], [Bop, Bo]];
    cond_prec = Bp' Dp^-1 Bp + Bop' Do^-1 Bop (:2419), mean = -cond_prec^-1 Bop' Do^-1 Bo y (:2422-2423), covariance = cond_prec^-1
    (:2424-2441; the reference's expression with the factor of the sparse Cholesky), nugget removed from its diagonal unless predict_response
    (re_model_template.h:4134-4150).  Dense: small cases.  Returns (mean, cov)."""
    co = np.asarray(coords_obs, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n_obs, n_pred = co.shape[0], cp.shape[0]
    call = np.vstack([cp, co])
    nn = neighbors(call, min(m_pred, n_obs + n_pred - 1))
    A, D, bad = vecchia_factor(call, nn, cov_type, pars_trans[1], pars_trans[2], gauss=True)
    B = _dense_B(nn, A)
    Bp, Bop, Bo = B[:n_pred, :n_pred], B[n_pred:, :n_pred], B[n_pred:, n_pred:]
    Dp, Do = D[:n_pred], D[n_pred:]
    cond_prec = Bp.T @ (Bp / Dp[:, None]) + Bop.T @ (Bop / Do[:, None])
    mean = -np.linalg.solve(cond_prec, Bop.T @ ((Bo @ np.asarray(y_obs, dtype=np.float64)) / Do))
    cov = pars_trans[0] * np.linalg.inv(cond_prec)
    if not predict_response:
        cov = cov - pars_trans[0] * np.eye(n_pred)
    return mean, cov


def predict_latent(coords_obs, y_obs, coords_pred, cov_type, pars_trans, m_pred, cond_obs_only=True, predict_response=False, weights=None):
    """Vecchia predictions 'latent_order_obs_first_cond_obs_only' / '..._cond_all', Gaussian likelihood
    (CalcPredVecchiaLatentObservedFirstOrder, src/GPBoost/Vecchia_utils.cpp:2446-2666), no duplicate locations: a Vecchia approximation of the
    LATENT process on (observed, prediction) points -- no nugget, diagonal x (1 + 1e-10) (:2589) -- every point searching among the
    preceding points, restricted to the observed ones if cond_obs_only (:2517-2561); Sigma = B^-1 D B^-T (:2597-2600) and the conditional
    distribution of the prediction points given y = b_obs + eps with R^-1 = diag(weights) (:2601-2650), restated literally.
    Dense: small cases.  Returns (mean, cov) with the error variance on the diagonal iff predict_response (:2624-2626)."""
    co = np.asarray(coords_obs, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n_obs, n_pred = co.shape[0], cp.shape[0]
    call = np.vstack([co, cp])
    m = min(m_pred, n_obs if cond_obs_only else n_obs + n_pred - 1)
    nn = neighbors_range(call, m, 0, n_obs - 1 if cond_obs_only else -1)
    A, D, bad = vecchia_factor(call, nn, cov_type, pars_trans[1], pars_trans[2], gauss=False)
    Binv = np.linalg.inv(_dense_B(nn, A))
    Sigma = Binv @ np.diag(D) @ Binv.T
    Rinv = np.ones(n_obs) if weights is None else np.asarray(weights, dtype=np.float64)
    Soo = Sigma[:n_obs, :n_obs] + np.diag(1.0 / Rinv)
    Spo = Sigma[n_obs:, :n_obs]
    K = np.linalg.solve(Soo, Spo.T).T                       # ZpSigmaZoT (ZoSigmaZoT + R)^-1
    mean = K @ np.asarray(y_obs, dtype=np.float64)
    cov = Sigma[n_obs:, n_obs:] - K @ Spo.T
    if predict_response:
        cov = cov + np.eye(n_pred)
    return mean, pars_trans[0] * cov


def fisher_std_errors(coords, nn, cov_type, cov_pars, num_rand_vec=50, seed_rand=1, run_id=0):
    """Standard errors of (sigma2, sigma1_2, rho) of a Gaussian Vecchia model from the stochastic Fisher information on the ORIGINAL scale:
    REModelTemplate::CalcFisherInformation_Vecchia, Hutchinson branch (include/GPBoost/re_model_template.h:10137-10230; default since
    use_stochastic_trace_for_Fisher_information_Vecchia_ = true, :6033), include_error_var, !transf_scale:
        v_0 = Psi^-1 z,  v_k = B' D^-1 (-dB_k Psi z + dD_k B^-T z) - dB_k' B^-T z,  FI_kl = mean_z (v_k . v_l) / 2,  se = sqrt(diag(FI^-1)).
    Probes as GenRandVecNormalParallel(seed_rand_vec_trace, run id).  coords / nn in Vecchia order.  Dense algebra: a checker for small n
    (no device path computes standard errors yet).  Derivatives on the original scale by the chain rule from the transformed ones."""
    co = np.asarray(coords, dtype=np.float64)
    n = co.shape[0]
    s2, s1, rho = [float(v) for v in cov_pars]
    pt = transform_cov_pars(cov_type, cov_pars)
    A, Dt, Ag, Dg, bad = vecchia_factor(co, nn, cov_type, pt[1], pt[2], gauss=True, grad=True)
    B = np.eye(n); dBt = [np.zeros((n, n)), np.zeros((n, n))]
    for i in range(n):
        for j, c in enumerate(nn[i]):
            if c >= 0:
                B[i, c] = -A[i, j]
                dBt[0][i, c] = -Ag[0][i, j]; dBt[1][i, c] = -Ag[1][i, j]
    D = s2 * Dt
    dB = [dBt[0] / s1, dBt[1] * (-1.0 / rho)]                  # d / d sigma1_2 = (d / d log ratio) / sigma1_2;  d / d rho = (d / d log a) (-1 / rho)
    dD = [s2 * Dg[0] / s1, s2 * Dg[1] * (-1.0 / rho)]
    Z = np.asarray(gen_rand_normal(n, num_rand_vec, seed=seed_rand, run_id=run_id))
    Bi = np.linalg.inv(B)
    BTiZ = Bi.T @ Z
    PsiZ = Bi @ (D[:, None] * BTiZ)
    V = [B.T @ ((B @ Z) / D[:, None])]
    for k in range(2):
        V.append(B.T @ ((-dB[k] @ PsiZ + dD[k][:, None] * BTiZ) / D[:, None]) - dB[k].T @ BTiZ)
    FI = np.array([[(V[k] * V[l]).sum(axis=0).mean() / 2 for l in range(3)] for k in range(3)])
    return np.sqrt(np.diag(np.linalg.inv(FI)))


def vecchia_factor(coords, nn, cov_type, var, a, gauss=True, grad=False):
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    m = nn.shape[1]
    A = np.empty((n, m)); D = np.empty(n)
    Ag = np.empty((2, n, m)) if grad else None
    Dg = np.empty((2, n)) if grad else None
    bad = lib().orc_vecchia_factor(_p(cm, C.c_double), C.c_int(n), C.c_int(d), _p(nn, C.c_int), C.c_int(m),
                                   C.c_int(cov_type), C.c_double(var), C.c_double(a), C.c_int(1 if gauss else 0),
                                   _p(A, C.c_double), _p(D, C.c_double), _p(Ag, C.c_double), _p(Dg, C.c_double))
    return (A, D, Ag, Dg, bad) if grad else (A, D, bad)


def vecchia_nll(coords, nn, cov_type, pars_trans, y):
    """Returns (yTPsiInvy, logdet, negll).  pars_trans = (sigma2, var, a)."""
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    out = np.empty(3)
    lib().orc_vecchia_nll(_p(cm, C.c_double), C.c_int(n), C.c_int(d), _p(nn, C.c_int), C.c_int(nn.shape[1]),
                          C.c_int(cov_type), _p(_f64(pars_trans), C.c_double), _p(_f64(y), C.c_double),
                          _p(out, C.c_double))
    return out


def vecchia_nll_grad(coords, nn, cov_type, pars_trans, y):
    """Returns (out3, grad3) -- grad wrt log(sigma2), log(var), log(a) as CalcGradPars does."""
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    out = np.empty(3); g = np.empty(3)
    lib().orc_vecchia_nll_grad(_p(cm, C.c_double), C.c_int(n), C.c_int(d), _p(nn, C.c_int), C.c_int(nn.shape[1]),
                               C.c_int(cov_type), _p(_f64(pars_trans), C.c_double), _p(_f64(y), C.c_double),
                               _p(out, C.c_double), _p(g, C.c_double))
    return out, g


def vecchia_yaux(A, D, nn, y):
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    out = np.empty(n)
    lib().orc_vecchia_yaux(_p(_f64(A), C.c_double), _p(_f64(D), C.c_double), _p(nn, C.c_int), C.c_int(n), C.c_int(m),
                           _p(_f64(y), C.c_double), _p(out, C.c_double))
    return out


def train_random_effects(A, D, nn, y, sigma2):
    """PredictTrainingDataRandomEffects, Gaussian Vecchia model (include/GPBoost/re_model_template.h:4496-4514): mean = y - y_aux,
    var_i = sigma2 (1 - sum_k (B o (D^-1 B))_ki) = sigma2 (1 - diag(B' D^-1 B)_i).  Vecchia order; A, D on the transformed scale."""
    n, m = nn.shape
    dg = 1.0 / D
    for i in range(n):
        for j in range(m):
            c = nn[i, j]
            if c >= 0:
                dg[c] += A[i, j] * A[i, j] / D[i]
    return np.asarray(y, dtype=np.float64) - vecchia_yaux(A, D, nn, y), sigma2 * (1.0 - dg)


def gls_coef(A, D, nn, X, y):
    """Generalised-least-squares coefficients beta = (X' Psi^-1 X)^-1 X' Psi^-1 y with Psi^-1 = B' D^-1 B of the Vecchia factor (Vecchia
    order): ProfileOutCoef / UpdateCoefGLS / CalcXTPsiInvX (re_model_template.h:2665-2683, :10012-10019, :6622-6628), solved by
    Cholesky as Eigen's llt() there.  Returns (beta, residual y - X beta)."""
    nn = np.asarray(nn)
    X = np.asarray(X, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    U = np.column_stack([X, y])
    valid = nn >= 0
    idx = np.where(valid, nn, 0)
    BU = U - np.einsum("ij,ijk->ik", np.where(valid, A, 0.0), U[idx])
    G = BU.T @ (BU / np.asarray(D)[:, None])
    p = X.shape[1]
    L = np.linalg.cholesky(G[:p, :p])
    beta = np.linalg.solve(L.T, np.linalg.solve(L, G[:p, p]))
    return beta, y - X @ beta


def exact_nll(coords, cov_type, pars_trans, y, want_yaux=False):
    cm = np.asfortranarray(coords, dtype=np.float64)
    n, d = cm.shape
    out = np.empty(3)
    ya = np.empty(n) if want_yaux else None
    rc = lib().orc_exact_nll(_p(cm, C.c_double), C.c_int(n), C.c_int(d), C.c_int(cov_type),
                             _p(_f64(pars_trans), C.c_double), _p(_f64(y), C.c_double), _p(out, C.c_double),
                             _p(ya, C.c_double))
    if rc != 0:
        raise FloatingPointError("matrix not SPD")
    return (out, ya) if want_yaux else out


def exact_predict(coords, y, coords_pred, cov_type, cov_pars, predict_response=True):
    """Prediction of the exact GP (dense Gaussian branch of REModelTemplate::Predict, include/GPBoost/re_model_template.h:4239-4330):
    mean = Sigma_po Psi^-1 y, covariance = Sigma_pp [+ sigma2 I] - Sigma_po Psi^-1 Sigma_op, Psi = Sigma + sigma2 I.  numpy, small n."""
    from scipy.spatial.distance import cdist
    s2, s12, rho = [float(v) for v in cov_pars]
    c = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[cov_type]

    def kern(A, B):
        x = c * cdist(A, B) / rho
        if cov_type == 0:
            return s12 * np.exp(-x)
        if cov_type == 1:
            return s12 * (1 + x) * np.exp(-x)
        return s12 * (1 + x + x * x / 3) * np.exp(-x)
    co = np.asarray(coords, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    Psi = kern(co, co) + s2 * np.eye(co.shape[0])
    Spo = kern(cp, co)
    mean = Spo @ np.linalg.solve(Psi, np.asarray(y, dtype=np.float64))
    cov = kern(cp, cp) - Spo @ np.linalg.solve(Psi, Spo.T)
    if predict_response:
        cov = cov + s2 * np.eye(cp.shape[0])
    return mean, cov


def exact_fisher_std_errors(coords, cov_type, cov_pars):
    """Standard errors of (sigma2, sigma1_2, rho) of the exact GP: CalcStdDevCovPar -> CalcFisherInformation, dense branch on the original
    scale with the error variance (include/GPBoost/re_model_template.h:10788-10815, 10066-10127): FI_ab = 1/2 tr(P dPsi_a P dPsi_b),
    P = (Sigma + sigma2 I)^-1, dPsi = {I, Sigma / sigma1_2, dSigma / d rho}; sqrt(diag(FI^-1)).  numpy, small n."""
    from scipy.spatial.distance import cdist
    s2, s12, rho = [float(v) for v in cov_pars]
    c = {0: 1.0, 1: np.sqrt(3.0), 2: np.sqrt(5.0)}[cov_type]
    dist = cdist(coords, coords)
    x = c * dist / rho
    if cov_type == 0:
        k = np.exp(-x); dk = np.exp(-x) * x / rho                      # d/d rho of exp(-d / rho)
    elif cov_type == 1:
        k = (1 + x) * np.exp(-x); dk = x * x * np.exp(-x) / rho
    else:
        k = (1 + x + x * x / 3) * np.exp(-x); dk = (x * x + x ** 3) / 3 * np.exp(-x) / rho
    n = dist.shape[0]
    P = np.linalg.inv(s12 * k + s2 * np.eye(n))
    G = [P, P @ k, P @ (s12 * dk)]
    FI = np.array([[0.5 * np.sum(G[a_].T * G[b_]) for b_ in range(3)] for a_ in range(3)])
    return np.sqrt(np.diag(np.linalg.inv(FI))), FI


def hist_build(bins, bin_offsets, data_indices, grad, hess=None, const_hess=1.0):
    """bins: (F, n) uint8.  Returns (hist_grad, hist_cnt(uint64), hist_hess)."""
    bins = np.ascontiguousarray(bins, dtype=np.uint8)
    F, n = bins.shape
    bo = np.ascontiguousarray(bin_offsets, dtype=np.int32)
    di = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
    nd = n if di is None else di.size
    tot = int(bo[-1])
    hg = np.empty(tot); hc = np.empty(tot, dtype=np.uint64); hh = np.empty(tot)
    h = None if hess is None else _f64(hess)
    lib().orc_hist_build(_p(bins, C.c_uint8), C.c_int(n), C.c_int(F), _p(bo, C.c_int), _p(di, C.c_int), C.c_int(nd),
                         _p(_f64(grad), C.c_double), _p(h, C.c_double), C.c_double(const_hess),
                         _p(hg, C.c_double), _p(hc, C.c_uint64), _p(hh, C.c_double))
    return hg, hc, hh


def hist_fix(hist, view_offset, num_bin, most_freq_bin, sum_gradient, sum_hessian):
    """Dataset::FixHistogram for every feature; hist (total_bins, 2), returns a fixed copy."""
    out = np.ascontiguousarray(hist, dtype=np.float64).copy()
    vo = np.ascontiguousarray(view_offset, dtype=np.int32); nb = np.ascontiguousarray(num_bin, dtype=np.int32)
    mf = np.ascontiguousarray(most_freq_bin, dtype=np.int32)
    lib().orc_hist_fix(_p(out, C.c_double), C.c_int(vo.size), _p(vo, C.c_int), _p(nb, C.c_int), _p(mf, C.c_int),
                       C.c_double(sum_gradient), C.c_double(sum_hessian))
    return out


def hist_subtract(parent, smaller):
    """FeatureHistogram::Subtract: larger = parent - smaller, entry by entry."""
    out = np.ascontiguousarray(parent, dtype=np.float64).copy()
    sm = np.ascontiguousarray(smaller, dtype=np.float64)
    lib().orc_hist_subtract(_p(out, C.c_double), _p(sm, C.c_double), C.c_int(out.shape[0]))
    return out


def root_parent_output(sum_gradient, sum_hessian, lambda_l1=0.0, lambda_l2=0.0, max_delta_step=0.0):
    """SerialTreeLearner::GetParentOutput for the root: its own output without smoothing (serial_tree_learner.cpp:758-770)."""
    lib().orc_root_parent_output.restype = C.c_double
    return float(lib().orc_root_parent_output(C.c_double(sum_gradient), C.c_double(sum_hessian), C.c_double(lambda_l1), C.c_double(lambda_l2),
                                              C.c_double(max_delta_step)))


def find_best_split(hist, view_offset, num_bin, offset, default_bin, missing, sum_gradient, sum_hessian, num_data, lambda_l2=0.0,
                    min_data_in_leaf=20, min_sum_hessian_in_leaf=1e-3, min_gain_to_split=0.0, lambda_l1=0.0, max_delta_step=0.0,
                    path_smooth=0.0, parent_output=0.0):
    """FeatureHistogram::FindBestThreshold for every (numerical) feature + the choice among features.
    -> (best_feature, out (F, 10), default_left (F,)); columns of out: gain, threshold, left_count, right_count, left_output,
    right_output, left_sum_gradient, left_sum_hessian, right_sum_gradient, right_sum_hessian."""
    h = np.ascontiguousarray(hist, dtype=np.float64)
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (view_offset, num_bin, offset, default_bin, missing)]
    F = arrs[0].size
    out = np.zeros((F, 10)); dl = np.zeros(F, dtype=np.int32)
    lib().orc_find_best_split_reg.restype = C.c_int
    best = lib().orc_find_best_split_reg(_p(h, C.c_double), C.c_int(F), *[_p(a, C.c_int) for a in arrs], C.c_double(sum_gradient),
                                         C.c_double(sum_hessian), C.c_int(int(num_data)), C.c_double(lambda_l2), C.c_int(int(min_data_in_leaf)),
                                         C.c_double(min_sum_hessian_in_leaf), C.c_double(min_gain_to_split), C.c_double(lambda_l1),
                                         C.c_double(max_delta_step), C.c_double(path_smooth), C.c_double(parent_output),
                                         _p(out, C.c_double), _p(dl, C.c_int))
    find_best_split.last_splittable = (dl >> 1) & 1
    return best, out, dl & 1


CAT_DEFAULTS = (4, 32, 10.0, 10.0, 100)     # max_cat_to_onehot, max_cat_threshold, cat_smooth, cat_l2, min_data_per_group (include/LightGBM/config.h)


def find_best_split_cat(hist, view_offset_f, num_bin_f, offset_f, sum_gradient, sum_hessian, num_data, lambda_l2=0.0, min_data_in_leaf=20,
                        min_sum_hessian_in_leaf=1e-3, min_gain_to_split=0.0, lambda_l1=0.0, max_delta_step=0.0, path_smooth=0.0, parent_output=0.0,
                        cat_cfg=CAT_DEFAULTS):
    """FeatureHistogram::FindBestThresholdCategoricalInner for ONE categorical feature on the fixed leaf histogram hist (total_bins, 2).
    -> (row (10,), flags (bit 0 default_left, bit 1 splittable), cat_bits (8,) uint32: bitset over the feature's bins of the categories going left);
    row[1] = the number of those categories."""
    h = np.ascontiguousarray(hist, dtype=np.float64)
    row = np.zeros(10); fl = C.c_int(0); bits = np.zeros(8, dtype=np.uint32)
    data = h[int(view_offset_f):].reshape(-1)
    lib().orc_find_best_split_cat(_p(np.ascontiguousarray(data), C.c_double), C.c_int(int(num_bin_f)), C.c_int(int(offset_f)), C.c_double(sum_gradient),
                                  C.c_double(sum_hessian), C.c_int(int(num_data)), C.c_double(lambda_l2), C.c_int(int(min_data_in_leaf)),
                                  C.c_double(min_sum_hessian_in_leaf), C.c_double(min_gain_to_split), C.c_double(lambda_l1), C.c_double(max_delta_step),
                                  C.c_double(path_smooth), C.c_double(parent_output), C.c_int(int(cat_cfg[0])), C.c_int(int(cat_cfg[1])),
                                  C.c_double(cat_cfg[2]), C.c_double(cat_cfg[3]), C.c_int(int(cat_cfg[4])), _p(row, C.c_double), C.byref(fl),
                                  _p(bits, C.c_uint32))
    return row, fl.value, bits


def split_leaf_layout(bins_col, min_bin, max_bin, use_min_bin, default_bin, most_freq_bin, missing_type, default_left, threshold, is_categorical,
                      cat_bits, data_indices):
    """FeatureGroup::Split for a feature inside a column that may hold several features (numerical: DenseBin::SplitInner<.., USE_MIN_BIN>,
    categorical: SplitCategoricalInner): (lte_indices, gt_indices) in the order of data_indices."""
    b = np.ascontiguousarray(bins_col, dtype=np.uint8)
    di = np.ascontiguousarray(data_indices, dtype=np.int32)
    bits = np.zeros(8, dtype=np.uint32) if cat_bits is None else np.ascontiguousarray(cat_bits, dtype=np.uint32)
    lte = np.empty(di.size, dtype=np.int32); gt = np.empty(di.size, dtype=np.int32)
    lib().orc_split_leaf_layout.restype = C.c_int
    nl = lib().orc_split_leaf_layout(_p(b, C.c_ubyte), C.c_int(int(min_bin)), C.c_int(int(max_bin)), C.c_int(int(bool(use_min_bin))), C.c_int(int(default_bin)),
                                     C.c_int(int(most_freq_bin)), C.c_int(int(missing_type)), C.c_int(int(bool(default_left))), C.c_uint(int(threshold)),
                                     C.c_int(int(bool(is_categorical))), _p(bits, C.c_uint32), _p(di, C.c_int), C.c_int(di.size), _p(lte, C.c_int),
                                     _p(gt, C.c_int))
    return lte[:nl].copy(), gt[:di.size - nl].copy()


def split_leaf(bins_f, max_bin, default_bin, most_freq_bin, missing_type, default_left, threshold, data_indices):
    """DenseBin::Split for a single-feature group: (lte_indices, gt_indices), both in the order of data_indices."""
    b = np.ascontiguousarray(bins_f, dtype=np.uint8)
    di = np.ascontiguousarray(data_indices, dtype=np.int32)
    lte = np.empty(di.size, dtype=np.int32); gt = np.empty(di.size, dtype=np.int32)
    lib().orc_split_leaf.restype = C.c_int
    nl = lib().orc_split_leaf(_p(b, C.c_ubyte), C.c_int(int(max_bin)), C.c_int(int(default_bin)), C.c_int(int(most_freq_bin)),
                              C.c_int(int(missing_type)), C.c_int(int(bool(default_left))), C.c_uint(int(threshold)), _p(di, C.c_int),
                              C.c_int(di.size), _p(lte, C.c_int), _p(gt, C.c_int))
    return lte[:nl].copy(), gt[:di.size - nl].copy()


def newton_leaf_values(A, D, nn, yaux, leaf, num_leaves):
    """REModelTemplate::NewtonUpdateLeafValues, Vecchia branch: leaf values of the Newton step.  A, D from
    vecchia_factor(gauss=True); yaux = B^T D^-1 B (F - y); leaf = leaf index per point; all in Vecchia order."""
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    A = np.ascontiguousarray(A, dtype=np.float64); D = np.ascontiguousarray(D, dtype=np.float64)
    ya = np.ascontiguousarray(yaux, dtype=np.float64); lf = np.ascontiguousarray(leaf, dtype=np.int32)
    out = np.zeros(num_leaves)
    rc = lib().orc_newton_leaf_values(_p(A, C.c_double), _p(D, C.c_double), _p(nn, C.c_int), C.c_int(n), C.c_int(m),
                                      _p(ya, C.c_double), _p(lf, C.c_int), C.c_int(num_leaves), _p(out, C.c_double))
    if rc != 0:
        raise RuntimeError("H^T Psi^-1 H is not positive definite")
    return out


def gen_rand_normal(n, t, seed=1, run_id=0):
    """GenRandVecNormalParallel (CG_utils.cpp:978-994): (n, t) Fortran-ordered N(0,1) probes."""
    out = np.empty((n, t), order="F")
    lib().orc_gen_rand_normal(C.c_int(seed), C.c_ulonglong(run_id), C.c_int(n), C.c_int(t), _p(out, C.c_double))
    return out


LINK_ID = {"bernoulli_logit": 0, "bernoulli_probit": 1, "poisson": 2, "gamma": 3, "negative_binomial": 4, "beta": 5, "t": 6, "lognormal": 7, "gaussian_latent": 8,
           # round 5: proportions under the logit / probit links (real-valued response in [0, 1]; binomial_*: trials = orc.sample_weights, binomial constant)
           "binomial_logit": 0, "binomial_probit": 1, "quasi_bernoulli_logit": 0, "quasi_bernoulli_probit": 1}
PROPORTION_LIKELIHOODS = ("binomial_logit", "binomial_probit", "quasi_bernoulli_logit", "quasi_bernoulli_probit")


def _responses(likelihood, y):
    """-> (int32 responses, float64 responses | None): gamma's response is real-valued (handed to the C side through orc_set_aux)."""
    if likelihood in ("gamma", "beta", "t", "lognormal", "gaussian_latent") or likelihood in PROPORTION_LIKELIHOODS:
        yd = np.ascontiguousarray(y, dtype=np.float64)
        lib().orc_set_binomial(C.c_int(1 if likelihood.startswith("binomial") else 0))
        return np.zeros(yd.shape[0], dtype=np.int32), yd
    lib().orc_set_binomial(C.c_int(0))
    return np.ascontiguousarray(y, dtype=np.int32), None


class sample_weights(object):
    """`with orc.sample_weights(w): ...` -- sample weights of the non-Gaussian likelihood (Likelihood::weights_, likelihoods.h:666-668) for every oracle call
    inside the block: data order as the response handed to those calls (Vecchia order; grouped by random effect with repeated locations).  None: no weights."""

    def __init__(self, w):
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)

    def __enter__(self):
        lib().orc_set_weights(_p(self.w, C.c_double))
        return self

    def __exit__(self, *exc):
        lib().orc_set_weights(None)
        return False


_PROBE_RUN_ID = 0      # cg_generator_counter_ at the time rand_vec_trace_I_ is drawn: 0, or 1 when rand_vec_trace_I2_ was drawn first (pivoted_cholesky)


def pivoted_cholesky_factor(coords, cov_type, var, a, rank=50, err_tol=1e-6):
    """PivotedCholsekyFactorizationSigma (CG_utils.h:438-486; PIV_CHOL_STOP_TOL = 1e-6, utils.h:44) of the non-approximated covariance matrix
    var * k(a * dist) of the points `coords` (Vecchia order): -> L_k (n, k) Fortran-ordered, k <= rank the number of columns computed."""
    co = np.asfortranarray(coords, dtype=np.float64)
    n, d = co.shape
    k = min(int(rank), n)
    L = np.zeros((n, k), order="F")
    fn = lib().orc_pivoted_cholesky
    fn.restype = C.c_int
    kk = fn(_p(co, C.c_double), C.c_int(n), C.c_int(d), C.c_int(cov_type), C.c_double(var), C.c_double(a), C.c_int(k), C.c_double(err_tol), _p(L, C.c_double))
    return L, int(kk)


class pivoted_cholesky_preconditioner(object):
    """`with orc.pivoted_cholesky_preconditioner(coords, cov_type, var, a, rank, num_rand_vec, seed_rand): ...` -- the Vecchia-Laplace oracle calls inside
    the block use cg_preconditioner_type = "pivoted_cholesky" (the (W^-1 + Sigma) form of the solves, P = W^-1 + L_k L_k^T; likelihoods.h:16277-16296,
    :16389-16465, :16554-16611, :16716-16736) at the covariance parameters (var, a): the factor is the one of THESE parameters, as the reference recomputes it
    before every mode finding (re_model_template.h:9317-9322).  rand_vec_trace_I2_ (k x t) is drawn first (generator counter 0), rand_vec_trace_I_ second (1)."""

    def __init__(self, coords, cov_type, var, a, rank=50, num_rand_vec=50, seed_rand=1):
        self.L, self.k = pivoted_cholesky_factor(coords, cov_type, var, a, rank)
        # the reference sizes rand_vec_trace_I2_ by fitc_piv_chol_preconditioner_rank_ (likelihoods.h:3996) and multiplies with all `rank` columns (zeros beyond k)
        self.rv2 = gen_rand_normal(self.L.shape[1], num_rand_vec, seed_rand, 0)

    def __enter__(self):
        global _PROBE_RUN_ID
        fn = lib().orc_set_pivchol
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        fn(self.L.ctypes.data, self.L.shape[1], self.rv2.ctypes.data)
        _PROBE_RUN_ID = 1
        return self

    def __exit__(self, *exc):
        global _PROBE_RUN_ID
        lib().orc_clear_pivchol()
        _PROBE_RUN_ID = 0
        return False


class fitc_preconditioner(object):
    """`with orc.fitc_preconditioner(coords, ip, cov_type, var, a, num_rand_vec, seed_rand): ...` -- cg_preconditioner_type = "fitc" for the Vecchia-Laplace oracle calls
    inside the block: P = diag(W^-1 + Sigma_m[0][0] - ||V_i||^2) + C Sigma_m^-1 C' with the k inducing points `ip` (k x d; the reference picks them by kmeans++ from the
    model's generator at the first covariance factor, Calc_FITC_Preconditioner_Vecchia, re_model_template.h:9502-9593 -- orc.vif_setup restates that draw), C the
    cross-covariance, Sigma_m the inducing points' covariance with the diagonal x (1 + 1e-6), V = (L_m^-1 C')'.  rand_vec_trace_I2_ (k x t) first, rand_vec_trace_I_ second."""

    def __init__(self, coords, ip, cov_type, var, a, num_rand_vec=50, seed_rand=1):
        from scipy.spatial.distance import cdist
        from scipy.linalg import cholesky, solve_triangular
        co = np.asarray(coords, dtype=np.float64); ip = np.asarray(ip, dtype=np.float64)
        self.k = ip.shape[0]
        Sm = _matern(cov_type, cdist(ip, ip), var, a)
        Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6                                   # JITTER_MULT_IP_FITC_FSA
        Lm = cholesky(Sm, lower=True)
        self.C = np.asfortranarray(_matern(cov_type, cdist(co, ip), var, a))          # n x k
        self.V = np.asfortranarray(solve_triangular(Lm, self.C.T, lower=True).T)      # n x k = chol_ip_cross_cov^T
        self.Sm = np.ascontiguousarray(Sm)
        self.logdet_Sm = 2.0 * np.log(np.diag(Lm)).sum()
        self.rv2 = gen_rand_normal(self.k, num_rand_vec, seed_rand, 0)

    def __enter__(self):
        global _PROBE_RUN_ID
        fn = lib().orc_set_fitc
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        fn(self.C.ctypes.data, self.V.ctypes.data, self.Sm.ctypes.data, float(self.logdet_Sm), self.k, self.rv2.ctypes.data)
        _PROBE_RUN_ID = 1
        return self

    def __exit__(self, *exc):
        global _PROBE_RUN_ID
        lib().orc_clear_pivchol()
        _PROBE_RUN_ID = 0
        return False


class vecchia_response_preconditioner(object):
    """`with orc.vecchia_response_preconditioner(coords, cov_type, var, a): ...` -- cg_preconditioner_type = "vecchia_response" for the Vecchia-Laplace oracle calls inside
    the block: the (W^-1 + Sigma) form of the solves with P = the Vecchia approximation of W^-1 + Sigma on the same neighbour sets, renewed for every W
    (CalcVecchiaApproxLatentAddDiagonal, re_model_template.h:5473-5492; likelihoods.h:16315-16323, :16439-16450, :16471-16473).  coords: Vecchia order.  Only
    rand_vec_trace_I_ is drawn (generator counter 0, likelihoods.h:3993-4009).  Evaluation only: the reference refuses the gradient (:6570-6572), orc_vecchia_laplace_grad too."""

    def __init__(self, coords, cov_type, var, a):
        self.co = np.asfortranarray(coords, dtype=np.float64)
        self.args = (int(cov_type), float(var), float(a))
        self.k = 0

    def __enter__(self):
        fn = lib().orc_set_vecchia_response
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        fn(self.co.ctypes.data, self.co.shape[1], *self.args)
        return self

    def __exit__(self, *exc):
        lib().orc_clear_pivchol()
        return False


class _aux_context(object):
    """orc_set_aux / orc_clear_aux around a call for the likelihoods with an auxiliary parameter (link >= 3); a no-op otherwise."""

    def __init__(self, link, aux, yd, aux_grad4):
        self.on = link >= 3 or yd is not None         # (a real-valued response travels the same way for the proportion likelihoods)
        av = np.atleast_1d(np.asarray(1.0 if aux is None else aux, dtype=np.float64))     # t: (scale, df)
        self.aux2 = float(av[1]) if av.size > 1 else None
        self.args = (float(av[0]), yd, aux_grad4)

    def __enter__(self):
        if self.on:
            aux, yd, g4 = self.args
            fn = lib().orc_set_aux
            fn.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
            fn(aux, None if yd is None else yd.ctypes.data, None if g4 is None else g4.ctypes.data)
            if self.aux2 is not None:
                lib().orc_set_aux2(C.c_double(self.aux2))
        return self

    def __exit__(self, *exc):
        if self.on:
            lib().orc_clear_aux()
        return False


def vecchia_laplace_logit(coords, nn, cov_type, var, a, y01, num_rand_vec=50, seed_rand=1, cg_max_num_it=1000,
                          cg_max_num_it_tridiag=1000, cg_delta_conv=1e-2, delta_conv_mode=1e-8, rand_vec=None, likelihood="bernoulli_logit", fixed_effects=None,
                          aux=None, factor=None):
    """Approximate negative log marginal likelihood of a Bernoulli-logit (or, likelihood="bernoulli_probit", -probit) Vecchia GP
    (Laplace, iterative, 'vadu').  Returns (negll, info dict).  coords / y01 in Vecchia order; var = sigma1^2, a = transformed range."""
    link = LINK_ID[likelihood]
    if factor is None:
        A, D, bad = vecchia_factor(coords, nn, cov_type, var, a, gauss=False)
    else:                                                   # (orc.vif_laplace: the residual process's factor)
        A, D = np.ascontiguousarray(factor[0]), np.ascontiguousarray(factor[1])
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    yi, yd = _responses(likelihood, y01)
    rv = gen_rand_normal(n, num_rand_vec, seed_rand, _PROBE_RUN_ID) if rand_vec is None else np.asfortranarray(rand_vec)
    out = np.empty(6); mode = np.empty(n)
    fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
    with _aux_context(link, aux, yd, None):
      rc = lib().orc_vecchia_laplace_binary_fe(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(nn, C.c_int), C.c_int(n), C.c_int(m),
                                          _p(yi, C.c_int), None if fe is None else _p(fe, C.c_double), _p(rv, C.c_double), C.c_int(rv.shape[1]), C.c_int(cg_max_num_it),
                                          C.c_int(cg_max_num_it_tridiag), C.c_double(cg_delta_conv), C.c_double(delta_conv_mode),
                                          _p(out, C.c_double), _p(mode, C.c_double))
    return -out[0], dict(rc=rc, newton_it=int(out[1]), cg_it=int(out[2]), log_det=out[3], lanczos_it=int(out[4]),
                         mll_no_det=out[5], mode=mode, A=A, D=D)


def unique_locations(coords):
    """DetermineUniqueDuplicateCoordsFast (src/GPBoost/GP_utils.cpp:472-548) as RECompGP uses it for one non-Gaussian GP
    (include/GPBoost/re_comp.h:863-885): -> (uniques, unique_idx); uniques = positions of the FIRST appearance of every distinct location
    (two locations are the same if their squared distance is < 1e-20), ascending; unique_idx[i] = index into uniques of point i.
    Candidates are found among points with the same coordinate sum, as the reference does."""
    co = np.asarray(coords, dtype=np.float64)
    n = co.shape[0]
    csum = co.sum(axis=1)
    order = sort_indices(csum)
    rep = np.arange(n)                        # representative (smallest index) of every point's location
    i = 0
    while i < n:
        j = i + 1
        while j < n and not (csum[order[i]] < csum[order[j]] - 1e-10 * max(1.0, abs(csum[order[i]]))) and csum[order[j]] - csum[order[i]] <= 1e-10 * max(1.0, abs(csum[order[i]])):
            j += 1
        grp = [int(order[k]) for k in range(i, j)]
        reps = []
        for p_ in sorted(grp):
            for r in reps:
                if ((co[p_] - co[r]) ** 2).sum() < 1e-20:
                    rep[p_] = r
                    break
            else:
                reps.append(p_)
        i = j
    uniques = np.array(sorted(set(rep.tolist())), dtype=np.int64)
    pos = {int(u): k for k, u in enumerate(uniques)}
    unique_idx = np.array([pos[int(r)] for r in rep], dtype=np.int32)
    return uniques, unique_idx


def _data_map(unique_idx):
    """CSR of the random effects' data and the order that groups the data by random effect (stable: ascending data position)."""
    unique_idx = np.asarray(unique_idx)
    order = np.argsort(unique_idx, kind="stable")
    counts = np.bincount(unique_idx)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), order


def vecchia_laplace_dup(coords_u, nn, cov_type, var, a, unique_idx, y, num_rand_vec=50, seed_rand=1, cg_max_num_it=1000, cg_max_num_it_tridiag=1000,
                        cg_delta_conv=1e-2, delta_conv_mode=1e-8, likelihood="bernoulli_logit", fixed_effects=None, grad=False):
    """Vecchia-Laplace approximation with REPEATED locations: the GP lives on the unique locations coords_u (Vecchia order), datum d belongs to
    random effect unique_idx[d] (Vecchia_utils.cpp:1156-1168; the likelihood terms of a random effect are sums over its data).
    y / fixed_effects per datum in the order unique_idx refers to.  -> (negll, info) or, grad = True, (negll, gradient wrt (log sigma1^2, log a), mode)."""
    link = {"bernoulli_logit": 0, "bernoulli_probit": 1, "poisson": 2}[likelihood]
    dptr, order = _data_map(unique_idx)
    yi = np.ascontiguousarray(np.asarray(y)[order], dtype=np.int32)
    fe = None if fixed_effects is None else np.ascontiguousarray(np.asarray(fixed_effects, dtype=np.float64)[order])
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    rv = gen_rand_normal(n, num_rand_vec, seed_rand, _PROBE_RUN_ID)
    out = np.empty(6); mode = np.zeros(n)
    if grad:
        A, D, Ag, Dg, bad = vecchia_factor(coords_u, nn, cov_type, var, a, gauss=False, grad=True)
        g = np.empty(2)
        rc = lib().orc_vecchia_laplace_grad_map(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(Ag, C.c_double), _p(Dg, C.c_double), _p(nn, C.c_int),
                                                C.c_int(n), C.c_int(m), _p(dptr, C.c_int), _p(yi, C.c_int), None if fe is None else _p(fe, C.c_double), _p(rv, C.c_double),
                                                C.c_int(rv.shape[1]), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag), C.c_double(cg_delta_conv),
                                                C.c_double(delta_conv_mode), _p(out, C.c_double), _p(g, C.c_double), _p(mode, C.c_double), C.c_int(0))
        if rc != 0:
            raise RuntimeError("orc_vecchia_laplace_grad_map failed")
        return -out[0], g, mode
    A, D, bad = vecchia_factor(coords_u, nn, cov_type, var, a, gauss=False)
    rc = lib().orc_vecchia_laplace_binary_fe_map(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(nn, C.c_int), C.c_int(n), C.c_int(m), _p(dptr, C.c_int),
                                                 _p(yi, C.c_int), None if fe is None else _p(fe, C.c_double), _p(rv, C.c_double), C.c_int(rv.shape[1]),
                                                 C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag), C.c_double(cg_delta_conv), C.c_double(delta_conv_mode),
                                                 _p(out, C.c_double), _p(mode, C.c_double))
    return -out[0], dict(rc=rc, newton_it=int(out[1]), cg_it=int(out[2]), log_det=out[3], lanczos_it=int(out[4]), mll_no_det=out[5], mode=mode)


def vecchia_laplace_grad(coords, nn, cov_type, var, a, y01, num_rand_vec=50, seed_rand=1, cg_max_num_it=1000, cg_max_num_it_tridiag=1000,
                         cg_delta_conv=1e-2, delta_conv_mode=1e-8, likelihood="bernoulli_logit", fixed_effects=None, mode_init=None,
                         want_mode=False, want_parts=False, aux=None):
    """(negll, gradient of negll wrt (log sigma1^2, log a)) of the Vecchia-Laplace approximation, iterative methods, 'vadu'
    (orc_vecchia_laplace_grad: the checker of the device gradient, tests/test_z_laplace_grad_gpu.py).  mode_init: start Newton's method there (the
    warm start of the reference's optimiser); want_mode: also return the mode found."""
    link = LINK_ID[likelihood]
    A, D, Ag, Dg, bad = vecchia_factor(coords, nn, cov_type, var, a, gauss=False, grad=True)
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    yi, yd = _responses(likelihood, y01)
    aux_g4 = np.zeros(8) if link >= 3 else None         # 4 per auxiliary parameter (t has two: scale, df)
    rv = gen_rand_normal(n, num_rand_vec, seed_rand, _PROBE_RUN_ID)
    fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
    out = np.empty(6); g = np.empty(2)
    mode = np.zeros(n) if mode_init is None else np.ascontiguousarray(mode_init, dtype=np.float64).copy()
    dbg = np.zeros(2 * n + 8) if want_parts else None
    with _aux_context(link, aux, yd, aux_g4):
        rc = lib().orc_vecchia_laplace_grad(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(Ag, C.c_double), _p(Dg, C.c_double),
                                            _p(nn, C.c_int), C.c_int(n), C.c_int(m), _p(yi, C.c_int), None if fe is None else _p(fe, C.c_double),
                                            _p(rv, C.c_double), C.c_int(rv.shape[1]), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag),
                                            C.c_double(cg_delta_conv), C.c_double(delta_conv_mode), _p(out, C.c_double), _p(g, C.c_double),
                                            _p(mode, C.c_double), C.c_int(0 if mode_init is None else 1), None if dbg is None else _p(dbg, C.c_double))
    if rc != 0:
        raise RuntimeError("orc_vecchia_laplace_grad failed")
    if link == 6:      # t: d(-mll) / d (log scale, log df)
        g = np.array([g[0], g[1], aux_g4[0], aux_g4[4]])
    elif link >= 3:      # likelihoods with an auxiliary parameter: the gradient's third entry is d(-mll) / d log(aux)
        g = np.array([g[0], g[1], aux_g4[0]])
    if want_parts:      # intermediate values for device parity tests
        parts = dict(dlogdet_dmode=dbg[:n].copy(), implicit_solve=dbg[n:2 * n].copy(), per_par=dbg[2 * n:].reshape(2, 4).copy(), mode=mode)
        return -out[0], g, parts
    return (-out[0], g, mode) if want_mode else (-out[0], g)


def gauss_hermite_adaptive(order=30):
    """Nodes x_j of the Gauss-Hermite rule (weight exp(-x^2)) and the ADAPTIVE weights w_j exp(x_j^2) the reference tabulates
    (GH_nodes_ / adaptive_GH_weights_, include/GPBoost/likelihoods.h:17472-17576; order_GH_ = 30), computed instead of copied."""
    x, w = np.polynomial.hermite.hermgauss(order)
    return x, w * np.exp(x * x)


def predict_response(likelihood, latent_mean, latent_var, predict_var=False, delta_conv_mode_finding=1e-8):
    """Likelihood::PredictResponse (include/GPBoost/likelihoods.h:9626-9672): response mean (and variance) from the latent predictive mean and
    variance.  probit: Phi(m / sqrt(1 + v)); logit: adaptive Gauss-Hermite quadrature of sigmoid(x) N(x; m, v) around the integrand's mode
    (RespMeanAdaptiveGHQuadrature, :10128-10160; Newton from 0, at most 100 steps, stop at |update| / |previous value| < delta_conv_mode_finding_);
    Poisson: exp(m + v / 2), variance pm ((exp(v) - 1) pm + 1).  Bernoulli variances: p (1 - p)."""
    from scipy.stats import norm
    m = np.asarray(latent_mean, dtype=np.float64); v = np.asarray(latent_var, dtype=np.float64)
    if likelihood == "bernoulli_probit":
        pm = norm.cdf(m / np.sqrt(1.0 + v))
        return pm, (pm * (1.0 - pm) if predict_var else None)
    if likelihood == "poisson":
        pm = np.exp(m + 0.5 * v)
        return pm, (pm * ((np.exp(v) - 1.0) * pm + 1.0) if predict_var else None)
    if likelihood != "bernoulli_logit":
        raise ValueError(likelihood)
    xs, aw = gauss_hermite_adaptive(30)
    out = np.empty_like(m)
    for i in range(m.size):
        s2i = 1.0 / v[i]
        mode = 0.0
        for _ in range(100):
            last = mode
            p = 1.0 / (1.0 + np.exp(-mode))
            # log CondMeanLikelihood = log sigmoid: first derivative 1 - p = 1 / (1 + e^x), second -p (1 - p)     (:10580-10640)
            upd = ((1.0 - p) - s2i * (mode - m[i])) / (-p * (1.0 - p) - s2i)
            mode -= upd
            with np.errstate(divide="ignore", invalid="ignore"):
                if abs(upd) / abs(last) < delta_conv_mode_finding:
                    break
        p = 1.0 / (1.0 + np.exp(-mode))
        sh = np.sqrt(2.0) / np.sqrt(p * (1.0 - p) + s2i)
        x = sh * xs + mode
        out[i] = np.sum(aw * (1.0 / (1.0 + np.exp(-x))) * norm.pdf(np.sqrt(s2i) * (x - m[i]))) * sh * np.sqrt(s2i)
    return out, (out * (1.0 - out) if predict_var else None)


def _laplace_mode_and_information(coords, nn, cov_type, var, a, y, likelihood, fixed_effects, unique_idx, **kw):
    """Mode of the latent process (iterative mode finder) and the information of the likelihood at the mode summed over every random effect's data
    -> (mode, W, A, D)."""
    co = np.asarray(coords, dtype=np.float64)
    n_obs = co.shape[0]
    if unique_idx is None:
        _, info = vecchia_laplace_logit(co, nn, cov_type, var, a, y, likelihood=likelihood, fixed_effects=fixed_effects, **kw)
        dptr, order = np.arange(n_obs + 1, dtype=np.int32), np.arange(n_obs)
    else:
        _, info = vecchia_laplace_dup(co, nn, cov_type, var, a, unique_idx, y, likelihood=likelihood, fixed_effects=fixed_effects, **kw)
        dptr, order = _data_map(unique_idx)
    mode = info["mode"]
    A, D, bad = vecchia_factor(co, nn, cov_type, var, a, gauss=False)
    yd = np.asarray(y, dtype=np.float64)[order]
    fe = np.zeros(yd.size) if fixed_effects is None else np.asarray(fixed_effects, dtype=np.float64)[order]
    re_of = np.repeat(np.arange(n_obs), np.diff(dptr))
    _, Wd, _ = _lik_terms(likelihood, yd, mode[re_of] + fe)
    return mode, np.bincount(re_of, weights=Wd, minlength=n_obs), A, D


def vecchia_laplace_train_re(coords, nn, cov_type, var, a, y, likelihood="bernoulli_logit", fixed_effects=None, unique_idx=None, **kw):
    """GPB_PredictREModelTrainingDataRandomEffects for a non-Gaussian Vecchia model (include/GPBoost/re_model_template.h:4683-4725): the mode and
    diag((Sigma^-1 + W)^-1) (CalcVarLaplaceApproxVecchia; the exact diagonal of its "cholesky" branch), per RANDOM EFFECT in Vecchia order.  Dense."""
    mode, W, A, D = _laplace_mode_and_information(coords, nn, cov_type, var, a, y, likelihood, fixed_effects, unique_idx, **kw)
    B = _dense_B(nn, A)
    M = B.T @ (B / D[:, None]) + np.diag(W)
    return mode, np.diag(np.linalg.inv(M)).copy()


def vecchia_laplace_predict(coords, nn, cov_type, var, a, y, coords_pred, m_pred, likelihood="bernoulli_logit", fixed_effects=None,
                            unique_idx=None, want_cov=False, cond_obs_only=True, **kw):
    """Latent prediction of a non-Gaussian Vecchia model, 'latent_order_obs_first_cond_obs_only' (the reference's default for these models):
    every prediction point conditions on its m_pred nearest OBSERVED points, factor rows without a nugget (CalcPredVecchiaObservedFirstOrder with
    CondObsOnly = true and gauss_likelihood = false, src/GPBoost/Vecchia_utils.cpp:1701-2060), then PredictLaplaceApproxVecchia
    (include/GPBoost/likelihoods.h:8563-8824):  mean = -Bpo mode (:8600-8602),  var = Dp + diag(Bpo (Sigma^-1 + W)^-1 Bpo') with W the information of
    the likelihood at the mode -- the value its "cholesky" branch computes (:8783-8821) and its "iterative" branch estimates with random vectors
    (:8637-8745).  Dense solve: small cases.  coords / y in Vecchia order (y per datum with unique_idx: repeated locations, as vecchia_laplace_dup).
    The mode comes from the iterative mode finder (kw: cg_delta_conv, delta_conv_mode).  -> (mean, var[, cov])."""
    co = np.asarray(coords, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n_obs, n_pred = co.shape[0], cp.shape[0]
    mode, W, A, D = _laplace_mode_and_information(co, nn, cov_type, var, a, y, likelihood, fixed_effects, unique_idx, **kw)
    B = _dense_B(nn, A)
    M = B.T @ (B / D[:, None]) + np.diag(W)
    call = np.vstack([co, cp])
    nnp = neighbors_range(call, m_pred, n_obs, n_obs - 1 if cond_obs_only else -1)
    Ap, Dp, bad = vecchia_factor(call, nnp, cov_type, var, a, gauss=False)
    rows = slice(n_obs, n_obs + n_pred)
    Bpo = np.zeros((n_pred, n_obs)); Bp = np.eye(n_pred)
    for k in range(n_pred):
        for j in range(nnp.shape[1]):
            c = nnp[n_obs + k, j]
            if c >= n_obs:
                Bp[k, c - n_obs] = -Ap[n_obs + k, j]           # 'latent_order_obs_first_cond_all': prediction points condition on each other
            elif c >= 0:
                Bpo[k, c] = -Ap[n_obs + k, j]
    # CondObsOnly = false (likelihoods.h:8603-8606, 8790-8821): mean = -Bp^-1 Bpo mode, cov = Bp^-1 Dp Bp^-T + (Bp^-1 Bpo) (Sigma^-1 + W)^-1 (Bp^-1 Bpo)'
    Bpi = np.linalg.inv(Bp)
    Cm = Bpi @ Bpo
    mean = -Cm @ mode
    prior = Bpi @ np.diag(Dp[rows]) @ Bpi.T
    cov = Cm @ np.linalg.solve(M, Cm.T)
    pvar = np.diag(prior) + np.diag(cov)
    if want_cov:
        return mean, pvar, cov + prior
    return mean, pvar


# ---------------------------------------------------------------------------
# High-level mirror of GPModel(gp_approx="vecchia").neg_log_likelihood for tests
# ---------------------------------------------------------------------------
def vecchia_laplace_grad_F(coords, nn, cov_type, var, a, y01, likelihood="bernoulli_logit", fixed_effects=None, **kw):
    """Boosting gradient for non-Gaussian data, d(-approximate marginal log-likelihood) / dF, Vecchia order
    (Likelihood::CalcGradNegMargLikelihoodLaplaceApproxVecchia with calc_F_grad, include/GPBoost/likelihoods.h:6996-7001):
        -d log p / d loc  +  d_mll_d_mode  -  W .* (Sigma^-1 + W)^-1 d_mll_d_mode,     d_mll_d_mode = 0.5 d logdet / d mode,
    from the by-products of vecchia_laplace_grad.  Checker only (no device path yet)."""
    from scipy.stats import norm
    kw = dict(kw); wts_kw = kw.pop("weights", None)
    negll, g, parts = vecchia_laplace_grad(coords, nn, cov_type, var, a, y01, likelihood=likelihood, fixed_effects=fixed_effects,
                                           want_parts=True, **kw)
    kw["weights"] = wts_kw
    y = np.asarray(y01, dtype=np.float64)
    loc = parts["mode"] + (0.0 if fixed_effects is None else np.asarray(fixed_effects, dtype=np.float64))
    if LINK_ID[likelihood] == 0:                    # (y may be a proportion: linear in y)
        p = 1.0 / (1.0 + np.exp(-loc)); first = y - p; W = p * (1.0 - p)
    elif likelihood == "poisson":
        e = np.exp(loc); first = y - e; W = e
    elif LINK_ID[likelihood] == 1:                  # y f(1) + (1 - y) f(0) of the two Bernoulli branches (exact at y = 0, 1)
        r1 = np.exp(norm.logpdf(loc) - norm.logcdf(loc)); r0 = np.exp(norm.logpdf(-loc) - norm.logcdf(-loc))
        first = y * r1 - (1.0 - y) * r0; W = y * r1 * (loc + r1) - (1.0 - y) * r0 * (loc - r0)
    else:
        raise ValueError(likelihood)
    d_mll_d_mode = 0.5 * parts["dlogdet_dmode"]
    if kw.get("weights") is not None:       # (the C side has them through orc.sample_weights; the per-datum terms here are weighted alike)
        wts = np.asarray(kw["weights"], dtype=np.float64); first = wts * first; W = wts * W
    return -first + d_mll_d_mode - W * parts["implicit_solve"]


def _lik_terms(likelihood, y, loc):
    """-> (d log p / d loc, information, d information / d loc) of the likelihoods on the path (likelihoods.h: CalcFirstDerivLogLik,
    CalcInformationLogLik, CalcFirstDerivInformationLocPar), elementwise."""
    from scipy.stats import norm
    y = np.asarray(y, dtype=np.float64); loc = np.asarray(loc, dtype=np.float64)
    if likelihood == "bernoulli_logit":
        p = 1.0 / (1.0 + np.exp(-loc))
        return y - p, p * (1.0 - p), p * (1.0 - p) * (1.0 - 2.0 * p)
    if likelihood == "poisson":
        e = np.exp(loc)
        return y - e, e, e
    if likelihood == "bernoulli_probit":
        z = np.where(y > 0, loc, -loc)
        r = np.exp(norm.logpdf(z) - norm.logcdf(z))
        first = np.where(y > 0, r, -r)
        info = r * (z + r)
        # d/dz [r (z + r)] with r' = -r (z + r); the chain rule through z = -loc for y = 0
        dz = -r * (z + r) * (z + r) + r * (1.0 - r * (z + r))
        return first, info, np.where(y > 0, dz, -dz)
    raise ValueError(likelihood)


def vecchia_laplace_dup_grad_F(coords_u, nn, cov_type, var, a, unique_idx, y, likelihood="bernoulli_logit", fixed_effects=None, num_rand_vec=50,
                               seed_rand=1, cg_max_num_it=1000, cg_max_num_it_tridiag=1000, cg_delta_conv=1e-2, delta_conv_mode=1e-8):
    """Boosting gradient d(-approximate marginal log-likelihood) / dF per DATUM for a non-Gaussian Vecchia model with REPEATED locations
    (use_random_effects_indices_of_data_; CalcGradNegMargLikelihoodLaplaceApproxVecchia, include/GPBoost/likelihoods.h:6944-6966 with the iterative
    method's estimate of diag((Sigma^-1 + W)^-1), :6700-6703):
        diag_r = (d logdet / d mode)_r / (sum of d information / d loc over the data of r)
        grad_d = -d log p_d / d loc + 0.5 (d information_d / d loc) diag_re(d) - information_d [(Sigma^-1 + W)^-1 d_mll_d_mode]_re(d).
    y / fixed_effects per datum in the order unique_idx refers to; result in the same order."""
    link = {"bernoulli_logit": 0, "bernoulli_probit": 1, "poisson": 2}[likelihood]
    dptr, order = _data_map(unique_idx)
    yi = np.ascontiguousarray(np.asarray(y)[order], dtype=np.int32)
    fe = None if fixed_effects is None else np.ascontiguousarray(np.asarray(fixed_effects, dtype=np.float64)[order])
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    rv = gen_rand_normal(n, num_rand_vec, seed_rand, _PROBE_RUN_ID)
    A, D, Ag, Dg, bad = vecchia_factor(coords_u, nn, cov_type, var, a, gauss=False, grad=True)
    out = np.empty(6); g = np.empty(2); mode = np.zeros(n); dbg = np.zeros(2 * n + 8)
    rc = lib().orc_vecchia_laplace_grad_map_dbg(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(Ag, C.c_double), _p(Dg, C.c_double), _p(nn, C.c_int),
                                                C.c_int(n), C.c_int(m), _p(dptr, C.c_int), _p(yi, C.c_int), None if fe is None else _p(fe, C.c_double),
                                                _p(rv, C.c_double), C.c_int(rv.shape[1]), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag),
                                                C.c_double(cg_delta_conv), C.c_double(delta_conv_mode), _p(out, C.c_double), _p(g, C.c_double),
                                                _p(mode, C.c_double), C.c_int(0), _p(dbg, C.c_double))
    if rc != 0:
        raise RuntimeError("orc_vecchia_laplace_grad_map_dbg failed")
    dld, sv = dbg[:n], dbg[n:2 * n]
    re_of = np.repeat(np.arange(n), np.diff(dptr))
    loc = mode[re_of] + (0.0 if fe is None else fe)
    first, info, dinfo = _lik_terms(likelihood, yi, loc)
    dW3 = np.bincount(re_of, weights=dinfo, minlength=n)
    diag = np.where(dW3 == 0.0, 0.0, dld / np.where(dW3 == 0.0, 1.0, dW3))
    gd = -first + 0.5 * dinfo * diag[re_of] - info * sv[re_of]
    res = np.empty_like(gd)
    res[order] = gd
    return res


def vecchia_setup(coords, m, ordering="random", seed=0):
    """Ordering + neighbour search (src/GPBoost/Vecchia_utils.cpp:1095-1221).
    Returns (perm, coords_ordered, nn)."""
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0]
    perm = shuffle(n, seed) if ordering == "random" else np.arange(n, dtype=np.int32)
    co = coords[perm]
    return perm, co, neighbors(co, m)


def gp_nll(coords, y, cov_pars, cov_function="exponential", shape=0.5, m=30, ordering="random", seed=0,
           setup=None):
    ct = cov_type_id(cov_function, shape)
    pt = transform_cov_pars(ct, cov_pars)
    perm, co, nn = setup if setup is not None else vecchia_setup(coords, m, ordering, seed)
    return vecchia_nll(co, nn, ct, pt, np.asarray(y, dtype=np.float64)[perm])[2]


# ---------------------------------------------------------------------------
# Full-scale Vecchia ("VIF") approximation, Gaussian likelihood, Euclidean neighbours
#   Psi = C_nm Sigma_m^-1 C_mn + Vecchia(residual process + nugget)
# Restates CalcSigmaComps (include/GPBoost/re_model_template.h:8151-8200: Sigma_m with its diagonal x (1 + 1e-6), V = L_m^-1 C_mn),
# CalcCovFactorGradientVecchia's full_scale_vecchia branches (src/GPBoost/Vecchia_utils.cpp:1463-1500, 1599-1623: every covariance of the
# per-point systems minus the predictive-process part V_a . V_b), CalcCovFactorFITC_FSA (:9646-9745: Woodbury matrix
# Sigma_m + (B C_nm)' D^-1 (B C_nm)), CalcYAux (:9785-9806) and the log-determinant (:2950-2966).  numpy; small n only.
# ---------------------------------------------------------------------------
def vif_setup(coords, m, num_ind_points, ordering="random", seed=0, max_it=1000, num_ind_points_preconditioner=0):
    """-> (perm, coords in Vecchia order, neighbour table, inducing points (k x d)): ordering shuffle + kmeans++ from the model's ONE
    generator (orc_stdlib.cpp: orc_vif_setup), Euclidean neighbour search among the ordered points.  num_ind_points_preconditioner > 0: a fifth
    return value, the inducing points of the "fitc" preconditioner of a full-scale Vecchia model with a non-Gaussian likelihood (a second
    kmeans++ run from the same generator, Calc_FITC_Preconditioner_Vecchia)."""
    coords = np.asarray(coords, dtype=np.float64)
    n, d = coords.shape
    cm = np.asfortranarray(coords)
    perm = np.empty(n, dtype=np.int32)
    ip = np.empty((num_ind_points, d), order="F")
    k2 = int(num_ind_points_preconditioner)
    ip2 = np.empty((max(k2, 1), d), order="F")
    rc = lib().orc_vif_setup2(C.c_int(n), C.c_int(d), _p(cm, C.c_double), C.c_int(seed), C.c_int(1 if ordering == "random" else 0),
                              C.c_int(int(num_ind_points)), C.c_int(max_it), _p(perm, C.c_int), _p(ip, C.c_double), C.c_int(k2), _p(ip2, C.c_double))
    if rc < 0:
        raise ValueError("more inducing points than data points")
    co = coords[perm]
    if k2 > 0:
        return perm, co, neighbors(co, m), np.ascontiguousarray(ip), np.ascontiguousarray(ip2)
    return perm, co, neighbors(co, m), np.ascontiguousarray(ip)


def _matern(cov_type, dist, var, a):
    r = a * dist
    if cov_type == 0:
        return var * np.exp(-r)
    if cov_type == 1:
        return var * (1.0 + r) * np.exp(-r)
    return var * (1.0 + r + r * r / 3.0) * np.exp(-r)


def vif_terms(co, nn, ip, cov_type, var, a, y):
    """-> (yTPsiInvy, log|Psi|, A, D): transformed scale (var = sigma1^2 / sigma^2, nugget 1)."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    n = co.shape[0]
    Sm = _matern(cov_type, cdist(ip, ip), var, a)
    Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6                                   # JITTER_MULT_IP_FITC_FSA
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, cdist(co, ip), var, a)                               # n x k
    V = solve_triangular(Lm, Cnm.T, lower=True)                                  # k x n  (chol_ip_cross_cov)
    A = np.zeros(nn.shape); D = np.empty(n)
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = var + 1.0 - V[:, i] @ V[:, i]
        if idx.size:
            Cnn = _matern(cov_type, cdist(co[idx], co[idx]), var, a) - V[:, idx].T @ V[:, idx]
            Cnn[np.diag_indices_from(Cnn)] += 1.0
            c = _matern(cov_type, cdist(co[idx], co[i:i + 1]), var, a)[:, 0] - V[:, idx].T @ V[:, i]
            Ai = cho_solve((cholesky(Cnn, lower=True), True), c)
            A[i, :idx.size] = Ai
            D[i] -= Ai @ c
    def B(x):                                                                    # B x, x: n or n x q
        out = x.copy()
        for i in range(n):
            idx = nn[i][nn[i] >= 0]
            out[i] -= A[i, :idx.size] @ x[idx]
        return out
    u = B(np.asarray(y, dtype=np.float64)); U = B(Cnm)
    W = Sm + U.T @ (U / D[:, None])
    Lw = cholesky(W, lower=True)
    r = U.T @ (u / D)
    quad = u @ (u / D) - r @ cho_solve((Lw, True), r)
    logdet = np.log(D).sum() - 2.0 * np.log(np.diag(Lm)).sum() + 2.0 * np.log(np.diag(Lw)).sum()
    return quad, logdet, A, D


def vif_resid_factor(co, nn, ip, cov_type, var, a, gauss=False):
    """Vecchia factor (A, D) of the RESIDUAL process of a full-scale Vecchia model + what goes with it: -> dict(A, D, C (n x k), V (n x k = chol_ip_cross_cov^T), Sm (k x k,
    diagonal x (1 + 1e-6)), Lm, logdet_Sm).  gauss = False: a latent process (non-Gaussian likelihood) -- no nugget, the neighbours' diagonal x (1 + 1e-10) AFTER the
    low-rank part is taken off (CalcCovFactorGradientVecchia, Vecchia_utils.cpp:1412-1414, :1461-1463, :1489-1500, :1599-1609)."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    co = np.asarray(co, dtype=np.float64); ip = np.asarray(ip, dtype=np.float64)
    n = co.shape[0]
    Sm = _matern(cov_type, cdist(ip, ip), var, a)
    Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, cdist(co, ip), var, a)
    V = solve_triangular(Lm, Cnm.T, lower=True)                                  # k x n
    A = np.zeros(nn.shape); D = np.empty(n)
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = (var + 1.0 if gauss else var) - V[:, i] @ V[:, i]
        if idx.size:
            Cnn = _matern(cov_type, cdist(co[idx], co[idx]), var, a) - V[:, idx].T @ V[:, idx]
            if gauss:
                Cnn[np.diag_indices_from(Cnn)] += 1.0
            else:
                Cnn[np.diag_indices_from(Cnn)] *= 1.0 + 1e-10
            c = _matern(cov_type, cdist(co[idx], co[i:i + 1]), var, a)[:, 0] - V[:, idx].T @ V[:, i]
            Ai = cho_solve((cholesky(Cnn, lower=True), True), c)
            A[i, :idx.size] = Ai
            D[i] -= Ai @ c
    return dict(A=A, D=D, C=np.asfortranarray(Cnm), V=np.asfortranarray(V.T), Sm=np.ascontiguousarray(Sm), Lm=Lm, logdet_Sm=2.0 * np.log(np.diag(Lm)).sum())


class vif_laplace(object):
    """`with orc.vif_laplace(co, nn, ip, cov_type, var, a, preconditioner, ip_preconditioner): f = ctx.factor; orc.vecchia_laplace_logit(co, nn, ..., factor=(f["A"], f["D"]))`
    -- the Vecchia-Laplace oracle calls inside the block evaluate a full-scale Vecchia (VIF) model with a non-Gaussian likelihood (FindModePostRandEffCalcMLLFSVA,
    likelihoods.h:3379-3750; gpb_oracle.c: orc_set_vif).  preconditioner: "fitc" (the reference's default; ip_preconditioner = its own inducing points, None: the model's,
    re_model_template.h:5249-5262), "vifdu" or "none".  Probe vectors (likelihoods.h:3633-3652): rand_vec_trace_I2_ (n x t) FIRST (generator counter 0), then
    rand_vec_trace_P_ (k x t, counter 1), then -- "vifdu" only -- rand_vec_trace_I3_ (n x t, counter 2)."""

    def __init__(self, co, nn, ip, cov_type, var, a, preconditioner="fitc", ip_preconditioner=None, num_rand_vec=50, seed_rand=1):
        self.factor = vif_resid_factor(co, nn, ip, cov_type, var, a)
        f = self.factor
        n = f["A"].shape[0]; self.k = f["Sm"].shape[0]
        self.pc = {"fitc": 0, "vifdu": 1, "none": 2}[preconditioner]
        self.fitc = None; self.rvP = None; self.rv3 = None
        if self.pc == 0:
            self.fitc = fitc_preconditioner(co, ip if ip_preconditioner is None else ip_preconditioner, cov_type, var, a, num_rand_vec, seed_rand)
            self.fitc.rv2 = gen_rand_normal(self.fitc.k, num_rand_vec, seed_rand, 1)
        elif self.pc == 1:
            self.rvP = gen_rand_normal(self.k, num_rand_vec, seed_rand, 1)
            self.rv3 = gen_rand_normal(n, num_rand_vec, seed_rand, 2)

    def __enter__(self):
        global _PROBE_RUN_ID
        if self.fitc is not None:
            self.fitc.__enter__()
        _PROBE_RUN_ID = 0
        f = self.factor
        fn = lib().orc_set_vif
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        fn(f["C"].ctypes.data, f["V"].ctypes.data, f["Sm"].ctypes.data, float(f["logdet_Sm"]), self.k, self.pc,
           None if self.rvP is None else self.rvP.ctypes.data, None if self.rv3 is None else self.rv3.ctypes.data)
        return self

    def __exit__(self, *exc):
        global _PROBE_RUN_ID
        lib().orc_clear_vif()
        if self.fitc is not None:
            self.fitc.__exit__()
        _PROBE_RUN_ID = 0
        return False


def _matern_grad_log_range(cov_type, dist, var, a):
    """d/d log a of var * k(a * dist), a the transformed range parameter (GradientRangeMaternShape0_5 / 1_5 / 2_5 with transf_scale,
    include/GPBoost/cov_fcts.h:2535-2554)."""
    r = a * dist
    if cov_type == 0:
        return -var * r * np.exp(-r)
    if cov_type == 1:
        return -var * r * r * np.exp(-r)
    return -var * r * r * (1.0 + r) / 3.0 * np.exp(-r)


def vif_grad_terms(co, nn, ip, cov_type, var, a, y):
    """Gradient of the full-scale Vecchia (VIF) likelihood, Gaussian data, wrt (log var, log a) on the transformed scale -- a numpy
    restatement of what the reference evaluates (small n):
      * the derivative factors B_grad = -dA, D_grad of the residual process: CalcCovFactorGradientVecchia's full_scale_vecchia branches
        (src/GPBoost/Vecchia_utils.cpp:1444-1458 set-up, :1503-1524 low-rank parts of dC_nn / dc, :1640-1656 dA_i = C_nn^-1 (dc - dC_nn A_i), dD_i,
        :1668-1679 the first point).  NOTE the reference differentiates the UN-jittered Sigma_m (GetZSigmaZtGrad) while it factorises
        Sigma_m with its diagonal x (1 + 1e-6) (CalcSigmaComps, include/GPBoost/re_model_template.h:8158-8160) -- restated as is;
      * CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i (re_model_template.h:2205-2330, 2447-2452), 'cholesky' branch of full_scale_vecchia.
    -> (quad, logdet, g[2][2], dA[2], dD[2]) with g[p] = (d(y' Psi^-1 y / 2) / d log theta_p, d(log|Psi| / 2) / d log theta_p), so that the
    reference's gradient entry of parameter p is g[p][0] / sigma2 + g[p][1]."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    import scipy.sparse as sp
    co = np.asarray(co, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    n, m = nn.shape
    k = ip.shape[0]
    dip = cdist(ip, ip); dnm = cdist(co, ip)
    Sm0 = _matern(cov_type, dip, var, a)
    Sm = Sm0.copy(); Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6                  # sigma_ip_stable
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, dnm, var, a)
    V = solve_triangular(Lm, Cnm.T, lower=True)                                  # chol_ip_cross_cov (k x n)
    SiCt = cho_solve((Lm, True), Cnm.T)                                          # sigma_ip_inv_cross_cov_T (k x n)
    dSm = [Sm0, _matern_grad_log_range(cov_type, dip, var, a)]                   # sigma_ip_grad (un-jittered)
    dCt = [Cnm.T, _matern_grad_log_range(cov_type, dnm, var, a).T]               # sigma_cross_cov_gradT (k x n)
    dSmSiCt = [dSm[p] @ SiCt for p in range(2)]                                  # sigma_ip_grad_sigma_ip_inv_cross_cov_T
    A = np.zeros((n, m)); D = np.empty(n)
    dA = [np.zeros((n, m)), np.zeros((n, m))]; dD = [np.empty(n), np.empty(n)]
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = var + 1.0 - V[:, i] @ V[:, i]
        low_self = [SiCt[:, i] @ (2.0 * dCt[p][:, i] - dSmSiCt[p][:, i]) for p in range(2)]     # :1654-1655
        if idx.size == 0:
            for p in range(2):
                dD[p][i] = (var if p == 0 else 0.0) - low_self[p]
            continue
        dn = cdist(co[idx], co[idx]); dc0 = cdist(co[idx], co[i:i + 1])[:, 0]
        Cnn = _matern(cov_type, dn, var, a) - V[:, idx].T @ V[:, idx]
        Cnn[np.diag_indices_from(Cnn)] += 1.0
        c = _matern(cov_type, dc0, var, a) - V[:, idx].T @ V[:, i]
        cf = (cholesky(Cnn, lower=True), True)
        Ai = cho_solve(cf, c)
        A[i, :idx.size] = Ai
        D[i] -= Ai @ c
        for p in range(2):
            dKnn = _matern(cov_type, dn, var, a) if p == 0 else _matern_grad_log_range(cov_type, dn, var, a)
            dKc = _matern(cov_type, dc0, var, a) if p == 0 else _matern_grad_log_range(cov_type, dc0, var, a)
            dc = dKc - (dCt[p][:, idx].T @ SiCt[:, i] + SiCt[:, idx].T @ (dCt[p][:, i] - dSmSiCt[p][:, i]))            # :1511-1512
            dCnn = dKnn - (dCt[p][:, idx].T @ SiCt[:, idx] + SiCt[:, idx].T @ (dCt[p][:, idx] - dSmSiCt[p][:, idx]))   # :1514-1516
            dAi = cho_solve(cf, dc - dCnn @ Ai)                                                                       # :1640-1641: (C_nn^-1 dc)' - A_i (C_nn^-1 dC_nn)'
            dA[p][i, :idx.size] = dAi
            dD[p][i] = (var if p == 0 else 0.0) - (dAi @ c + Ai @ dc) - low_self[p]                                    # :1571, 1646-1655
    rows = np.repeat(np.arange(n), m); cols = nn.ravel(); ok = cols >= 0
    B = (sp.identity(n, format="csr") - sp.csr_matrix((A.ravel()[ok], (rows[ok], cols[ok])), shape=(n, n))).tocsr()
    Dinv = 1.0 / D
    BC = B @ Cnm                                                                 # B_cross_cov_
    DiBC = Dinv[:, None] * BC                                                    # D_inv_B_cross_cov_
    SC = B.T @ DiBC                                                              # B_T_D_inv_B_cross_cov_
    W = Sm + BC.T @ DiBC                                                         # sigma_woodbury (:9728-9731)
    Lw = cholesky(W, lower=True)
    u = B @ y
    Sy = B.T @ (Dinv * u)
    r = Cnm.T @ Sy
    yaux = Sy - B.T @ (Dinv * (B @ (Cnm @ cho_solve((Lw, True), r))))           # CalcYAux (:9785-9806)
    quad = y @ yaux
    logdet = np.log(D).sum() - 2.0 * np.log(np.diag(Lm)).sum() + 2.0 * np.log(np.diag(Lw)).sum()
    s = cho_solve((Lm, True), Cnm.T @ yaux)                                      # sigma_ip_inv_cross_cov_y_aux (:2222)
    vy = Dinv * u                                                                # vecchia_y (:2262)
    wv = cho_solve((Lw, True), SC.T @ y)                                         # woodbury_vecchia_y (:2263-2264)
    g = np.zeros((2, 2))
    for p in range(2):
        Bg = -sp.csr_matrix((dA[p].ravel()[ok], (rows[ok], cols[ok])), shape=(n, n))
        g[p, 1] -= 0.5 * np.trace(cho_solve((Lm, True), dSm[p]))                                                       # :2274-2275
        g[p, 0] += 0.5 * s @ (dSm[p] @ s) - (dCt[p] @ yaux) @ s                                                        # :2277-2278
        CBg = Bg @ Cnm                                                                                                 # :2286-2290
        DgDiBC = dD[p][:, None] * DiBC                                                                                 # :2294-2298
        X1 = dCt[p] @ SC                                                                                               # :2301
        X2 = CBg.T @ DiBC                                                                                              # :2305
        X3 = DiBC.T @ DgDiBC                                                                                           # :2308
        dW = X1 + X1.T + X2.T + X2 - X3                                                                                # :2309-2310
        vgy = Bg.T @ vy - B.T @ (Dinv * (dD[p] * vy)) + B.T @ (Dinv * (Bg @ y))                                        # :2312-2313
        g[p, 1] += 0.5 * Dinv @ dD[p]                                                                                  # :2314
        g[p, 0] += 0.5 * y @ vgy - (Cnm.T @ vgy) @ wv + wv @ (X2 @ wv) - 0.5 * (DiBC @ wv) @ (DgDiBC @ wv)             # :2315-2317
        g[p, 1] += 0.5 * np.trace(cho_solve((Lw, True), dW + dSm[p]))                                                  # :2449-2451
    return quad, logdet, g, dA, dD, A, D


def vif_resid_factor_grad(co, nn, ip, cov_type, var, a):
    """LATENT residual-process factor of a full-scale Vecchia model with its derivatives wrt (log var, log a): vif_grad_terms' first half without the nugget and with
    the neighbours' diagonal x (1 + 1e-10) (vif_resid_factor) -- the derivative matrices carry no jitter, as in the reference (Vecchia_utils.cpp:1503-1524, :1599-1609,
    :1640-1656).  -> dict(A, D, dA[2], dD[2], Sm0, Sm, Lm, C (n x k), SiCt (k x n), dSm[2], dCt[2] (k x n), V (k x n))."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    co = np.asarray(co, dtype=np.float64); ip = np.asarray(ip, dtype=np.float64)
    n, m = nn.shape
    dip = cdist(ip, ip); dnm = cdist(co, ip)
    Sm0 = _matern(cov_type, dip, var, a)
    Sm = Sm0.copy(); Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, dnm, var, a)
    V = solve_triangular(Lm, Cnm.T, lower=True)
    SiCt = cho_solve((Lm, True), Cnm.T)
    dSm = [Sm0, _matern_grad_log_range(cov_type, dip, var, a)]
    dCt = [Cnm.T, _matern_grad_log_range(cov_type, dnm, var, a).T]
    dSmSiCt = [dSm[p] @ SiCt for p in range(2)]
    A = np.zeros((n, m)); D = np.empty(n)
    dA = [np.zeros((n, m)), np.zeros((n, m))]; dD = [np.empty(n), np.empty(n)]
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = var - V[:, i] @ V[:, i]
        low_self = [SiCt[:, i] @ (2.0 * dCt[p][:, i] - dSmSiCt[p][:, i]) for p in range(2)]
        if idx.size == 0:
            for p in range(2):
                dD[p][i] = (var if p == 0 else 0.0) - low_self[p]
            continue
        dn = cdist(co[idx], co[idx]); dc0 = cdist(co[idx], co[i:i + 1])[:, 0]
        Cnn = _matern(cov_type, dn, var, a) - V[:, idx].T @ V[:, idx]
        Cnn[np.diag_indices_from(Cnn)] *= 1.0 + 1e-10
        c = _matern(cov_type, dc0, var, a) - V[:, idx].T @ V[:, i]
        cf = (cholesky(Cnn, lower=True), True)
        Ai = cho_solve(cf, c)
        A[i, :idx.size] = Ai
        D[i] -= Ai @ c
        for p in range(2):
            dKnn = _matern(cov_type, dn, var, a) if p == 0 else _matern_grad_log_range(cov_type, dn, var, a)
            dKc = _matern(cov_type, dc0, var, a) if p == 0 else _matern_grad_log_range(cov_type, dc0, var, a)
            dc = dKc - (dCt[p][:, idx].T @ SiCt[:, i] + SiCt[:, idx].T @ (dCt[p][:, i] - dSmSiCt[p][:, i]))
            dCnn = dKnn - (dCt[p][:, idx].T @ SiCt[:, idx] + SiCt[:, idx].T @ (dCt[p][:, idx] - dSmSiCt[p][:, idx]))
            dAi = cho_solve(cf, dc - dCnn @ Ai)
            dA[p][i, :idx.size] = dAi
            dD[p][i] = (var if p == 0 else 0.0) - (dAi @ c + Ai @ dc) - low_self[p]
    return dict(A=A, D=D, dA=dA, dD=dD, Sm0=Sm0, Sm=Sm, Lm=Lm, C=Cnm, SiCt=SiCt, dSm=dSm, dCt=dCt, V=V)


def _optimal_c(a, b, tr_a, tr_b):
    """CalcOptimalC (CG_utils.cpp:1053-1069): cov(a, b) / var(b) of the per-probe samples, 1 when var(b) = 0."""
    cov = np.mean((a - tr_a) * (b - tr_b)); var = np.mean((b - tr_b) ** 2)
    return 1.0 if var == 0.0 else cov / var


def vif_laplace_grad(co, nn, ip, ip_pc, cov_type, var, a, y, likelihood="bernoulli_logit", aux=None, num_rand_vec=50, seed_rand=1, cg_max_num_it=1000,
                     cg_max_num_it_tridiag=1000, cg_delta_conv=1e-2, delta_conv_mode=1e-8, want_parts=False, mode_init=None):
    """(negll, gradient of negll wrt (log sigma1^2, log a)[, d / d log aux]) of a full-scale Vecchia (VIF) model with a non-Gaussian likelihood, iterative methods, "fitc"
    preconditioner: Likelihood::CalcGradNegMargLikelihoodLaplaceApproxFSVA (likelihoods.h:5279-5520).  The mode, the log-determinant's block CG, d logdet / d mode with its
    variance reduction, the implicit solve and the auxiliary parameter's part come from gpb_oracle.c (orc_vecchia_laplace_grad with orc_set_vif: the code the Vecchia path
    shares); the covariance parameters' part (:5413-5520) is restated here with dense / sparse numpy on those by-products.  co / y in Vecchia order."""
    from scipy.linalg import cho_solve, cholesky, solve_triangular
    from scipy.spatial.distance import cdist
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    link = LINK_ID[likelihood]
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, m = nn.shape
    t = num_rand_vec
    F = vif_resid_factor_grad(co, nn, ip, cov_type, var, a)
    A = np.ascontiguousarray(F["A"]); D = np.ascontiguousarray(F["D"])
    ctx = vif_laplace.__new__(vif_laplace)
    ctx.factor = dict(A=A, D=D, C=np.asfortranarray(F["C"]), V=np.asfortranarray(F["V"].T), Sm=np.ascontiguousarray(F["Sm"]), Lm=F["Lm"],
                      logdet_Sm=2.0 * np.log(np.diag(F["Lm"])).sum())
    ctx.k = F["Sm"].shape[0]; ctx.pc = 0; ctx.rvP = None; ctx.rv3 = None
    ctx.fitc = fitc_preconditioner(co, ip_pc, cov_type, var, a, t, seed_rand)
    ctx.fitc.rv2 = gen_rand_normal(ctx.fitc.k, t, seed_rand, 1)
    kp = ctx.fitc.k
    yi, yd = _responses(likelihood, y)
    aux_g4 = np.zeros(8) if link >= 3 else None
    parts = np.zeros(2 * n * t + 6 * n + kp * kp)
    Ag = np.zeros((2, n, m)); Dg = np.zeros((2, n))
    out = np.empty(6); g = np.empty(2)
    mode = np.zeros(n) if mode_init is None else np.ascontiguousarray(mode_init, dtype=np.float64).copy()      # (mode_init: Newton's method starts there -- a warm start)
    with ctx:
        rv = gen_rand_normal(n, t, seed_rand, 0)
        lib().orc_vif_set_parts_out.argtypes = [C.c_void_p]
        lib().orc_vif_set_parts_out(parts.ctypes.data)
        with _aux_context(link, aux, yd, aux_g4):
            rc = lib().orc_vecchia_laplace_grad(C.c_int(link), _p(A, C.c_double), _p(D, C.c_double), _p(Ag, C.c_double), _p(Dg, C.c_double), _p(nn, C.c_int), C.c_int(n),
                                                C.c_int(m), _p(yi, C.c_int), None, _p(rv, C.c_double), C.c_int(t), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag),
                                                C.c_double(cg_delta_conv), C.c_double(delta_conv_mode), _p(out, C.c_double), _p(g, C.c_double), _p(mode, C.c_double),
                                                C.c_int(0 if mode_init is None else 1), None)
        lib().orc_vif_set_parts_out(None)
    if rc != 0:
        raise RuntimeError("orc_vecchia_laplace_grad failed")
    q = 0
    U = parts[q:q + n * t].reshape(t, n).T; q += n * t
    WIPIZ = parts[q:q + n * t].reshape(t, n).T; q += n * t
    dld = parts[q:q + n]; q += n
    sv = parts[q:q + n]; q += n
    W = parts[q:q + n]; q += n
    dW = parts[q:q + n]; q += n
    wp = parts[q:q + n]; q += n                     # diagonal_approx_inv_preconditioner_ (of the last Newton step)
    md = parts[q:q + n]; q += n
    cholk = np.tril(parts[q:q + kp * kp].reshape(kp, kp))      # chol_fact_woodbury_preconditioner_ (lower)
    PIZ = W[:, None] * WIPIZ                                   # PI_Z = P^-1 Z
    # --- the model's matrices ---
    rows = np.repeat(np.arange(n), m); cols = nn.ravel(); ok = cols >= 0
    B = (sp.identity(n, format="csr") - sp.csr_matrix((A.ravel()[ok], (rows[ok], cols[ok])), shape=(n, n))).tocsr()
    Bt = B.T.tocsr()
    Dinv = 1.0 / D
    Cm = F["C"]; SiCt = F["SiCt"]; Sm = F["Sm"]
    SigI = lambda X: Bt @ (Dinv[:, None] * (B @ X)) if X.ndim == 2 else Bt @ (Dinv * (B @ X))          # B' D^-1 B X
    def SigV(X):                                                                                        # B^-1 D B^-T X
        Y = spl.spsolve_triangular(Bt, X, lower=False, unit_diagonal=True)
        Y = (D[:, None] * Y) if X.ndim == 2 else D * Y
        return spl.spsolve_triangular(B, Y, lower=True, unit_diagonal=True)
    Q = SigI(Cm)                                               # Bt_D_inv_B_cross_cov
    Mw = Sm + Cm.T @ Q                                         # sigma_woodbury
    Lw = (cholesky(Mw, lower=True), True)
    # --- the preconditioner's matrices ---
    ipp = np.asarray(ip_pc, dtype=np.float64)
    dpp = cdist(ipp, ipp); dnp_ = cdist(np.asarray(co, dtype=np.float64), ipp)
    Smp0 = _matern(cov_type, dpp, var, a)
    Smp = Smp0.copy(); Smp[np.diag_indices_from(Smp)] *= 1.0 + 1e-6
    Lmp = (cholesky(Smp, lower=True), True)
    Cp = _matern(cov_type, dnp_, var, a)
    dSmp = [Smp0, _matern_grad_log_range(cov_type, dpp, var, a)]
    dCp = [Cp, _matern_grad_log_range(cov_type, dnp_, var, a)]
    sipc = cho_solve(Lmp, Cp.T)                                # sigma_ip_inv_sigma_cross_cov_preconditioner (kp x n)
    sipc_PIZ = sipc @ PIZ
    SiCt_PIZ = SiCt @ PIZ                                      # sigma_ip_inv_cross_cov_PI_Z
    SigI_mode = SigI(md)
    grad = np.zeros(2); info = []
    for p in range(2):
        dC = F["dCt"][p].T; dSm = F["dSm"][p]
        if p == 0:
            SId = lambda X: -SigI(X)
        else:
            Bg = -sp.csr_matrix((F["dA"][1].ravel()[ok], (rows[ok], cols[ok])), shape=(n, n))
            Bgt = Bg.T.tocsr(); dDp = F["dD"][1]
            def SId(X, Bg=Bg, Bgt=Bgt, dDp=dDp):
                sc = (lambda v, Y: v[:, None] * Y) if X.ndim == 2 else (lambda v, Y: v * Y)
                BX = B @ X
                return Bgt @ sc(Dinv, BX) + Bt @ sc(Dinv, Bg @ X) - Bt @ sc(Dinv * dDp * Dinv, BX)
        SIdC = SId(Cm)
        Mg = dSm + Cm.T @ SIdC
        X1 = Q.T @ dC
        Mg = Mg + X1 + X1.T                                    # sigma_woodbury_grad
        a1 = SId(md)
        wsol = cho_solve(Lw, Cm.T @ SigI_mode)
        SIdm = (a1 - SigI(Cm @ cho_solve(Lw, Cm.T @ a1)) - SId(Cm @ wsol) - SigI(dC @ wsol) - SigI(Cm @ cho_solve(Lw, dC.T @ SigI_mode))
                + SigI(Cm @ cho_solve(Lw, Mg @ wsol)))        # SigmaI_deriv_mode
        explicit = 0.5 * (md @ SIdm)
        PPd = dC @ SiCt_PIZ + SiCt.T @ (dC.T @ PIZ) - SiCt.T @ (dSm @ SiCt_PIZ)
        Sd = PPd - SigV(SId(SigV(PIZ)))                        # SigmaI_deriv_sample_vec = d Sigma / d theta PI_Z
        sample_Sigma = (U * Sd).sum(axis=0)
        stoch_tr = sample_Sigma.mean()
        # variance reduction with d P / d theta
        PgPIZ = dCp[p] @ sipc_PIZ + sipc.T @ (dCp[p].T @ PIZ) - sipc.T @ (dSmp[p] @ sipc_PIZ)
        dSs = dSmp[p] @ sipc
        dgrad = dSmp[p][0, 0] - (2.0 * np.einsum("ki,ik->i", sipc, dCp[p]) - np.einsum("ki,ki->i", sipc, dSs))
        PgPIZ = PgPIZ + dgrad[:, None] * PIZ
        tr_PI_P = (dgrad * wp).sum() - np.trace(cho_solve(Lmp, dSmp[p]))
        DiC = wp[:, None] * Cp
        X2 = dCp[p].T @ DiC
        Mgp = dSmp[p] + X2 + X2.T - DiC.T @ (dgrad[:, None] * DiC)
        tr_PI_P += np.trace(cho_solve((cholk, True), Mgp))
        sample_P = (PIZ * PgPIZ).sum(axis=0)
        c_opt = _optimal_c(sample_Sigma, sample_P, stoch_tr, tr_PI_P)
        dl = stoch_tr - c_opt * (sample_P.mean() - tr_PI_P)
        grad[p] = explicit + 0.5 * dl - sv @ SIdm
        info.append((md @ SIdm, dl, c_opt, -(sv @ SIdm)))
    negll = -out[0]
    if link == 6:
        grad = np.array([grad[0], grad[1], aux_g4[0], aux_g4[4]])
    elif link >= 3:
        grad = np.array([grad[0], grad[1], aux_g4[0]])
    if want_parts:
        return negll, grad, dict(per_par=np.array(info), dlogdet_dmode=dld.copy(), implicit_solve=sv.copy(), mode=md.copy())
    return negll, grad


def vif_predict_obs_only(co, nn, ip, cov_type, pars_trans, y, coords_pred, m_pred, predict_response=True):
    """Prediction of a full-scale Vecchia (VIF) model, 'order_obs_first_cond_obs_only' (CalcPredVecchiaObservedFirstOrder with the
    full_scale_vecchia arguments, src/GPBoost/Vecchia_utils.cpp:1701-2060, called from re_model_template.h:4041-4056): the conditional law of
    y_p = C_p Sigma_m^-1 eta + e_p given y = C Sigma_m^-1 eta + e under the model -- eta ~ N(0, Sigma_m) the inducing values, (e, e_p) the
    residual process + nugget in its Vecchia form, every prediction point conditioning on its m_pred nearest OBSERVED points:
        mean_p = A_p y_nn + (B C)_p W^-1 (B C)' D^-1 B y,      var_p = sigma2 (D_p + (B C)_p W^-1 (B C)_p')   [- sigma2 for the latent process]
    with (B C)_p = C_p - A_p C_nn and W the Woodbury matrix of the observed points.  numpy, small n.  -> (mean, var)."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    sigma2, var, a = pars_trans
    co = np.asarray(co, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n, npd = co.shape[0], cp.shape[0]
    Sm = _matern(cov_type, cdist(ip, ip), var, a)
    Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, cdist(co, ip), var, a)
    Cpm = _matern(cov_type, cdist(cp, ip), var, a)
    V = solve_triangular(Lm, Cnm.T, lower=True); Vp = solve_triangular(Lm, Cpm.T, lower=True)
    quad, logdet, A, D = vif_terms(co, nn, ip, cov_type, var, a, y)
    y = np.asarray(y, dtype=np.float64)

    def Bmul(x):
        out = x.copy()
        for i in range(n):
            idx = nn[i][nn[i] >= 0]
            out[i] -= A[i, :idx.size] @ x[idx]
        return out
    u = Bmul(y); U = Bmul(Cnm)
    W = Sm + U.T @ (U / D[:, None])
    Lw = cholesky(W, lower=True)
    v = cho_solve((Lw, True), U.T @ (u / D))
    nnp = neighbors_range(np.vstack([co, cp]), min(m_pred, n), n, n - 1)[n:]
    mean = np.empty(npd); varp = np.empty(npd)
    for i in range(npd):
        idx = nnp[i][nnp[i] >= 0]
        Cnn = _matern(cov_type, cdist(co[idx], co[idx]), var, a) - V[:, idx].T @ V[:, idx]
        Cnn[np.diag_indices_from(Cnn)] += 1.0
        c = _matern(cov_type, cdist(co[idx], cp[i:i + 1]), var, a)[:, 0] - V[:, idx].T @ Vp[:, i]
        Ai = cho_solve((cholesky(Cnn, lower=True), True), c)
        Dp = var + 1.0 - Vp[:, i] @ Vp[:, i] - Ai @ c
        bc = Cpm[i] - Ai @ Cnm[idx]
        mean[i] = Ai @ y[idx] + bc @ v
        t = solve_triangular(Lw, bc, lower=True)
        varp[i] = sigma2 * (Dp + t @ t - (0.0 if predict_response else 1.0))
    return mean, varp


def vif_predict_cond_all(co, nn, ip, cov_type, pars_trans, y, coords_pred, m_pred, predict_response=True, want_cov=False):
    """Prediction of a full-scale Vecchia (VIF) model, 'order_obs_first_cond_all' (CalcPredVecchiaObservedFirstOrder with CondObsOnly = false and the
    full_scale_vecchia arguments, src/GPBoost/Vecchia_utils.cpp:1803-1826 neighbours among observed AND preceding prediction points, :1889-1925 the
    residual covariances of such neighbours, :1975-2046 mean and (co)variances; called from re_model_template.h:4057-4071).  With Bpo / Bp / Dp the
    factor rows of the appended points in the residual process + nugget (Bp unit lower triangular) and C~ = [C; C_p]:
        mean = Bp^-1 (-Bpo y + (B~ C~)_p v),   v = W^-1 (B C)' D^-1 B y          (:1976-1981)
        cov  = sigma2 (Bp^-1 Dp Bp^-T + T W^-1 T'),   T = Bp^-1 (B~ C~)_p = C_p + Bp^-1 Bpo C     [- sigma2 I for the latent process]
    -- the reference's eight-term expression (:2028-2046) collapses to T W^-1 T' with W = Sigma_m + (B C)' D^-1 (B C).  numpy, small n.
    -> (mean, var) or (mean, var, cov)."""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cholesky, solve_triangular, cho_solve
    sigma2, var, a = pars_trans
    co = np.asarray(co, dtype=np.float64); cp = np.asarray(coords_pred, dtype=np.float64)
    n, npd = co.shape[0], cp.shape[0]
    Sm = _matern(cov_type, cdist(ip, ip), var, a)
    Sm[np.diag_indices_from(Sm)] *= 1.0 + 1e-6
    Lm = cholesky(Sm, lower=True)
    Cnm = _matern(cov_type, cdist(co, ip), var, a)
    Cpm = _matern(cov_type, cdist(cp, ip), var, a)
    call = np.vstack([co, cp]); Call = np.vstack([Cnm, Cpm])
    Vall = solve_triangular(Lm, Call.T, lower=True)
    quad, logdet, A, D = vif_terms(co, nn, ip, cov_type, var, a, y)
    y = np.asarray(y, dtype=np.float64)

    def Bmul(x):
        out = x.copy()
        for i in range(n):
            idx = nn[i][nn[i] >= 0]
            out[i] -= A[i, :idx.size] @ x[idx]
        return out
    u = Bmul(y); U = Bmul(Cnm)
    W = Sm + U.T @ (U / D[:, None])
    Lw = cholesky(W, lower=True)
    v = cho_solve((Lw, True), U.T @ (u / D))
    nnp = neighbors_range(call, min(m_pred, n + npd - 1), n, -1)[n:]
    yall = np.concatenate([y, np.zeros(npd)])
    mean = np.empty(npd); Dp = np.empty(npd)
    T = np.empty((npd, ip.shape[0])); Linv = np.zeros((npd, npd))
    for i in range(npd):
        idx = nnp[i][nnp[i] >= 0]
        Cnn = _matern(cov_type, cdist(call[idx], call[idx]), var, a) - Vall[:, idx].T @ Vall[:, idx]
        Cnn[np.diag_indices_from(Cnn)] += 1.0
        c = _matern(cov_type, cdist(call[idx], cp[i:i + 1]), var, a)[:, 0] - Vall[:, idx].T @ Vall[:, n + i]
        Ai = cho_solve((cholesky(Cnn, lower=True), True), c)
        Dp[i] = var + 1.0 - Vall[:, n + i] @ Vall[:, n + i] - Ai @ c
        bc = Cpm[i] - Ai @ Call[idx]                       # row i of B~ C~
        w = Ai @ yall[idx] + bc @ v                        # -(Bpo y)_i + (B~ C~)_i v
        Linv[i, i] = 1.0
        for aij, c_ in zip(Ai, idx):
            if c_ >= n:                                    # a preceding prediction point: forward substitution with Bp = I - A_pp
                q = c_ - n
                w += aij * mean[q]; bc = bc + aij * T[q]; Linv[i, :q + 1] += aij * Linv[q, :q + 1]
        mean[i] = w; T[i] = bc
    Tw = solve_triangular(Lw, T.T, lower=True)             # k x np
    cov = sigma2 * ((Linv * Dp[None, :]) @ Linv.T + Tw.T @ Tw)
    if not predict_response:
        cov[np.diag_indices_from(cov)] -= sigma2
    if want_cov:
        return mean, np.diag(cov).copy(), cov
    return mean, np.diag(cov).copy()


def vif_nll(coords, y, cov_pars, cov_function="exponential", shape=0.5, m=30, num_ind_points=200, ordering="random", seed=0, setup=None):
    ct = cov_type_id(cov_function, shape)
    pt = transform_cov_pars(ct, cov_pars)
    perm, co, nn, ip = setup if setup is not None else vif_setup(coords, m, num_ind_points, ordering, seed)
    quad, logdet, _, _ = vif_terms(co, nn, ip, ct, pt[1], pt[2], np.asarray(y, dtype=np.float64)[perm])
    n = co.shape[0]
    return quad / 2.0 / pt[0] + logdet / 2.0 + n / 2.0 * (np.log(pt[0]) + np.log(2 * np.pi))


def vecchia_nll_weighted(co, nn, cov_type, pars_trans, y, nug):
    """Gaussian Vecchia likelihood with sample weights: observation i has the nugget nug[i] = 1 / w_i on the transformed scale
    (GetGaussianNuggetDiagFromWeights, include/GPBoost/re_model_template.h:6393-6417; src/GPBoost/Vecchia_utils.cpp:1418-1422, 1610-1614).
    numpy, small n.  -> (yTPsiInvy, log|Psi|, negll, A, D)"""
    from scipy.spatial.distance import cdist
    from scipy.linalg import cho_solve, cholesky
    sigma2, var, a = pars_trans
    n = co.shape[0]
    A = np.zeros(nn.shape); D = np.empty(n); u = np.empty(n)
    for i in range(n):
        idx = nn[i][nn[i] >= 0]
        D[i] = var + nug[i]; u[i] = y[i]
        if idx.size:
            Cnn = _matern(cov_type, cdist(co[idx], co[idx]), var, a)
            Cnn[np.diag_indices_from(Cnn)] += nug[idx]
            c = _matern(cov_type, cdist(co[idx], co[i:i + 1]), var, a)[:, 0]
            Ai = cho_solve((cholesky(Cnn, lower=True), True), c)
            A[i, :idx.size] = Ai
            D[i] -= Ai @ c
            u[i] -= Ai @ y[idx]
    quad = float(u @ (u / D)); logdet = float(np.log(D).sum())
    return quad, logdet, quad / 2.0 / sigma2 + logdet / 2.0 + n / 2.0 * (np.log(sigma2) + np.log(2 * np.pi)), A, D


# ---------------------------------------------------------------------------
# The R test-suite's deterministic fixture
# (R-package/tests/testthat/test_GPModel_gaussian_process.R:36-60)
# ---------------------------------------------------------------------------
def sim_rand_unif(n, init_c):
    """LCG x_{k+1} = (22695477 x_k + 1) mod 2^32, x_0 = floor(init_c 2^32); returns x / 2^32.
    NB: R evaluates this in *double* arithmetic (the product exceeds 2^53 and is rounded), so the
    recursion is done in float64 here too -- exact integer arithmetic gives a different sequence."""
    s = np.empty(n)
    s[0] = np.floor(init_c * 2 ** 32)
    for i in range(1, n):
        s[i] = (22695477 * s[i - 1] + 1) % 2 ** 32
    return s / 2 ** 32


def r_fixture():
    from scipy.spatial.distance import cdist
    from scipy.stats import norm
    n, d = 100, 2
    coords = sim_rand_unif(n * d, 0.1).reshape((n, d), order="F")
    Sigma = np.exp(-cdist(coords, coords) / 0.1) + 1e-20 * np.eye(n)
    Cc = np.linalg.cholesky(Sigma)
    y = Cc @ norm.ppf(sim_rand_unif(n, 0.8)) + norm.ppf(sim_rand_unif(n, 0.1)) / 5
    return coords, y


def r_fixture_multiple():
    """R-package/tests/testthat/test_GPModel_gaussian_process.R:72-78, :1678-1679: 25 locations, each observed four times
    (coords_multiple, eps_multiple + xi) and the suite's initial values c(var(y)/2, var(y)/2, mean(dist(unique(coords)))/3)."""
    from scipy.spatial.distance import cdist, pdist
    from scipy.stats import norm
    n, d = 100, 2
    cu = sim_rand_unif(n * d // 4, 0.1).reshape((n // 4, d), order="F")
    coords = np.vstack([cu, cu, cu, cu])
    Sigma = np.exp(-cdist(coords, coords) / 0.1) + 1e-10 * np.eye(n)
    y = np.linalg.cholesky(Sigma) @ norm.ppf(sim_rand_unif(n, 0.8)) + norm.ppf(sim_rand_unif(n, 0.1)) / 5
    init = np.array([np.var(y, ddof=1) / 2, np.var(y, ddof=1) / 2, pdist(cu).mean() / 3])
    return coords, y, init


def r_fixture_logit():
    """Binary fixture of R-package/tests/testthat/test_GPModel_non_Gaussian_data.R:52-62, 2510-2513 (same coords / L / b_1
    as above; y = 1{u < sigmoid(L b_1)} with u from the LCG started at 0.2341)."""
    from scipy.spatial.distance import cdist
    from scipy.stats import norm
    n, d = 100, 2
    coords = sim_rand_unif(n * d, 0.1).reshape((n, d), order="F")
    Sigma = np.exp(-cdist(coords, coords) / 0.1) + 1e-20 * np.eye(n)
    Cc = np.linalg.cholesky(Sigma)
    probs = 1.0 / (1.0 + np.exp(-(Cc @ norm.ppf(sim_rand_unif(n, 0.8)))))
    return coords, (sim_rand_unif(n, 0.2341) < probs).astype(np.float64)


def r_fixture_probit():
    """Binary fixture of R-package/tests/testthat/test_GPModel_non_Gaussian_data.R:52-62, 1391-1392 (same coords / L / b_1;
    y = 1{u < Phi(L b_1)} with u from the LCG started at 0.19341); expected nll 67.18342059 at cov_pars (1, 0.2) (:1405, :1426)."""
    from scipy.spatial.distance import cdist
    from scipy.stats import norm
    n, d = 100, 2
    coords = sim_rand_unif(n * d, 0.1).reshape((n, d), order="F")
    Sigma = np.exp(-cdist(coords, coords) / 0.1) + 1e-20 * np.eye(n)
    Cc = np.linalg.cholesky(Sigma)
    probs = norm.cdf(Cc @ norm.ppf(sim_rand_unif(n, 0.8)))
    return coords, (sim_rand_unif(n, 0.19341) < probs).astype(np.float64)

