"""ctypes view of the low-level C ABI (include/gpb_hip.h): what the reference's C++ host would call.

``VecchiaState`` and ``HistBuilder`` are thin RAII wrappers; every method is one C call.  Used by the
parity tests and by bench.py (resident and sharded evaluation); the high-level mirror of the
reference's Python API is :class:`gpboost_amd.GPModel`.
"""
import ctypes as C

import numpy as np

from .basic import GPBoostError, _lib, _shim_call

MODE_NLL, MODE_FACTOR, MODE_GRAD = 0, 1, 2


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class VecchiaState(object):
    """coords: (n, d) array ALREADY in Vecchia order; m: number of neighbours."""

    def __init__(self, coords, m):
        coords = np.asarray(coords, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords.reshape(-1, 1)
        self.n, self.d = coords.shape
        self.m = max(min(int(m), self.n - 1), 1)
        cm = np.asfortranarray(coords)
        self.h = C.c_void_p()
        _shim_call(_lib().gpb_hip_vecchia_create(C.c_int(self.n), C.c_int(self.d), C.c_int(int(m)), _p(cm),
                                                 C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value is not None:
            _lib().gpb_hip_vecchia_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_handle(cls, ptr, n, d, m):
        """Non-owning view of a gpb_hip_vecchia_t* owned by a GPModel (GPB_HIP_GetVecchiaHandle)."""
        self = cls.__new__(cls)
        self.n, self.d, self.m = int(n), int(d), int(m)
        self.h = C.c_void_p(ptr.value if isinstance(ptr, C.c_void_p) else int(ptr))
        self.close = lambda: None
        return self

    def set_stream(self, hip_stream_ptr):
        _shim_call(_lib().gpb_hip_vecchia_set_stream(self.h, C.c_void_p(int(hip_stream_ptr))))

    def find_neighbors(self):
        dup = C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_find_neighbors(self.h, C.byref(dup)))
        return bool(dup.value)

    def set_neighbors(self, nn):
        nn = np.ascontiguousarray(nn, dtype=np.int32)
        assert nn.shape == (self.n, self.m), (nn.shape, self.n, self.m)
        _shim_call(_lib().gpb_hip_vecchia_set_neighbors(self.h, _p(nn, C.c_int32)))

    def get_neighbors(self):
        nn = np.empty((self.n, self.m), dtype=np.int32)
        _shim_call(_lib().gpb_hip_vecchia_get_neighbors(self.h, _p(nn, C.c_int32)))
        return nn

    def set_shard(self, i_begin, i_end):
        _shim_call(_lib().gpb_hip_vecchia_set_shard(self.h, C.c_int(int(i_begin)), C.c_int(int(i_end))))

    def set_y(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.shape == (self.n,)
        _shim_call(_lib().gpb_hip_vecchia_set_y(self.h, _p(y)))

    def set_y_dev(self, y_dev_ptr):
        _shim_call(_lib().gpb_hip_vecchia_set_y_dev(self.h, C.c_void_p(int(y_dev_ptr))))

    def sync(self):
        _shim_call(_lib().gpb_hip_vecchia_sync(self.h))

    def nll_terms(self, cov_type, var, a, gauss=True):
        """-> array {y^T Psi^-1 y, log|Psi|, #bad} over this handle's shard."""
        out = np.empty(3)
        _shim_call(_lib().gpb_hip_vecchia_nll_terms(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                    C.c_int(1 if gauss else 0), _p(out)))
        return out

    def nll_terms_dev(self, cov_type, var, a, out_dev_ptr, gauss=True):
        _shim_call(_lib().gpb_hip_vecchia_nll_terms_dev(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                        C.c_int(1 if gauss else 0), C.c_void_p(int(out_dev_ptr))))

    def grad_terms(self, cov_type, var, a):
        out = np.empty(7)
        _shim_call(_lib().gpb_hip_vecchia_grad_terms(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), _p(out)))
        return out

    def grad_terms_dev(self, cov_type, var, a, out_dev_ptr):
        _shim_call(_lib().gpb_hip_vecchia_grad_terms_dev(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                         C.c_void_p(int(out_dev_ptr))))

    def comm_init(self, id128, rank, world):
        """Collective: bootstrap the in-library RCCL communicator from the 128-byte unique id of rank 0."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(id128))
        _shim_call(_lib().gpb_hip_vecchia_comm_init(self.h, buf, C.c_int(int(rank)), C.c_int(int(world))))

    def mailbox_attach(self, name, rank, world):
        """Collective: attach to the node-local shared-memory mailbox `name` (mailbox_create() on rank 0); the 3 / 7 sums of the sharded
        evaluations then travel through it instead of ncclAllReduce."""
        _shim_call(_lib().gpb_hip_vecchia_mailbox_attach(self.h, C.c_char_p(name if isinstance(name, bytes) else name.encode()), C.c_int(int(rank)), C.c_int(int(world))))

    def mailbox_info(self):
        r, w = C.c_int(0), C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_mailbox_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def mailbox_detach(self):
        _shim_call(_lib().gpb_hip_vecchia_mailbox_detach(self.h))

    def timing(self, enable):
        """enable=True: record HIP events around every point-kernel launch; enable=False: -> (launches, mean kernel ms of the last <= 256)"""
        cnt = C.c_int64(0); ms = C.c_double(0.0)
        _shim_call(_lib().gpb_hip_vecchia_timing(self.h, C.c_int(1 if enable else 0), C.byref(cnt), C.byref(ms)))
        return cnt.value, ms.value

    def comm_init_local(self, group, rank):
        """Collective over the threads of an in-process group (LocalGroup) instead of RCCL."""
        _shim_call(_lib().gpb_hip_vecchia_comm_init_local(self.h, group.g, C.c_int(int(rank))))

    def comm_info(self):
        """(rank, number of ranks) as RCCL reports them for the handle's communicator (ncclCommUserRank / ncclCommCount); (0, 0) without one."""
        r, w = C.c_int(0), C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def find_neighbors_part(self, part, nparts):
        dup = C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_find_neighbors_part(self.h, C.c_int(int(part)), C.c_int(int(nparts)), C.byref(dup)))
        return bool(dup.value)

    def neighbors_allreduce(self):
        dup = C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_neighbors_allreduce(self.h, C.byref(dup)))
        return bool(dup.value)

    def yaux_allreduce(self):
        out = np.empty(self.n)
        _shim_call(_lib().gpb_hip_vecchia_yaux_allreduce(self.h, _p(out)))
        return out

    def nll_terms_allreduce(self, cov_type, var, a, gauss=True):
        out = np.empty(3)
        _shim_call(_lib().gpb_hip_vecchia_nll_terms_allreduce(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                              C.c_int(1 if gauss else 0), _p(out)))
        return out

    def grad_terms_allreduce(self, cov_type, var, a):
        out = np.empty(7)
        _shim_call(_lib().gpb_hip_vecchia_grad_terms_allreduce(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), _p(out)))
        return out

    def bench(self, mode, cov_type, var, a, warmup, steps):
        """-> (ms_total, ms_point_kernel_avg, last_terms[7])"""
        t = C.c_double(0); k = C.c_double(0); out = np.empty(7)
        _shim_call(_lib().gpb_hip_vecchia_bench(self.h, C.c_int(mode), C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                C.c_int(warmup), C.c_int(steps), C.byref(t), C.byref(k), _p(out)))
        return t.value, k.value, out

    def factor(self, cov_type, var, a, gauss=True):
        _shim_call(_lib().gpb_hip_vecchia_factor(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a),
                                                 C.c_int(1 if gauss else 0)))

    def get_factor(self):
        A = np.empty((self.n, self.m)); D = np.empty(self.n); u = np.empty(self.n)
        _shim_call(_lib().gpb_hip_vecchia_get_factor(self.h, _p(A), _p(D), _p(u)))
        return A, D, u

    def yaux(self):
        out = np.empty(self.n)
        _shim_call(_lib().gpb_hip_vecchia_yaux(self.h, _p(out)))
        return out

    def predict_obs_only(self, coords_pred, num_neighbors_pred, cov_type, var, a):
        """-> (pred_mean, Dp, has_duplicates) for prediction points conditioning on the observed points only."""
        cp = np.asarray(coords_pred, dtype=np.float64)
        if cp.ndim == 1:
            cp = cp.reshape(-1, 1)
        cm = np.asfortranarray(cp)
        mu = np.empty(cp.shape[0]); Dp = np.empty(cp.shape[0]); dup = C.c_int(0)
        _shim_call(_lib().gpb_hip_vecchia_predict_obs_only(self.h, C.c_int(cp.shape[0]), _p(cm), C.c_int(int(num_neighbors_pred)),
                                                           C.c_int(cov_type), C.c_double(var), C.c_double(a), _p(mu), _p(Dp), C.byref(dup)))
        return mu, Dp, bool(dup.value)

    def newton_leaf_values(self, leaf_index, num_leaves):
        """Needs factor(gauss=True) and yaux() for the current y = F - y; leaf_index in Vecchia order."""
        leaf = np.ascontiguousarray(leaf_index, dtype=np.int32)
        out = np.empty(int(num_leaves))
        _shim_call(_lib().gpb_hip_vecchia_newton_leaf_values(self.h, _p(leaf, C.c_int), C.c_int(int(num_leaves)), _p(out)))
        return out

    def laplace_set_likelihood(self, likelihood):
        lid = {"bernoulli_logit": 0, "bernoulli_probit": 1, "poisson": 2, "gamma": 3, "negative_binomial": 4, "beta": 5, "t": 6, "lognormal": 7, "gaussian_latent": 8,
               "binomial_logit": 0, "binomial_probit": 1, "quasi_bernoulli_logit": 0, "quasi_bernoulli_probit": 1}[likelihood]
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_likelihood(self.h, C.c_int(lid)))
        self._lap_link = lid

    def laplace_set_preconditioner(self, cg_preconditioner_type="vadu", rank=-999):
        """cg_preconditioner_type of the iterative methods: "vadu", "pivoted_cholesky" with `rank` columns, "fitc" with `rank` inducing points, or "vecchia_response"
        (evaluation only) -- gpb_hip_vecchia_laplace_set_preconditioner."""
        t = {"vadu": 0, "pivoted_cholesky": 1, "fitc": 2, "vecchia_response": 3, "vifdu": 4, "none": 5}[cg_preconditioner_type]      # (vifdu / none: full-scale Vecchia handles only)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_preconditioner(self.h, C.c_int(t), C.c_int(int(rank))))

    def vif_set_inducing_points(self, ip):
        """Inducing points (k x d) of a full-scale Vecchia ("VIF") model (gpb_hip_vecchia_vif_set_inducing_points): the handle's Laplace evaluations then use
        Sigma = C Sigma_m^-1 C' + the Vecchia approximation of the residual process."""
        ipf = np.asfortranarray(ip, dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_vif_set_inducing_points(self.h, C.c_int(ipf.shape[0]), _p(ipf)))

    def laplace_set_inducing_points(self, ip):
        """Inducing points (k x d) of the "fitc" preconditioner (gpb_hip_vecchia_laplace_set_inducing_points)."""
        ipf = np.asfortranarray(ip, dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_inducing_points(self.h, C.c_int(ipf.shape[0]), _p(ipf)))

    def laplace_set_response_real(self, y):
        """gamma: the real-valued response (> 0), in the order laplace_set_labels takes its labels."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_response_real(self.h, _p(y)))

    def set_sorted_gather(self, mode):
        """-1: default (n >= 32768, from the third evaluation), 0: never, 1: always -- the spatially sorted copy of the records the neighbour gathers read."""
        _shim_call(_lib().gpb_hip_vecchia_set_sorted_gather(self.h, C.c_int(int(mode))))

    def laplace_set_weights(self, w):
        """Sample weights of the non-Gaussian likelihood, in the order of the labels (gpb_hip_vecchia_laplace_set_weights); None removes them."""
        ww = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_weights(self.h, _p(ww)))

    def laplace_set_binomial(self, on=True):
        """binomial_logit / binomial_probit: the binomial normalising constant over (proportions, trials = sample weights) (gpb_hip_vecchia_laplace_set_binomial)."""
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_binomial(self.h, C.c_int(int(bool(on)))))

    def laplace_set_aux(self, aux):
        """gamma / negative_binomial: the shape parameter (gpb_hip_vecchia_laplace_set_aux_pars)."""
        a = np.ascontiguousarray(np.atleast_1d(aux), dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_aux_pars(self.h, _p(a), C.c_int(a.size)))

    def laplace_grad_aux(self):
        """-> {gradient wrt log(shape), CalcGradNegLogLikAuxPars part, log-determinant part, implicit part} at the state of the last laplace_eval_grad."""
        o = np.zeros(8)          # 4 per auxiliary parameter (t has two: scale, df)
        _shim_call(_lib().gpb_hip_vecchia_laplace_grad_aux_current(self.h, _p(o)))
        return o if getattr(self, "_lap_link", 0) == 6 else o[:4]

    def laplace_set_fixed_effects(self, fixed_effects):
        """Offset of the location parameter, Vecchia order (None removes it)."""
        fe = None if fixed_effects is None else np.ascontiguousarray(fixed_effects, dtype=np.float64)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_fixed_effects(self.h, _p(fe)))

    def laplace_set_labels(self, y01):
        y01 = np.ascontiguousarray(y01, dtype=np.int32)
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_labels(self.h, _p(y01, C.c_int)))

    def laplace_logit(self, cov_type, var, a, num_rand_vec=50, seed_rand_vec=1, cg_max_num_it=1000, cg_max_num_it_tridiag=1000,
                      cg_delta_conv=1e-2, delta_conv_mode_finding=1e-8, reset_mode=True, want_mode=False):
        """-> (negll, info): Vecchia-Laplace approximation, Bernoulli-logit, iterative methods (gpb_hip_vecchia_laplace_logit)."""
        o = np.empty(9)
        mode = np.empty(self.n) if want_mode else None
        _shim_call(_lib().gpb_hip_vecchia_laplace_logit(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), C.c_int(num_rand_vec),
                                                        C.c_int(seed_rand_vec), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag),
                                                        C.c_double(cg_delta_conv), C.c_double(delta_conv_mode_finding),
                                                        C.c_int(1 if reset_mode else 0), _p(o), _p(mode)))
        return -o[0], dict(newton_it=int(o[1]), cg_it=int(o[2]), log_det=o[3], lanczos_it=int(o[4]), mll_no_det=o[5],
                           ms_factor=o[6], ms_mode=o[7], ms_logdet=o[8], mode=mode)

    def laplace_eval_grad(self, cov_type, var, a, num_rand_vec=50, seed_rand_vec=1, cg_max_num_it=1000, cg_max_num_it_tridiag=1000,
                          cg_delta_conv=1e-2, delta_conv_mode_finding=1e-8, reset_mode=True, want_parts=False):
        """-> (negll, grad wrt (log var, log a)[, parts]): Vecchia-Laplace approximation and its gradient on the device
        (gpb_hip_vecchia_laplace_eval + gpb_hip_vecchia_laplace_grad_current)."""
        o = np.empty(9)
        _shim_call(_lib().gpb_hip_vecchia_laplace_eval(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), C.c_int(num_rand_vec),
                                                       C.c_int(seed_rand_vec), C.c_int(cg_max_num_it), C.c_int(cg_max_num_it_tridiag),
                                                       C.c_double(cg_delta_conv), C.c_double(delta_conv_mode_finding),
                                                       C.c_int(1 if reset_mode else 0), C.c_int(1), _p(o), None))
        g = np.empty(2)
        parts = np.empty(8) if want_parts else None
        vecs = np.empty(2 * self.n) if want_parts else None
        _shim_call(_lib().gpb_hip_vecchia_laplace_grad_current(self.h, C.c_int(cg_max_num_it), C.c_double(cg_delta_conv), _p(g), _p(parts), _p(vecs)))
        if getattr(self, "_lap_link", 0) == 6:      # t: d / d (log scale, log df)
            ga = self.laplace_grad_aux()
            g = np.array([g[0], g[1], ga[0], ga[4]])
        elif getattr(self, "_lap_link", 0) >= 3:      # likelihoods with an auxiliary parameter: third entry = d / d log(aux)
            g = np.array([g[0], g[1], self.laplace_grad_aux()[0]])
        if want_parts:
            return -o[0], g, dict(per_par=parts.reshape(2, 4), dlogdet_dmode=vecs[:self.n], implicit_solve=vecs[self.n:])
        return -o[0], g

    def laplace_set_data_map(self, re_ptr):
        """Repeated locations: the state's n points are the unique locations, re_ptr (n + 1) the CSR of their data; labels / fixed effects are
        then handed over grouped by random effect (gpb_hip_vecchia_laplace_set_data_map).  None: one datum per random effect."""
        rp = None if re_ptr is None else np.ascontiguousarray(re_ptr, dtype=np.int32)
        self._n_data = self.n if rp is None else int(rp[-1])
        _shim_call(_lib().gpb_hip_vecchia_laplace_set_data_map(self.h, _p(rp, C.c_int)))

    def laplace_grad_F(self):
        """Boosting gradient d(-mll)/dF at the state of the last laplace_eval_grad (gpb_hip_vecchia_laplace_grad_F_current): Vecchia order, or --
        with a data map -- per datum in the grouped order of the labels."""
        out = np.empty(getattr(self, "_n_data", self.n))
        _shim_call(_lib().gpb_hip_vecchia_laplace_grad_F_current(self.h, _p(out)))
        return out

    def laplace_reset_mode_to_previous(self):
        _shim_call(_lib().gpb_hip_vecchia_laplace_reset_mode_to_previous(self.h))

    def laplace_range_deriv(self, cov_type, var, a):
        """-> (dA / d log a [n, m], dD / d log a [n]) of the factor without nugget (test seam, gpb_hip_vecchia_laplace_range_deriv)."""
        dA = np.empty((self.n, self.m)); dD = np.empty(self.n)
        _shim_call(_lib().gpb_hip_vecchia_laplace_range_deriv(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), _p(dA), _p(dD)))
        return dA, dD

    def yaux_partial_dev(self, w_dev_ptr):
        """This shard's contribution to y_aux as a full n-vector on the device (sum over ranks = y_aux)."""
        _shim_call(_lib().gpb_hip_vecchia_yaux_partial_dev(self.h, C.c_void_p(int(w_dev_ptr))))


class DeviceBuffer(object):
    """n doubles of device memory (gpb_hip_dev_alloc); .ptr is what the *_dev entry points take."""

    def __init__(self, n):
        self.n = int(n)
        self.p = C.c_void_p()
        _shim_call(_lib().gpb_hip_dev_alloc(C.c_uint64(8 * self.n), C.byref(self.p)))

    @property
    def ptr(self):
        return self.p.value

    def to_host(self):
        out = np.empty(self.n)
        _shim_call(_lib().gpb_hip_dev_to_host(_p(out), self.p, C.c_uint64(8 * self.n)))
        return out

    def __del__(self):
        try:
            if self.p.value:
                _lib().gpb_hip_dev_free(self.p)
                self.p = C.c_void_p()
        except Exception:
            pass


class ExactState(object):
    """Exact (dense) GP: coords (n, d) in data order."""

    def __init__(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords.reshape(-1, 1)
        self.n, self.d = coords.shape
        cm = np.asfortranarray(coords)
        self.h = C.c_void_p()
        _shim_call(_lib().gpb_hip_exact_create(C.c_int(self.n), C.c_int(self.d), _p(cm), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value is not None:
            _lib().gpb_hip_exact_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_y(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.shape == (self.n,)
        _shim_call(_lib().gpb_hip_exact_set_y(self.h, _p(y)))

    def nll_terms(self, cov_type, var, a, want_yaux=False):
        """-> (array {y^T Psi^-1 y, log|Psi|}, yaux or None, ms {assembly, factorisation, solves})"""
        out = np.empty(2); ms = np.empty(3)
        ya = np.empty(self.n) if want_yaux else None
        _shim_call(_lib().gpb_hip_exact_nll_terms(self.h, C.c_int(cov_type), C.c_double(var), C.c_double(a), _p(out),
                                                  _p(ya), _p(ms)))
        return out, ya, ms


class LocalGroup(object):
    """In-process group of `world` ranks (threads of this process; their handles may share one device): the second transport behind the
    library's collectives (gpb_hip_local_group_create).  `run(fn)` calls fn(rank) on one thread per rank and returns the results in rank
    order; an exception on any rank aborts the group so that no peer waits in a barrier for ever."""

    def __init__(self, world):
        self.world = int(world)
        self.g = C.c_void_p()
        _shim_call(_lib().gpb_hip_local_group_create(C.c_int(self.world), C.byref(self.g)))

    def close(self):
        if getattr(self, "g", None) is not None and self.g.value is not None:
            _lib().gpb_hip_local_group_free(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def abort(self):
        _lib().gpb_hip_local_group_abort(self.g)

    def run(self, fn):
        import threading
        res, err = [None] * self.world, [None] * self.world

        def work(r):
            try:
                res[r] = fn(r)
            except BaseException as e:        # noqa: BLE001 -- re-raised on the calling thread
                err[r] = e
                self.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return res


def mailbox_create(world):
    """rank 0: create the node-local mailbox segment for `world` ranks -> its name (bytes; hand it to every rank's mailbox_attach)"""
    buf = C.create_string_buffer(64)
    _shim_call(_lib().gpb_hip_mailbox_create(C.c_int(int(world)), buf))
    return buf.value


def comm_unique_id():
    """128-byte ncclUniqueId (call on rank 0, broadcast to the other ranks)."""
    buf = (C.c_ubyte * 128)()
    _shim_call(_lib().gpb_hip_comm_get_unique_id(buf))
    return bytes(buf)


def nll_from_terms(n, yPy, logdet, sigma2):
    """include/GPBoost/re_model_template.h:3132"""
    return yPy / 2. / sigma2 + logdet / 2. + n / 2. * (np.log(sigma2) + np.log(2 * np.pi))


def grad_from_terms(n, t7, sigma2):
    """include/GPBoost/re_model_template.h:1994,2004 -> d nll / d log(sigma2, var, a)"""
    return np.array([-t7[0] / sigma2 / 2. + n / 2., t7[3] / sigma2 + t7[4], t7[5] / sigma2 + t7[6]])


class HistBuilder(object):
    """bins: (F, n) uint8 feature-major; bin_offsets: F+1 prefix sums."""

    def __init__(self, bins, bin_offsets):
        bins = np.ascontiguousarray(bins, dtype=np.uint8)
        self.F, self.n = bins.shape
        self.bin_offsets = np.ascontiguousarray(bin_offsets, dtype=np.int32)
        self.total_bins = int(self.bin_offsets[-1])
        self.h = C.c_void_p()
        _shim_call(_lib().gpb_hip_hist_create(C.c_int(self.n), C.c_int(self.F), _p(bins, C.c_uint8),
                                              _p(self.bin_offsets, C.c_int32), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value is not None:
            _lib().gpb_hip_hist_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_gradients(self, grad, hess=None):
        grad = np.ascontiguousarray(grad, dtype=np.float64)
        hess = None if hess is None else np.ascontiguousarray(hess, dtype=np.float64)
        _shim_call(_lib().gpb_hip_hist_set_gradients(self.h, _p(grad), _p(hess)))

    def bench(self, data_indices=None, const_hess=1.0, reps=10):
        """mean ms of one leaf build (kernels only, HIP events)"""
        di = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
        nd = self.n if di is None else di.size
        ms = C.c_double(0)
        _shim_call(_lib().gpb_hip_hist_bench(self.h, _p(di, C.c_int32), C.c_int(nd), C.c_double(const_hess), C.c_int(reps),
                                             C.byref(ms)))
        return ms.value

    def comm_init(self, id128, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(id128))
        _shim_call(_lib().gpb_hip_hist_comm_init(self.h, buf, C.c_int(int(rank)), C.c_int(int(world))))

    def comm_init_local(self, group, rank):
        _shim_call(_lib().gpb_hip_hist_comm_init_local(self.h, group.g, C.c_int(int(rank))))

    def build_allreduce(self, data_indices=None, const_hess=1.0):
        """Local leaf histogram of this rank's rows + all-reduce over the ranks -> (hist (total_bins, 2), cnt)."""
        idx = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
        nd = self.n if idx is None else idx.size
        hist = np.empty((self.total_bins, 2)); cnt = np.empty(self.total_bins, dtype=np.uint64)
        _shim_call(_lib().gpb_hip_hist_build_allreduce(self.h, _p(idx, C.c_int), C.c_int(nd), C.c_double(const_hess), _p(hist),
                                                       _p(cnt, C.c_uint64)))
        return hist, cnt

    # ---- resident leaf histograms (row a12) ----
    def pool_resize(self, num_slots):
        _shim_call(_lib().gpb_hip_hist_pool_resize(self.h, C.c_int(int(num_slots))))

    def set_fix_info(self, view_offset, num_bin, most_freq_bin):
        vo = np.ascontiguousarray(view_offset, dtype=np.int32); nb = np.ascontiguousarray(num_bin, dtype=np.int32)
        mf = np.ascontiguousarray(most_freq_bin, dtype=np.int32)
        _shim_call(_lib().gpb_hip_hist_set_fix_info(self.h, _p(vo, C.c_int), _p(nb, C.c_int), _p(mf, C.c_int)))

    def build_slot(self, slot, data_indices=None, const_hess=1.0):
        idx = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
        nd = self.n if idx is None else idx.size
        _shim_call(_lib().gpb_hip_hist_build_slot(self.h, C.c_int(int(slot)), _p(idx, C.c_int), C.c_int(nd), C.c_double(const_hess)))

    def fix_slot(self, slot, sum_gradient, sum_hessian):
        _shim_call(_lib().gpb_hip_hist_fix_slot(self.h, C.c_int(int(slot)), C.c_double(sum_gradient), C.c_double(sum_hessian)))

    def subtract_slots(self, parent, smaller, out):
        _shim_call(_lib().gpb_hip_hist_subtract_slots(self.h, C.c_int(int(parent)), C.c_int(int(smaller)), C.c_int(int(out))))

    def set_split_info(self, offset, default_bin, missing_type):
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (offset, default_bin, missing_type)]
        _shim_call(_lib().gpb_hip_hist_set_split_info(self.h, *[_p(x, C.c_int) for x in a]))

    def set_categorical(self, is_categorical, max_cat_to_onehot=4, max_cat_threshold=32, cat_smooth=10.0, cat_l2=10.0, min_data_per_group=100):
        """Categorical features: flags per feature (None: all numerical) and the reference's Config values of the same names
        (FindBestThresholdCategoricalInner, feature_histogram.hpp:278-519); they stay set for find_best_split and grow_tree."""
        m = None if is_categorical is None else np.ascontiguousarray(np.asarray(is_categorical) != 0, dtype=np.int8)
        if m is not None and m.size != self.F:
            raise ValueError("is_categorical must have one flag per feature")
        self.is_categorical = np.zeros(self.F, dtype=np.int8) if m is None else m
        _shim_call(_lib().gpb_hip_hist_set_categorical(self.h, _p(m, C.c_int8), C.c_int(int(max_cat_to_onehot)), C.c_int(int(max_cat_threshold)),
                                                       C.c_double(cat_smooth), C.c_double(cat_l2), C.c_int(int(min_data_per_group))))

    def set_feature_block_exchange(self, on=True):
        """Data-parallel tree grower: reduce-scatter by feature block + exchange of the ranks' best splits (default) or all-reduce of every histogram."""
        _shim_call(_lib().gpb_hip_hist_set_feature_block_exchange(self.h, C.c_int(-1 if on is None else int(bool(on)))))

    def set_regularisation(self, lambda_l1=0.0, max_delta_step=0.0, path_smooth=0.0, parent_output=0.0):
        """lambda_l1 / max_delta_step / path_smooth of the split search (they stay set for find_best_split and grow_tree); parent_output: the
        leaf's own output, for the following find_best_split calls (path smoothing; grow_tree tracks it itself)."""
        _shim_call(_lib().gpb_hip_hist_set_regularisation(self.h, C.c_double(lambda_l1), C.c_double(max_delta_step), C.c_double(path_smooth),
                                                          C.c_double(parent_output)))

    def set_root_rows(self, rows=None):
        """Bagging: the (ascending) rows the root of the following trees holds; None = all rows."""
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        _shim_call(_lib().gpb_hip_hist_set_root_rows(self.h, _p(r, C.c_int), C.c_int(0 if r is None else r.size)))

    def set_feature_mask(self, is_feature_used=None):
        """The columns grow_tree may split on (feature_fraction's per-tree sample); None = all."""
        m = None if is_feature_used is None else np.ascontiguousarray(is_feature_used, dtype=np.int8)
        if m is not None and m.size != self.F:
            raise ValueError("is_feature_used must have one flag per feature")
        _shim_call(_lib().gpb_hip_hist_set_feature_mask(self.h, _p(m, C.c_int8)))

    def set_max_depth(self, max_depth):
        """config max_depth of grow_tree (<= 0: no limit)."""
        _shim_call(_lib().gpb_hip_hist_set_max_depth(self.h, C.c_int(int(max_depth))))

    def find_best_split(self, slot, sum_gradient, sum_hessian, num_data, lambda_l2=0.0, min_data_in_leaf=20,
                        min_sum_hessian_in_leaf=1e-3, min_gain_to_split=0.0, is_feature_used=None):
        """-> (best_feature, out (F, 10), default_left (F,)): FeatureHistogram::FindBestThreshold per feature + the winner;
        self.last_splittable holds is_splittable() per feature."""
        out = np.empty((self.F, 10)); dl = np.empty(self.F, dtype=np.int32); best = C.c_int(-1)
        self.last_splittable = np.empty(self.F, dtype=np.int32)
        used = None if is_feature_used is None else np.ascontiguousarray(is_feature_used, dtype=np.int8)
        _shim_call(_lib().gpb_hip_hist_find_best_split(self.h, C.c_int(int(slot)), C.c_double(sum_gradient), C.c_double(sum_hessian),
                                                       C.c_int(int(num_data)), C.c_double(lambda_l2), C.c_int(int(min_data_in_leaf)),
                                                       C.c_double(min_sum_hessian_in_leaf), C.c_double(min_gain_to_split),
                                                       _p(used, C.c_int8), C.byref(best), _p(out), _p(dl, C.c_int),
                                                       _p(self.last_splittable, C.c_int)))
        # categorical features: the sets of bins going left (8 words per feature; zeros for numerical features)
        self.last_cat_bits = np.zeros((self.F, 8), dtype=np.uint32)
        _shim_call(_lib().gpb_hip_hist_last_split_cat_bits(self.h, _p(self.last_cat_bits, C.c_uint32)))
        return best.value, out, dl

    def split_leaf(self, data_indices, feature, threshold, default_left, cat_bits=None):
        """-> (lte_indices, gt_indices), both in the order of data_indices (None = all rows).  cat_bits (8 words): a categorical split, the rows whose
        bin is in the set go left."""
        idx = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
        cnt = self.n if idx is None else idx.size
        lte = np.empty(cnt, dtype=np.int32); gt = np.empty(cnt, dtype=np.int32); nl = C.c_int(0)
        if cat_bits is not None:
            w = np.ascontiguousarray(cat_bits, dtype=np.uint32)
            _shim_call(_lib().gpb_hip_hist_split_leaf_categorical(self.h, _p(idx, C.c_int), C.c_int(cnt), C.c_int(int(feature)), _p(w, C.c_uint32),
                                                                  _p(lte, C.c_int), _p(gt, C.c_int), C.byref(nl)))
            return lte[:nl.value].copy(), gt[:cnt - nl.value].copy()
        _shim_call(_lib().gpb_hip_hist_split_leaf(self.h, _p(idx, C.c_int), C.c_int(cnt), C.c_int(int(feature)), C.c_uint(int(threshold)),
                                                  C.c_int(int(bool(default_left))), _p(lte, C.c_int), _p(gt, C.c_int), C.byref(nl)))
        return lte[:nl.value].copy(), gt[:cnt - nl.value].copy()

    def grow_tree(self, num_leaves, sum_gradient, sum_hessian, lambda_l2=0.0, min_data_in_leaf=20, min_sum_hessian_in_leaf=1e-3,
                  min_gain_to_split=0.0, const_hess=1.0, want_leaf_index=True):
        """One leaf-wise tree with device-resident row lists (gpb_hip_hist_grow_tree).  Returns the arrays of tests/tree_harness.grow_tree
        plus 'data_leaf_index'."""
        L = int(num_leaves)
        nl = C.c_int(0)
        ia = {k: np.zeros(L, dtype=np.int32) for k in ("split_feature_inner", "default_left", "left_child", "right_child", "internal_count", "leaf_count")}
        thr = np.zeros(L, dtype=np.uint32); gain = np.zeros(L); lv = np.zeros(L)
        dli = np.empty(self.n, dtype=np.int32) if want_leaf_index else None
        _shim_call(_lib().gpb_hip_hist_grow_tree(
            self.h, C.c_int(L), C.c_double(sum_gradient), C.c_double(sum_hessian), C.c_double(lambda_l2), C.c_int(int(min_data_in_leaf)),
            C.c_double(min_sum_hessian_in_leaf), C.c_double(min_gain_to_split), C.c_double(const_hess), C.byref(nl),
            _p(ia["split_feature_inner"], C.c_int), _p(thr, C.c_uint32), _p(ia["default_left"], C.c_int), _p(ia["left_child"], C.c_int),
            _p(ia["right_child"], C.c_int), _p(gain), _p(ia["internal_count"], C.c_int), _p(lv), _p(ia["leaf_count"], C.c_int),
            None if dli is None else _p(dli, C.c_int)))
        k = nl.value
        out = dict(num_leaves=k, threshold_in_bin=thr[:k - 1].astype(np.int64), split_gain=gain[:k - 1].copy(), leaf_value=lv[:k].copy(),
                   leaf_count=ia["leaf_count"][:k].copy(), data_leaf_index=dli)
        for key in ("split_feature_inner", "default_left", "left_child", "right_child", "internal_count"):
            out[key] = ia[key][:k - 1].copy()
        # categorical nodes: flags and the sets of bins going left (threshold_in_bin of such a node = its index among them, as in the reference's Tree)
        nic = np.zeros(max(k - 1, 1), dtype=np.int32); ncb = np.zeros((max(k - 1, 1), 8), dtype=np.uint32)
        if k > 1:
            _shim_call(_lib().gpb_hip_hist_last_tree_cat_nodes(self.h, C.c_int(k - 1), _p(nic, C.c_int), _p(ncb, C.c_uint32)))
        out["node_is_cat"] = nic[:k - 1].copy(); out["node_cat_bits"] = ncb[:k - 1].copy()
        return out

    def get_slot(self, slot):
        out = np.empty((self.total_bins, 2))
        _shim_call(_lib().gpb_hip_hist_get_slot(self.h, C.c_int(int(slot)), _p(out)))
        return out

    def build(self, data_indices=None, const_hess=1.0):
        """-> (hist[total_bins, 2] = {grad sum, hess sum}, cnt[total_bins] uint64)"""
        di = None if data_indices is None else np.ascontiguousarray(data_indices, dtype=np.int32)
        nd = self.n if di is None else di.size
        hist = np.empty((self.total_bins, 2)); cnt = np.empty(self.total_bins, dtype=np.uint64)
        _shim_call(_lib().gpb_hip_hist_build(self.h, _p(di, C.c_int32), C.c_int(nd), C.c_double(const_hess), _p(hist),
                                             _p(cnt, C.c_uint64)))
        return hist, cnt
