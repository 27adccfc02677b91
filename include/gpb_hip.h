/*
 * include/gpb_hip.h -- C ABI of the MI355X (gfx950) GP + histogram hot path.
 *
 * This is the `extern "C"` shim the reference's host C++ calls instead of running its
 * per-point OpenMP loops; plain pointers and sizes only, no C++/torch types.  Each
 * entry point cites the reference interface it replaces (paths relative to the
 * fabsig/GPBoost v1.7.3 tree).  INTEGRATION.md shows the call sites a maintainer patches.
 *
 * Conventions (same as the reference's C API, include/LightGBM/c_api.h:1837-1849):
 *   - every function returns 0 on success, -1 on failure; the message is kept in a
 *     thread-local buffer read with gpb_hip_get_last_error();
 *   - never aborts/exits; any HIP error becomes a -1 (the reference host then raises
 *     through Log::REFatal, include/LightGBM/utils/log.h:149,190);
 *   - one host thread per handle (REModel has no locking); several handles
 *     may be alive; a handle is bound to the HIP device current at creation time;
 *   - host input arrays are borrowed for the duration of the call only;
 *   - "dev" variants take device pointers valid on the handle's device and are enqueued
 *     on the handle's stream without synchronising (use gpb_hip_vecchia_sync()).
 *
 * The library needs a gfx950 device: gpb_hip_device_count() == 0 makes every *_create
 * fail loudly (there is no CPU fallback in this library by design).
 */
#ifndef GPB_HIP_H_
#define GPB_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPB_HIP_EXPORT __attribute__((visibility("default")))

/* covariance functions of the hot path: include/GPBoost/cov_fcts.h:2100-2118 */
enum { GPB_HIP_COV_MATERN_0_5 = 0 /* == "exponential" */, GPB_HIP_COV_MATERN_1_5 = 1, GPB_HIP_COV_MATERN_2_5 = 2 };

/* number of doubles gpb_hip_vecchia_grad_terms writes */
#define GPB_HIP_NUM_TERMS 7

typedef struct gpb_hip_vecchia gpb_hip_vecchia_t;
typedef struct gpb_hip_hist gpb_hip_hist_t;
typedef struct gpb_hip_exact gpb_hip_exact_t;
typedef struct gpb_hip_local_group gpb_hip_local_group_t;   /* in-process group of ranks (threads), see gpb_hip_local_group_create */

GPB_HIP_EXPORT const char* gpb_hip_get_last_error(void);

/* Diagnostics.  With GPB_HIP_API_TIMING=1 in the environment every entry point of this header accumulates its wall time and call count (inclusive);
 * the table is written to stderr at process exit and by this call (reset != 0 clears it).  It plays the part of the reference's TIMETAG build
 * (include/LightGBM/utils/common.h:989-1068) for an integration: time inside this library against time in the caller's host code.  -1 if the
 * variable is not set. */
GPB_HIP_EXPORT int gpb_hip_api_timing_report(int reset);
/* GPB_HIP_API_TIMING=2 additionally records the TIMELINE of the calls (name, duration, the caller's time since the previous call returned) and prints the last 400 entries
 * with the table.  gpb_hip_api_mark puts a named entry of zero duration on that timeline from the caller's side -- an integration brackets its own host passes with it
 * (`name` must outlive the report: a string literal).  A no-op unless the timeline is on. */
GPB_HIP_EXPORT int gpb_hip_api_mark(const char* name);
GPB_HIP_EXPORT int gpb_hip_device_count(int* count);
/* Make `device` current for the calling thread (handles bind to the device current at their creation). */
GPB_HIP_EXPORT int gpb_hip_set_device(int device);
/* 1 if the current device is gfx950 */
GPB_HIP_EXPORT int gpb_hip_device_is_gfx950(int* yes);
/* Plain device-buffer helpers for hosts that have no allocator of their own (the "_dev" entry points take such
 * pointers; a torch/RCCL host passes tensor.data_ptr() instead). gpb_hip_dev_to_host synchronises the device first. */
GPB_HIP_EXPORT int gpb_hip_dev_alloc(uint64_t bytes, void** out);
GPB_HIP_EXPORT int gpb_hip_dev_free(void* p);
GPB_HIP_EXPORT int gpb_hip_dev_to_host(void* dst_host, const void* src_dev, uint64_t bytes);
/* Runs the fp64-DPP primitives against their compiler-scheduled equivalents on the device. */
GPB_HIP_EXPORT int gpb_hip_selftest(void);

/* ------------------------------------------------------------------------------------
 * Vecchia state: replaces what CreateREComponentsVecchia builds on the host
 * (src/GPBoost/Vecchia_utils.cpp:1095-1289): coordinates in Vecchia order + neighbour table.
 *   coords_colmajor  n x d, column-major (as RECompGP::coords_, include/GPBoost/re_comp.h:837),
 *                    ALREADY in Vecchia order; d in {1,2,3}
 *   num_neighbors    m <= 62
 * ---------------------------------------------------------------------------------- */
GPB_HIP_EXPORT int gpb_hip_vecchia_create(int32_t n, int32_t d, int32_t num_neighbors, const double* coords_colmajor,
                                          gpb_hip_vecchia_t** out);
GPB_HIP_EXPORT int gpb_hip_vecchia_free(gpb_hip_vecchia_t* h);
GPB_HIP_EXPORT int gpb_hip_vecchia_sync(gpb_hip_vecchia_t* h);
/* Run this handle's work on a caller-owned hipStream_t (e.g. the stream an RCCL all-reduce of the partial
 * terms is enqueued on), instead of the handle's private stream.  NULL = the legacy default stream. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_stream(gpb_hip_vecchia_t* h, void* hip_stream);

/* Ordered nearest-neighbour search on the device, bit-identical to
 * find_nearest_neighbors_Vecchia_fast(neighbor_selection = "nearest", start_at = 0, end_search_at = -1)
 * (src/GPBoost/Vecchia_utils.cpp:733-985, inner loop :1029-1093).  The coordinate-sum argsort is done
 * on the host with std::sort and the reference's comparator (include/GPBoost/utils.h:230-238) because
 * its tie order is libstdc++-specific.  has_duplicates (may be NULL) receives the flag of :812,:905. */
GPB_HIP_EXPORT int gpb_hip_vecchia_find_neighbors(gpb_hip_vecchia_t* h, int* has_duplicates);
/* Alternatively hand over a neighbour table computed elsewhere: n x m int32, row-major, -1 padded. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_neighbors(gpb_hip_vecchia_t* h, const int32_t* nn);
/* Spatially sorted gather (round 5): the m neighbour records a point reads (32 bytes each) lie at random positions of the Vecchia ordering -- one 64-byte sector per
 * record from L2 / Infinity Cache / HBM; the neighbours are close in SPACE, so in a copy of the records sorted along a Morton curve they share sectors.  The library keeps
 * that copy behind the records and a neighbour table that points into it; own records, outputs and the arithmetic are unchanged, bit for bit.  mode: -1 (default) for
 * n >= 32768 from the third evaluation on a neighbour table (the host sort costs ~0.1 s per million points), 0 never, 1 always.  Not used with sample weights.
 * GPB_VECCHIA_SORTED_GATHER=0 in the environment switches the default off. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_sorted_gather(gpb_hip_vecchia_t* h, int mode);
GPB_HIP_EXPORT int gpb_hip_vecchia_get_neighbors(gpb_hip_vecchia_t* h, int32_t* nn);

/* Multi-GPU: this handle evaluates points [i_begin, i_end) of the ordering only (default: all).
 * Coordinates, y and the neighbour table stay replicated. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_shard(gpb_hip_vecchia_t* h, int32_t i_begin, int32_t i_end);

/* Page-locked host memory for staging buffers that are uploaded every call (the C API host keeps y in Vecchia order there: an
 * 8 MB upload from pinned memory takes about half the time of one from pageable memory). */
GPB_HIP_EXPORT int gpb_hip_pinned_alloc(size_t bytes, void** out);
GPB_HIP_EXPORT int gpb_hip_pinned_free(void* p);
/* y in Vecchia order (REModelTemplate::SetY, include/GPBoost/re_model_template.h:6185-6222).  A factor computed by gpb_hip_vecchia_factor stays
 * valid (A, D do not depend on y); u = B y is renewed for the new response at its next use (y_aux, get_factor) -- the GPBoost algorithm sets a new
 * response every boosting iteration at unchanged parameters (CalcGradientF, :3313-3316).  The full-scale (VIF) factor carries the response: it goes.
 * Rounding: the renewed u is y_i - sum_j A_ij y_nn(i,j) accumulated by fma over the STORED A (vecchia_By_pts_kernel), the factor kernel's u comes out of
 * its elimination: equal to ~1e-16 relative, not bit for bit -- y_aux after set_y is not bit-identical to factor-then-y_aux at the same response. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_y(gpb_hip_vecchia_t* h, const double* y_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_set_y_dev(gpb_hip_vecchia_t* h, const double* y_dev);

/* Fused factor + likelihood terms: replaces CalcCovFactorGradientVecchia(calc_cov_factor = true,
 * calc_gradient = false) (src/GPBoost/Vecchia_utils.cpp:1367-1699) followed by CalcYTPsiIInvY
 * (include/GPBoost/re_model_template.h:9960-9968) and the log-determinant (:2946-2948).
 *   var = sigma1^2 / sigma^2, a = transformed range (re_model.cpp:768-778, cov_fcts.h:500-516)
 *   gauss_likelihood: 1 -> nugget 1 on the diagonal (:1601), 0 -> diag *= 1 + 1e-10 (:1608)
 *   out3 = { y^T Psi^-1 y, log|Psi|, #points with D_i <= 0 (:1685-1698) }  (sums over this handle's shard) */
GPB_HIP_EXPORT int gpb_hip_vecchia_nll_terms(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                             int gauss_likelihood, double* out3_host);
/* Enqueue only; out3_dev is a device pointer (3 doubles). */
/* Linear-regression covariates, Gaussian likelihood (GPB_OptimLinRegrCoefCovPar with the default optimizer_coef "wls": the coefficients are profiled
 * out by generalised least squares at every evaluation, optim_utils.h:296-302 -> ProfileOutCoef / UpdateCoefGLS, re_model_template.h:2665-2683, 10012-10019).
 *   set_covariates  X: p columns of n values in VECCHIA order, column-major [p][n]; resident until replaced (p = 0 removes them)
 *   gram            after gpb_hip_vecchia_factor: G = (B [X, y0])' D^-1 (B [X, y0]), (p+1) x (p+1) row-major, y0 = the response last uploaded with
 *                   gpb_hip_vecchia_set_y: X' Psi^-1 X (CalcXTPsiInvX, :6624-6628), X' Psi^-1 y0 and y0' Psi^-1 y0 in one pass
 *   set_resid       response := y0 - X beta (UpdateFixedEffects, :2859-2871); beta = NULL restores y0 */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_covariates(gpb_hip_vecchia_t* h, int32_t p, const double* X_colmajor);
GPB_HIP_EXPORT int gpb_hip_vecchia_gram(gpb_hip_vecchia_t* h, double* G_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_set_resid(gpb_hip_vecchia_t* h, const double* beta_host);

/* K evaluations with one synchronisation (and, on a sharded handle, ONE ncclAllReduce of 3 K doubles): the trial points of a line search
 * or any batch of parameter sets.  var[k], a[k]: transformed parameters of evaluation k; out: K x {y' Psi^-1 y, log|Psi|, #(D <= 0)} job-wide. */
GPB_HIP_EXPORT int gpb_hip_vecchia_nll_terms_batch(gpb_hip_vecchia_t* h, int cov_type, int32_t K, const double* var, const double* a,
                                                   int gauss_likelihood, double* out3K_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_nll_terms_dev(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                                 int gauss_likelihood, double* out3_dev);

/* Same launch shape plus the covariance-parameter gradient pieces: replaces
 * CalcCovFactorGradientVecchia(calc_gradient = true) + the Vecchia branch of CalcGradPars
 * (include/GPBoost/re_model_template.h:1988-2011), Gaussian likelihood, transf_scale = true.
 *   out7 = { yPy, logdet, #bad, G1_var, G2_var, G1_range, G2_range } with
 *   d nll / d log(par_p) = G1_p / sigma2 + G2_p          (the :2004 expression, summed over the shard) */
GPB_HIP_EXPORT int gpb_hip_vecchia_grad_terms(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                              double* out7_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_grad_terms_dev(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                                  double* out7_dev);

/* Sample weights of a Gaussian model (include/GPBoost/re_model_template.h:403-431): observation i has error variance sigma^2 / w_i --
 * on the transformed scale the nugget 1 / w_i on its diagonal entries (GetGaussianNuggetDiagFromWeights, :6393-6417;
 * src/GPBoost/Vecchia_utils.cpp:1418-1422, 1610-1614, 1952-1958).  nug: n values 1 / w_i in Vecchia order (NULL: uniform nugget again).
 * Honoured by the likelihood, gradient, factor / y_aux and both prediction entry points; standard errors are not. */
GPB_HIP_EXPORT int gpb_hip_vecchia_set_nugget_diag(gpb_hip_vecchia_t* h, const double* nug_host);

/* Full-scale Vecchia ("VIF": Vecchia-inducing-points full-scale) approximation, Gaussian likelihood, Euclidean neighbours -- the
 * device part of CalcSigmaComps (include/GPBoost/re_model_template.h:8151-8200: cross-covariances, V = L_m^-1 C_mn), of the
 * full_scale_vecchia branches of CalcCovFactorGradientVecchia (src/GPBoost/Vecchia_utils.cpp:1463-1524, 1599-1656: the Vecchia factor
 * of the residual process and its derivatives), of CalcCovFactorFITC_FSA (re_model_template.h:9646-9745: B C_nm and (B C_nm)' D^-1 (B C_nm))
 * and of CalcGradPars_FITC_FSA_GaussLikelihood_Cluster_i (:2205-2330, 2447-2452).  The k x k matrices (Sigma_m, its Cholesky factor, the
 * Woodbury matrix, their inverses) stay on the host, k <= 256.
 *   set_inducing_points   ip: column-major k x d (the host's kmeans++, GP_utils.cpp:208-308)
 *   vif_factor            Linv: k x k row-major inverse of chol(Sigma_m with its diagonal x (1 + 1e-6)); with_grad: also the range derivative of
 *                         C_nm and B dC_nm (inputs of vif_grad_sums);
 *                         out3 = { sum u_i^2 / D_i, sum log D_i, #(D_i <= 0) } of the residual factor;
 *                         G = (B [C_nm, y])' D^-1 (B [C_nm, y]), (k + 1) x (k + 1) row-major; A / D / u / B C_nm stay on the device
 *   vif_grad_sums         after vif_factor(with_grad = 1) at the same parameters.  Winv = W^-1 (W = Sigma_m + (B C)' D^-1 (B C)), Si = Sigma_m^-1,
 *                         N0 = 2 Si - Si dSigma_m/dlog(var) Si, negMp1 = -Si dSigma_m/dlog(a) Si (derivatives of the UN-jittered Sigma_m as in the
 *                         reference), all k x k row-major; w = W^-1 (B C)' D^-1 B y.  sums12[2 s + p], p = 0 variance / 1 range, over the points i:
 *                           s = 0: dD_i / D_i                         s = 1: 2 (dB z)_i v_i - v_i^2 dD_i     (z = y - C w, v = D^-1 B z)
 *                           s = 2: (dB C)_i . Hm_i / D_i  (Hm = B C W^-1)   s = 3: dD_i (B C)_i . Hm_i / D_i^2
 *                           s = 4: (B dC)_i . Hm_i / D_i               s = 5: v_i (B dC)_i . w
 *                         from which  d(y' Psi^-1 y) = S2 - 2 S6 + w' dSigma_m w  and
 *                         d log|Psi| = S1 - tr(Si dSigma_m) + tr(Winv dSigma_m) + 2 S5 + 2 S3 - S4   (DESIGN.md 4.12).
 *   vif_get_grad_factor   dA_i (n x m) and dD_i (n) of parameter p of the last vif_grad_sums(keep_factor = 1): the reference's -B_grad / D_grad */
/* Lloyd iterations of kmeans_plusplus (src/GPBoost/GP_utils.cpp:237-308) after the host's random_plusplus start: assignment step on the device with the
 * reference's arithmetic, ordered mean update on the host -- the reference's means bit for bit.  x: column-major n x d; means: row-major k x d in / out. */
GPB_HIP_EXPORT int gpb_hip_kmeans_lloyd(int32_t n, int32_t d, const double* x_colmajor, int32_t k, double* means_rowmajor, int32_t max_it,
                                        int32_t* iterations);
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_set_inducing_points(gpb_hip_vecchia_t* h, int32_t k, const double* ip_colmajor);
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_factor(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Linv_rowmajor, int with_grad,
                                              double* out3_host, double* G_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_grad_sums(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Winv, const double* Si,
                                                 const double* N0, const double* negMp1, const double* w_host, int keep_factor, double* sums12_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_get_grad_factor(gpb_hip_vecchia_t* h, int p, double* dA_host, double* dD_host);

/* Route B seam switch: thread-local flag the reference's patched find_nearest_neighbors_Vecchia_fast consults -- set by
 * REModelTemplate's constructor around CreateREComponentsVecchia when GPU_use was requested, so that the ordered neighbour search of model creation
 * (src/GPBoost/Vecchia_utils.cpp:733-985: 24 s at n = 1e6 on 8 cores) runs on the device (gpb_hip_vecchia_find_neighbors: bit-identical tables). */
GPB_HIP_EXPORT int gpb_hip_route_b_set_device_search(int on);
GPB_HIP_EXPORT int gpb_hip_route_b_get_device_search(void);

/* In-loop timing of the dominant kernel: enable = 1 records a HIP event pair around every point-kernel launch of the handle (ring of 256 pairs);
 * enable = 0 stops, synchronises the stream and returns the launches seen and the mean kernel time (ms) of the last min(count, 256) launches. */
GPB_HIP_EXPORT int gpb_hip_vecchia_timing(gpb_hip_vecchia_t* h, int enable, int64_t* launches, double* mean_kernel_ms);

/* Node-local MAILBOX for the 3 / 7 sums of a sharded likelihood / gradient evaluation: a POSIX
 * shared-memory segment mapped and page-locked by every rank of the node.  Each rank's finisher workgroup stores its shard sums straight into its
 * slot (system-scope stores, no collective kernel, no copy), every host polls all slots and adds them in rank order: identical bits on all ranks.
 * With a mailbox attached, gpb_hip_vecchia_{nll,grad}_terms_allreduce use it instead of ncclAllReduce; RCCL stays for y_aux, histograms and the
 * neighbour table.  Every rank must make the same sequence of evaluations (SPMD), as with any collective.
 *   gpb_hip_mailbox_create          rank 0: creates the segment for `world` ranks, returns its name (64 bytes; hand it to the other ranks)
 *   gpb_hip_vecchia_mailbox_attach  every rank: maps it, waits for all ranks (120 s), rank 0 unlinks the name */
GPB_HIP_EXPORT int gpb_hip_mailbox_create(int world, char* name_out64);
GPB_HIP_EXPORT int gpb_hip_vecchia_mailbox_attach(gpb_hip_vecchia_t* h, const char* name, int rank, int world);
GPB_HIP_EXPORT int gpb_hip_vecchia_mailbox_info(gpb_hip_vecchia_t* h, int* rank, int* world);
GPB_HIP_EXPORT int gpb_hip_vecchia_mailbox_detach(gpb_hip_vecchia_t* h);

/* In-library RCCL reduction over the ranks of a node (one process per GPU; xGMI): the communicator is bootstrapped from a
 * 128-byte ncclUniqueId made on rank 0 and handed to every rank by the host (e.g. a torch.distributed broadcast).
 * The *_allreduce calls run point kernel -> fixed-order reduction -> ncclAllReduce(sum) of the 3 / 7 terms on the handle's
 * stream and deliver the job-wide terms to every rank's host -- one launch sequence, one collective, one sync per evaluation. */
GPB_HIP_EXPORT int gpb_hip_comm_get_unique_id(unsigned char* id128);
GPB_HIP_EXPORT int gpb_hip_vecchia_comm_init(gpb_hip_vecchia_t* h, const unsigned char* id128, int rank, int world);
/* Second transport behind the same collectives: an IN-PROCESS group whose ranks are threads of one process (their handles may share one
 * device).  Every all-reduce = publish the buffer, barrier, reduce all published buffers in rank order into scratch, barrier, copy back:
 * identical results on all ranks for every type.  It exists so that the sharded code paths -- neighbour-search parts, likelihood terms,
 * y_aux, data-parallel histograms and trees -- run with SEVERAL ranks on the single MI355X of a test box through the same host code and
 * kernels as under RCCL (tests/test_multirank_gpu.py); every rank must call the collective entry points from its own thread.
 * gpb_hip_local_group_abort wakes ranks waiting in a barrier (their call fails) after an error on another rank. */
GPB_HIP_EXPORT int gpb_hip_local_group_create(int world, gpb_hip_local_group_t** out);
GPB_HIP_EXPORT int gpb_hip_local_group_abort(gpb_hip_local_group_t* g);
GPB_HIP_EXPORT int gpb_hip_local_group_free(gpb_hip_local_group_t* g);
GPB_HIP_EXPORT int gpb_hip_vecchia_comm_init_local(gpb_hip_vecchia_t* h, gpb_hip_local_group_t* g, int rank);
/* rank / world of the handle's communicator; world = 0 while there is none.  GPB_OptimCovPar (gpboost_c_api_subset.h) evaluates
 * through the *_allreduce forms whenever a communicator exists, so a sharded fit is the same host loop on every rank. */
GPB_HIP_EXPORT int gpb_hip_vecchia_comm_info(gpb_hip_vecchia_t* h, int* rank, int* world);
/* Multi-GPU neighbour search: rank r searches the r-th of `nparts` equal blocks of query positions in
 * coordinate-sum order (balanced: that order is random with respect to the index that decides a query's cost), rows of the other
 * queries are left at a value < -1; gpb_hip_vecchia_neighbors_allreduce completes the table on every rank with ONE
 * ncclAllReduce(max) of n x m int32 (and the duplicates flag).  Without a communicator the parts can be merged by the host
 * (elementwise max of gpb_hip_vecchia_get_neighbors) and handed back through gpb_hip_vecchia_set_neighbors. */
GPB_HIP_EXPORT int gpb_hip_vecchia_find_neighbors_part(gpb_hip_vecchia_t* h, int32_t part, int32_t nparts, int* has_duplicates);
GPB_HIP_EXPORT int gpb_hip_vecchia_neighbors_allreduce(gpb_hip_vecchia_t* h, int* has_duplicates);
/* Multi-GPU y_aux: the shard's contribution (gpb_hip_vecchia_yaux_partial_dev) summed over the ranks with one all-reduce of n
 * doubles, delivered to the host of every rank. */
GPB_HIP_EXPORT int gpb_hip_vecchia_yaux_allreduce(gpb_hip_vecchia_t* h, double* yaux_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_nll_terms_allreduce(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                                       int gauss_likelihood, double* out3_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_grad_terms_allreduce(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                                        double* out7_host);

/* Measurement helper (bench.py): `steps` back-to-back evaluations (mode 0 = nll terms, 2 = gradient terms) on the
 * handle's stream after `warmup` untimed ones, covariance parameters perturbed every step.  ms_total: HIP events
 * around the whole timed region (point kernel + final reduction, no host sync inside); ms_point_kernel_avg: mean of
 * per-launch event pairs around the fused point kernel alone; out7_host (may be NULL): terms of the last step. */
GPB_HIP_EXPORT int gpb_hip_vecchia_bench(gpb_hip_vecchia_t* h, int mode, int cov_type, double var, double a, int warmup,
                                         int steps, double* ms_total, double* ms_point_kernel_avg, double* out7_host);

/* Materialise the factor on the device: A (n x m, B = I - A, Vecchia_utils.cpp:1620-1622), D (the
 * reference keeps D^-1, :1682) and u = B y.  Needed by y_aux / prediction-type consumers only. */
GPB_HIP_EXPORT int gpb_hip_vecchia_factor(gpb_hip_vecchia_t* h, int cov_type, double var, double a,
                                          int gauss_likelihood);
GPB_HIP_EXPORT int gpb_hip_vecchia_get_factor(gpb_hip_vecchia_t* h, double* A_host, double* D_host, double* u_host);

/* y_aux = B^T D^-1 B y (CalcYAux, include/GPBoost/re_model_template.h:9771-9773), Vecchia order.
 * Requires gpb_hip_vecchia_factor() at the current parameters; the response may have been replaced since (see gpb_hip_vecchia_set_y). */
GPB_HIP_EXPORT int gpb_hip_vecchia_yaux(gpb_hip_vecchia_t* h, double* yaux_host);
/* diag(Psi^-1) = diag(B^T D^-1 B) (transformed scale, Vecchia order) from the stored factor: the predictive variances of the training-data
 * random effects are sigma2 (1 - diag) (PredictTrainingDataRandomEffects with calc_var, include/GPBoost/re_model_template.h:4508-4514). */
GPB_HIP_EXPORT int gpb_hip_vecchia_psi_inv_diag(gpb_hip_vecchia_t* h, double* diag_host);

/* Multi-GPU form: this handle's shard contributes w = B_s^T D_s^-1 B_s y (rows of the shard only) as a full n-vector
 * in device memory (enqueued on the handle's stream); the sum over ranks (one all-reduce of n doubles)
 * is y_aux.  Requires gpb_hip_vecchia_factor() on the same shard. */
GPB_HIP_EXPORT int gpb_hip_vecchia_yaux_partial_dev(gpb_hip_vecchia_t* h, double* w_dev);

/* Prediction at new locations with vecchia_pred_type = "order_obs_first_cond_obs_only", Gaussian likelihood:
 * replaces CalcPredVecchiaObservedFirstOrder(CondObsOnly = true) (src/GPBoost/Vecchia_utils.cpp:1701-2060): neighbour search of the
 * prediction points among the OBSERVED points only (find_nearest_neighbors_Vecchia_fast with start_at = n_obs, end_search_at =
 * n_obs - 1, :1792-1799, bit-identical) + the per-point factorisation (:1883-1975) on the device.
 *   h                    the fitted state: observed coordinates (Vecchia order) and y (gpb_hip_vecchia_set_y)
 *   coords_pred_colmajor n_pred x d, column-major
 *   pred_mean            out: A_p y_nn  (= -Bpo y, :1989)
 *   pred_D               out: Dp on the transformed scale, nugget 1 included (:1875-1877): predictive variance of the response is
 *                        sigma^2 * Dp, of the latent process sigma^2 * (Dp - 1) */
GPB_HIP_EXPORT int gpb_hip_vecchia_predict_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                    int32_t num_neighbors_pred, int cov_type, double var, double a, double* pred_mean,
                                                    double* pred_D, int* has_duplicates);
/* Latent predictive mean of a non-Gaussian (Vecchia-Laplace) model at new locations, every prediction point conditioning on its nearest
 * OBSERVED points: pred_mean = -Bpo mode (PredictLaplaceApproxVecchia with CondObsOnly, include/GPBoost/likelihoods.h:8600-8602), the rows
 * of Bpo from the latent covariance (no nugget, diagonal x (1 + 1e-10): Vecchia_utils.cpp:1963-1965).  The response on the handle must be
 * the mode in Vecchia order (gpb_hip_vecchia_set_y). */
GPB_HIP_EXPORT int gpb_hip_vecchia_predict_latent_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                           int32_t num_neighbors_pred, int cov_type, double var, double a, double* pred_mean,
                                                           int* has_duplicates);
/* The same prediction with the predictive VARIANCES (pred_var, n_pred) and / or the covariance matrix (pred_cov, n_pred x n_pred row-major) of the
 * latent process; either may be NULL:  Dp + Bpo (Sigma^-1 + W)^-1 Bpo'  with W the information of the likelihood at the mode
 * (PredictLaplaceApproxVecchia, include/GPBoost/likelihoods.h:8563-8824).  The reference computes this exactly in its "cholesky" branch (:8783-8821)
 * and estimates it with nsim_var_pred random vectors in its "iterative" branch (:8637-8745); this entry point returns the exact value, solved by
 * preconditioned conjugate gradients ('vadu') on blocks of right-hand sides until every residual norm is below tol.  Needs the state of the
 * likelihood evaluation that found the mode (gpb_hip_vecchia_laplace_logit / _eval on this handle) and the mode as the handle's response.
 * cg_iterations (optional): block CG iterations used. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_predict(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                   int32_t num_neighbors_pred, int cov_type, double var, double a, int cg_max_num_it, double tol,
                                                   double* pred_mean, double* pred_var, double* pred_cov, int* has_duplicates, int* cg_iterations);

/* 'latent_order_obs_first_cond_all' for non-Gaussian models (prediction points condition on observed AND preceding prediction points): the factor
 * rows of the appended points for the LATENT process (gpb_hip_vecchia_predict_cond_all without a nugget), and the quadratic forms
 * c_r' (Sigma^-1 + W)^-1 c_s of sparse rows c_r handed over by the host -- the rows of Bp^-1 Bpo, PredictLaplaceApproxVecchia with CondObsOnly = false
 * (include/GPBoost/likelihoods.h:8603-8606, 8790-8821): cols / vals n_rows x mmax (Vecchia positions, -1 padded); out n_rows (diagonal) or, want_cov,
 * n_rows x n_rows row-major.  Same block solves and state requirements as gpb_hip_vecchia_laplace_predict. */
GPB_HIP_EXPORT int gpb_hip_vecchia_predict_cond_all_latent(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                           int32_t num_neighbors_pred, int cov_type, double var, double a, int32_t* m_used,
                                                           int32_t* nn_pred, double* A_pred, double* D_pred, int* has_duplicates);
/* Full-scale Vecchia model with a non-Gaussian likelihood: latent mean (and variance, pred_var != NULL) at new locations, 'order_obs_first_cond_obs_only' (PredictLaplaceApproxFSVA,
 * include/GPBoost/likelihoods.h:7999-8535), at the state of the last evaluation at THESE covariance parameters.  The variance is the exact expression of the reference's Cholesky
 * branch (:8455-8527) with every (B'D^-1B + W)^-1 product by block CG to the residual bound `tol`; the reference's iterative branch estimates it by simulation. */
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_laplace_predict(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred, int cov_type, double var,
                                                       double a, int cg_max_num_it, double tol, double* pred_mean, double* pred_var, double* pred_cov, int* has_duplicates,
                                                       int* cg_iterations);      /* pred_var (n_pred) / pred_cov (n_pred x n_pred, :8489-8504) may be NULL; a NEGATIVE num_neighbors_pred asks for
                                                                                  'latent_order_obs_first_cond_all' with |num_neighbors_pred| neighbours (the prediction points condition on the preceding ones too) */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_quad_forms(gpb_hip_vecchia_t* h, int32_t n_rows, int32_t mmax, const int32_t* cols_host, const double* vals_host,
                                                      int cg_max_num_it, double tol, int want_cov, double* out_host, int* cg_iterations);
/* diag((Sigma^-1 + W)^-1) at the mode (Vecchia order of the random effects): variances of the latent process at the TRAINING locations,
 * Likelihood::CalcVarLaplaceApproxVecchia behind GPB_PredictREModelTrainingDataRandomEffects (include/GPBoost/re_model_template.h:4683-4725).  Exact
 * (the reference's "cholesky" value), by the same block solves on the unit vectors: ceil(n / 52) block solves, for moderate n. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_mode_var(gpb_hip_vecchia_t* h, int cg_max_num_it, double tol, double* var_host, int* cg_iterations);

/* Full-scale Vecchia (VIF) prediction, 'order_obs_first_cond_obs_only' (CalcPredVecchiaObservedFirstOrder with the full_scale_vecchia arguments,
 * src/GPBoost/Vecchia_utils.cpp:1701-2060, called from include/GPBoost/re_model_template.h:4041-4056): neighbour search of the appended prediction points
 * among the observed ones, cross-covariances with the inducing points (ip_colmajor: k x d; Linv_rowmajor: inverse Cholesky factor of Sigma_m, as
 * gpb_hip_vecchia_vif_factor takes it) and the residual-process factor rows of the appended points.  Outputs: u_pred = -A_p y_nn and D_pred (n_pred),
 * BC_pred = C_p - A_p C_nn (n_pred x k, row-major).  With W, (B C)' D^-1 B y of the observed points (the likelihood evaluation):
 * mean = -u_pred + BC_pred W^-1 (B C)' D^-1 B y,  var = sigma2 (D_pred + BC_pred W^-1 BC_pred' [- 1 for the latent process]). */
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_predict_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                        int32_t num_neighbors_pred, const double* ip_colmajor, int cov_type, double var, double a,
                                                        const double* Linv_rowmajor, double* u_pred, double* D_pred, double* BC_pred,
                                                        int* has_duplicates);
/* The same for 'order_obs_first_cond_all' (round 5; CalcPredVecchiaObservedFirstOrder with CondObsOnly = false, src/GPBoost/Vecchia_utils.cpp:1803-1826,
 * 1889-1925, 1975-2046; the reference's only other prediction type for full-scale Vecchia models, re_model_template.h:4057-4085): neighbours among the
 * observed AND the preceding prediction points.  In addition to the outputs above (BC_pred = row of [Bpo Bp] [C; C_p], u_pred = (Bpo y)): the rows of
 * [Bpo Bp] themselves -- nn_pred (n_pred x *m_used, indices into (observed, prediction) points, -1 padded) and A_pred.  With Bp = I - A_pp (unit lower
 * triangular):  mean = Bp^-1 (-u_pred + BC_pred W^-1 (B C)' D^-1 B y),  cov = sigma2 (Bp^-1 D_pred Bp^-T + T W^-1 T'),  T = Bp^-1 BC_pred. */
GPB_HIP_EXPORT int gpb_hip_vecchia_vif_predict_cond_all(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                                        const double* ip_colmajor, int cov_type, double var, double a, const double* Linv_rowmajor,
                                                        int32_t* m_used, int32_t* nn_pred, double* A_pred, double* u_pred, double* D_pred, double* BC_pred,
                                                        int* has_duplicates);

/* Device half of the Vecchia prediction types that factor EVERY point of a joint (observed, prediction) ordering again:
 *   layout_pred_first = 1  'order_pred_first' (CalcPredVecchiaPredictedFirstOrder, src/GPBoost/Vecchia_utils.cpp:2203-2444): prediction points
 *                          first, observed points (Vecchia order) after them, neighbours among all preceding points, nugget on every diagonal
 *   layout_pred_first = 0  'latent_order_obs_first_cond_obs_only' (cond_all = 0) / '..._cond_all' (cond_all = 1)
 *                          (CalcPredVecchiaLatentObservedFirstOrder, :2446-2666), gauss_likelihood = 0: the latent process, diagonal x (1 + 1e-10)
 * Neighbour search (num_neighbors_pred candidates, capped like :755-758) and vecchia_point_kernel<MODE_FACTOR> over all n_obs + n_pred rows.
 * Outputs row-major over the joint ordering, caller allocates (n_obs + n_pred) x num_neighbors_pred: neighbour indices (-1 padded), A_i; D_i;
 * u_i = (B y)_i with y = 0 at the prediction points (may be NULL).  The conditional precision of the prediction points is assembled from
 * these rows by the caller (GPB_PredictREModel) and inverted by gpb_hip_dense_spd_solve. */
GPB_HIP_EXPORT int gpb_hip_vecchia_predict_joint_factor(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                        int32_t num_neighbors_pred, int layout_pred_first, int cond_all, int gauss_likelihood,
                                                        int cov_type, double var, double a, int32_t* m_used, int32_t* nn_all, double* A_all,
                                                        double* D_all, double* u_all, int* has_duplicates);

/* Dense symmetric positive definite solve / inverse on the device (blocked MFMA Cholesky of the exact-GP path): what the reference gives to
 * its sparse Cholesky in the prediction types above (Vecchia_utils.cpp:2419-2441, 2601-2650).  M_host row-major n x n, lower triangle
 * significant.  x_host (n, with rhs_host) = M^-1 rhs; inv_sub_host ((n - sub0)^2, row-major, symmetric) = rows / columns [sub0, n) of M^-1;
 * either may be NULL.  n <= 24000 with the inverse (one partial factorisation of [[M, .], [I, 0]]), <= 60000 without; -1 if M is not
 * positive definite. */
GPB_HIP_EXPORT int gpb_hip_dense_spd_solve(int32_t n, const double* M_host, const double* rhs_host, double* x_host, int32_t sub0,
                                           double* inv_sub_host);

/* Vecchia prediction 'order_obs_first_cond_all' (CalcPredVecchiaObservedFirstOrder with CondObsOnly = false,
 * src/GPBoost/Vecchia_utils.cpp:1701-2093): the prediction points condition on their num_neighbors_pred nearest points among the observed AND
 * the preceding prediction points (neighbour search with end_search_at = -1, :1806-1822).  The device does the search and the per-point
 * factor of the appended rows; outputs (row-major [n_pred][*m_used], caller allocates for num_neighbors_pred columns): neighbour indices
 * into (observed, prediction) (-1 padded), A_p, D_p (nugget included).  mean = Bp^-1 (-Bpo y) and the rows of Bp^-1 (:2061-2090) are
 * host work: GPB_HIP_PredictCondAllHost / GPB_PredictREModel. */
GPB_HIP_EXPORT int gpb_hip_vecchia_predict_cond_all(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor,
                                                    int32_t num_neighbors_pred, int cov_type, double var, double a, int32_t* m_used,
                                                    int32_t* nn_pred, double* A_pred, double* D_pred, int* has_duplicates);

/* Standard errors of (sigma2, sigma1_2, rho) of a Gaussian Vecchia model: stochastic (Hutchinson) Fisher information on the original scale,
 * REModelTemplate::CalcFisherInformation_Vecchia (include/GPBoost/re_model_template.h:10137-10230) behind CalculateStandardErrorsCovPars /
 * GPB_GetCovPar(calc_std_dev = true), with the reference's probe vectors (num_rand_vec, seed_rand_vec as in GPB_SetOptimConfig).
 * (ratio, a) = the transformed parameters (sigma1_2 / sigma2, transformed range) belonging to (sigma2, sigma1_2, rho). */
GPB_HIP_EXPORT int gpb_hip_vecchia_fisher_std_errors(gpb_hip_vecchia_t* h, int cov_type, double sigma2, double sigma1_2, double rho, double ratio,
                                                     double a, int num_rand_vec, int seed_rand_vec, double* se3_host);

/* Newton update of the tree leaf values in the GPBoost algorithm: replaces
 * REModelTemplate::NewtonUpdateLeafValues, Vecchia branch (include/GPBoost/re_model_template.h:4982-5063; B H and
 * (B H)^T D^-1 (B H) at :5005-5008, the L x L solve at :5056-5062).
 *   leaf_index   leaf of every point, VECCHIA order, values in [0, num_leaves), num_leaves <= 64
 *   leaf_values  out: (H^T Psi^-1 H)^-1 (- H^T y_aux)
 * Precondition (as in the reference, :4989): gpb_hip_vecchia_factor(gauss_likelihood = 1) and gpb_hip_vecchia_yaux have run for
 * the current y = F - y (the objective's gradient call, regression_objective.hpp:153-201, does exactly that). */
GPB_HIP_EXPORT int gpb_hip_vecchia_newton_leaf_values(gpb_hip_vecchia_t* h, const int32_t* leaf_index, int32_t num_leaves,
                                                      double* leaf_values);

/* ------------------------------------------------------------------------------------
 * Vecchia-Laplace approximation, Bernoulli-logit likelihood, iterative methods ("vadu" preconditioner) --
 * BASELINE config 4 / SURVEY.md 8 row a13.  Replaces, for one evaluation of the approximate marginal likelihood,
 *   Likelihood::FindModePostRandEffCalcMLLVecchia          include/GPBoost/likelihoods.h:3773-4059
 *   Likelihood::Inv_SigmaI_plus_ZtWZ_Vecchia_iterative     :16264-16348  -> CGVecchiaLaplaceVec, src/GPBoost/CG_utils.cpp:21-108
 *   Likelihood::CalcLogDetStochVecchia                     :16376-16525  -> CGTridiagVecchiaLaplace, CG_utils.cpp:110-229,
 *                                                                           LogDetStochTridiag, :1035-1051
 *   GenRandVecNormalParallel                               CG_utils.cpp:978-994 (probes; same libstdc++ engine/distribution)
 * on top of the device factor (gpb_hip_vecchia_factor with gauss_likelihood = 0, computed inside).
 *   set_labels  y in {0,1}, Vecchia order (likelihoods.h:1046-1052 rejects anything else)
 *   logit       var = sigma_1^2 (no nugget), a = transformed range; num_rand_vec / seed_rand_vec / cg_* / delta_conv_*
 *               as in GPB_SetOptimConfig (c_api.h:1437-1467; reference defaults 50 / 1 / 1000 / 1000 / 1e-2 / 1e-8);
 *               reset_mode != 0 starts Newton's method at 0 (first evaluation of a model), 0 warm-starts at the last mode.
 *               out9_host = { approximate marginal LOG-likelihood (the reference returns its negative),
 *                             #Newton iterations, total #CG iterations in the mode finding, log|Sigma W + I|,
 *                             #CG-Lanczos iterations, log p(y|mode) - 0.5 mode^T Sigma^-1 mode,
 *                             ms factor, ms mode finding, ms log-determinant (host wall clock) };
 *               mode_host (optional) receives the mode, Vecchia order. */
/* Likelihood of the Laplace path (set it BEFORE the labels, which are validated against it): 2 = "poisson" (counts >= 0; LogLikPoisson,
 * FirstDerivLogLikPoisson, SecondDerivNegLogLikPoisson, likelihoods.h:11407-11415, :12481-12483, :13315-13317, and the normalising
 * constant -sum log(y!), :10750-10757), 0 = "bernoulli_logit" (default), 1 = "bernoulli_probit" (LogLikBernoulliProbit /
 * FirstDerivLogLikBernoulliProbit / SecondDerivNegLogLikBernoulliProbit, likelihoods.h:11385-11392, :12459-12466, :13282-13291,
 * with GPBoost::normalLogCDF, DF_utils.h:74-92).  gpb_hip_vecchia_laplace_logit then evaluates that likelihood.
 * Round 5: 3 = "gamma", 4 = "negative_binomial" (auxiliary shape parameter, below), 5 = "beta" (mean = sigmoid(location), auxiliary precision; real-valued response
 * strictly inside (0, 1) through gpb_hip_vecchia_laplace_set_response_real; LogLikBeta :11903-11913, FirstDerivLogLikBeta :12501-12507, SecondDerivNegLogLikBeta
 * :13336-13346, third derivative :13892-13917, auxiliary-parameter gradient :14229-14241, :14816-14845; GPBoost::digamma / trigamma / tetragamma, src/GPBoost/DF_utils.cpp:82-201),
 * 6 = "t" (location = latent value; two auxiliary parameters scale, df; approximation_type "fisher_laplace", likelihoods.h:384-423),
 * 7 = "lognormal" (mean of y = exp(location), one auxiliary parameter: the variance of log y; real-valued response > 0; constant information 1 / aux, likelihoods.h:30-34,
 * :505-513; LogLikLogNormal :11950-11958, FirstDerivLogLikLogNormal :12534-12538, normalising constant :10623-10631 / :10887-10889, auxiliary gradient :14275-14286, :14891-14900). */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_likelihood(gpb_hip_vecchia_t* h, int likelihood_id);
/* Repeated locations for the non-Gaussian models (the reference's unique-location mapping: RECompGP with use_Z_for_duplicates,
 * include/GPBoost/re_comp.h:863-885; src/GPBoost/Vecchia_utils.cpp:1156-1168): the handle's n points are the UNIQUE locations (random effects),
 * re_ptr (n + 1, re_ptr[0] = 0, every random effect has at least one datum) is the CSR of their data.  After this call
 * gpb_hip_vecchia_laplace_set_labels / _set_fixed_effects take re_ptr[n] values GROUPED BY RANDOM EFFECT (Vecchia order of the random effects), and
 * every likelihood term of a random effect -- log-likelihood, first derivative, information, its derivative -- is the sum over its data
 * (CalcZtVGivenIndices on first_deriv_ll_ / information_ll_, likelihoods.h).  re_ptr = NULL restores one datum per random effect.  Labels and fixed
 * effects must be set again afterwards.  The boosting gradient gpb_hip_vecchia_laplace_grad_F_current then returns re_ptr[n] values, per datum in the
 * same grouped order. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_data_map(gpb_hip_vecchia_t* h, const int32_t* re_ptr);

/* Fixed effects F (offset of the location parameter, Vecchia order; NULL removes them): the likelihood is evaluated at mode + F
 * (likelihoods.h:3861-3870), which is how the GPBoost algorithm passes the tree ensemble's scores for non-Gaussian data. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_fixed_effects(gpb_hip_vecchia_t* h, const double* fixed_effects);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_labels(gpb_hip_vecchia_t* h, const int32_t* y01);
/* Likelihoods with an auxiliary parameter (round 5): likelihood ids 3 = "gamma" (shape; log link, response > 0, real-valued:
 * LogLikGamma / FirstDerivLogLikGamma / SecondDerivNegLogLikGamma, likelihoods.h:11872-11880, :12485-12487, :13319-13321; normalising constant
 * :10998-11007) and 4 = "negative_binomial" (shape; counts >= 0 through gpb_hip_vecchia_laplace_set_labels: LogLikNegBin / FirstDerivLogLikNegBin /
 * SecondDerivNegLogLikNegBin, :11882-11890, :12489-12492, :13323-13327; normalising constant :11019-11030).
 *   set_response_real  gamma's response, in the order set_labels takes its labels (Vecchia order / grouped by random effect)
 *   set_aux_pars       the shape (> 0; default 1): Likelihood::SetAuxPars; every later evaluation uses it
 *   grad_aux_current   d(-approximate marginal log-likelihood) / d log(shape) at the state of the last gpb_hip_vecchia_laplace_grad_current:
 *                      CalcGradNegLogLikAuxPars (:14185-14215) + 0.5 sum_d (d information_d / d log aux) diag_r(d) + sum_d (d^2 log p_d / d loc d log aux)
 *                      [(Sigma^-1 + W)^-1 d_mll_d_mode]_r(d)  (:6743-6808, CalcSecondDerivLogLikFirstDerivInformationAuxPar :14777-14799);
 *                      out4 = { the gradient, its three parts } */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_response_real(gpb_hip_vecchia_t* h, const double* y);
/* Sample weights of a non-Gaussian likelihood (round 5; Likelihood::weights_, include/GPBoost/likelihoods.h:666-668): log-likelihood, its derivatives wrt the
 * location parameter, the per-datum parts of the normalising constants (:10573-10600, :10750-10757, :11019-11030) and of the auxiliary-parameter gradients
 * (:14185-14215, :14777-14799) are multiplied by w_d.  Same order as the labels; NULL removes them.  Finite and >= 0. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_weights(gpb_hip_vecchia_t* h, const double* w);
/* Proportions under the logit / probit links (round 5): gpb_hip_vecchia_laplace_set_response_real also serves likelihood ids 0 / 1 with y in [0, 1] -- binomial_logit /
 * binomial_probit (y = successes / trials, the trials are the sample weights) and quasi_bernoulli_logit / _probit (LogLikBernoulliLogit<double>, LogLikBinomialProbit,
 * include/GPBoost/likelihoods.h:11394-11404 and their derivatives :12468-12474, :13293-13305, :13800-13820).  on != 0 here adds the binomial normalising constant
 * sum lgamma(w + 1) - lgamma(k + 1) - lgamma(w - k + 1), k = w y (:10612-10622). */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_binomial(gpb_hip_vecchia_t* h, int on);
/* cg_preconditioner_type of the iterative methods (round 5; SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, include/GPBoost/re_model_template.h:5906; type 2 = "fitc": below): type 0 = "vadu"
 * (P = B^T (D^-1 + W) B; the solves in the form (Sigma^-1 + W) u = rhs, src/GPBoost/CG_utils.cpp:21-229), 1 = "pivoted_cholesky" (P = W^-1 + L_k L_k^T with the
 * rank-k pivoted Cholesky factor of the non-approximated covariance matrix, include/GPBoost/CG_utils.h:438-486; the solves in the form (W^-1 + Sigma) u' = Sigma rhs,
 * u = W^-1 u', CG_utils.cpp:231-499; log-determinant and gradients: include/GPBoost/likelihoods.h:16389-16465, :16554-16611, :16716-16736).  rank =
 * fitc_piv_chol_preconditioner_rank_ (<= 0: the reference's default 50, re_model_template.h:5922); it may not exceed the number of random effects.  Takes effect at the
 * next evaluation; evaluation, gradients (covariance parameters, auxiliary parameter, fixed effects) follow it, predictions keep solving with "vadu". */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_preconditioner(gpb_hip_vecchia_t* h, int type, int rank);
/* type 2 = "fitc" (round 5): P = diag(W^-1 + Sigma_m[0][0] - ||V_i||^2) + C Sigma_m^-1 C' with the cross-covariance C of k inducing points, Sigma_m their covariance
 * (diagonal x (1 + 1e-6)), V = (L_m^-1 C')' -- Calc_FITC_Preconditioner_Vecchia (include/GPBoost/re_model_template.h:9502-9593), the fitc branches of
 * include/GPBoost/likelihoods.h:16296-16311, :16419-16437, :16465-16470, :16576-16583, :16612-16633; the same (W^-1 + Sigma) solves and the same device kernels as
 * "pivoted_cholesky" with C in place of L_k.  The inducing points come from the host (k x d, column-major; the reference draws them by kmeans++ from the model's
 * generator at the first covariance factor -- libstdc++'s distributions); rank <= 0 in set_preconditioner: the reference's default 200. */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_inducing_points(gpb_hip_vecchia_t* h, int32_t k, const double* ip_colmajor);
/* type 3 = "vecchia_response" (round 6; the fifth entry of SUPPORTED_PRECONDITIONERS_NONGAUSS_VECCHIA_, include/GPBoost/re_model_template.h:5906): the same (W^-1 + Sigma)
 * solves with P = the Vecchia approximation of W^-1 + Sigma on the model's neighbour sets, P^-1 = B_p' D_p^-1 B_p -- CalcVecchiaApproxLatentAddDiagonal
 * (re_model_template.h:5473-5492; src/GPBoost/Vecchia_utils.cpp:1418-1422, :1610-1614) renewed for every W (include/GPBoost/likelihoods.h:16315-16323), applied as
 * src/GPBoost/CG_utils.cpp:300-303 / :410-416, probes B_p^-1 D_p^1/2 r (likelihoods.h:16439-16450), log|P| = sum log D_p (:16471-16473).  On the device the factor is ONE launch
 * of the Gaussian path's point kernel (MODE_FACTOR with per-point diagonal additions 1 / W_i) per Newton iteration.  `rank` is ignored.  EVALUATION ONLY:
 * gpb_hip_vecchia_laplace_grad_current fails with the reference's message (likelihoods.h:6570-6572 refuses gradients with this preconditioner). */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_set_aux_pars(gpb_hip_vecchia_t* h, const double* aux, int32_t num_aux);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_get_aux_pars(gpb_hip_vecchia_t* h, double* aux_out, int32_t* num_aux);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_grad_aux_current(gpb_hip_vecchia_t* h, double* out4_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_logit(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int num_rand_vec,
                                                 int seed_rand_vec, int cg_max_num_it, int cg_max_num_it_tridiag,
                                                 double cg_delta_conv, double delta_conv_mode_finding, int reset_mode,
                                                 double* out9_host, double* mode_host);

/* Gradient of the NEGATIVE approximate marginal log-likelihood wrt (log sigma_1^2, log a) -- what the covariance-parameter optimiser
 * needs for non-Gaussian data.  Replaces
 *   Likelihood::CalcGradNegMargLikelihoodLaplaceApproxVecchia   include/GPBoost/likelihoods.h:6521-6700 (iterative methods, "vadu")
 *   CalcLogDetStochDerivModeVecchia / CalcLogDetStochDerivCovParVecchia   :16636-16690, :16706-16790
 *   CalcOptimalC / CalcOptimalCVectorized                       src/GPBoost/CG_utils.cpp:1053-1090
 *   eval          = gpb_hip_vecchia_laplace_logit plus keep_grad_state: the log-determinant's block CG also accumulates
 *                   U = (Sigma^-1 + W)^-1 Z (CG_utils.cpp:173) and P^-1 Z is kept, so that the gradient can follow
 *   grad_current  gradient at the state the last eval(keep_grad_state = 1) left; cg_* as in GPB_SetOptimConfig (one more preconditioned CG
 *                 solve for the implicit derivative).  parts8_host (optional): per parameter { mode' SigmaI_deriv mode, d logdet / d theta,
 *                 optimal c, implicit part }; vecs2n_host (optional, Vecchia order): d logdet / d mode, (Sigma^-1 + W)^-1 d_mll_d_mode
 *   reset_mode_to_previous   Likelihood::ResetModeToPreviousValue (likelihoods.h:997-1004): the mode goes back to its value before the
 *                 last mode finding (rejected optimiser steps)
 *   range_deriv   test seam: dA / d log a (n x m) and dD / d log a (n) of the factor without nugget (Vecchia_utils.cpp:1640-1652) */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_eval(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int num_rand_vec,
                                                int seed_rand_vec, int cg_max_num_it, int cg_max_num_it_tridiag, double cg_delta_conv,
                                                double delta_conv_mode_finding, int reset_mode, int keep_grad_state,
                                                double* out9_host, double* mode_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_grad_current(gpb_hip_vecchia_t* h, int cg_max_num_it, double cg_delta_conv,
                                                        double* grad2_host, double* parts8_host, double* vecs2n_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_reset_mode_to_previous(gpb_hip_vecchia_t* h);
/* Boosting gradient for non-Gaussian data, d(-approximate marginal log-likelihood) / dF in Vecchia order, at the state of the last
 * grad_current: -d log p / d loc + 0.5 d logdet / d mode - W .* (Sigma^-1 + W)^-1 (0.5 d logdet / d mode)
 * (CalcGradNegMargLikelihoodLaplaceApproxVecchia with calc_F_grad, likelihoods.h:6996-7001; what REModel::CalcGradient hands to the
 * boosting objective for non-Gaussian likelihoods, re_model_template.h:3298-3321).  With a data map (repeated locations): the data-scale form
 * -d log p_d / d loc + 0.5 (d information_d / d loc) diag_r - information_d [(Sigma^-1 + W)^-1 d_mll_d_mode]_r per datum d of random effect r,
 * diag_r = (d logdet / d mode)_r / (d information / d loc summed over r's data) (likelihoods.h:6944-6966, :6700-6703). */
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_grad_F_current(gpb_hip_vecchia_t* h, double* gradF_host);
GPB_HIP_EXPORT int gpb_hip_vecchia_laplace_range_deriv(gpb_hip_vecchia_t* h, int cov_type, double var, double a, double* dA_host,
                                                       double* dD_host);

/* ------------------------------------------------------------------------------------
 * Exact (dense) GP, Gaussian likelihood -- BASELINE config 1: replaces CalcSigmaComps / CalcZSigmaZt / CalcChol /
 * chol.solve(y) / log-det (include/GPBoost/re_model_template.h:8151, :9273-9287, :6491-6494, :9894, :3127).
 * Covariance assembly (HBM-write-bound), blocked Cholesky with fp64 MFMA trailing updates, triangular solves.
 *   coords_colmajor n x d in DATA order, d in {1,2,3}
 *   out2 = { y^T Psi^-1 y, log|Psi| } with Psi = Sigma(var, a) + I;  yaux (may be NULL) = Psi^-1 y
 *   ms3 (may be NULL) = HIP-event durations {assembly, factorisation, solves} of this call
 * ---------------------------------------------------------------------------------- */
GPB_HIP_EXPORT int gpb_hip_exact_create(int32_t n, int32_t d, const double* coords_colmajor, gpb_hip_exact_t** out);
GPB_HIP_EXPORT int gpb_hip_exact_free(gpb_hip_exact_t* h);
GPB_HIP_EXPORT int gpb_hip_exact_set_y(gpb_hip_exact_t* h, const double* y_host);
GPB_HIP_EXPORT int gpb_hip_exact_nll_terms(gpb_hip_exact_t* h, int cov_type, double var, double a, double* out2_host,
                                           double* yaux_host, double* ms3);
/* Likelihood terms and covariance-parameter gradient sums of the exact GP: replaces CalcPsiInv + the trace / quadratic forms of
 * CalcGradPars' dense branch (include/GPBoost/re_model_template.h:6586-6614, 2016-2040).  out7 as gpb_hip_vecchia_grad_terms:
 * {y' Psi^-1 y, log|Psi|, 0, g1_var, g2_var, g1_range, g2_range}; d nll / d log(theta_k) = g1_k / sigma2 + g2_k (transformed scale). */
GPB_HIP_EXPORT int gpb_hip_exact_grad_terms(gpb_hip_exact_t* h, int cov_type, double var, double a, double* out7_host);

/* diag(Psi^-1) of the exact GP (transformed scale): predictive variances of the training-data random effects = sigma2 (1 - diag)
 * (PredictTrainingDataRandomEffects, dense branch, include/GPBoost/re_model_template.h:4515-4620: Sigma - Sigma Psi^-1 Sigma); n <= 24000. */
GPB_HIP_EXPORT int gpb_hip_exact_psi_inv_diag(gpb_hip_exact_t* h, int cov_type, double var, double a, double* diag_host);

/* Prediction of the exact GP at new locations (dense Gaussian branch of REModelTemplate::Predict, include/GPBoost/re_model_template.h:4239-4330):
 * mean_out (n_pred) = C Psi^-1 y and q_out (n_pred^2, row-major, symmetric; may be NULL) = C Psi^-1 C' on the transformed scale
 * (Psi = Sigma / sigma2 + I, C = Sigma_pred,obs / sigma2; var and a as gpb_hip_exact_nll_terms takes them), from one partial factorisation of
 * [[Psi, ., .], [C, 0, .], [y', 0, 0]].  predictive covariance = sigma2 (Sigma_pp / sigma2 [+ I for the response] - q). */
GPB_HIP_EXPORT int gpb_hip_exact_predict(gpb_hip_exact_t* h, int cov_type, double var, double a, int32_t n_pred, const double* coords_pred_colmajor,
                                         double* mean_out, double* q_out);

/* Standard errors of (sigma2, sigma1_2, rho) of the exact GP: sqrt(diag(FI^-1)) with the Fisher information on the original scale
 * (CalcStdDevCovPar -> CalcFisherInformation, dense branch: include/GPBoost/re_model_template.h:10788-10815, 10066-10127).  sigma2 = error
 * variance, ratio = sigma1_2 / sigma2 and a = the transformed range parameter (as gpb_hip_exact_nll_terms takes them), rho = the range on
 * the original scale.  The six traces 1/2 tr(Psi^-1 dPsi_a Psi^-1 dPsi_b) are blocks of one Schur complement of a (4 n)^2 augmented matrix:
 * n <= 24000.  NaN where the Fisher information is not positive definite. */
GPB_HIP_EXPORT int gpb_hip_exact_fisher_std_errors(gpb_hip_exact_t* h, int cov_type, double sigma2, double ratio, double a, double rho,
                                                   double* se3_host);

/* ------------------------------------------------------------------------------------
 * LightGBM feature histograms: replaces Dataset::ConstructHistogramsInner for dense uint8
 * features (src/LightGBM/io/dataset.cpp:1143-1245 -> DenseBin<uint8_t>::ConstructHistogramInner,
 * src/LightGBM/io/dense_bin.hpp:98-141), i.e. what a HIP TreeLearner's ConstructHistograms
 * (src/LightGBM/treelearner/serial_tree_learner.cpp:351-373) calls.
 *   bins         F x n uint8, feature-major (the col-wise DenseBin storage)
 *   bin_offsets  F+1 prefix sums of #bins per feature (<= 256 each)
 * ---------------------------------------------------------------------------------- */
GPB_HIP_EXPORT int gpb_hip_hist_create(int32_t n, int32_t num_features, const uint8_t* bins,
                                       const int32_t* bin_offsets, gpb_hip_hist_t** out);
GPB_HIP_EXPORT int gpb_hip_hist_free(gpb_hip_hist_t* h);
/* Opt-in page-locking of caller buffers (round 4): a caller that passes the SAME gradient / hessian / leaf-index arrays to
 * gpb_hip_hist_set_gradients / gpb_hip_hist_grow_tree every iteration (the reference's Booster through route B) may register them once: the per-tree
 * copies then run at the PCIe rate instead of through the runtime's staging buffer (8 MB at n = 1e6: ~2.5 ms pageable, ~0.35 ms registered).
 * LIFETIME CONTRACT: the registered arrays must stay allocated until gpb_hip_hist_unregister_host_buffers or gpb_hip_hist_free; arrays smaller
 * than 1 MB and NULL pointers are skipped.  Without this call the library never page-locks caller memory. */
GPB_HIP_EXPORT int gpb_hip_hist_register_host_buffers(gpb_hip_hist_t* h, const double* grad, const double* hess, const int32_t* data_leaf_index);
GPB_HIP_EXPORT int gpb_hip_hist_unregister_host_buffers(gpb_hip_hist_t* h);

/* gradients/hessians of all n rows (score_t = double, include/LightGBM/meta.h:32-40); hess may be NULL
 * for a constant hessian (RegressionL2loss::IsConstantHessian, regression_objective.hpp:230). */
GPB_HIP_EXPORT int gpb_hip_hist_set_gradients(gpb_hip_hist_t* h, const double* grad, const double* hess);
/* Build the histogram of one leaf.  data_indices (int32, num_data of them; NULL = all rows).
 * hist_out: sum(bins) entries of {double grad_sum; double hess_sum} (hist_t pairs, include/LightGBM/bin.h:33-39);
 * with a constant hessian hess_sum = count * const_hess (dataset.cpp:1223-1226).
 * cnt_out (may be NULL): sum(bins) uint64 exact row counts. */
GPB_HIP_EXPORT int gpb_hip_hist_build(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data,
                                      double const_hess, double* hist_out, uint64_t* cnt_out);

/* Measurement helper: `reps` back-to-back leaf histogram builds (build + chunk reduction kernels, no D2H),
 * mean duration from HIP events on the handle's stream. */
GPB_HIP_EXPORT int gpb_hip_hist_bench(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess,
                                      int reps, double* ms_avg);

/* Data-parallel histograms (the scheme of DataParallelTreeLearner, data_parallel_tree_learner.cpp:155-173): each
 * rank's handle holds a shard of the rows (all features).  With a communicator on the handle
 *   - gpb_hip_hist_set_gradients (collective) agrees ONE fixed-point scale: all-reduce(max) of the bits of max |grad|, max |hess|;
 *   - every build (gpb_hip_hist_build, _build_slot, _build_allreduce, the tree grower) is the JOB's histogram: the rank's integer
 *     totals (two 64-bit limbs per sum + the count, 5 words per bin) are all-reduced(sum) as integers and converted once, by the same
 *     expression as on one GPU.  Counts are exact and the sums are bit-identical to the one-GPU histogram of the same rows, whatever the
 *     number of ranks and the way rows are dealt to them (tests/test_multirank_gpu.py).
 * Communicator bootstrap as for gpb_hip_vecchia_comm_init (128-byte ncclUniqueId from gpb_hip_comm_get_unique_id on rank 0), or an
 * in-process group (gpb_hip_local_group_create).  Setting a communicator invalidates the gradients: call set_gradients again. */
GPB_HIP_EXPORT int gpb_hip_hist_comm_init(gpb_hip_hist_t* h, const unsigned char* id128, int rank, int world);
GPB_HIP_EXPORT int gpb_hip_hist_comm_init_local(gpb_hip_hist_t* h, gpb_hip_local_group_t* g, int rank);
GPB_HIP_EXPORT int gpb_hip_hist_build_allreduce(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess,
                                                double* hist_out, uint64_t* cnt_out);

/* Resident leaf histograms (the role of HistogramPool, src/LightGBM/treelearner/feature_histogram.hpp:1086-1330) and the two
 * per-leaf post-processing steps of SerialTreeLearner::FindBestSplitsFromHistograms (serial_tree_learner.cpp:375-449) -- SURVEY.md 8
 * row a12:
 *   pool_resize     num_slots histograms of 2 * total_bins doubles in HBM
 *   set_fix_info    per feature: first bin of its view of the histogram (FeatureHistogram::data_, train_share_states.cpp:296-300),
 *                   BinMapper::num_bin(), BinMapper::GetMostFreqBin() (<= 0: nothing to fix)
 *   build_slot      gpb_hip_hist_build into a slot, no copy to the host
 *   fix_slot        Dataset::FixHistogram for every feature (src/LightGBM/io/dataset.cpp:1272-1290): view[mfb] = leaf sum -
 *                   sum of the other bins, subtracted in ascending bin order as the reference does (bit-identical given the histogram)
 *   subtract_slots  out = parent - smaller, FeatureHistogram::Subtract (feature_histogram.hpp:79-83); out may be parent
 *   get_slot        copy one slot to the host, (grad, hess) pairs */
GPB_HIP_EXPORT int gpb_hip_hist_pool_resize(gpb_hip_hist_t* h, int32_t num_slots);
GPB_HIP_EXPORT int gpb_hip_hist_set_fix_info(gpb_hip_hist_t* h, const int32_t* view_offset, const int32_t* num_bin,
                                             const int32_t* most_freq_bin);
GPB_HIP_EXPORT int gpb_hip_hist_build_slot(gpb_hip_hist_t* h, int32_t slot, const int32_t* data_indices, int32_t num_data,
                                           double const_hess);
GPB_HIP_EXPORT int gpb_hip_hist_fix_slot(gpb_hip_hist_t* h, int32_t slot, double sum_gradient, double sum_hessian);
GPB_HIP_EXPORT int gpb_hip_hist_subtract_slots(gpb_hip_hist_t* h, int32_t parent_slot, int32_t smaller_slot, int32_t out_slot);
/* One regression tree grown leaf-wise on the primitives above with the leaves' row lists RESIDENT on the device (the reference keeps
 * them in DataPartition on the host): restates the control flow of SerialTreeLearner::Train (serial_tree_learner.cpp:159-210, :283-323,
 * :325-449, :565-690; DataPartition::Split, data_partition.hpp:101-130; SplitInfo::operator>, split_info.hpp:126-153) for numerical
 * features and the default regularisation path.  Needs set_gradients, set_fix_info, set_split_info and a pool of >= num_leaves slots.
 *   sum_gradient / sum_hessian   root sums as LeafSplits::Init computes them (leaf_splits.hpp:73-86); sum_hessian = n * const_hess for a
 *                                constant hessian
 *   outputs                      the arrays of include/LightGBM/tree.h: per node (num_leaves - 1) split_feature_inner, threshold_in_bin,
 *                                default_left, left_child / right_child (~leaf for leaves), split_gain, internal_count; per leaf
 *                                (num_leaves) leaf_value (before shrinkage), leaf_count; data_leaf_index (n, optional): leaf of every row
 * Per split only the left count and the F x 10 split candidates of the two children cross PCIe.
 * Data-parallel form: with a communicator on the handle (gpb_hip_hist_comm_init / _comm_init_local) the rows are a shard; every freshly
 * built histogram (integer totals, see above) and every left count are all-reduced (DataParallelTreeLearner's scheme,
 * data_parallel_tree_learner.cpp:55-80, :155-173, :240-260); the root's sums and row count are read off the root histogram's integer
 * totals -- sum_gradient / sum_hessian are IGNORED then --, so that the tree does not depend on the rank layout either.  All ranks
 * return the same tree with GLOBAL counts, data_leaf_index covers the rank's own rows. */
GPB_HIP_EXPORT int gpb_hip_hist_grow_tree(gpb_hip_hist_t* h, int32_t num_leaves, double sum_gradient, double sum_hessian, double lambda_l2,
                                          int32_t min_data_in_leaf, double min_sum_hessian_in_leaf, double min_gain_to_split,
                                          double const_hess, int32_t* out_num_leaves, int32_t* split_feature_inner,
                                          uint32_t* threshold_in_bin, int32_t* default_left, int32_t* left_child, int32_t* right_child,
                                          double* split_gain, int32_t* internal_count, double* leaf_value, int32_t* leaf_count,
                                          int32_t* data_leaf_index);
/* Per node of the LAST tree grown on the handle: {left output, right output, left count, right count, left sum of hessians, right sum of
 * hessians} -- what Tree::Split (include/LightGBM/tree.h:63-66) takes next to the arrays above (integration/hip_tree_learner.h builds the
 * reference's own Tree object from them).  out6: 6 x num_nodes doubles. */
GPB_HIP_EXPORT int gpb_hip_hist_last_tree_node_info(gpb_hip_hist_t* h, int32_t num_nodes, double* out6);
/* Data-parallel form of gpb_hip_hist_grow_tree, the exchange below the root (round 5).  on != 0 (the default): DataParallelTreeLearner's scheme
 * (src/LightGBM/treelearner/data_parallel_tree_learner.cpp:131, :155-173 reduce-scatter of the smaller leaf's histogram by feature block, :244 sync of the best
 * split) with the integer totals on the wire -- ONE reduce-scatter of 3 (constant hessian) or 5 int64 words per bin, every rank converts / fixes / subtracts /
 * searches the features whose bins it received, then one all-reduce of world x 2 zero-padded candidate records (24 doubles each) gives every rank the job's best
 * split of the two children.  on == 0: every rank all-reduces the whole histogram and searches every feature.  on < 0 (the default): feature blocks when a
 * histogram message is at least 2 MB, the all-reduce below (latency-bound messages: one collective and one synchronisation fewer per split).  Identical trees
 * either way, for every rank layout. */
GPB_HIP_EXPORT int gpb_hip_hist_set_feature_block_exchange(gpb_hip_hist_t* h, int on);
GPB_HIP_EXPORT int gpb_hip_hist_get_slot(gpb_hip_hist_t* h, int32_t slot, double* hist_out);

/* Split search on a device-resident (fixed) leaf histogram -- SURVEY.md 8f rank 2: FeatureHistogram::FindBestThreshold for every
 * numerical feature (src/LightGBM/treelearner/feature_histogram.hpp:85-95, :857-1084; all three missing-value types :163-207) and the
 * choice among features (serial_tree_learner.cpp:725-756, split_info.hpp:126-153), default regularisation path only (lambda_l1 = 0,
 * max_delta_step = 0, path_smooth = 0, no monotone constraints / extra_trees / CEGB).  Bit-identical to the reference given the histogram.
 *   set_split_info   per feature: FeatureMetainfo::offset (1 iff most_freq_bin == 0), BinMapper::GetDefaultBin(), missing type
 *                    (0 None, 1 Zero, 2 NaN; include/LightGBM/bin.h:27-31); views / num_bin come from gpb_hip_hist_set_fix_info
 *   find_best_split  leaf totals sum_gradient / sum_hessian / num_data as LeafSplits holds them; config values lambda_l2,
 *                    min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split; is_feature_used (may be NULL) as in :753
 *                    best_feature: inner feature index (-1 if the mask is empty); per_feature_out10 (may be NULL): F x 10 doubles =
 *                    SplitInfo {gain, threshold, left_count, right_count, left_output, right_output, left_sum_gradient,
 *                    left_sum_hessian, right_sum_gradient, right_sum_hessian}; per_feature_default_left (may be NULL);
 *                    per_feature_splittable (may be NULL): FeatureHistogram::is_splittable() after the search -- the children of a
 *                    leaf skip the features that were not splittable in it (serial_tree_learner.cpp:328-334) */
/* Partition of a leaf's rows by a numerical split: DataPartition::Split (src/LightGBM/treelearner/data_partition.hpp:101-130) ->
 * Dataset::Split (dataset.h:506-516) -> DenseBin::Split / SplitInner (src/LightGBM/io/dense_bin.hpp:176-307; single-feature groups,
 * all missing-value variants).  Stable: lte_out and gt_out (each sized cnt by the caller) keep the order of data_indices
 * (NULL = all rows); *lte_count = rows going left.  threshold / default_left as SplitInfo holds them (inner feature index). */
GPB_HIP_EXPORT int gpb_hip_hist_split_leaf(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature,
                                           uint32_t threshold, int default_left, int32_t* lte_out, int32_t* gt_out, int32_t* lte_count);
GPB_HIP_EXPORT int gpb_hip_hist_set_split_info(gpb_hip_hist_t* h, const int32_t* offset, const int32_t* default_bin,
                                               const int32_t* missing_type);
/* Categorical features (round 5).  is_categorical[f] != 0: feature f is searched by FeatureHistogram::FindBestThresholdCategoricalInner
 * (src/LightGBM/treelearner/feature_histogram.hpp:278-519: one-hot for num_bin <= max_cat_to_onehot, otherwise the bins with >= cat_smooth rows sorted by
 * sum_grad / (sum_hess + cat_smooth) and accumulated from both ends, at most max_cat_threshold per side, a candidate every min_data_per_group rows, l2
 * raised by cat_l2) instead of the threshold scans, in gpb_hip_hist_find_best_split AND gpb_hip_hist_grow_tree; its splits are SETS of bins
 * (DataPartition::Split -> DenseBin::SplitCategorical, dense_bin.hpp:305-362).  The five configuration values are the reference's Config fields of the
 * same names (defaults 4, 32, 10, 10, 100).  is_categorical == NULL: every feature numerical.  Columns stay single-feature columns: a feature the
 * reference keeps inside a bundle (EFB) or a multi-value group is handed over as its own column (0 = most frequent bin; BinIterator::Get gives the bin).
 *   per_feature_out10 of a categorical feature: column 1 (threshold) holds the NUMBER of bins going left; gpb_hip_hist_last_split_cat_bits returns the sets
 *   (8 words per feature, bit b = the feature's bin b; zeros for numerical features) of the last gpb_hip_hist_find_best_split
 *   gpb_hip_hist_split_leaf_categorical: gpb_hip_hist_split_leaf with such a set instead of (threshold, default_left)
 *   gpb_hip_hist_last_tree_cat_nodes: per node of the last gpb_hip_hist_grow_tree tree, is it categorical and its set (8 words) -- the cat_bitset_inner of
 *   Tree::SplitCategorical (serial_tree_learner.cpp:617-640); threshold_in_bin of such a node = its running index among the categorical nodes */
GPB_HIP_EXPORT int gpb_hip_hist_set_categorical(gpb_hip_hist_t* h, const int8_t* is_categorical, int32_t max_cat_to_onehot, int32_t max_cat_threshold,
                                                double cat_smooth, double cat_l2, int32_t min_data_per_group);
GPB_HIP_EXPORT int gpb_hip_hist_last_split_cat_bits(gpb_hip_hist_t* h, uint32_t* bits_out);
GPB_HIP_EXPORT int gpb_hip_hist_split_leaf_categorical(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature,
                                                       const uint32_t* cat_bits8, int32_t* lte_out, int32_t* gt_out, int32_t* lte_count);
GPB_HIP_EXPORT int gpb_hip_hist_last_tree_cat_nodes(gpb_hip_hist_t* h, int32_t num_nodes, int32_t* is_categorical, uint32_t* cat_bits8);
/* The other regularisation paths of the search (config lambda_l1, max_delta_step, path_smooth: feature_histogram.hpp:137-161 picks the template
 * instance of FindBestThresholdSequentially; ThresholdL1 :737-741, CalculateSplittedLeafOutput :743-765, GetLeafGain :826-857).  They stay set on
 * the handle for gpb_hip_hist_find_best_split and gpb_hip_hist_grow_tree; all zero (the default) is the plain lambda_l2 path.  parent_output is
 * the `parent_output` argument of FindBestThreshold (:85-95) for the following gpb_hip_hist_find_best_split calls -- the leaf's own output,
 * used by path smoothing only; gpb_hip_hist_grow_tree keeps track of it itself (SerialTreeLearner::GetParentOutput,
 * serial_tree_learner.cpp:758-770). */
GPB_HIP_EXPORT int gpb_hip_hist_set_regularisation(gpb_hip_hist_t* h, double lambda_l1, double max_delta_step, double path_smooth,
                                                   double parent_output);
/* config max_depth for gpb_hip_hist_grow_tree (<= 0: no limit): the children of a split at depth max_depth - 1 are not searched
 * (SerialTreeLearner::BeforeFindBestSplit, serial_tree_learner.cpp:286-295). */
GPB_HIP_EXPORT int gpb_hip_hist_set_max_depth(gpb_hip_hist_t* h, int32_t max_depth);
/* Bagging: the rows the root of the following trees holds (ascending row indices; NULL or cnt <= 0: all rows) -- what
 * DataPartition::SetUsedDataIndices / Init do for the reference's learner (data_partition.hpp:57-63, serial_tree_learner.cpp SetBaggingData).
 * The caller passes the bag's own sum_gradient / sum_hessian to gpb_hip_hist_grow_tree; data_leaf_index is -1 for rows outside the bag. */
GPB_HIP_EXPORT int gpb_hip_hist_set_root_rows(gpb_hip_hist_t* h, const int32_t* rows, int32_t cnt);
/* The columns gpb_hip_hist_grow_tree may split on (config feature_fraction: ColSampler::is_feature_used_bytree(), col_sampler.hpp:181; the
 * caller samples, as the reference's learner does in BeforeTrain, serial_tree_learner.cpp:258): F flags by inner feature index, NULL = all. */
GPB_HIP_EXPORT int gpb_hip_hist_set_feature_mask(gpb_hip_hist_t* h, const int8_t* is_feature_used);
GPB_HIP_EXPORT int gpb_hip_hist_find_best_split(gpb_hip_hist_t* h, int32_t slot, double sum_gradient, double sum_hessian, int32_t num_data,
                                                double lambda_l2, int32_t min_data_in_leaf, double min_sum_hessian_in_leaf,
                                                double min_gain_to_split, const int8_t* is_feature_used, int32_t* best_feature,
                                                double* per_feature_out10, int32_t* per_feature_default_left,
                                                int32_t* per_feature_splittable);

#ifdef __cplusplus
}
#endif
#endif /* GPB_HIP_H_ */
