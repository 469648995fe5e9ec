/*
 * gpmi.h — C-ABI of the MI355X-native Gaussian-process regression hot path.
 *
 * Drop-in boundary for the `inference.gp` path of C-bowman/inference-tools
 * (reference @ 2025-06-14).  The reference is pure Python and has NO FFI of its
 * own; each entry point below therefore cites the
 * reference *Python* function it replaces (file:line relative to the reference
 * root).  The Python host package `inference_amd` binds these through ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, 64-bit sizes; no torch / numpy types.
 *   - all arithmetic is IEEE fp64; matrices are row-major (NumPy C order).
 *   - pointers named *_host are host memory owned by the caller; the library
 *     owns every device buffer behind the handle.  Pointers named *_dev in the
 *     `gpmi_dev_*` block are device pointers (tests / micro-benchmarks).
 *   - every function returns an int status: 0 = ok, <0 = error
 *     (GPMI_ERR_*; text via gpmi_last_error).  Numerical failure of the
 *     Cholesky factorisation is NOT an error status: it is reported LAPACK-style
 *     through `info` (0 = ok, k>0 = leading minor of order k is not positive
 *     definite / not finite), which the host maps to the reference's
 *     `LinAlgError` handling (regression.py:536-542).
 *   - calls on one handle are serialised on the handle's HIP stream; several
 *     handles may be used concurrently from different threads / on different
 *     devices.  No callbacks into the caller.
 */
#ifndef GPMI_H
#define GPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMI_VERSION 100

/* covariance kernels (covariance.py:181-279 SquaredExponential, 282-368 RationalQuadratic) */
#define GPMI_KERNEL_SE 0 /* theta = [ln a, ln l_1..ln l_d]            n_theta = d+1 */
#define GPMI_KERNEL_RQ 1 /* theta = [ln a, ln kappa, ln l_1..ln l_d]  n_theta = d+2 */

#define GPMI_OK 0
#define GPMI_ERR_ARG (-1)     /* bad argument / call order */
#define GPMI_ERR_HIP (-2)     /* HIP runtime error */
#define GPMI_ERR_NOMEM (-3)   /* device allocation failed */
#define GPMI_ERR_NODEVICE (-4)/* no usable gfx950 device */
#define GPMI_ERR_INTERNAL (-5) /* internal consistency check failed (a bug: please report) */

typedef struct gpmi_ctx gpmi_ctx;

/* ---- lifecycle ------------------------------------------------------------------ */
int gpmi_version(void);
/* number of visible HIP devices (does not create a context) */
int gpmi_device_count(int* count);
/* PCI bus id ("0000:c1:00.0") of visible device `device` into buf (cap >= 16 bytes, NUL-terminated): a physical identity
 * that does not depend on HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, used by the sharded drivers to tell whether two
 * ranks share a device (no reference counterpart: regression.py:597-601 farms its work over CPU processes) */
int gpmi_device_pci_bus_id(int device, char* buf, int cap);
/* create a handle bound to `device` (own stream + workspaces) */
int gpmi_create(int device, gpmi_ctx** ctx);
int gpmi_destroy(gpmi_ctx* ctx);
/* text of the last error on this handle (ctx may be NULL: last create error) */
const char* gpmi_last_error(const gpmi_ctx* ctx);
/* block until all work queued on the handle's stream has finished */
int gpmi_sync(gpmi_ctx* ctx);

/* ---- data -------------------------------------------------------------------------
 * Replaces GpRegressor.__init__ storing x, y and sig (regression.py:94-133, 246-322) and
 * CovarianceFunction.pass_spatial_data (covariance.py:212-226, 309-321): only x (n x d),
 * y (n) and the data-error covariance are uploaded — the N x N x d tensors of the
 * reference are never formed.
 *   noise_var_host : n values y_err**2 (regression.py:320) or NULL for zeros (regression.py:322)
 *   y_cov_host     : dense n x n y-covariance (regression.py:262-293) or NULL; replaces noise_var
 */
int gpmi_set_data(gpmi_ctx* ctx, const double* x_host, const double* y_host,
                  const double* noise_var_host, const double* y_cov_host, int64_t n, int64_t d);

/* ---- fit ---------------------------------------------------------------------------
 * Replaces GpRegressor.set_hyperparameters (regression.py:218-244):
 *   K_xx = K(theta) + extra_diag*I + sig ; L = chol(K_xx) ; alpha = L^-T L^-1 (y - mu)
 *   theta_host  : covariance hyper-parameters (log space), n_theta values
 *   extra_diag  : variance added to the diagonal by a `+ WhiteNoise()` component
 *                 (covariance.py:163-169), 0 otherwise
 *   mu_host     : the mean vector MeanFunction.build_mean(theta_mean) (mean.py:47-48), n values
 *   alpha_host  : out, n values          logdet_host : out, sum_i ln L_ii
 * The factor stays resident on the device for predict / posterior / gradients.
 */
int gpmi_fit(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta, double extra_diag,
             const double* mu_host, double* alpha_host, double* logdet_host, int* info);

/* ---- log-marginal likelihood ------------------------------------------------------
 * Replaces GpRegressor.marginal_likelihood (regression.py:528-542):
 *   lml = -1/2 |L^-1 (y-mu)|^2 - sum ln L_ii      (no -(n/2) ln 2 pi term, as the reference)
 * Uses a scratch matrix: the fitted state of gpmi_fit is not disturbed.
 * If info != 0 the value written is -1e50 (regression.py:540-542 sentinel; the host warns).
 */
int gpmi_lml(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta, double extra_diag,
             const double* mu_host, double* lml_host, int* info);

/* T independent evaluations of gpmi_lml (hyper-parameter grid / DE population / PT chains):
 *   thetas_host : T x n_theta      extra_diag_host : T values or NULL
 *   mus_host    : T x n mean vectors, or NULL with mu_const_host : T constants (ConstantMean)
 *   lml_host    : T out            info_host : T out
 * Evaluations are spread over the handle's worker streams (gpmi_set_streams). */
int gpmi_lml_batch(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                   const double* extra_diag_host, const double* mus_host,
                   const double* mu_const_host, double* lml_host, int* info_host);
/* Asynchronous form of gpmi_lml_batch for the lockstep sizes (n <= 4096, diagonal data errors): submit enqueues the T
 * (<= 256; GPMI_ASYNC_SLOT_MAX) evaluations of slot 0 or 1 and returns at once, wait blocks until they are through and delivers lml[T] /
 * info[T] in the order submitted.  The two slots are the two halves of the lockstep workspace and run side by side; a
 * value is bit-identical to what gpmi_lml_batch returns for the same hyper-parameters.  What it is for: a tempering
 * driver (mcmc/parallel.py:190-231 in the reference: one process per chain, results over pipes) splits its ladders
 * in two groups and does the accept / reject bookkeeping of one group while the device evaluates the other.
 * While a slot is pending, gpmi_lml_batch / gpmi_lml_grad_batch on the same handle are refused (GPMI_ERR_ARG). */
int gpmi_lml_batch_submit(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                          const double* extra_diag, const double* mus_host, const double* mu_const, int slot);
int gpmi_lml_batch_wait(gpmi_ctx* ctx, int slot, double* lml, int* info);

/* number of concurrent worker streams (each with its own n x n scratch) used by gpmi_lml_batch */
int gpmi_set_streams(gpmi_ctx* ctx, int n_streams);
/* Handle options.
 *   GPMI_OPT_LOCKSTEP_ALWAYS  1: gpmi_lml_batch / gpmi_lml take the lockstep path (all evaluations of a call in one
 *       launch sequence, batch in blockIdx.z) for ANY number of evaluations while n <= 4096, also for T = 1.  A
 *       value then does not depend on which other evaluations share its call - what the lockstep MCMC drivers
 *       (GibbsChain retries make the batches ragged, reference gibbs.py:635-648) need for trajectories that are
 *       bit-identical however the chains are grouped.  0 (default): a single evaluation uses the lane path. */
#define GPMI_OPT_LOCKSTEP_ALWAYS 1
/*   GPMI_OPT_RESERVE_POINTS   r >= 0, read by the NEXT gpmi_set_data: room for r more training points in the padded
 *       device matrices (identity in the padding: the factor is unaffected), to be filled by gpmi_append_point */
#define GPMI_OPT_RESERVE_POINTS 2
/*   GPMI_OPT_NO_FLOW          1: factorisations of this handle keep the stream-ordered schedule for their chain-bound
 *     part instead of the flag-ordered tile-task launch (same factor, bit for bit; slower).  The flag-ordered launch
 *     needs its two kernels side by side on the device; when something outside the library prevents that for about
 *     a second (a heavily oversubscribed device, a tool that serialises kernels) the call fails with
 *     GPMI_ERR_INTERNAL instead of hanging - a caller can then set this option and repeat the call (the Python layer
 *     does, once, with a warning). */
#define GPMI_OPT_NO_FLOW 3
int gpmi_set_option(gpmi_ctx* ctx, int option, int value);

/* Allocate now what gpmi_lml_grad would allocate inside its first call - the evaluation lane (a second n x n matrix,
 * its streams), the matrix of L^-T, the contraction's partial sums for n_theta covariance parameters - so that the first
 * gradient evaluation of an optimiser run costs what the others cost (N = 16384: 169 ms -> 78 ms; the reference has no
 * counterpart: regression.py:544-567 allocates its N x N temporaries in every call). */
int gpmi_prepare_gradient(gpmi_ctx* ctx, int n_theta);

/* Replaces GpRegressor.marginal_likelihood_gradient (regression.py:544-567):
 *   grad_theta_host : n_theta values  1/2 sum (alpha alpha^T - K^-1) o dK/dtheta_j
 *                     (dK_j recomputed from x on the fly: covariance.py:268-276, 350-365)
 *   trace_q_host    : sum_i (alpha_i^2 - K^-1_ii)  (= d/d extra_diag * 2; WhiteNoise gradient,
 *                     covariance.py:171-175) or NULL
 *   alpha_host      : n values K^-1 (y-mu) for the mean-parameter gradients (regression.py:563)
 * No sentinel here: info != 0 is returned to the host, which raises like the reference
 * (regression.py:555 has no LinAlgError guard). */
int gpmi_lml_grad(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta,
                  double extra_diag, const double* mu_host, double* lml_host,
                  double* grad_theta_host, double* trace_q_host, double* alpha_host, int* info);
/* T evaluations of gpmi_lml_grad (marginal_likelihood_gradient, regression.py:544-567) in one call - the L-BFGS-B starts
 * of the hyper-parameter search (regression.py:585-605: the reference farms them over a multiprocessing.Pool).  For
 * padded N <= 4096 the evaluations of a chunk advance in LOCKSTEP (every launch carries the chunk, as in
 * gpmi_lml_batch); larger problems run one after another.  thetas: T x n_theta; extra, mu_const: T (or mus: T x n);
 * outputs lml (T), grad_theta (T x n_theta), trace_q (T, may be NULL), alpha_host (T x n, may be NULL), info (T). */
int gpmi_lml_grad_batch(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                        const double* extra_diag_host, const double* mus_host, const double* mu_const_host,
                        double* lml, double* grad_theta, double* trace_q, double* alpha_host, int* info);

/* ---- prediction (needs a prior gpmi_fit) --------------------------------------------
 * Replaces the per-point loop of GpRegressor.__call__ (regression.py:188-216) by one batched
 * cross-covariance + triangular solve:
 *   mu_host[m]  = k(q_m, x) . alpha            (the host adds MeanFunction(q), regression.py:212)
 *   var_host[m] = a^2 - |L^-1 k(q_m, x)|^2     (the host takes sqrt(abs(.)), regression.py:216)
 */
int gpmi_predict(gpmi_ctx* ctx, const double* pts_host, int64_t m, double* mu_host,
                 double* var_host);
/* Replaces GpRegressor.build_posterior (regression.py:421-449): mu (m) and
 * Sigma = K_qq - Q^T Q (m x m), Q = L^-1 K_qx^T.  cov_host may be NULL (mean_only). */
int gpmi_posterior(gpmi_ctx* ctx, const double* pts_host, int64_t m, double* mu_host,
                   double* cov_host);
/* Replaces GpRegressor.spatial_derivatives (regression.py:387-419), SE kernel only
 * (RationalQuadratic has no gradient_terms: covariance.py:38-44): dmu (m x d), dvar (m x d). */
int gpmi_spatial_derivatives(gpmi_ctx* ctx, const double* pts_host, int64_t m, double* dmu_host,
                             double* dvar_host);
/* Replaces GpRegressor.gradient (regression.py:351-385), SE only: mean (m x d) and covariance
 * (m x d x d) of the gradient of the regression estimate. */
int gpmi_gradient(gpmi_ctx* ctx, const double* pts_host, int64_t m, double* gmu_host,
                  double* gcov_host);

/* Replaces GpRegressor.loo_likelihood_gradient (regression.py:489-526): besides alpha and diag(K^-1)
 * returns p = K^-1 (alpha / diag) (for the mean-parameter gradients, regression.py:516-520) and
 *   grad_theta_host[j] = sum_ab dK_j[a][b] (sym(p alpha^T) - K^-1 diag(c2) K^-1)_ab   (regression.py:511-514)
 *   trace_q_host       = trace of that matrix (WhiteNoise gradient = 2 sigma^2 trace).
 * No sentinel: info != 0 makes the host raise, as regression.py:501 has no LinAlgError guard. */
int gpmi_loo_grad(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta, double extra_diag,
                  const double* mu_host, double* alpha_host, double* ikdiag_host, double* p_host,
                  double* grad_theta_host, double* trace_q_host, int* info);
/* gpmi_lml_grad_batch with noise variances of their own for every evaluation (noise_var_host: T x n) and the diagonal of
 * alpha alpha^T - K^-1 back (qdiag_host: T x n): what HeteroscedasticNoise (covariance.py:608-690: one variance per point IS
 * a hyper-parameter, its gradient is exp(2 theta_i) qdiag_i, :683-689) needs from a batch - the single-evaluation form is
 * gpmi_set_noise + gpmi_lml_grad + gpmi_lml_grad_qdiag. */
int gpmi_lml_grad_batch_noise(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                              const double* extra_diag_host, const double* mus_host, const double* mu_const_host,
                              const double* noise_var_host, double* lml_host, double* grad_theta_host,
                              double* trace_q_host, double* alpha_host, double* qdiag_host, int* info);

/* The same for T hyper-parameter vectors at once (thetas: T x n_theta, extra / mu_const: T values or mus: T x n; outputs
 * T x n, T x n_theta, T): for n <= 4096 every launch carries the whole chunk (lockstep, like gpmi_lml_grad_batch); the
 * multi-start search with the cross-validation objective (regression.py:159-164, 585-605) advances all its starts on it. */
int gpmi_loo_grad_batch(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                        const double* extra_diag_host, const double* mus_host, const double* mu_const_host,
                        double* alpha_host, double* ikdiag_host, double* p_host, double* grad_theta_host,
                        double* trace_q_host, int* info);
/* gpmi_loo_grad_batch with noise variances of their own for every evaluation (noise_var_host: T x n) and, back, the diagonal of
 * M = K^-1 diag(c2) K^-1 (mdiag_host: T x n): the leave-one-out gradient with respect to HeteroscedasticNoise's parameter of
 * point i is 2 exp(2 theta_i) (p_i alpha_i - M_ii) (regression.py:509-514 with dK = 2 s_i^2 e_i e_i^T, covariance.py:683-689).
 * Lockstep sizes only (n <= 4096, diagonal data errors); the cross_val = True search of such a model advances its starts on it. */
int gpmi_loo_grad_batch_noise(gpmi_ctx* ctx, int kernel, int64_t T, const double* thetas_host, int n_theta,
                              const double* extra_diag_host, const double* mus_host, const double* mu_const_host,
                              const double* noise_var_host, double* alpha_host, double* ikdiag_host, double* p_host,
                              double* mdiag_host, double* grad_theta_host, double* trace_q_host, int* info);

/* ---- covariance plugin surface -----------------------------------------------------
 * Replaces CovarianceFunction.build_covariance (covariance.py:247-255, 343-348): the n x n matrix
 * a^2 (C + 1e-12 I) + extra_diag I, plus the data-error covariance when with_noise != 0. */
int gpmi_covariance(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta,
                    double extra_diag, int with_noise, double* K_host);
/* Replaces CovarianceFunction.__call__(u, v = x, theta) (covariance.py:240-245, 335-341):
 * the m x n cross-covariance between pts and the stored x, no jitter. */
int gpmi_cross_covariance(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta,
                          const double* pts_host, int64_t m, double* out_host);

/* ---- lazy attribute downloads (GpRegressor.K_xx / .L, regression.py:239-241) ------- */
int gpmi_get_K(gpmi_ctx* ctx, double* K_host); /* n x n, rebuilt from the fitted theta */
int gpmi_get_L(gpmi_ctx* ctx, double* L_host); /* n x n lower factor, upper triangle zero */

/* ---- leave-one-out (regression.py:451-526) ------------------------------------------ */
/* diag(K^-1) of the fitted model: var = 1/diag, mu = y - alpha*var on the host (regression.py:460-466) */
int gpmi_loo_diag(gpmi_ctx* ctx, double* ikdiag_host);
/* alpha = K^-1 (y - mu) and diag(K^-1) at an arbitrary theta (scratch matrix; the fitted state is not
 * disturbed): the O(n^3) part of GpRegressor.loo_likelihood (regression.py:468-487); the host forms
 * -1/2 sum(var alpha^2 + ln var), var = 1/diag.  info != 0 -> the host returns the -1e50 sentinel. */
int gpmi_loo_terms(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta,
                   double extra_diag, const double* mu_host, double* alpha_host,
                   double* ikdiag_host, int* info);

/* ---- per-point noise hyper-parameters (HeteroscedasticNoise, covariance.py:608-690) ----
 * K = K_stationary + diag(sigma_i^2) + data errors: the host adds exp(2 theta_i) to the data variances and
 * replaces the diagonal term with this call before gpmi_fit / gpmi_lml / gpmi_lml_grad (n values; ignored
 * when a dense y_cov was given to gpmi_set_data). */
int gpmi_set_noise(gpmi_ctx* ctx, const double* noise_var_host);
/* q_i = alpha_i^2 - (K^-1)_ii of the most recent gpmi_lml_grad call (n values): the gradient with respect to
 * ln sigma_i is 1/2 Q_ii 2 sigma_i^2 = sigma_i^2 q_i (covariance.py:682-686, regression.py:561-565). */
int gpmi_lml_grad_qdiag(gpmi_ctx* ctx, double* qdiag_host);

/* ---- mixture covariance: ChangePoint (covariance.py:371-606) ---------------------------
 * K = sum_m diag(g_m) K_m(theta_m) diag(g_m) + extra_diag I + data errors, with up to four stationary
 * sub-kernels K_m and per-point weights g_m (products of the logistic windows of covariance.py:529-559,
 * evaluated on the host in O(N)).  kernels[nk]: GPMI_KERNEL_*; thetas: the sub-kernels' parameter vectors
 * concatenated; n_thetas[nk]; g_host: nk x n row-major. */
int gpmi_fit_mix(gpmi_ctx* ctx, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                 const double* g_host, double extra_diag, const double* mu_host, double* alpha_host,
                 double* logdet, int* info);
int gpmi_lml_mix(gpmi_ctx* ctx, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                 const double* g_host, double extra_diag, const double* mu_host, double* lml, int* info);
/* LML and the gradient pieces (covariance.py:561-594 + regression.py:544-567): grad_thetas = the sub-kernels'
 * parameter gradients, concatenated; hrows: the window row sums, Q = alpha alpha^T - K^-1, from which the host forms the
 * window-parameter gradients.  hw_host == NULL: hrows is nk x n, h_m(i) = sum_j Q_ij K_m,ij g_m(j) (two regions: the host
 * contracts sum_i dw_i (h_1 - h_0)_i).  hw_host != NULL (nk x 2 x n, round 5): the caller's own row-sum weights, two per
 * sub-kernel, hrows nk x 2 x n with h_{m,r}(i) = sum_j Q_ij K_m,ij hw_{m,r}(j) - covariance.py:588-593 differentiates
 * change-point c through the factors (1 - f_c) of K_c and f_c of K_{c+1} ONLY, whatever other windows multiply them, so
 * for any number of regions the host passes hw_{m,0} = f_{m-1}, hw_{m,1} = 1 - f_m and contracts
 * sum_i df_c(i) (h_{c+1,0} - h_{c,1})(i); the same expression as hw == NULL when nk = 2.
 * q_i for additive noise terms comes from gpmi_lml_grad_qdiag as usual. */
int gpmi_lml_grad_mix(gpmi_ctx* ctx, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                      const double* g_host, const double* hw_host, double extra_diag, const double* mu_host, double* lml,
                      double* grad_thetas, double* hrows_host, double* alpha_host, int* info);

/* gpmi_lml_grad_mix for T hyper-parameter vectors in one call (round 4; regression.py:544-567 with a ChangePoint kernel,
 * covariance.py:529-594, for the T starts of multistart_bfgs, regression.py:585-605).  N <= 4096: the evaluations advance
 * in lockstep, every launch carrying all of them.  thetas: T rows, each the sub-kernels' parameters back to back
 * (sum of n_thetas values); g: T x nk x n window weights; hw: NULL or T x nk x 2 x n row-sum weights (see
 * gpmi_lml_grad_mix); extra: T WhiteNoise variances (or NULL); one of mus (T x n) / mu_const (T).  Out: lml[T],
 * grad_thetas[T x sum n_thetas], hrows[T x nk x n] (hw == NULL: h_m(i) = sum_j Q_ij K_m,ij g_m(j)) or [T x nk x 2 x n],
 * optional alpha_out[T x n], qdiag_out[T x n] (diag(alpha alpha^T - K^-1)), info[T]. */
int gpmi_lml_grad_batch_mix(gpmi_ctx* ctx, int nk, const int* kernels, int64_t T, const double* thetas,
                            const int* n_thetas, const double* g, const double* hw, const double* extra,
                            const double* mus, const double* mu_const, double* lml, double* grad_thetas, double* hrows,
                            double* alpha_out, double* qdiag_out, int* info);
/* The leave-one-out counterpart (regression.py:489-526 with covariance.py:529-594) for T hyper-parameter vectors in lockstep:
 * alpha, diag(K^-1), p = K^-1 (alpha / diag) and diag(M), M = K^-1 diag(c2) K^-1, per evaluation (T x n each); the
 * sub-kernels' gradient components (T x sum n_thetas) and the window row sums h_m(i) = sum_j (sym(p alpha^T) - M)_ij K_m,ij g_m(j)
 * (T x nk x n; window parameters: 2 sum_i dw_i (h_1 - h_0)_i on the host; WhiteNoise: 2 s^2 sum(p o alpha - diag M)); with
 * hw_host (T x nk x 2 x n, see gpmi_lml_grad_mix) the row sums use the caller's weights, T x nk x 2 x n: any number of regions.
 * Lockstep sizes only (n <= 4096, diagonal data errors). */
int gpmi_loo_grad_batch_mix(gpmi_ctx* ctx, int nk, const int* kernels, int64_t T, const double* thetas_host,
                            const int* n_thetas, const double* g_host, const double* hw_host,
                            const double* extra_diag_host, const double* mus_host, const double* mu_const_host,
                            double* alpha_host, double* ikdiag_host, double* p_host, double* mdiag_host,
                            double* grad_thetas_host, double* hrows_host, int* info);
/* alpha and diag(K^-1) at arbitrary hyper-parameters: the O(n^3) part of loo_likelihood (regression.py:468-487) */
int gpmi_loo_terms_mix(gpmi_ctx* ctx, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                       const double* g_host, double extra_diag, const double* mu_host, double* alpha_host,
                       double* ikdiag_host, int* info);
/* prediction with the model of gpmi_fit_mix: gq_host (nk x m) are the weights of the query points;
 * mu* = k.alpha (the host adds the mean function), negsumsq = -|L^-1 k|^2 (the host adds K_qq[0, 0]) */
int gpmi_predict_mix(gpmi_ctx* ctx, const double* pts_host, int64_t m, const double* gq_host,
                     double* mu_host, double* negsumsq_host);
/* build_posterior (regression.py:421-449) with the model of gpmi_fit_mix: mu = K_qx alpha, cov = K_qq - Q^T Q */
int gpmi_posterior_mix(gpmi_ctx* ctx, const double* pts_host, int64_t m, const double* gq_host,
                       double* mu_host, double* cov_host);

/* ---- Gaussian-process linear inversion (inference/gp/inversion.py) --------------------
 * The model parameters (n of them, at the positions given to gpmi_set_data as x; y / noise of that call are
 * unused) have the GP prior N(mu, K(theta)); the data y (m values, independent errors y_err) are
 * A mu_true + noise with the m x n model matrix A (row-major).  All entry points work on
 * J = A K A^T + diag(y_err^2) = L L^T (inversion.py:189, 198) with the same kernels as the regression path. */
int gpmi_linv_set(gpmi_ctx* ctx, const double* A_host, int64_t m, const double* y_host,
                  const double* y_err_host);
/* Replaces GpLinearInverter.marginal_likelihood (inversion.py:177-191): -1/2 |L^-1 (y - A mu)|^2 - sum ln L_ii */
int gpmi_linv_lml(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta, double extra_diag,
                  const double* mu_host, double* lml, int* info);
/* Replaces GpLinearInverter.marginal_likelihood_gradient (inversion.py:193-217).  grad_theta: the n_theta
 * covariance-parameter gradients 1/2 sum (alpha alpha^T - J^-1) o (A dK_j A^T), evaluated as
 * 1/2 sum (w w^T - A^T J^-1 A) o dK_j with w = A^T alpha; trace_q = sum_a (w_a^2 - (A^T J^-1 A)_aa) for an
 * additive WhiteNoise term; at_alpha (n values) = w, from which the host forms the mean-parameter gradients
 * sum_i alpha_i (A dmu_j)_i = w . dmu_j. */
int gpmi_linv_lml_grad(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta, double extra_diag,
                       const double* mu_host, double* lml, double* grad_theta, double* trace_q,
                       double* at_alpha_host, int* info);
/* Replaces GpLinearInverter.calculate_posterior / calculate_posterior_mean (inversion.py:138-175):
 * mean = mu + K A^T J^-1 (y - A mu), cov = K - K A^T J^-1 A K (the Woodbury form of the reference's
 * solve(I + K A^T Sigma^-1 A, K)); cov_host may be NULL. */
int gpmi_linv_posterior(gpmi_ctx* ctx, int kernel, const double* theta_host, int n_theta,
                        double extra_diag, const double* mu_host, double* mean_host, double* cov_host,
                        int* info);
/* The same three with a prior covariance the CALLER evaluated (K_host: n x n, row-major, symmetric) - any
 * CovarianceFunction object the reference accepts as `prior_covariance_function` (inversion.py:117-127, plugin ABC
 * covariance.py:8-44): the host calls the object's own build_covariance / covariance_and_gradients, the device does
 * A K A^T + Sigma, the factorisation, the solves and the inverse.  gpmi_linv_lml_grad_dense returns
 * G = A^T J^-1 A (n x n) and w = A^T alpha (at_alpha_host, n): grad_j = 1/2 sum (w w^T - G) o dK_j on the host
 * (inversion.py:202-216 with Q = alpha alpha^T - J^-1 folded through A). */
int gpmi_linv_lml_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* lml, int* info);
int gpmi_linv_lml_grad_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* lml,
                             double* G_host, double* at_alpha_host, int* info);
int gpmi_linv_posterior_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* mean_host,
                              double* cov_host, int* info);

/* ---- multi-GPU result gather (RCCL over xGMI) ----------------------------------------
 * Replaces the result return of multiprocessing.Pool.map (regression.py:600-601) and the
 * (theta, log-prob) messages of the tempering processes (mcmc/parallel.py:195-201).  One process per
 * GPU.  Rank 0 calls gpmi_comm_unique_id (128 bytes), the caller distributes the id over any host
 * channel, every rank calls gpmi_comm_init, then gpmi_comm_allgather gathers `count` doubles per
 * rank (recv_host: world * count, in rank order).  librccl is loaded on first use.
 * No data-path collective exists: the units (hyper-parameter vectors, optimiser starts, tempering ladders) are
 * independent; the communicator carries the data set once at start-up and a few doubles per unit at the end. */
int gpmi_comm_unique_id(char* id_out_128);
int gpmi_comm_init(gpmi_ctx* ctx, int rank, int world, const char* id_128);
int gpmi_comm_allgather(gpmi_ctx* ctx, const double* send_host, double* recv_host, int64_t count);
/* Start-up distribution of the data set: `count` doubles of rank `root`'s buf_host to every rank's buf_host (one
 * ncclBroadcast).  Replaces the pickling of the whole regressor into every worker process
 * (regression.py:597-601, mcmc/parallel.py:127-136): x, y, y_err are all a rank needs to build its own. */
int gpmi_comm_broadcast(gpmi_ctx* ctx, double* buf_host, int64_t count, int root);
/* The number of ranks RCCL sees in the communicator (ncclCommCount). */
int gpmi_comm_count(gpmi_ctx* ctx, int* ranks);
int gpmi_comm_destroy(gpmi_ctx* ctx);

/* Append ONE training point to the fitted model at unchanged hyper-parameters in O(n^2): the new row of the Cholesky
 * factor is l = L^-1 k(X, x_new), l_nn = sqrt(k_nn - l.l) (one triangular sweep), alpha is re-solved (two sweeps).
 * The reference has no counterpart: GpOptimiser.add_evaluation re-fits from scratch (optimisation.py:177-186,
 * O(n^3) per added point even when the hyper-parameters are kept).  Needs a fit by gpmi_fit (SE / RQ, diagonal
 * data errors) and free capacity (GPMI_OPT_RESERVE_POINTS).
 *   x_new_host : d values    mu_host : n + 1 prior means (the old points' first)    alpha_host : n + 1 out
 *   info : 0, or n + 1 if the enlarged matrix is not positive definite (the model is then left unchanged) */
int gpmi_append_point(gpmi_ctx* ctx, const double* x_new_host, double y_new, double noise_var_new,
                      const double* mu_host, double* alpha_host, double* logdet_host, int* info);
/* points the handle has room for (n + free capacity) */
int gpmi_capacity(gpmi_ctx* ctx, int64_t* capacity);

/* ---- dense entry points: covariance functions that only implement the plugin ABC -------------------------
 * Reference: CovarianceFunction (inference/gp/covariance.py:8-44) is an open plugin contract and GpRegressor
 * (regression.py:134-155) accepts any object implementing it.  For a kernel the library has no device code for, the
 * host evaluates the plugin's own build_covariance / __call__ and passes the dense matrices; every O(N^3) step
 * (numpy.linalg.cholesky regression.py:241, solve_triangular :242-244 / :213 / :447, the explicit inverse :556-557)
 * runs on the device.  gpmi_set_data must have been called (x is unused, y and n are).
 *   K_host : n x n row-major, the complete K(theta) + Sigma (only its lower triangle is read) */
/* fit (regression.py:218-244) on lane 0; afterwards gpmi_predict_dense / gpmi_solve_rows / gpmi_get_L /
 * gpmi_loo_diag use this factor */
int gpmi_fit_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* alpha_host,
                   double* logdet_host, int* info);
/* LML (regression.py:528-542) of a dense K on a scratch lane; alpha_host (n) and iK_host (n x n, K^-1, what
 * marginal_likelihood_gradient :555-565 and the LOO expressions :460-526 contract with the plugin's own dK) are
 * optional (NULL) */
int gpmi_lml_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* lml_host,
                   double* alpha_host, double* iK_host, int* info);
/* leave-one-out pieces (regression.py:468-526) of a dense K on a scratch lane: alpha, diag(K^-1), and for the
 * gradient the two parameter-independent pieces p = K^-1 c1 (n) and W = K^-1 diag(c2) K^-1 (n x n) with
 * c1 = alpha var, c2 = var (1 + var alpha^2) / 2, var = 1 / diag(K^-1): d LOO / d theta_j = p . (dK_j alpha) -
 * sum dK_j o W for the plugin's own dK_j (p_host, W_host may be NULL) */
int gpmi_loo_dense(gpmi_ctx* ctx, const double* K_host, const double* mu_host, double* alpha_host,
                   double* ikdiag_host, double* p_host, double* W_host, int* info);
/* predict pieces (regression.py:208-214) for m query points given their cross-covariances Kq (m x n):
 * kalpha_host[i] = Kq[i] . alpha, sumsq_host[i] = |L^-1 Kq[i]|^2 (either may be NULL) */
int gpmi_predict_dense(gpmi_ctx* ctx, const double* Kq_host, int64_t m, double* kalpha_host, double* sumsq_host);
/* X = Q L^-T for m right-hand sides given as rows (m x n), and / or their Gram matrix X X^T (m x m): the building
 * block of build_posterior (regression.py:447-448) and of gradient / spatial_derivatives (:374-383, :410-417) */
int gpmi_solve_rows(gpmi_ctx* ctx, const double* Q_host, int64_t m, double* X_host, double* gram_host);

/* ---- instrumentation ---------------------------------------------------------------
 * HIP-event timing on the handle's own stream (torch.cuda.Event would not see it). */
int gpmi_timer_start(gpmi_ctx* ctx);
int gpmi_timer_stop(gpmi_ctx* ctx, float* ms);
/* per-kernel-class accounting: when enabled every launch of the class is bracketed by events; the two
 * trailing-update classes (SYRK, SYRK_REST, SYRK_SLICE) are timed by in-kernel stamps instead and cost nothing */
/* Host-only (no device call): the task lists of the flag-ordered factorisation of a tail of `m` tile rows dealt to `nwg`
 * workgroups (csrc/potrf_flow.hip; replaces nothing in the reference - it is the schedule of numpy.linalg.cholesky's
 * replacement, regression.py:241).  out: 8 ints per task {type (0 panel TRSM slab, 1 one-column update of a 64 x 64
 * sub-tile, 2 K = 512 chunk), i, j, k, s, flag increment, owner workgroup, 0}, list after list; out == NULL: only the
 * count.  tests/test_flow_cpu.py replays them with NumPy tile operations. */
int gpmi_flow_task_lists(int m, int nwg, int64_t cap, int32_t* out, int64_t* ntasks);

#define GPMI_PROF_KBUILD 0  /* covariance build            (HBM-write bound) */
#define GPMI_PROF_SYRK 1    /* potrf trailing SYRK/GEMM    (fp64 MFMA bound) */
#define GPMI_PROF_PANEL 2   /* potrf diagonal block + panel TRSM (latency bound) */
#define GPMI_PROF_SOLVE 3   /* single right-hand side triangular sweeps (HBM-read bound) */
#define GPMI_PROF_SYRK_REST 4 /* trailing-update launches that do not run the 128 x 128-tile kernel: the 64 x 64-tile
                                 remainder of a split launch and the launches with fewer than 384 tiles */
#define GPMI_PROF_TRSM 5    /* many right-hand side solves: predict, posterior, L^-T (fp64 MFMA bound) */
#define GPMI_PROF_SYRK_SLICE 6 /* the tiles of a trailing update that run, concurrently with the rest, on the 32 CUs
                                  reserved for the panel chain (same 128 x 128-tile kernel as SYRK) */
#define GPMI_PROF_FLOW 7    /* the flag-ordered tile-task launch of the chain-bound part of a factorisation (potrf_flow.hip):
                               FLOPs of its update tasks over the launch's whole duration - it waits for the panel
                               chain most of the time, so this is the overlap achieved, not a kernel rate */
#define GPMI_PROF_NCLASS 8
/* on = 0: off; 1: every class; otherwise a class bitmask shifted left by one (2 << klass) */
int gpmi_profile_enable(gpmi_ctx* ctx, int on);
/* accumulated since the last reset: launches, total ms, algorithmic flops and bytes */
int gpmi_profile_read(gpmi_ctx* ctx, int klass, int64_t* launches, double* ms, double* flops,
                      double* bytes);
int gpmi_profile_reset(gpmi_ctx* ctx);
/* shader clock (GHz) held during the stamped trailing-update launches since the last reset: cycles (s_memtime)
 * over wall time (s_memrealtime) of the first workgroups of every launch; 0 if nothing was stamped */
int gpmi_profile_clock(gpmi_ctx* ctx, double* ghz);

/* ---- device-pointer entry points (kernel tests / micro-benchmarks) -------------------
 * Matrices are row-major with leading dimension ld (multiple of 128); n multiple of 128. */
int gpmi_dev_alloc(gpmi_ctx* ctx, int64_t bytes, void** ptr_dev);
int gpmi_dev_free(gpmi_ctx* ctx, void* ptr_dev);
int gpmi_dev_upload(gpmi_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int gpmi_dev_download(gpmi_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
/* in-place lower Cholesky of the n x n matrix at A_dev (upper triangle untouched) */
int gpmi_dev_potrf(gpmi_ctx* ctx, double* A_dev, int64_t n, int64_t ld, int* info);
/* C (m x n, lower tiles only if lower != 0) -= A (m x k) * B (n x k)^T */
int gpmi_dev_gemm_nt(gpmi_ctx* ctx, double* C_dev, int64_t ldc, const double* A_dev, int64_t lda,
                     const double* B_dev, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower);

#ifdef __cplusplus
}
#endif
#endif /* GPMI_H */
