// C-ABI of libgpmi (include/gpmi.h): appending one training point at fixed hyper-parameters, and the dense entry points for covariance functions
// that only implement the plugin ABC.
// (split from api.hip in round 4; the handle, the lanes and the helpers these entry points are built from: api.hip,
// api_internal.h)
#include "api_internal.h"

// ---- append one training point at fixed hyper-parameters (O(n^2)) ------------------------------------------------
namespace {

// row n of L <- [l_0 .. l_{n-1}, sqrt(knn - l.l)]; red[0] = the new pivot (<= 0: not positive definite, nothing written)
__global__ __launch_bounds__(1024) void append_row_kernel(double* __restrict__ L, int64_t ld, int64_t n,
                                                          const double* __restrict__ l, double knn,
                                                          double* __restrict__ red) {
  __shared__ double part[16];
  __shared__ double pivot;
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 1024) s = fma(l[j], l[j], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    pivot = knn - t;
    red[0] = pivot;
  }
  __syncthreads();
  if (!(pivot > 0.0)) return;
  for (int64_t j = threadIdx.x; j < n; j += 1024) L[n * ld + j] = l[j];
  if (threadIdx.x == 0) L[n * ld + n] = sqrt(pivot);
}

// inverse of the 128 x 128 diagonal block that holds row n: only its row i = n - r0 changes (the rows below are
// still identity rows): invD[i][t] = (delta_it - sum_{c<i} T[i][c] invD[c][t]) / T[i][i]
__global__ void append_invd_kernel(const double* __restrict__ L, int64_t ld, int64_t n, double* __restrict__ invD,
                                   const double* __restrict__ red) {
  if (!(red[0] > 0.0)) return;
  const int64_t r0 = n / GPMI_NB * GPMI_NB;
  const int i = (int)(n - r0), t = threadIdx.x;
  double* D = invD + (n / GPMI_NB) * GPMI_NB * GPMI_NB;
  double acc = (t == i) ? 1.0 : 0.0;
  for (int cc = 0; cc < i; ++cc) acc = fma(-L[n * ld + r0 + cc], D[cc * GPMI_NB + t], acc);
  D[i * GPMI_NB + t] = acc / L[n * ld + n];
}

}  // namespace

extern "C" {

int gpmi_capacity(gpmi_ctx* c, int64_t* capacity) {
  if (!c || !capacity) return GPMI_ERR_ARG;
  *capacity = c->np;
  return GPMI_OK;
}

int gpmi_append_point(gpmi_ctx* c, const double* x_new, double y_new, double noise_var_new, const double* mu,
                      double* alpha_out, double* logdet_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && c->fit_params.kernel >= 0 && c->mix_nk == 0, "gpmi_append_point needs a fit by gpmi_fit (SE / RQ)");
  ARGCHK(c, !c->ycov, "gpmi_append_point: diagonal data errors only");
  ARGCHK(c, x_new && mu, "x_new / mu is NULL");
  ARGCHK(c, c->n < c->np, "no capacity left: set GPMI_OPT_RESERVE_POINTS before gpmi_set_data");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int64_t n = c->n;
  const KParams p = c->fit_params;
  if (int rc = ensure_query_ws(c, GPMI_NB)) return rc;
  HIPCHK(c, hipMemcpyAsync(c->x + n * c->d, x_new, sizeof(double) * c->d, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->pts, x_new, sizeof(double) * c->d, hipMemcpyHostToDevice, s));
  // k = K(x_new, X) against the n points present (zeros beyond), then l = L^-1 k (rows >= n of L are identity rows)
  launch_kbuild_cross(s, p, c->pts, 1, GPMI_NB, c->x, n, c->np, c->Q, c->ld);
  double* lvec = L.vec + 2 * c->np;
  HIPCHK(c, hipMemsetAsync(L.info, 0, sizeof(int), s));
  trsv_forward(c, s, L.A, c->np, c->ld, L.invD, c->Q, lvec, L.info);
  // K_nn = a^2 (1 + 1e-12) + WhiteNoise + data variance (covariance.py:254-255, regression.py:239)
  const double knn = p.a2 * (1.0 + 1e-12) + p.extra_diag + noise_var_new;
  hipLaunchKernelGGL(append_row_kernel, dim3(1), dim3(1024), 0, s, L.A, c->ld, n, lvec, knn, L.red + 4);
  hipLaunchKernelGGL(append_invd_kernel, dim3(1), dim3(GPMI_NB), 0, s, L.A, c->ld, n, L.invD, L.red + 4);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red + 4, L.red + 4, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (!(L.h_red[4] > 0.0)) {
    if (info) *info = (int)(n + 1);
    return GPMI_OK;  // nothing was written: the fitted model is unchanged
  }
  // the point is in: data vectors, then alpha = L^-T L^-1 (y - mu) and the log-determinant as in gpmi_fit
  HIPCHK(c, hipMemcpyAsync(c->y + n, &y_new, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->noise + n, &noise_var_new, sizeof(double), hipMemcpyHostToDevice, s));
  c->n = n + 1;
  L.inv2_valid = false;
  double* mu_dev = L.vec + 3 * c->np;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  launch_residual(s, c->y, mu_dev, 0.0, L.vec + 2 * c->np, c->n, c->np);
  trsv_forward(c, s, L.A, c->np, c->ld, L.invD, L.vec + 2 * c->np, L.vec, L.info);
  launch_lml_reduce(s, L.vec, L.A, c->ld, c->np, L.red);
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, c->alpha, L.info);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out) HIPCHK(c, hipMemcpyAsync(alpha_out, c->alpha, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  INFOCHK(c, L.h_info[0]);
  if (logdet_out) *logdet_out = L.h_red[1];
  if (info) *info = 0;
  return GPMI_OK;
}

}  // extern "C"

// ---- dense entry points: covariance functions that only implement the plugin ABC ---------------------------
// Reference: CovarianceFunction (covariance.py:8-44) is an open plugin contract; GpRegressor accepts any object
// that implements it.  For such kernels the host evaluates the plugin's own build_covariance / __call__ /
// covariance_and_gradients (there is no device code for an unknown kernel) and hands the dense matrices over;
// everything of O(N^3) - potrf, solves, K^-1, the many-right-hand-side solves of predict / posterior - runs on the
// device exactly as for the built-in kernels.  No CPU solve anywhere.
namespace {

// zero the padding rows / columns of an np x ld matrix whose n x n block has just been uploaded, identity on the
// padded diagonal (chol(blockdiag(K, I)) = blockdiag(L, I))
__global__ void dense_pad_kernel(double* __restrict__ A, int64_t ld, int64_t n, int64_t np) {
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np) return;
  if (i >= n || j >= n) A[i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// rows [rows_valid, rows_padded) and the columns [n, np) of every row of a query panel <- 0
__global__ void panel_pad_kernel(double* __restrict__ Q, int64_t ld, int64_t rows_valid, int64_t n, int64_t np) {
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np) return;
  if (i >= rows_valid || j >= n) Q[i * ld + j] = 0.0;
}

int upload_dense_matrix(gpmi_ctx* c, Lane& L, const double* K_host) {
  hipStream_t s = L.stream;
  HIPCHK(c, hipMemcpy2DAsync(L.A, sizeof(double) * c->ld, K_host, sizeof(double) * c->n, sizeof(double) * c->n,
                             (size_t)c->n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(dense_pad_kernel, dim3((unsigned)((c->np + 255) / 256), (unsigned)c->np), dim3(256), 0, s, L.A,
                     c->ld, c->n, c->np);
  return GPMI_OK;
}

// factorise the matrix already in L.A, forward-solve the residual, reduce (the dense twin of
// enqueue_factor_and_forward)
int enqueue_dense_factor_and_forward(gpmi_ctx* c, Lane& L, const double* mu_dev, int slot) {
  hipStream_t s = L.stream;
  L.inv2_valid = false;
  HIPCHK(c, hipMemsetAsync(L.info + slot, 0, sizeof(int), s));
  potrf_lower(c, L, L.A, c->np, c->ld, L.invD, L.info + slot, true);
  launch_residual(s, c->y, mu_dev, 0.0, L.vec + 2 * c->np, c->n, c->np);
  trsv_forward(c, s, L.A, c->np, c->ld, L.invD, L.vec + 2 * c->np, L.vec, L.info + slot);
  launch_lml_reduce(s, L.vec, L.A, c->ld, c->np, L.red + 2 * slot);
  HIPCHK(c, hipGetLastError());
  return GPMI_OK;
}

int upload_query_panel(gpmi_ctx* c, hipStream_t s, double* Q, const double* rows_host, int64_t mc, int64_t mp) {
  HIPCHK(c, hipMemcpy2DAsync(Q, sizeof(double) * c->ld, rows_host, sizeof(double) * c->n, sizeof(double) * c->n,
                             (size_t)mc, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(panel_pad_kernel, dim3((unsigned)((c->np + 255) / 256), (unsigned)mp), dim3(256), 0, s, Q, c->ld,
                     mc, c->n, c->np);
  return GPMI_OK;
}

}  // namespace

extern "C" {

int gpmi_fit_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* alpha_out, double* logdet_out,
                   int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, K_host && mu, "K / mu is NULL");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = upload_dense_matrix(c, L, K_host)) return rc;
  if (int rc = enqueue_dense_factor_and_forward(c, L, mu_dev, 0)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, c->alpha, L.info);
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out) HIPCHK(c, hipMemcpyAsync(alpha_out, c->alpha, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (logdet_out) *logdet_out = L.h_red[1];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  c->fit_params = KParams{};
  c->fit_params.kernel = -1;  // dense: the kernel-specific entry points (gpmi_predict, ...) do not apply
  c->fitted = (L.h_info[0] == 0);
  c->mix_nk = 0;
  return GPMI_OK;
}

int gpmi_lml_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* lml, double* alpha_out,
                   double* iK_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, K_host && mu && lml, "K / mu / lml is NULL");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = upload_dense_matrix(c, L, K_host)) return rc;
  if (int rc = enqueue_dense_factor_and_forward(c, L, mu_dev, 0)) return rc;
  if (alpha_out || iK_out) trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  if (iK_out) {
    // K^-1 = L^-T L^-1 (regression.py:556-557): L^-T by forward substitution on the identity, then the k-skipped SYRK
    if (int rc = ensure_second_matrix(c, L)) return rc;
    if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, true, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, (int)(c->np / GPMI_NB),
                (int)(c->np / GPMI_NB), (int)c->np);
    launch_mirror_lower(s, L.A, c->ld, c->np);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy2DAsync(iK_out, sizeof(double) * c->n, L.A, sizeof(double) * c->ld, sizeof(double) * c->n,
                               (size_t)c->n, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out) HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  INFOCHK(c, L.h_info[0]);
  // -1/2 v.v - sum ln L_ii (regression.py:539); the caller applies the -1e50 convention (regression.py:540-542)
  *lml = (L.h_info[0] == 0) ? (-0.5 * L.h_red[0] - L.h_red[1]) : -1e50;
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_loo_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* alpha_out, double* ikdiag,
                   double* p_out, double* W_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, K_host && mu && alpha_out && ikdiag, "NULL argument");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  const int64_t need = 4 * c->np;
  if (L.gws_doubles < need) {
    if (L.gws) (void)hipFree(L.gws);
    L.gws = nullptr;
    L.gws_doubles = 0;
    HIPCHK(c, hipMalloc(&L.gws, sizeof(double) * need));
    L.gws_doubles = need;
  }
  hipStream_t s = L.stream;
  const int nt = (int)(c->np / GPMI_NB);
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  double *diag_dev = L.gws, *c1_dev = L.gws + c->np, *sc2_dev = L.gws + 2 * c->np, *p_dev = L.gws + 3 * c->np;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = upload_dense_matrix(c, L, K_host)) return rc;
  if (int rc = enqueue_dense_factor_and_forward(c, L, mu_dev, 0)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_rows_sumsq(s, L.B2, c->ld, c->np, c->np, 0.0, diag_dev);  // = -diag(K^-1)   (regression.py:503)
  launch_negate(s, diag_dev, c->np);
  HIPCHK(c, hipMemcpyAsync(ikdiag, diag_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  if (p_out || W_out) {
    // the gradient's two parameter-independent pieces (regression.py:507-514 regrouped):
    //   sum_i c1_i (K^-1 dK alpha)_i = p . (dK alpha),  p = K^-1 c1
    //   sum_i c2_i (K^-1 dK K^-1)_ii = sum dK o W,       W = K^-1 diag(c2) K^-1 = G G^T, G = K^-1 diag(sqrt c2)
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, true, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, nt, nt, (int)c->np);
    launch_mirror_lower(s, L.A, c->ld, c->np);
    launch_loo_vectors(s, alpha_dev, diag_dev, c1_dev, sc2_dev, c->n, c->np);
    launch_rows_dot(s, L.A, c->ld, c->np, c->np, c1_dev, p_dev);
    if (p_out) HIPCHK(c, hipMemcpyAsync(p_out, p_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
    if (W_out) {
      launch_scale_columns(s, L.A, sc2_dev, L.B2, c->ld, c->np);
      launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, false, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, nt, nt, (int)c->np);
      launch_mirror_lower(s, L.A, c->ld, c->np);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpy2DAsync(W_out, sizeof(double) * c->n, L.A, sizeof(double) * c->ld, sizeof(double) * c->n,
                                 (size_t)c->n, hipMemcpyDeviceToHost, s));
    }
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_predict_dense(gpmi_ctx* c, const double* Kq_host, int64_t m, double* kalpha_out, double* sumsq_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_predict_dense needs a successful fit");
  ARGCHK(c, Kq_host && m > 0, "Kq is NULL or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int64_t chunk = 2048;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t mp = round_up(mc, GPMI_NB);
    if (int rc = ensure_query_ws(c, mp)) return rc;
    if (int rc = upload_query_panel(c, s, c->Q, Kq_host + m0 * c->n, mc, mp)) return rc;
    double* mu_dev = c->pvec;
    double* ss_dev = c->pvec + mp;
    if (kalpha_out) launch_rows_dot(s, c->Q, c->ld, mp, c->np, c->alpha, mu_dev);
    if (sumsq_out) {
      if (int rc = ensure_inv2(c, L, s)) return rc;
      trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);
      launch_rows_sumsq(s, c->Q2, c->ld, mp, c->np, 0.0, ss_dev);  // - |L^-1 k|^2
    }
    HIPCHK(c, hipGetLastError());
    if (kalpha_out) HIPCHK(c, hipMemcpyAsync(kalpha_out + m0, mu_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    if (sumsq_out) HIPCHK(c, hipMemcpyAsync(sumsq_out + m0, ss_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  if (sumsq_out)
    for (int64_t i = 0; i < m; ++i) sumsq_out[i] = -sumsq_out[i];
  return GPMI_OK;
}

int gpmi_solve_rows(gpmi_ctx* c, const double* Q_host, int64_t m, double* X_host, double* gram_host) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_solve_rows needs a successful fit");
  ARGCHK(c, Q_host && m > 0 && (X_host || gram_host), "Q is NULL, m <= 0 or nothing requested");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int64_t mp = round_up(m, GPMI_NB);
  ARGCHK(c, mp <= 8192, "at most 8192 right-hand sides per call");
  if (int rc = ensure_query_ws(c, mp)) return rc;
  if (int rc = upload_query_panel(c, s, c->Q, Q_host, m, mp)) return rc;
  if (int rc = ensure_inv2(c, L, s)) return rc;
  trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);  // X = Q L^-T
  HIPCHK(c, hipGetLastError());
  if (X_host)
    HIPCHK(c, hipMemcpy2DAsync(X_host, sizeof(double) * c->n, c->Q2, sizeof(double) * c->ld, sizeof(double) * c->n,
                               (size_t)m, hipMemcpyDeviceToHost, s));
  if (gram_host) {
    // G = X X^T (m x m): what posterior covariances are made of (regression.py:447-448)
    double* G = nullptr;
    const int64_t ldg = mp + 32;
    HIPCHK(c, hipMalloc(&G, sizeof(double) * mp * ldg));
    launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, G, ldg, c->Q2, c->ld, c->Q2, c->ld, (int)(mp / GPMI_NB),
                   (int)(mp / GPMI_NB), (int)c->np);
    hipError_t e = hipMemcpy2DAsync(gram_host, sizeof(double) * m, G, sizeof(double) * ldg, sizeof(double) * m,
                                    (size_t)m, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(G);
    HIPCHK(c, e);
  }
  HIPCHK(c, hipStreamSynchronize(s));
  return GPMI_OK;
}

}  // extern "C"

