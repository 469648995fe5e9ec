// C-ABI of libgpmi (include/gpmi.h): Gaussian-process linear inversion (GpLinearInverter).
// (split from api.hip in round 4; the handle, the lanes and the helpers these entry points are built from: api.hip,
// api_internal.h)
#include "api_internal.h"

// ---- Gaussian-process linear inversion --------------------------------------------------------------
// (called by api.hip when a handle's data set is replaced or the handle destroyed)
void linv_free(LinvState& S) {
  for (double** p : {&S.A, &S.At, &S.y, &S.sig2, &S.zero, &S.K, &S.T, &S.J, &S.J2, &S.Q, &S.X, &S.invD, &S.inv2,
                     &S.inv2_t, &S.panel, &S.vec, &S.gws}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  S = LinvState();
}

namespace {

__global__ void linv_add_diag_kernel(double* __restrict__ J, int64_t ld, const double* __restrict__ sig2,
                                     int64_t mp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mp) J[i * ld + i] += sig2[i];
}


int linv_alloc(gpmi_ctx* c, double** p, int64_t doubles) {
  if (*p) return GPMI_OK;
  HIPCHK(c, hipMalloc(p, sizeof(double) * doubles));
  ZERO_SYNC(c, *p, sizeof(double) * doubles);  // (the inverses' upper 16-blocks rely on it, see lane_alloc)
  return GPMI_OK;
}

// J = A K(theta) A^T + Sigma -> L, v = L^-1 (y - A mu) in vec[0:mp], red = {v.v, sum ln L_ii} in lanes[0].red
// prior covariance: built on the device from kernel parameters (K_host == nullptr), or a dense n x n matrix the host
// evaluated with the covariance object's own build_covariance (user-defined kernels, plugin ABC covariance.py:8-44)
int linv_factor(gpmi_ctx* c, const KParams& p, const double* mu_host, const double* K_host = nullptr) {
  LinvState& S = c->linv;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int nt = (int)(c->np / GPMI_NB), mt = (int)(S.mp / GPMI_NB);
  const int64_t vmax = S.mp > c->np ? S.mp : c->np;
  double* v = S.vec;               // mp
  double* r = S.vec + vmax;        // mp
  double* mu_dev = S.vec + 2 * vmax;  // np
  double* amu = S.vec + 3 * vmax;  // mp
  HIPCHK(c, hipMemsetAsync(L.info, 0, sizeof(int), s));
  HIPCHK(c, hipMemsetAsync(mu_dev, 0, sizeof(double) * c->np, s));
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu_host, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  // prior covariance, both triangles (it is a GEMM operand), identity in the padding (A's padding is zero)
  if (K_host) {
    launch_set_identity(s, S.K, c->ld, c->np);
    HIPCHK(c, hipMemcpy2DAsync(S.K, sizeof(double) * c->ld, K_host, sizeof(double) * c->n, sizeof(double) * c->n, c->n,
                               hipMemcpyHostToDevice, s));
  } else {
    launch_kbuild_square(s, p, c->x, c->n, c->np, S.zero, S.K, c->ld, false);
  }
  launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, S.T, c->ld, S.A, c->ld, S.K, c->ld, mt, nt, (int)c->np);   // T = A K
  launch_gemm_nt(s, TILES_LOWER, OP_ASSIGN, S.J, S.ldm, S.T, c->ld, S.A, c->ld, mt, mt, (int)c->np);  // J = T A^T
  hipLaunchKernelGGL(linv_add_diag_kernel, dim3((unsigned)((S.mp + 255) / 256)), dim3(256), 0, s, S.J, S.ldm,
                     S.sig2, S.mp);
  potrf_lower(c, L, S.J, S.mp, S.ldm, S.invD, L.info);
  launch_rows_dot(s, S.A, c->ld, S.mp, c->np, mu_dev, amu);            // A mu
  launch_residual(s, S.y, amu, 0.0, r, S.m, S.mp);                      // y - A mu (zero padded)
  trsv_forward(c, s, S.J, S.mp, S.ldm, S.invD, r, v, L.info);
  launch_lml_reduce(s, v, S.J, S.ldm, S.mp, L.red);
  HIPCHK(c, hipGetLastError());
  return GPMI_OK;
}

int linv_ready(gpmi_ctx* c) {
  ARGCHK(c, c->n > 0, "gpmi_set_data (parameter positions) has not been called");
  ARGCHK(c, c->linv.m > 0, "gpmi_linv_set has not been called");
  return GPMI_OK;
}

}  // namespace

extern "C" {

int gpmi_linv_set(gpmi_ctx* c, const double* A, int64_t m, const double* y, const double* y_err) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data (parameter positions) has not been called");
  ARGCHK(c, A && y && y_err && m > 0, "A, y, y_err must be non-NULL and m positive");
  if (int rc = set_device(c)) return rc;
  LinvState& S = c->linv;
  linv_free(S);
  S.m = m;
  S.mp = round_up(m, GPMI_NB);
  S.ldm = S.mp + 32;
  const int64_t n = c->n, np = c->np, ld = c->ld, mp = S.mp, ldm = S.ldm;
  const int64_t vmax = mp > np ? mp : np;
  if (int rc = linv_alloc(c, &S.A, mp * ld)) return rc;
  if (int rc = linv_alloc(c, &S.At, np * ldm)) return rc;
  if (int rc = linv_alloc(c, &S.y, mp)) return rc;
  if (int rc = linv_alloc(c, &S.sig2, mp)) return rc;
  if (int rc = linv_alloc(c, &S.zero, np)) return rc;
  if (int rc = linv_alloc(c, &S.K, np * ld)) return rc;
  if (int rc = linv_alloc(c, &S.T, mp * ld)) return rc;
  if (int rc = linv_alloc(c, &S.J, mp * ldm)) return rc;
  if (int rc = linv_alloc(c, &S.invD, (mp / GPMI_NB) * GPMI_NB * GPMI_NB)) return rc;
  if (int rc = linv_alloc(c, &S.vec, 8 * vmax)) return rc;
  ZERO_SYNC(c, S.A, sizeof(double) * mp * ld);
  ZERO_SYNC(c, S.At, sizeof(double) * np * ldm);
  ZERO_SYNC(c, S.y, sizeof(double) * mp);
  ZERO_SYNC(c, S.zero, sizeof(double) * np);
  HIPCHK(c, hipMemcpy2D(S.A, sizeof(double) * ld, A, sizeof(double) * n, sizeof(double) * n, m,
                        hipMemcpyHostToDevice));
  std::vector<double> at((size_t)n * m), s2((size_t)mp, 1.0);
  for (int64_t i = 0; i < m; ++i) {
    for (int64_t j = 0; j < n; ++j) at[(size_t)j * m + i] = A[i * n + j];
    s2[(size_t)i] = y_err[i] * y_err[i];
  }
  HIPCHK(c, hipMemcpy2D(S.At, sizeof(double) * ldm, at.data(), sizeof(double) * m, sizeof(double) * m, n,
                        hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(S.y, y, sizeof(double) * m, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(S.sig2, s2.data(), sizeof(double) * mp, hipMemcpyHostToDevice));
  return GPMI_OK;
}

static int linv_lml_impl(gpmi_ctx* c, const KParams& p, const double* K_host, const double* mu, double* lml,
                         int* info) {
  ARGCHK(c, mu && lml, "mu / lml is NULL");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  if (int rc = linv_factor(c, p, mu, K_host)) return rc;
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  c->fitted = false;  // lane 0's streams / result slots were used; the regression fit (if any) is gone
  *lml = -0.5 * L.h_red[0] - L.h_red[1];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_linv_lml(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                  const double* mu, double* lml, int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  return linv_lml_impl(c, p, nullptr, mu, lml, info);
}

int gpmi_linv_lml_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* lml, int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  ARGCHK(c, K_host, "K is NULL");
  return linv_lml_impl(c, KParams{}, K_host, mu, lml, info);
}

// n_theta >= 0: fused contraction with the kernel's own derivatives (grad_theta, trace_q);
// n_theta < 0 (dense prior): G = A^T J^-1 A goes back to the host (G_host, n x n), which contracts it with the
// covariance object's dK_j
static int linv_lml_grad_impl(gpmi_ctx* c, const KParams& p, int n_theta, const double* K_host, const double* mu,
                              double* lml, double* grad_theta, double* trace_q, double* G_host, double* at_alpha,
                              int* info) {
  ARGCHK(c, mu && lml && (grad_theta || G_host), "mu / lml / output is NULL");
  if (int rc = set_device(c)) return rc;
  LinvState& S = c->linv;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int nt = (int)(c->np / GPMI_NB), mt = (int)(S.mp / GPMI_NB);
  const int64_t vmax = S.mp > c->np ? S.mp : c->np;
  if (int rc = linv_alloc(c, &S.J2, S.mp * S.ldm)) return rc;
  if (int rc = linv_alloc(c, &S.inv2, (int64_t)((mt + 3) / 4) * GPMI_OB * GPMI_OB)) return rc;
  if (int rc = linv_alloc(c, &S.inv2_t, (int64_t)((mt + 3) / 4) * 256 * 256)) return rc;
  if (int rc = linv_alloc(c, &S.panel, S.mp * (GPMI_OB + 32))) return rc;
  const int64_t need = n_theta >= 0 ? grad_ws_doubles(c->np, n_theta) : 0;
  if (S.gws_doubles < need) {
    if (S.gws) (void)hipFree(S.gws);
    S.gws = nullptr;
    S.gws_doubles = 0;
    HIPCHK(c, hipMalloc(&S.gws, sizeof(double) * need));
    S.gws_doubles = need;
  }
  if (int rc = linv_factor(c, p, mu, K_host)) return rc;
  double* v = S.vec;
  double* alpha = S.vec + 4 * vmax;  // mp
  double* w = S.vec + 5 * vmax;      // np
  double* gout = L.red + 16;
  trsv_backward(c, s, S.J, S.mp, S.ldm, S.invD, v, alpha, L.info);   // alpha = J^-1 (y - A mu)
  launch_rows_dot(s, S.At, S.ldm, c->np, S.mp, alpha, w);            // w = A^T alpha
  // J^-1 = L^-T L^-1 (inversion.py:205-206): L^-T by forward substitution on the identity, then a k-skipped SYRK
  build_inv2(s, S.J, S.mp, S.ldm, S.invD, S.inv2, S.inv2_t);
  launch_set_identity(s, S.J2, S.ldm, S.mp);
  trsm_rows_forward(c, s, S.J, S.mp, S.ldm, S.inv2, S.J2, S.mp, true, nullptr, S.panel);
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, S.J, S.ldm, S.J2, S.ldm, S.J2, S.ldm, mt, mt, (int)S.mp);
  launch_mirror_lower(s, S.J, S.ldm, S.mp);
  // A^T J^-1 A: U = J^-1 A (into T), then A^T U (into K; B = U is k-major)
  launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, S.T, c->ld, S.J, S.ldm, S.At, S.ldm, mt, nt, (int)S.mp);
  launch_gemm(s, TILES_RECT, OP_ASSIGN, true, 0, S.K, c->ld, S.At, S.ldm, S.T, c->ld, nt, nt, (int)S.mp);
  if (n_theta >= 0) launch_lml_grad(s, p, n_theta, c->x, c->n, c->np, S.K, c->ld, w, w, S.gws, gout);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (n_theta >= 0)
    HIPCHK(c, hipMemcpyAsync(L.h_red + 16, gout, sizeof(double) * (n_theta + 1), hipMemcpyDeviceToHost, s));
  if (G_host)
    HIPCHK(c, hipMemcpy2DAsync(G_host, sizeof(double) * c->n, S.K, sizeof(double) * c->ld, sizeof(double) * c->n, c->n,
                               hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (at_alpha) HIPCHK(c, hipMemcpyAsync(at_alpha, w, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  c->fitted = false;
  *lml = -0.5 * L.h_red[0] - L.h_red[1];
  for (int j = 0; j < n_theta; ++j) grad_theta[j] = L.h_red[16 + j];
  if (trace_q) *trace_q = L.h_red[16 + n_theta];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_linv_lml_grad(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                       const double* mu, double* lml, double* grad_theta, double* trace_q,
                       double* at_alpha, int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, grad_theta, "grad_theta is NULL");
  return linv_lml_grad_impl(c, p, n_theta, nullptr, mu, lml, grad_theta, trace_q, nullptr, at_alpha, info);
}

int gpmi_linv_lml_grad_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* lml, double* G_host,
                             double* at_alpha, int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  ARGCHK(c, K_host && G_host, "K / G is NULL");
  return linv_lml_grad_impl(c, KParams{}, -1, K_host, mu, lml, nullptr, nullptr, G_host, at_alpha, info);
}

static int linv_posterior_impl(gpmi_ctx* c, const KParams& p, const double* K_host, const double* mu, double* mean,
                               double* cov, int* info) {
  ARGCHK(c, mu && mean, "mu / mean is NULL");
  if (int rc = set_device(c)) return rc;
  LinvState& S = c->linv;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int nt = (int)(c->np / GPMI_NB), mt = (int)(S.mp / GPMI_NB);
  const int64_t vmax = S.mp > c->np ? S.mp : c->np;
  if (int rc = linv_alloc(c, &S.Q, c->np * S.ldm)) return rc;
  if (int rc = linv_alloc(c, &S.X, c->np * S.ldm)) return rc;
  if (int rc = linv_alloc(c, &S.inv2, (int64_t)((mt + 3) / 4) * GPMI_OB * GPMI_OB)) return rc;
  if (int rc = linv_alloc(c, &S.inv2_t, (int64_t)((mt + 3) / 4) * 256 * 256)) return rc;
  if (int rc = linv_factor(c, p, mu, K_host)) return rc;
  double* v = S.vec;
  double* dm = S.vec + 5 * vmax;  // np
  build_inv2(s, S.J, S.mp, S.ldm, S.invD, S.inv2, S.inv2_t);
  // X = K A^T L^-T  (rows = parameters): mean = mu + X v, cov = K - X X^T
  launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, S.Q, S.ldm, S.K, c->ld, S.A, c->ld, nt, mt, (int)c->np);
  trsm_rows_forward(c, s, S.J, S.mp, S.ldm, S.inv2, S.Q, c->np, false, S.X, nullptr);
  launch_rows_dot(s, S.X, S.ldm, c->np, S.mp, v, dm);
  if (cov) launch_gemm_nt(s, TILES_RECT, OP_SUB, S.K, c->ld, S.X, S.ldm, S.X, S.ldm, nt, nt, (int)S.mp);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(mean, dm, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  if (cov)
    HIPCHK(c, hipMemcpy2DAsync(cov, sizeof(double) * c->n, S.K, sizeof(double) * c->ld, sizeof(double) * c->n,
                               c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  c->fitted = false;
  for (int64_t i = 0; i < c->n; ++i) mean[i] += mu[i];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_linv_posterior(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                        const double* mu, double* mean, double* cov, int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  return linv_posterior_impl(c, p, nullptr, mu, mean, cov, info);
}

int gpmi_linv_posterior_dense(gpmi_ctx* c, const double* K_host, const double* mu, double* mean, double* cov,
                              int* info) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = linv_ready(c)) return rc;
  ARGCHK(c, K_host, "K is NULL");
  return linv_posterior_impl(c, KParams{}, K_host, mu, mean, cov, info);
}

}  // extern "C"

