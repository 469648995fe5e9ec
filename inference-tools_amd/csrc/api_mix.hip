// C-ABI of libgpmi (include/gpmi.h): mixture covariance (ChangePoint) and per-point noise hyper-parameters (HeteroscedasticNoise).
// (split from api.hip in round 4; the handle, the lanes and the helpers these entry points are built from: api.hip,
// api_internal.h)
#include "api_internal.h"

// ---- mixture covariance (ChangePoint) -------------------------------------------------------------------
namespace {

// parse the sub-kernels, upload the weights: fills ps[nk] and the device buffers of the context
int mix_prepare(gpmi_ctx* c, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                const double* g_host, KParams* ps) {
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, nk >= 1 && nk <= GPMI_MAX_MIX, "number of sub-kernels out of range (1..4)");
  ARGCHK(c, kernels && thetas && n_thetas && g_host, "NULL argument");
  int off = 0;
  for (int m = 0; m < nk; ++m) {
    if (int rc = make_params(c, kernels[m], thetas + off, n_thetas[m], 0.0, ps[m])) return rc;
    off += n_thetas[m];
  }
  if (int rc = set_device(c)) return rc;
  if (!c->mix_g) HIPCHK(c, hipMalloc(&c->mix_g, sizeof(double) * GPMI_MAX_MIX * c->np));
  if (!c->mix_scratch) HIPCHK(c, hipMalloc(&c->mix_scratch, sizeof(double) * c->np * c->ld));
  if (!c->mix_zero) {
    HIPCHK(c, hipMalloc(&c->mix_zero, sizeof(double) * c->np));
    ZERO_SYNC(c, c->mix_zero, sizeof(double) * c->np);
  }
  if (int rc = gpmi_sync(c)) return rc;  // nothing may still be reading the previous weights
  std::vector<double> g((size_t)nk * c->np);
  for (int m = 0; m < nk; ++m)
    for (int64_t i = 0; i < c->np; ++i)
      g[(size_t)m * c->np + i] = i < c->n ? g_host[(size_t)m * c->n + i] : (m == 0 ? 1.0 : 0.0);
  HIPCHK(c, hipMemcpy(c->mix_g, g.data(), sizeof(double) * nk * c->np, hipMemcpyHostToDevice));
  return GPMI_OK;
}

int ensure_q3(gpmi_ctx* c) {
  if (c->q3_cap >= c->mq_cap && c->Q3) return GPMI_OK;
  if (c->Q3) (void)hipFree(c->Q3);
  c->Q3 = nullptr;
  HIPCHK(c, hipMalloc(&c->Q3, sizeof(double) * c->mq_cap * c->ld));
  c->q3_cap = c->mq_cap;
  return GPMI_OK;
}

// c->Q (mp x ld) = sum_m diag(gq_m) K_m(pts, x) diag(g_m); gq_dev: nk x mp device weights of the query points
void build_mix_cross(gpmi_ctx* c, hipStream_t s, const double* gq_dev, int64_t mc, int64_t mp) {
  for (int m = 0; m < c->mix_nk; ++m) {
    launch_kbuild_cross(s, c->mix_p[m], c->pts, mc, mp, c->x, c->n, c->np, c->Q3, c->ld);
    launch_scale_add(s, c->Q, c->ld, c->Q3, c->ld, gq_dev + (int64_t)m * mp, c->mix_g + (int64_t)m * c->np, mp,
                     c->np, m > 0);
  }
}

}  // namespace

extern "C" {

int gpmi_fit_mix(gpmi_ctx* c, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                 const double* g_host, double extra_diag, const double* mu, double* alpha_out,
                 double* logdet_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams ps[GPMI_MAX_MIX];
  if (int rc = mix_prepare(c, nk, kernels, thetas, n_thetas, g_host, ps)) return rc;
  ARGCHK(c, mu != nullptr, "mu is NULL");
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  const MixEval mx{nk, ps, c->mix_g, extra_diag, c->mix_scratch, c->mix_zero};
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, ps[0], mu_dev, 0.0, 0, true, &mx)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, c->alpha, L.info);
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out)
    HIPCHK(c, hipMemcpyAsync(alpha_out, c->alpha, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (logdet_out) *logdet_out = L.h_red[1];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  c->mix_nk = nk;
  for (int m = 0; m < nk; ++m) c->mix_p[m] = ps[m];
  c->fit_params = ps[0];
  c->fitted = (L.h_info[0] == 0);
  return GPMI_OK;
}

int gpmi_lml_mix(gpmi_ctx* c, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                 const double* g_host, double extra_diag, const double* mu, double* lml, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams ps[GPMI_MAX_MIX];
  if (int rc = ensure_lanes(c, 2)) return rc;
  // the weights of a fitted mixture live in the same device buffer: an evaluation at other hyper-parameters
  // invalidates them for gpmi_predict_mix until the next gpmi_fit_mix (the host wrapper re-fits lazily)
  if (int rc = mix_prepare(c, nk, kernels, thetas, n_thetas, g_host, ps)) return rc;
  ARGCHK(c, mu && lml, "mu / lml is NULL");
  c->fitted = false;
  Lane& L = c->lanes[1];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  const MixEval mx{nk, ps, c->mix_g, extra_diag, c->mix_scratch, c->mix_zero};
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, ps[0], mu_dev, 0.0, 0, true, &mx)) return rc;
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  *lml = (L.h_info[0] == 0) ? (-0.5 * L.h_red[0] - L.h_red[1]) : -1e50;
  return GPMI_OK;
}

int gpmi_lml_grad_mix(gpmi_ctx* c, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                      const double* g_host, const double* hw_host, double extra_diag, const double* mu, double* lml,
                      double* grad_thetas, double* hrows, double* alpha_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams ps[GPMI_MAX_MIX];
  if (int rc = ensure_lanes(c, 2)) return rc;
  if (int rc = mix_prepare(c, nk, kernels, thetas, n_thetas, g_host, ps)) return rc;
  ARGCHK(c, mu && lml && grad_thetas && hrows, "mu / lml / grad_thetas / hrows is NULL");
  c->fitted = false;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  int max_nt = 0, tot_nt = 0;
  for (int m = 0; m < nk; ++m) {
    max_nt = n_thetas[m] > max_nt ? n_thetas[m] : max_nt;
    tot_nt += n_thetas[m];
  }
  const int64_t need = grad_ws_doubles(c->np, max_nt);
  if (L.gws_doubles < need) {
    if (L.gws) (void)hipFree(L.gws);
    L.gws = nullptr;
    L.gws_doubles = 0;
    HIPCHK(c, hipMalloc(&L.gws, sizeof(double) * need));
    L.gws_doubles = need;
  }
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  double* ua = L.vec + 2 * c->np;  // g_m o alpha
  double* gout = L.red + 16;       // (n_theta_m + 1) values per sub-kernel, consecutive
  // row sums (nk x np, or nk x 2 x np with the caller's own weights hw behind them): the scratch matrix is free once
  // K is factorised
  const int nrow = hw_host ? 2 : 1;
  double* hdev = c->mix_scratch;
  double* wdev = c->mix_scratch + (int64_t)2 * GPMI_MAX_MIX * c->np;
  const MixEval mx{nk, ps, c->mix_g, extra_diag, c->mix_scratch, c->mix_zero};
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, ps[0], mu_dev, 0.0, 0, true, &mx)) return rc;
  if (hw_host) {
    HIPCHK(c, hipMemsetAsync(wdev, 0, sizeof(double) * 2 * nk * c->np, s));
    HIPCHK(c, hipMemcpy2DAsync(wdev, sizeof(double) * c->np, hw_host, sizeof(double) * c->n, sizeof(double) * c->n,
                               (size_t)2 * nk, hipMemcpyHostToDevice, s));
  }
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  // K^-1 = L^-T L^-1, lower tiles over L, then both triangles (it is read row-wise below)
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, L.A, c->ld, L.B2, c->ld, L.B2, c->ld,
              (int)(c->np / GPMI_NB), (int)(c->np / GPMI_NB), (int)c->np);
  launch_mirror_lower(s, L.A, c->ld, c->np);
  int goff = 0;
  for (int m = 0; m < nk; ++m) {
    const double* gm = c->mix_g + (int64_t)m * c->np;
    // sub-kernel parameters: 1/2 sum Q o (D_m dK_m D_m) = 1/2 sum (D_m Q D_m) o dK_m, the fused contraction on
    // the scaled inverse with u = v = g_m o alpha
    launch_scale_add(s, L.B2, c->ld, L.A, c->ld, gm, gm, c->np, c->np, false);
    launch_vec_mul(s, gm, alpha_dev, ua, c->np);
    launch_lml_grad(s, ps[m], n_thetas[m], c->x, c->n, c->np, L.B2, c->ld, ua, ua, L.gws, gout + goff);
    goff += n_thetas[m] + 1;
    // window parameters: h_m(i) = sum_j Q_ij K_m,ij g_m(j)  (the host contracts it with d g_m / d phi)
    KParams pm = ps[m];
    pm.extra_diag = 0.0;
    launch_kbuild_square(s, pm, c->x, c->n, c->np, c->mix_zero, L.B2, c->ld, false);
    // (the caller's two window factors of a sub-kernel - rows 2 m, 2 m + 1, np apart - in one pass)
    if (hw_host)
      launch_mix_rowsum(s, L.A, L.B2, c->ld, alpha_dev, wdev + (int64_t)m * 2 * c->np, hdev + (int64_t)m * 2 * c->np, c->n, 1, 0,
                        0, 0, nullptr, 0, c->np);
    else
      launch_mix_rowsum(s, L.A, L.B2, c->ld, alpha_dev, gm, hdev + (int64_t)m * c->np, c->n);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_red + 16, gout, sizeof(double) * goff, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  for (int row = 0; row < nk * nrow; ++row)
    HIPCHK(c, hipMemcpyAsync(hrows + (int64_t)row * c->n, hdev + (int64_t)row * c->np, sizeof(double) * c->n,
                             hipMemcpyDeviceToHost, s));
  if (alpha_out) HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  *lml = -0.5 * L.h_red[0] - L.h_red[1];
  goff = 0;
  int o = 0;
  for (int m = 0; m < nk; ++m) {
    for (int j = 0; j < n_thetas[m]; ++j) grad_thetas[o++] = L.h_red[16 + goff + j];
    goff += n_thetas[m] + 1;
  }
  (void)tot_nt;
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

// gpmi_lml_grad_mix for T hyper-parameter vectors at once (round 4).  For N <= 4096 the evaluations advance in lockstep:
// every sub-kernel build and fold, the factorisation, both sweeps, L^-T, the k-skipped SYRK, and per sub-kernel the
// weight-scaled inverse, the fused contraction and the window row sums carry the chunk in blockIdx.z.  thetas: T rows of
// the sub-kernels' parameters back to back (sum n_thetas each); g_host: T x nk x n window weights; hrows: T x nk x n;
// qdiag_out (optional, T x n): diag(alpha alpha^T - K^-1) for the WhiteNoise term.  Larger problems: one at a time.
int gpmi_lml_grad_batch_mix(gpmi_ctx* c, int nk, const int* kernels, int64_t T, const double* thetas,
                            const int* n_thetas, const double* g_host, const double* hw_host, const double* extra,
                            const double* mus, const double* mu_const, double* lml, double* grad_thetas, double* hrows,
                            double* alpha_out, double* qdiag_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, nk >= 1 && nk <= GPMI_MAX_MIX, "number of sub-kernels out of range (1..4)");
  ARGCHK(c, kernels && thetas && n_thetas && g_host && lml && grad_thetas && hrows, "NULL argument");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  int tot_nt = 0, max_nt = 0;
  for (int m = 0; m < nk; ++m) {
    tot_nt += n_thetas[m];
    max_nt = n_thetas[m] > max_nt ? n_thetas[m] : max_nt;
  }
  if (int rc = set_device(c)) return rc;
  const bool lockstep = (T >= 2 || c->lockstep_always) && c->np <= 4096 && !c->ycov;
  if (!lockstep) {
    std::vector<double> mu_row((size_t)c->n);
    for (int64_t t = 0; t < T; ++t) {
      const double* mu_t = mus ? mus + t * c->n : mu_row.data();
      if (!mus) std::fill(mu_row.begin(), mu_row.end(), mu_const[t]);
      int inf = 0;
      const int nrow1 = hw_host ? 2 : 1;
      const int rc = gpmi_lml_grad_mix(c, nk, kernels, thetas + t * tot_nt, n_thetas, g_host + t * nk * c->n,
                                       hw_host ? hw_host + t * nk * 2 * c->n : nullptr, extra ? extra[t] : 0.0, mu_t,
                                       lml + t, grad_thetas + t * tot_nt, hrows + t * nk * nrow1 * c->n,
                                       alpha_out ? alpha_out + t * c->n : nullptr, &inf);
      if (info) info[t] = inf;
      if (rc != GPMI_OK) return rc;
      if (qdiag_out)
        if (int rc2 = gpmi_lml_grad_qdiag(c, qdiag_out + t * c->n)) return rc2;
    }
    return GPMI_OK;
  }
  if (c->lanes.size() < 2)
    if (int rc = ensure_lanes(c, 2)) return rc;
  // parameters: ps[m][t]
  std::vector<KParams> ps((size_t)nk * T);
  for (int64_t t = 0; t < T; ++t) {
    int off = 0;
    for (int m = 0; m < nk; ++m) {
      if (int rc = make_params(c, kernels[m], thetas + t * tot_nt + off, n_thetas[m], 0.0, ps[(size_t)m * T + t])) return rc;
      off += n_thetas[m];
    }
  }
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "gpmi_lml_grad_batch_mix: an asynchronous batch is pending on this handle (gpmi_lml_batch_wait first)");
  if (int rc = ensure_batch_ws(c, (int)(T < 64 ? (T < 2 ? 2 : T) : 64))) return rc;
  const int W = max_nt + 1;  // values per sub-kernel and problem from the contraction
  if (int rc = ensure_batch_grad_ws(c, c->bcap, nk * W - 1)) return rc;
  if (!c->mix_zero) {
    HIPCHK(c, hipMalloc(&c->mix_zero, sizeof(double) * c->np));
    ZERO_SYNC(c, c->mix_zero, sizeof(double) * c->np);
  }
  const int cap = c->bgrad_cap;
  if (c->bMix_cap < cap) {
    auto fr = [](double*& p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    fr(c->bMixG);
    fr(c->bMixH);
    fr(c->bMixW);
    fr(c->bMixExtra);
    if (c->bMixP) (void)hipFree(c->bMixP);
    c->bMixP = nullptr;
    c->bMix_cap = 0;
    HIPCHK(c, hipMalloc(&c->bMixG, sizeof(double) * cap * GPMI_MAX_MIX * c->np));
    // (row sums and the caller's row-sum weights: up to two per sub-kernel)
    HIPCHK(c, hipMalloc(&c->bMixH, sizeof(double) * cap * 2 * GPMI_MAX_MIX * c->np));
    HIPCHK(c, hipMalloc(&c->bMixW, sizeof(double) * cap * 2 * GPMI_MAX_MIX * c->np));
    HIPCHK(c, hipMalloc(&c->bMixExtra, sizeof(double) * cap));
    HIPCHK(c, hipMalloc(&c->bMixP, sizeof(KParams) * cap * GPMI_MAX_MIX));
    c->bMix_cap = cap;
  }
  if (qdiag_out && c->bNoise_cap < cap) {
    if (c->bNoise) (void)hipFree(c->bNoise);
    c->bNoise = nullptr;
    c->bNoise_cap = 0;
    HIPCHK(c, hipMalloc(&c->bNoise, sizeof(double) * c->np * cap));
    c->bNoise_cap = cap;
  }
  hipStream_t s = c->lanes[1].stream;
  const int nt = (int)(c->np / GPMI_NB);
  const BatchShape shape0{1, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
  const int64_t sG = (int64_t)GPMI_MAX_MIX * c->np;  // between the problems' weight sets
  // row sums per sub-kernel: one with the sub-kernel's own weights, or two with the caller's (hw); stride between the
  // problems' row-sum (and row-sum weight) sets
  const int nrow = hw_host ? 2 : 1;
  const int64_t sW = (int64_t)nrow * GPMI_MAX_MIX * c->np;
  static const bool stage_times = std::getenv("GPMI_DEBUG_STAGES") != nullptr;  // debugging aid: host time per stage
  RowStage stage(c, s);  // strided outputs reach the caller's arrays through pinned memory (api_internal.h)
  if (int rc = stage.reserve((size_t)cap * (sizeof(double) * (3 * GPMI_MAX_MIX * c->np + c->n) + GPMI_MAX_MIX * sizeof(KParams) + 4096) +
                             (size_t)(2 * GPMI_MAX_MIX + 2) * cap * (sizeof(double) * c->n + 256)))
    return rc;
  for (int64_t t0 = 0; t0 < T; t0 += cap) {
    const auto h0 = std::chrono::steady_clock::now();
    const int B = (int)((T - t0 < cap) ? T - t0 : cap);
    BatchShape bs = shape0;
    bs.count = B;
    // weights, padded like mix_prepare does (the identity in the padding belongs to sub-kernel 0); every input of the
    // chunk is laid out in the pinned staging buffer and leaves from there (RowStage, api_internal.h)
    double* gpad = stage.host(sizeof(double) * B * sG);
    std::fill(gpad, gpad + (size_t)B * sG, 0.0);
    for (int b = 0; b < B; ++b)
      for (int m = 0; m < nk; ++m) {
        double* dst = gpad + (size_t)b * sG + (size_t)m * c->np;
        const double* src = g_host + ((t0 + b) * nk + m) * c->n;
        for (int64_t i = 0; i < c->np; ++i) dst[i] = i < c->n ? src[i] : (m == 0 ? 1.0 : 0.0);
      }
    if (int rc = stage.put(c->bMixG, gpad, sizeof(double) * B * sG)) return rc;
    if (hw_host) {
      double* wpad = stage.host(sizeof(double) * B * sW);
      std::fill(wpad, wpad + (size_t)B * sW, 0.0);
      for (int b = 0; b < B; ++b)
        for (int row = 0; row < 2 * nk; ++row)
          std::copy(hw_host + ((t0 + b) * 2 * nk + row) * c->n, hw_host + ((t0 + b) * 2 * nk + row + 1) * c->n,
                    wpad + (size_t)b * sW + (size_t)row * c->np);
      if (int rc = stage.put(c->bMixW, wpad, sizeof(double) * B * sW)) return rc;
    }
    double* ex = stage.host(sizeof(double) * B);
    for (int b = 0; b < B; ++b) ex[b] = extra ? extra[t0 + b] : 0.0;
    if (int rc = stage.put(c->bMixExtra, ex, sizeof(double) * B)) return rc;
    for (int m = 0; m < nk; ++m)
      if (int rc = stage.put_copy(c->bMixP + (int64_t)m * cap, ps.data() + (size_t)m * T + t0, sizeof(KParams) * B)) return rc;
    if (mus) {
      if (int rc = stage.put_copy(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n)) return rc;
    } else {
      if (int rc = stage.put_copy(c->bMu, mu_const + t0, sizeof(double) * B)) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    // K = sum_m D_m K_m D_m + noise + extra: the sub-kernels' lower tiles into the second matrix, folded into the first
    // (the upper triangle of the sum is never read before the mirror below)
    for (int m = 0; m < nk; ++m) {
      launch_kbuild_square_batched(s, kernels[m], c->bMixP + (int64_t)m * cap, B, c->x, c->n, c->np, c->mix_zero, c->bB2,
                                   c->ld, bs.sMat, (int)c->d, 0);
      launch_scale_add(s, c->bA, c->ld, c->bB2, c->ld, c->bMixG + (int64_t)m * c->np, c->bMixG + (int64_t)m * c->np,
                       c->np, c->np, m > 0, B, bs.sMat, bs.sMat, sG);
    }
    launch_add_diag_vec(s, c->bA, c->ld, c->noise, 0.0, c->n, B, bs.sMat, c->bMixExtra);
    potrf_lower_batched(c, s, c->bA, c->np, c->ld, c->bInv, c->bInfo, bs);
    launch_residual_batched(s, c->y, mus ? c->bMu : nullptr, mus ? nullptr : c->bMu, c->bVec + 2 * c->np, c->n, c->np,
                            bs);
    trsv_forward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec + 2 * c->np, c->bVec, c->bInfo, bs);
    launch_lml_reduce(s, c->bVec, c->bA, c->ld, c->np, c->bRed, bs);
    double* alpha_dev = c->bVec + c->np;   // slot 1 of every problem's four work vectors
    double* ua_dev = c->bVec + 2 * c->np;  // slot 2 (the residual is spent): g_m o alpha
    trsv_backward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec, alpha_dev, c->bInfo, bs);
    trsm_identity_batched(s, c->bA, c->np, c->ld, c->bInv, c->bB2, bs);
    const GemmBatch syrk{B, bs.sMat, bs.sMat, bs.sMat};
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_mirror_lower(s, c->bA, c->ld, c->np, B, bs.sMat);
    if (qdiag_out) launch_qdiag_batched(s, B, c->bA, c->ld, alpha_dev, c->bNoise, c->n, bs.sMat, bs.sVec, c->np);
    for (int m = 0; m < nk; ++m) {
      const double* gm = c->bMixG + (int64_t)m * c->np;
      const KParams* pm = c->bMixP + (int64_t)m * cap;
      // 1/2 sum (D_m Q D_m) o dK_m: the fused contraction on the weight-scaled inverse with u = v = g_m o alpha
      launch_scale_add(s, c->bB2, c->ld, c->bA, c->ld, gm, gm, c->np, c->np, false, B, bs.sMat, bs.sMat, sG);
      launch_vec_mul(s, gm, alpha_dev, ua_dev, c->np, B, sG, bs.sVec, bs.sVec);
      launch_lml_grad_batched(s, pm, B, n_thetas[m], c->x, c->n, c->np, c->bB2, c->ld, bs.sMat, ua_dev, ua_dev, bs.sVec,
                              c->bGws, c->bGout + (int64_t)m * cap * W);
      // window parameters: h_m(i) = sum_j Q_ij K_m,ij g_m(j) on the full K_m (lower tiles built, then mirrored)
      launch_kbuild_square_batched(s, kernels[m], pm, B, c->x, c->n, c->np, c->mix_zero, c->bB2, c->ld, bs.sMat,
                                   (int)c->d, 0);
      launch_mirror_lower(s, c->bB2, c->ld, c->np, B, bs.sMat);
      if (!hw_host) {
        launch_mix_rowsum(s, c->bA, c->bB2, c->ld, alpha_dev, gm, c->bMixH + (int64_t)m * c->np, c->n, B, bs.sMat,
                          bs.sVec, sG);
      } else {
        // (weights and rows share the stride, bMixG's differs; the sub-kernel's two window factors in one pass)
        launch_mix_rowsum(s, c->bA, c->bB2, c->ld, alpha_dev, c->bMixW + (int64_t)(2 * m) * c->np,
                          c->bMixH + (int64_t)(2 * m) * c->np, c->n, B, bs.sMat, bs.sVec, sW, nullptr, 0, c->np);
      }
    }
    HIPCHK(c, hipGetLastError());
    const auto h1 = std::chrono::steady_clock::now();
    HIPCHK(c, hipMemcpyAsync(c->h_bRed, c->bRed, sizeof(double) * 2 * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * nk * cap * W, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    const size_t rowb = sizeof(double) * c->n;
    for (int row = 0; row < nk * nrow; ++row)
      if (int rc = stage.down(hrows + (t0 * nk * nrow + row) * c->n, rowb * nk * nrow, c->bMixH + (int64_t)row * c->np,
                              sizeof(double) * sW, rowb, B))
        return rc;
    if (alpha_out)
      if (int rc = stage.down(alpha_out + t0 * c->n, rowb, alpha_dev, sizeof(double) * bs.sVec, rowb, B)) return rc;
    if (qdiag_out)
      if (int rc = stage.down(qdiag_out + t0 * c->n, rowb, c->bNoise, sizeof(double) * c->np, rowb, B)) return rc;
    const auto h2 = std::chrono::steady_clock::now();
    if (int rc = stage.flush()) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    stage.finish();
    if (stage_times) {
      const auto h3 = std::chrono::steady_clock::now();
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "[gpmi stages] lml_grad_batch_mix B=%d: uploads+launches %.2f ms, download enqueue %.2f ms, wait %.2f ms\n", B,
                   ms(h0, h1), ms(h1, h2), ms(h2, h3));
    }
    for (int b = 0; b < B; ++b) {
      const int inf = c->h_bInfo[b];
      INFOCHK(c, inf);
      lml[t0 + b] = -0.5 * c->h_bRed[2 * b] - c->h_bRed[2 * b + 1];
      int o = 0;
      for (int m = 0; m < nk; ++m)
        for (int j = 0; j < n_thetas[m]; ++j)
          grad_thetas[(t0 + b) * tot_nt + o++] = c->h_bGout[(int64_t)m * cap * W + (int64_t)b * (n_thetas[m] + 1) + j];
      if (info) info[t0 + b] = inf;
    }
  }
  return GPMI_OK;
}

// Leave-one-out terms and gradient (regression.py:489-526) of a two-or-more-region ChangePoint mixture for T hyper-parameter
// vectors in lockstep (round 5): gpmi_lml_grad_batch_mix's build / factorisation / inverse, then the leave-one-out
// vectors, M = K^-1 diag(c2) K^-1 and, per sub-kernel, the weight-scaled M with u = g_m o p, v = g_m o alpha in the
// fused contraction and the window row sums h_m(i) = sum_j (sym(p alpha^T) - M)_ij K_m,ij g_m(j).  Outputs per evaluation:
// alpha, diag(K^-1), p = K^-1 c1, diag(M) (WhiteNoise: 2 s^2 sum(p o alpha - diag M)), the sub-kernels' gradients and the
// row sums (window parameters: 2 sum_i dw_i (h_1 - h_0)_i, contracted on the host).  Lockstep sizes only.
int gpmi_loo_grad_batch_mix(gpmi_ctx* c, int nk, const int* kernels, int64_t T, const double* thetas,
                            const int* n_thetas, const double* g_host, const double* hw_host, const double* extra,
                            const double* mus, const double* mu_const, double* alpha_out, double* ikdiag_out,
                            double* p_out, double* mdiag_out, double* grad_thetas, double* hrows, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, nk >= 1 && nk <= GPMI_MAX_MIX, "number of sub-kernels out of range (1..4)");
  ARGCHK(c, kernels && thetas && n_thetas && g_host && alpha_out && ikdiag_out && p_out && mdiag_out && grad_thetas && hrows,
         "NULL argument");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  int tot_nt = 0, max_nt = 0;
  for (int m = 0; m < nk; ++m) {
    tot_nt += n_thetas[m];
    max_nt = n_thetas[m] > max_nt ? n_thetas[m] : max_nt;
  }
  if (int rc = set_device(c)) return rc;
  ARGCHK(c, c->np <= 4096 && !c->ycov, "lockstep sizes only (n <= 4096, diagonal data errors)");
  if (c->lanes.size() < 2)
    if (int rc = ensure_lanes(c, 2)) return rc;
  // parameters: ps[m][t]
  std::vector<KParams> ps((size_t)nk * T);
  for (int64_t t = 0; t < T; ++t) {
    int off = 0;
    for (int m = 0; m < nk; ++m) {
      if (int rc = make_params(c, kernels[m], thetas + t * tot_nt + off, n_thetas[m], 0.0, ps[(size_t)m * T + t])) return rc;
      off += n_thetas[m];
    }
  }
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "gpmi_loo_grad_batch_mix: an asynchronous batch is pending on this handle (gpmi_lml_batch_wait first)");
  if (int rc = ensure_batch_ws(c, (int)(T < 64 ? (T < 2 ? 2 : T) : 64))) return rc;
  const int W = max_nt + 1;  // values per sub-kernel and problem from the contraction
  if (int rc = ensure_batch_grad_ws(c, c->bcap, nk * W - 1)) return rc;
  if (!c->mix_zero) {
    HIPCHK(c, hipMalloc(&c->mix_zero, sizeof(double) * c->np));
    ZERO_SYNC(c, c->mix_zero, sizeof(double) * c->np);
  }
  const int cap = c->bgrad_cap;
  if (c->bMix_cap < cap) {
    auto fr = [](double*& p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    fr(c->bMixG);
    fr(c->bMixH);
    fr(c->bMixW);
    fr(c->bMixExtra);
    if (c->bMixP) (void)hipFree(c->bMixP);
    c->bMixP = nullptr;
    c->bMix_cap = 0;
    HIPCHK(c, hipMalloc(&c->bMixG, sizeof(double) * cap * GPMI_MAX_MIX * c->np));
    // (row sums and the caller's row-sum weights: up to two per sub-kernel)
    HIPCHK(c, hipMalloc(&c->bMixH, sizeof(double) * cap * 2 * GPMI_MAX_MIX * c->np));
    HIPCHK(c, hipMalloc(&c->bMixW, sizeof(double) * cap * 2 * GPMI_MAX_MIX * c->np));
    HIPCHK(c, hipMalloc(&c->bMixExtra, sizeof(double) * cap));
    HIPCHK(c, hipMalloc(&c->bMixP, sizeof(KParams) * cap * GPMI_MAX_MIX));
    c->bMix_cap = cap;
  }
  // four more vectors per problem: diag(K^-1), c1, sqrt(c2) - later diag(M) -, p = K^-1 c1 (regression.py:505-513).
  // (the same stride as the work vectors': the fused contraction takes u = p and v = alpha with ONE stride)
  const int64_t sLoo = 4 * c->np;
  if (c->bLoo_cap < cap) {
    if (c->bLoo) (void)hipFree(c->bLoo);
    c->bLoo = nullptr;
    c->bLoo_cap = 0;
    HIPCHK(c, hipMalloc(&c->bLoo, sizeof(double) * sLoo * cap));
    c->bLoo_cap = cap;
  }
  hipStream_t s = c->lanes[1].stream;
  const int nt = (int)(c->np / GPMI_NB);
  const BatchShape shape0{1, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
  const int64_t sG = (int64_t)GPMI_MAX_MIX * c->np;  // between the problems' weight sets
  // row sums per sub-kernel: one with the sub-kernel's own weights, or two with the caller's (hw); stride between the
  // problems' row-sum (and row-sum weight) sets
  const int nrow = hw_host ? 2 : 1;
  const int64_t sW = (int64_t)nrow * GPMI_MAX_MIX * c->np;
  RowStage stage(c, s);  // strided outputs reach the caller's arrays through pinned memory (api_internal.h)
  if (int rc = stage.reserve((size_t)cap * (sizeof(double) * (3 * GPMI_MAX_MIX * c->np + c->n) + GPMI_MAX_MIX * sizeof(KParams) + 4096) +
                             (size_t)(2 * GPMI_MAX_MIX + 4) * cap * (sizeof(double) * c->n + 256)))
    return rc;
  for (int64_t t0 = 0; t0 < T; t0 += cap) {
    const int B = (int)((T - t0 < cap) ? T - t0 : cap);
    BatchShape bs = shape0;
    bs.count = B;
    // weights, padded like mix_prepare does (the identity in the padding belongs to sub-kernel 0); every input of the
    // chunk is laid out in the pinned staging buffer and leaves from there (RowStage, api_internal.h)
    double* gpad = stage.host(sizeof(double) * B * sG);
    std::fill(gpad, gpad + (size_t)B * sG, 0.0);
    for (int b = 0; b < B; ++b)
      for (int m = 0; m < nk; ++m) {
        double* dst = gpad + (size_t)b * sG + (size_t)m * c->np;
        const double* src = g_host + ((t0 + b) * nk + m) * c->n;
        for (int64_t i = 0; i < c->np; ++i) dst[i] = i < c->n ? src[i] : (m == 0 ? 1.0 : 0.0);
      }
    if (int rc = stage.put(c->bMixG, gpad, sizeof(double) * B * sG)) return rc;
    if (hw_host) {
      double* wpad = stage.host(sizeof(double) * B * sW);
      std::fill(wpad, wpad + (size_t)B * sW, 0.0);
      for (int b = 0; b < B; ++b)
        for (int row = 0; row < 2 * nk; ++row)
          std::copy(hw_host + ((t0 + b) * 2 * nk + row) * c->n, hw_host + ((t0 + b) * 2 * nk + row + 1) * c->n,
                    wpad + (size_t)b * sW + (size_t)row * c->np);
      if (int rc = stage.put(c->bMixW, wpad, sizeof(double) * B * sW)) return rc;
    }
    double* ex = stage.host(sizeof(double) * B);
    for (int b = 0; b < B; ++b) ex[b] = extra ? extra[t0 + b] : 0.0;
    if (int rc = stage.put(c->bMixExtra, ex, sizeof(double) * B)) return rc;
    for (int m = 0; m < nk; ++m)
      if (int rc = stage.put_copy(c->bMixP + (int64_t)m * cap, ps.data() + (size_t)m * T + t0, sizeof(KParams) * B)) return rc;
    if (mus) {
      if (int rc = stage.put_copy(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n)) return rc;
    } else {
      if (int rc = stage.put_copy(c->bMu, mu_const + t0, sizeof(double) * B)) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    // K = sum_m D_m K_m D_m + noise + extra: the sub-kernels' lower tiles into the second matrix, folded into the first
    // (the upper triangle of the sum is never read before the mirror below)
    for (int m = 0; m < nk; ++m) {
      launch_kbuild_square_batched(s, kernels[m], c->bMixP + (int64_t)m * cap, B, c->x, c->n, c->np, c->mix_zero, c->bB2,
                                   c->ld, bs.sMat, (int)c->d, 0);
      launch_scale_add(s, c->bA, c->ld, c->bB2, c->ld, c->bMixG + (int64_t)m * c->np, c->bMixG + (int64_t)m * c->np,
                       c->np, c->np, m > 0, B, bs.sMat, bs.sMat, sG);
    }
    launch_add_diag_vec(s, c->bA, c->ld, c->noise, 0.0, c->n, B, bs.sMat, c->bMixExtra);
    potrf_lower_batched(c, s, c->bA, c->np, c->ld, c->bInv, c->bInfo, bs);
    launch_residual_batched(s, c->y, mus ? c->bMu : nullptr, mus ? nullptr : c->bMu, c->bVec + 2 * c->np, c->n, c->np,
                            bs);
    trsv_forward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec + 2 * c->np, c->bVec, c->bInfo, bs);
    double* alpha_dev = c->bVec + c->np;   // slot 1 of every problem's four work vectors
    double* ua_dev = c->bVec + 2 * c->np;  // slot 2 (the residual is spent): g_m o alpha
    trsv_backward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec, alpha_dev, c->bInfo, bs);
    double* va_dev = c->bVec + 3 * c->np;  // slot 3: g_m o alpha (slot 2: g_m o p)
    double* diag_dev = c->bLoo;
    double* c1_dev = c->bLoo + c->np;
    double* sc2_dev = c->bLoo + 2 * c->np;
    double* p_dev = c->bLoo + 3 * c->np;
    double* mdiag_dev = sc2_dev;  // (sqrt(c2) is spent once G = K^-1 diag(sqrt c2) exists)
    // L^-T, its row sums of squares = diag(K^-1), K^-1 in full, the leave-one-out vectors, p = K^-1 c1, then
    // M = K^-1 diag(c2) K^-1 = G G^T with G = K^-1 diag(sqrt c2), in full as well (it is read row-wise below)
    trsm_identity_batched(s, c->bA, c->np, c->ld, c->bInv, c->bB2, bs);
    launch_rows_sumsq(s, c->bB2, c->ld, c->np, c->np, 0.0, diag_dev, B, bs.sMat, sLoo, -1.0);
    const GemmBatch syrk{B, bs.sMat, bs.sMat, bs.sMat};
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_mirror_lower(s, c->bA, c->ld, c->np, B, bs.sMat);
    launch_loo_vectors(s, alpha_dev, diag_dev, c1_dev, sc2_dev, c->n, c->np, B, bs.sVec, sLoo);
    launch_rows_dot(s, c->bA, c->ld, c->np, c->np, c1_dev, p_dev, B, bs.sMat, sLoo);
    launch_scale_columns(s, c->bA, sc2_dev, c->bB2, c->ld, c->np, B, bs.sMat, sLoo);
    launch_rows_sumsq(s, c->bB2, c->ld, c->np, c->np, 0.0, mdiag_dev, B, bs.sMat, sLoo, -1.0);  // M_ii = |row i of G|^2
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 0, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_mirror_lower(s, c->bA, c->ld, c->np, B, bs.sMat);
    for (int m = 0; m < nk; ++m) {
      const double* gm = c->bMixG + (int64_t)m * c->np;
      const KParams* pm = c->bMixP + (int64_t)m * cap;
      // sum (D_m Q D_m) o dK_m with Q = sym(p alpha^T) - M: the fused contraction on the weight-scaled M with
      // u = g_m o p, v = g_m o alpha (it returns 1/2 of the sum; the leave-one-out gradient has no 1/2: doubled below)
      launch_scale_add(s, c->bB2, c->ld, c->bA, c->ld, gm, gm, c->np, c->np, false, B, bs.sMat, bs.sMat, sG);
      launch_vec_mul(s, gm, p_dev, ua_dev, c->np, B, sG, sLoo, bs.sVec);
      launch_vec_mul(s, gm, alpha_dev, va_dev, c->np, B, sG, bs.sVec, bs.sVec);
      launch_lml_grad_batched(s, pm, B, n_thetas[m], c->x, c->n, c->np, c->bB2, c->ld, bs.sMat, ua_dev, va_dev, bs.sVec,
                              c->bGws, c->bGout + (int64_t)m * cap * W);
      // window parameters: h_m(i) = sum_j Q_ij K_m,ij g_m(j) on the full K_m (lower tiles built, then mirrored)
      launch_kbuild_square_batched(s, kernels[m], pm, B, c->x, c->n, c->np, c->mix_zero, c->bB2, c->ld, bs.sMat,
                                   (int)c->d, 0);
      launch_mirror_lower(s, c->bB2, c->ld, c->np, B, bs.sMat);
      if (!hw_host) {
        launch_mix_rowsum(s, c->bA, c->bB2, c->ld, alpha_dev, gm, c->bMixH + (int64_t)m * c->np, c->n, B, bs.sMat,
                          bs.sVec, sG, p_dev, sLoo);
      } else {
        launch_mix_rowsum(s, c->bA, c->bB2, c->ld, alpha_dev, c->bMixW + (int64_t)(2 * m) * c->np,
                          c->bMixH + (int64_t)(2 * m) * c->np, c->n, B, bs.sMat, bs.sVec, sW, p_dev, sLoo, c->np);
      }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * nk * cap * W, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    const size_t rowb = sizeof(double) * c->n;
    for (int row = 0; row < nk * nrow; ++row)
      if (int rc = stage.down(hrows + (t0 * nk * nrow + row) * c->n, rowb * nk * nrow, c->bMixH + (int64_t)row * c->np,
                              sizeof(double) * sW, rowb, B))
        return rc;
    if (int rc = stage.down(alpha_out + t0 * c->n, rowb, alpha_dev, sizeof(double) * bs.sVec, rowb, B)) return rc;
    if (int rc = stage.down(ikdiag_out + t0 * c->n, rowb, diag_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (int rc = stage.down(p_out + t0 * c->n, rowb, p_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (int rc = stage.down(mdiag_out + t0 * c->n, rowb, mdiag_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (int rc = stage.flush()) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    stage.finish();
    for (int b = 0; b < B; ++b) {
      const int inf = c->h_bInfo[b];
      INFOCHK(c, inf);
      int o = 0;
      for (int m = 0; m < nk; ++m)
        for (int j = 0; j < n_thetas[m]; ++j)
          grad_thetas[(t0 + b) * tot_nt + o++] = 2.0 * c->h_bGout[(int64_t)m * cap * W + (int64_t)b * (n_thetas[m] + 1) + j];
      if (info) info[t0 + b] = inf;
    }
  }
  return GPMI_OK;
}

int gpmi_loo_terms_mix(gpmi_ctx* c, int nk, const int* kernels, const double* thetas, const int* n_thetas,
                       const double* g_host, double extra_diag, const double* mu, double* alpha_out,
                       double* ikdiag, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams ps[GPMI_MAX_MIX];
  if (int rc = ensure_lanes(c, 2)) return rc;
  if (int rc = mix_prepare(c, nk, kernels, thetas, n_thetas, g_host, ps)) return rc;
  ARGCHK(c, mu && alpha_out && ikdiag, "mu / alpha / ikdiag is NULL");
  c->fitted = false;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  double* diag_dev = L.vec + 2 * c->np;
  const MixEval mx{nk, ps, c->mix_g, extra_diag, c->mix_scratch, c->mix_zero};
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, ps[0], mu_dev, 0.0, 0, true, &mx)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_rows_sumsq(s, L.B2, c->ld, c->np, c->np, 0.0, diag_dev);  // -diag(K^-1): squared row norms of L^-T
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(ikdiag, diag_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  for (int64_t i = 0; i < c->n; ++i) ikdiag[i] = -ikdiag[i];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_predict_mix(gpmi_ctx* c, const double* pts, int64_t m, const double* gq_host, double* mu_out,
                     double* negsumsq_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && c->mix_nk > 0, "gpmi_predict_mix needs a successful gpmi_fit_mix");
  ARGCHK(c, pts && gq_host && m > 0, "NULL argument or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int64_t chunk = 2048;
  const int nk = c->mix_nk;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t mp = round_up(mc, GPMI_NB);
    if (int rc = ensure_query_ws(c, mp)) return rc;
    if (int rc = ensure_q3(c)) return rc;
    double* gq_dev = c->pvec + 2 * mp;  // nk x mp  (pvec holds mp (2 + 2 d + d^2 + 4) doubles)
    HIPCHK(c, hipMemsetAsync(gq_dev, 0, sizeof(double) * nk * mp, s));
    HIPCHK(c, hipMemcpyAsync(c->pts, pts + m0 * c->d, sizeof(double) * mc * c->d, hipMemcpyHostToDevice, s));
    for (int k = 0; k < nk; ++k)
      HIPCHK(c, hipMemcpyAsync(gq_dev + (int64_t)k * mp, gq_host + (int64_t)k * m + m0, sizeof(double) * mc,
                               hipMemcpyHostToDevice, s));
    build_mix_cross(c, s, gq_dev, mc, mp);
    double* mu_dev = c->pvec;
    double* var_dev = c->pvec + mp;
    if (mu_out) launch_rows_dot(s, c->Q, c->ld, mp, c->np, c->alpha, mu_dev);
    if (negsumsq_out) {
      if (int rc = ensure_inv2(c, L, s)) return rc;
      trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);
      launch_rows_sumsq(s, c->Q2, c->ld, mp, c->np, 0.0, var_dev);  // -|L^-1 k|^2; the host adds K_qq[0, 0]
    }
    HIPCHK(c, hipGetLastError());
    if (mu_out) HIPCHK(c, hipMemcpyAsync(mu_out + m0, mu_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    if (negsumsq_out)
      HIPCHK(c, hipMemcpyAsync(negsumsq_out + m0, var_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  return GPMI_OK;
}

int gpmi_posterior_mix(gpmi_ctx* c, const double* pts, int64_t m, const double* gq_host, double* mu_out,
                       double* cov_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && c->mix_nk > 0, "gpmi_posterior_mix needs a successful gpmi_fit_mix");
  ARGCHK(c, pts && gq_host && m > 0, "NULL argument or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int nk = c->mix_nk;
  const int64_t mp = round_up(m, GPMI_NB);
  if (int rc = ensure_query_ws(c, mp)) return rc;
  if (int rc = ensure_q3(c)) return rc;
  double* gq_dev = c->pvec + 2 * mp;
  HIPCHK(c, hipMemsetAsync(gq_dev, 0, sizeof(double) * nk * mp, s));
  HIPCHK(c, hipMemcpyAsync(c->pts, pts, sizeof(double) * m * c->d, hipMemcpyHostToDevice, s));
  for (int k = 0; k < nk; ++k)
    HIPCHK(c, hipMemcpyAsync(gq_dev + (int64_t)k * mp, gq_host + (int64_t)k * m, sizeof(double) * m,
                             hipMemcpyHostToDevice, s));
  build_mix_cross(c, s, gq_dev, m, mp);
  double* mu_dev = c->pvec;
  launch_rows_dot(s, c->Q, c->ld, mp, c->np, c->alpha, mu_dev);
  if (mu_out) HIPCHK(c, hipMemcpyAsync(mu_out, mu_dev, sizeof(double) * m, hipMemcpyDeviceToHost, s));
  if (cov_out) {
    const int64_t ldq = mp + 32;
    double *Kqq = nullptr, *tmp = nullptr;
    HIPCHK(c, hipMalloc(&Kqq, sizeof(double) * mp * ldq));
    hipError_t e = hipMalloc(&tmp, sizeof(double) * mp * ldq);
    if (e == hipSuccess) {
      if (int rc = ensure_inv2(c, L, s)) {
        (void)hipFree(Kqq);
        (void)hipFree(tmp);
        return rc;
      }
      trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);
      // K_qq = sum_m diag(gq_m) K_m(pts, pts) diag(gq_m): no jitter, no noise (covariance.py:529-544)
      for (int k = 0; k < nk; ++k) {
        launch_kbuild_cross(s, c->mix_p[k], c->pts, m, mp, c->pts, m, mp, tmp, ldq);
        launch_scale_add(s, Kqq, ldq, tmp, ldq, gq_dev + (int64_t)k * mp, gq_dev + (int64_t)k * mp, mp, mp, k > 0);
      }
      launch_gemm_nt(s, TILES_RECT, OP_SUB, Kqq, ldq, c->Q2, c->ld, c->Q2, c->ld, (int)(mp / GPMI_NB),
                     (int)(mp / GPMI_NB), (int)c->np);
      e = hipMemcpy2DAsync(cov_out, sizeof(double) * m, Kqq, sizeof(double) * ldq, sizeof(double) * m, m,
                           hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(Kqq);
    if (tmp) (void)hipFree(tmp);
    HIPCHK(c, e);
  }
  HIPCHK(c, hipStreamSynchronize(s));
  return GPMI_OK;
}

}  // extern "C"

// ---- per-point noise hyper-parameters ----------------------------------------------------------------
extern "C" {

int gpmi_set_noise(gpmi_ctx* c, const double* noise_var) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, noise_var != nullptr, "noise_var is NULL");
  if (int rc = set_device(c)) return rc;
  if (int rc = gpmi_sync(c)) return rc;  // nothing may still be reading the old values
  HIPCHK(c, hipMemcpy(c->noise, noise_var, sizeof(double) * c->n, hipMemcpyHostToDevice));
  return GPMI_OK;
}

int gpmi_lml_grad_qdiag(gpmi_ctx* c, double* qdiag) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, qdiag != nullptr, "qdiag is NULL");
  ARGCHK(c, c->lanes.size() >= 2 && c->lanes[1].B2, "gpmi_lml_grad has not been called");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[1];  // after gpmi_lml_grad: L.A = K^-1 (lower tiles), vec + np = alpha
  double* out = L.vec + 2 * c->np;
  launch_qdiag(L.stream, L.A, c->ld, L.vec + c->np, out, c->n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(qdiag, out, sizeof(double) * c->n, hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  return GPMI_OK;
}

}  // extern "C"

