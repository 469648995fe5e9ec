// C-ABI of libgpmi (include/gpmi.h): the GpRegressor entry points - fit, likelihood (single, lockstep batches, asynchronous slots), likelihood
// and leave-one-out gradients, predict, posterior, spatial gradients, covariance downloads.
// (split from api.hip in round 4; the handle, the lanes and the helpers these entry points are built from: api.hip,
// api_internal.h)
#include "api_internal.h"

extern "C" {

int gpmi_fit(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
             const double* mu, double* alpha_out, double* logdet_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, mu != nullptr, "mu is NULL");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  const bool dbg = std::getenv("GPMI_DEBUG_TIMING") != nullptr;
  const auto h0 = std::chrono::steady_clock::now();
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0, true, nullptr, true, c->alpha)) return rc;
  const auto h1 = std::chrono::steady_clock::now();
  // alpha = L^-T v  (c->alpha already holds the sweep's sentinel: enqueue_factor_and_forward, backward_out)
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, c->alpha, L.info, BatchShape(), true);
  if (L.inv2_valid) HIPCHK(c, hipStreamWaitEvent(s, L.ev_main, 0));  // (the inverse blocks built beside the sweeps)
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out)
    HIPCHK(c, hipMemcpyAsync(alpha_out, c->alpha, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  const auto h2 = std::chrono::steady_clock::now();
  HIPCHK(c, hipStreamSynchronize(s));
  if (dbg) {
    const auto h3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[gpmi_fit] host enqueue factor+forward %.2f ms, backward+copies %.2f ms, final sync %.2f ms\n",
                 ms(h0, h1), ms(h1, h2), ms(h2, h3));
  }
  if (logdet_out) *logdet_out = L.h_red[1];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  c->fit_params = p;
  c->fitted = (L.h_info[0] == 0);
  return GPMI_OK;
}

int gpmi_lml(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
             const double* mu, double* lml, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, mu != nullptr && lml != nullptr, "mu / lml is NULL");
  int inf = 0;
  int rc = gpmi_lml_batch(c, kernel, 1, theta, n_theta, &extra_diag, mu, nullptr, lml, &inf);
  if (info) *info = inf;
  return rc;
}

int gpmi_lml_batch(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta,
                   const double* extra, const double* mus, const double* mu_const, double* lml,
                   int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, thetas && lml, "thetas / lml is NULL");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  if (int rc = set_device(c)) return rc;
  if (c->lanes.size() < 2)
    if (int rc = ensure_lanes(c, 2)) return rc;
  const int S = (int)c->lanes.size() - 1;
  std::vector<KParams> ps((size_t)T);
  for (int64_t t = 0; t < T; ++t)
    if (int rc = make_params(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0,
                             ps[(size_t)t]))
      return rc;
  if ((T >= 2 || c->lockstep_always) && c->np <= 4096 && !c->ycov) {
    // small problems: all evaluations of a chunk advance in lockstep, one launch per step for the
    // whole chunk (blockIdx.z), instead of one latency-bound launch sequence per evaluation
    ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
           "gpmi_lml_batch: an asynchronous batch is pending on this handle (gpmi_lml_batch_wait first)");
    if (int rc = ensure_batch_ws(c, (int)(T < 2 ? 2 : (T > 4096 ? 4096 : T)))) return rc;  // (capped by ensure_batch_ws)
    // A chunk runs as TWO half-batches on two streams (the chunk's workspace split in the middle): while one half is in
    // a latency-bound step - potrf_diag: one workgroup per matrix, 32 of 256 CUs for a half of 32 - the other half's
    // GEMM launches fill the chip.  A value does not depend on the batch it is evaluated in (§4.3), so the split is
    // invisible in the results.  GPMI_BATCH_SPLIT=0: one stream (A/B).
    static const bool split_ok = [] {
      const char* e = std::getenv("GPMI_BATCH_SPLIT");
      return !e || std::atoi(e) != 0;
    }();
    const bool two = split_ok && T >= 16 && ensure_lanes(c, 3) == GPMI_OK;
    const BatchShape shape0{1, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
    auto enqueue = [&](hipStream_t s, int64_t t_first, int off, int B) -> int {
      BatchShape bs = shape0;
      bs.count = B;
      double* A = c->bA + (int64_t)off * bs.sMat;
      double* Inv = c->bInv + (int64_t)off * bs.sInv;
      double* Vec = c->bVec + (int64_t)off * bs.sVec;
      double* Mu = c->bMu + (int64_t)off * (mus ? c->n : 1);
      HIPCHK(c, hipMemcpyAsync(c->bParams + off, ps.data() + t_first, sizeof(KParams) * B, hipMemcpyHostToDevice, s));
      if (mus)
        HIPCHK(c, hipMemcpyAsync(Mu, mus + t_first * c->n, sizeof(double) * B * c->n, hipMemcpyHostToDevice, s));
      else
        HIPCHK(c, hipMemcpyAsync(Mu, mu_const + t_first, sizeof(double) * B, hipMemcpyHostToDevice, s));
      HIPCHK(c, hipMemsetAsync(c->bInfo + off, 0, sizeof(int) * B, s));
      launch_kbuild_square_batched(s, ps[0].kernel, c->bParams + off, B, c->x, c->n, c->np, c->noise, A, c->ld, bs.sMat,
                                   (int)c->d);
      potrf_lower_batched(c, s, A, c->np, c->ld, Inv, c->bInfo + off, bs);
      launch_residual_batched(s, c->y, mus ? Mu : nullptr, mus ? nullptr : Mu, Vec + 2 * c->np, c->n, c->np, bs);
      trsv_forward(c, s, A, c->np, c->ld, Inv, Vec + 2 * c->np, Vec, c->bInfo + off, bs);
      launch_lml_reduce(s, Vec, A, c->ld, c->np, c->bRed + 2 * off, bs);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(c->h_bRed + 2 * off, c->bRed + 2 * off, sizeof(double) * 2 * B, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipMemcpyAsync(c->h_bInfo + off, c->bInfo + off, sizeof(int) * B, hipMemcpyDeviceToHost, s));
      return GPMI_OK;
    };
    for (int64_t t0 = 0; t0 < T; t0 += c->bcap) {
      const int B = (int)((T - t0 < c->bcap) ? T - t0 : c->bcap);
      const int B1 = (two && B >= 16) ? B / 2 : B;
      if (int rc = enqueue(c->lanes[1].stream, t0, 0, B1)) return rc;
      if (B1 < B)
        if (int rc = enqueue(c->lanes[2].stream, t0 + B1, B1, B - B1)) return rc;
      HIPCHK(c, hipStreamSynchronize(c->lanes[1].stream));
      if (B1 < B) HIPCHK(c, hipStreamSynchronize(c->lanes[2].stream));
      for (int b = 0; b < B; ++b) {
        const int inf = c->h_bInfo[b];
        INFOCHK(c, inf);
        lml[t0 + b] = (inf == 0) ? (-0.5 * c->h_bRed[2 * b] - c->h_bRed[2 * b + 1]) : -1e50;
        if (info) info[t0 + b] = inf;
      }
    }
    return GPMI_OK;
  }
  std::vector<int> slot_of((size_t)T);
  std::vector<int> used((size_t)S, 0);
  for (int64_t t = 0; t < T; ++t) {
    const int li = (int)(t % S);
    Lane& L = c->lanes[1 + li];
    const int slot = used[li]++;
    slot_of[(size_t)t] = slot;
    double* mu_dev = nullptr;
    if (mus) {
      mu_dev = L.vec + 3 * c->np;
      HIPCHK(c, hipMemcpyAsync(mu_dev, mus + t * c->n, sizeof(double) * c->n,
                               hipMemcpyHostToDevice, L.stream));
    }
    if (int rc = enqueue_factor_and_forward(c, L, ps[(size_t)t], mu_dev,
                                            mu_const ? mu_const[t] : 0.0, slot, S == 1 || T == 1))
      return rc;
  }
  for (int li = 0; li < S; ++li) {
    if (!used[li]) continue;
    Lane& L = c->lanes[1 + li];
    HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double) * used[li], hipMemcpyDeviceToHost,
                             L.stream));
    HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int) * used[li], hipMemcpyDeviceToHost,
                             L.stream));
  }
  for (int li = 0; li < S; ++li)
    if (used[li]) HIPCHK(c, hipStreamSynchronize(c->lanes[1 + li].stream));
  for (int64_t t = 0; t < T; ++t) {
    Lane& L = c->lanes[1 + (int)(t % S)];
    const int slot = slot_of[(size_t)t];
    const int inf = L.h_info[slot];
    INFOCHK(c, inf);
    // -1/2 v.v - sum ln L_ii (regression.py:539); sentinel on failure (regression.py:540-542)
    lml[t] = (inf == 0) ? (-0.5 * L.h_red[2 * slot] - L.h_red[2 * slot + 1]) : -1e50;
    if (info) info[t] = inf;
  }
  return GPMI_OK;
}

// Asynchronous lockstep batches: gpmi_lml_batch_submit enqueues the T evaluations of a slot and returns; the caller does
// its own work (a tempering driver: the accept / reject bookkeeping of the OTHER half of its chains) and collects the
// values with gpmi_lml_batch_wait.  Two slots, the two halves of the lockstep workspace, on two streams - the same
// device work as one gpmi_lml_batch call of both halves (which runs them as two half-batches side by side), so a value
// is bit-identical either way.  Lockstep sizes only (np <= 4096, no dense y covariance).
int gpmi_lml_batch_submit(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                          const double* mus, const double* mu_const, int slot) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, slot == 0 || slot == 1, "slot must be 0 or 1");
  // evaluations per slot: GPMI_ASYNC_SLOT_MAX (default 256 - 128 until round 6; the workspace budget GPMI_BATCH_GIB /
  // GPMI_BATCH_MAX decides how many of them a slot really gets: api.hip, ensure_batch_ws)
  static const int slot_max = [] {
    const char* e = std::getenv("GPMI_ASYNC_SLOT_MAX");
    const int v = e ? std::atoi(e) : 256;
    return v >= 1 && v <= 2048 ? v : 256;
  }();
  ARGCHK(c, T >= 1 && T <= slot_max, "T out of range (1 .. GPMI_ASYNC_SLOT_MAX per slot, default 256)");
  ARGCHK(c, thetas, "thetas is NULL");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  ARGCHK(c, c->np <= 4096 && !c->ycov, "asynchronous batches: lockstep sizes only (n <= 4096, diagonal data errors)");
  ARGCHK(c, c->bpend[slot] == 0, "this slot has a batch pending (gpmi_lml_batch_wait first)");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 3)) return rc;
  // each slot owns half of the workspace: at least 2 T matrices - and, because the workspace can only grow while nothing
  // is pending, all that a slot may ever be asked for (128 evaluations) from the first submission on, as far as the
  // memory cap of ensure_batch_ws allows: a later, larger group of chains then fits whatever the first one was
  if (c->bpend[1 - slot] == 0) {
    if (int rc = ensure_batch_ws(c, 2 * slot_max)) return rc;  // (a no-op once it is that large)
    ARGCHK(c, c->bcap >= 2 * T, "not enough device memory for two batches of this size");
  } else {
    ARGCHK(c, c->bcap >= 2 * T,
           "this batch does not fit the slot (half of the lockstep workspace) and the workspace cannot grow while the other slot is pending");
  }
  const int off = slot * (c->bcap / 2);
  // inputs through pinned staging that lives until the wait (the copies are asynchronous)
  const int64_t mu_doubles = mus ? T * c->n : T;
  const int64_t need = (int64_t)sizeof(KParams) * T + (int64_t)sizeof(double) * mu_doubles;
  if (c->h_bStage_bytes[slot] < need) {
    if (c->h_bStage[slot]) (void)hipHostFree(c->h_bStage[slot]);
    c->h_bStage[slot] = nullptr;
    c->h_bStage_bytes[slot] = 0;
    HIPCHK(c, hipHostMalloc(&c->h_bStage[slot], (size_t)need));
    c->h_bStage_bytes[slot] = need;
  }
  KParams* ps = reinterpret_cast<KParams*>(c->h_bStage[slot]);
  double* mu_stage = reinterpret_cast<double*>(c->h_bStage[slot] + sizeof(KParams) * T);
  for (int64_t t = 0; t < T; ++t)
    if (int rc = make_params(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0, ps[t])) return rc;
  std::memcpy(mu_stage, mus ? mus : mu_const, sizeof(double) * mu_doubles);
  hipStream_t s = c->lanes[1 + slot].stream;
  BatchShape bs{(int)T, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
  double* A = c->bA + (int64_t)off * bs.sMat;
  double* Inv = c->bInv + (int64_t)off * bs.sInv;
  double* Vec = c->bVec + (int64_t)off * bs.sVec;
  double* Mu = c->bMu + (int64_t)off * (mus ? c->n : 1);
  HIPCHK(c, hipMemcpyAsync(c->bParams + off, ps, sizeof(KParams) * T, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(Mu, mu_stage, sizeof(double) * mu_doubles, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(c->bInfo + off, 0, sizeof(int) * T, s));
  launch_kbuild_square_batched(s, ps[0].kernel, c->bParams + off, (int)T, c->x, c->n, c->np, c->noise, A, c->ld, bs.sMat,
                               (int)c->d);
  potrf_lower_batched(c, s, A, c->np, c->ld, Inv, c->bInfo + off, bs);
  launch_residual_batched(s, c->y, mus ? Mu : nullptr, mus ? nullptr : Mu, Vec + 2 * c->np, c->n, c->np, bs);
  trsv_forward(c, s, A, c->np, c->ld, Inv, Vec + 2 * c->np, Vec, c->bInfo + off, bs);
  launch_lml_reduce(s, Vec, A, c->ld, c->np, c->bRed + 2 * off, bs);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_bRed + 2 * off, c->bRed + 2 * off, sizeof(double) * 2 * T, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_bInfo + off, c->bInfo + off, sizeof(int) * T, hipMemcpyDeviceToHost, s));
  c->bpend[slot] = (int)T;
  return GPMI_OK;
}

int gpmi_lml_batch_wait(gpmi_ctx* c, int slot, double* lml, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, slot == 0 || slot == 1, "slot must be 0 or 1");
  ARGCHK(c, c->bpend[slot] > 0, "nothing pending in this slot");
  ARGCHK(c, lml, "lml is NULL");
  if (int rc = set_device(c)) return rc;
  const int T = c->bpend[slot];
  const int off = slot * (c->bcap / 2);
  c->bpend[slot] = 0;  // whatever happens below, the slot is free again
  HIPCHK(c, hipStreamSynchronize(c->lanes[1 + slot].stream));
  for (int b = 0; b < T; ++b) {
    const int inf = c->h_bInfo[off + b];
    INFOCHK(c, inf);
    lml[b] = (inf == 0) ? (-0.5 * c->h_bRed[2 * (off + b)] - c->h_bRed[2 * (off + b) + 1]) : -1e50;
    if (info) info[b] = inf;
  }
  return GPMI_OK;
}

// the evaluation lane of the likelihood gradient with everything gpmi_lml_grad allocates lazily: the lane itself
// (matrix, inverse blocks, streams - for a large problem the CU-masked pair), the second matrix, the contraction's
// partial sums for n_theta parameters
static int ensure_gradient_lane(gpmi_ctx* c, int n_theta) {
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  const int64_t need = grad_ws_doubles(c->np, n_theta);
  if (L.gws_doubles < need) {
    if (L.gws) (void)hipFree(L.gws);
    L.gws = nullptr;
    L.gws_doubles = 0;
    HIPCHK(c, hipMalloc(&L.gws, sizeof(double) * need));
    L.gws_doubles = need;
  }
  return GPMI_OK;
}

int gpmi_prepare_gradient(gpmi_ctx* c, int n_theta) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, n_theta >= 1 && n_theta <= GPMI_MAX_D + 2, "n_theta out of range");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_gradient_lane(c, n_theta)) return rc;
  HIPCHK(c, hipDeviceSynchronize());  // the allocations have happened when this returns
  return GPMI_OK;
}

int gpmi_lml_grad(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                  const double* mu, double* lml, double* grad_theta, double* trace_q,
                  double* alpha_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, mu && lml && grad_theta, "mu / lml / grad_theta is NULL");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_gradient_lane(c, n_theta)) return rc;
  Lane& L = c->lanes[1];
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  double* gout = L.red + 16;  // n_theta + 1 values (n_theta <= GPMI_MAX_D + 2)
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0, true, nullptr, false, nullptr, L.B2)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  // K^-1 = L^-T L^-1 (regression.py:556-557), lower tiles, overwriting L
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, true, L.A, c->ld, L.B2, c->ld, L.B2, c->ld,
              (int)(c->np / GPMI_NB), (int)(c->np / GPMI_NB), (int)c->np);
  launch_lml_grad(s, p, n_theta, c->x, c->n, c->np, L.A, c->ld, alpha_dev, alpha_dev, L.gws, gout);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_red + 16, gout, sizeof(double) * (n_theta + 1),
                           hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (alpha_out)
    HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  *lml = -0.5 * L.h_red[0] - L.h_red[1];
  for (int j = 0; j < n_theta; ++j) grad_theta[j] = L.h_red[16 + j];
  if (trace_q) *trace_q = L.h_red[16 + n_theta];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

// noise_batch (T x n, host) / qdiag_out (T x n, host): the per-problem noise variances of HeteroscedasticNoise
// (covariance.py:608-690: they are hyper-parameters) and the diagonal of Q = alpha alpha^T - K^-1 their gradient needs;
// both NULL for the plain form
static int lml_grad_batch_impl(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                               const double* mus, const double* mu_const, const double* noise_batch, double* lml,
                               double* grad_theta, double* trace_q, double* alpha_out, double* qdiag_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, thetas && lml && grad_theta, "thetas / lml / grad_theta is NULL");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  if (int rc = set_device(c)) return rc;
  const bool lockstep = (T >= 2 || c->lockstep_always) && c->np <= 4096 && !c->ycov;
  if (!lockstep) {
    // large problems are throughput-bound one at a time: the single-evaluation path, one after another
    std::vector<double> mu_row((size_t)c->n);
    for (int64_t t = 0; t < T; ++t) {
      const double* mu_t = mus ? mus + t * c->n : mu_row.data();
      if (!mus) std::fill(mu_row.begin(), mu_row.end(), mu_const[t]);
      int inf = 0;
      if (noise_batch)
        if (int rc = gpmi_set_noise(c, noise_batch + t * c->n)) return rc;
      const int rc = gpmi_lml_grad(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0, mu_t, lml + t,
                                   grad_theta + t * n_theta, trace_q ? trace_q + t : nullptr,
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
  std::vector<KParams> ps((size_t)T);
  for (int64_t t = 0; t < T; ++t)
    if (int rc = make_params(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0, ps[(size_t)t])) return rc;
  // lockstep: every launch carries the chunk in blockIdx.z - K-build, factorisation, both sweeps, L^-T by forward
  // substitution on the identity, the k-skipped SYRK K^-1 = L^-T L^-1 (regression.py:556-557) and the fused contraction
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "gpmi_lml_grad_batch: an asynchronous batch is pending on this handle (gpmi_lml_batch_wait first)");
  if (int rc = ensure_batch_ws(c, (int)(T < 64 ? (T < 2 ? 2 : T) : 64))) return rc;
  if (int rc = ensure_batch_grad_ws(c, c->bcap, n_theta)) return rc;
  if ((noise_batch || qdiag_out) && c->bNoise_cap < c->bgrad_cap) {
    if (c->bNoise) (void)hipFree(c->bNoise);
    c->bNoise = nullptr;
    c->bNoise_cap = 0;
    HIPCHK(c, hipMalloc(&c->bNoise, sizeof(double) * c->np * c->bgrad_cap));
    c->bNoise_cap = c->bgrad_cap;
  }
  hipStream_t s = c->lanes[1].stream;
  const int nt = (int)(c->np / GPMI_NB);
  const BatchShape shape0{1, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
  const int W = n_theta + 1;
  const size_t rowb = sizeof(double) * c->n;
  RowStage stage(c, s);  // strided rows travel between the caller's arrays and the device through pinned memory (api_internal.h)
  if (int rc = stage.reserve((size_t)c->bgrad_cap * (7 * (rowb + 256) + sizeof(KParams) + 512))) return rc;
  for (int64_t t0 = 0; t0 < T; t0 += c->bgrad_cap) {
    const int B = (int)((T - t0 < c->bgrad_cap) ? T - t0 : c->bgrad_cap);
    BatchShape bs = shape0;
    bs.count = B;
    if (int rc = stage.put_copy(c->bParams, ps.data() + t0, sizeof(KParams) * B)) return rc;  // (inputs leave from pinned memory)
    if (mus) {
      if (int rc = stage.put_copy(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n)) return rc;
    } else {
      if (int rc = stage.put_copy(c->bMu, mu_const + t0, sizeof(double) * B)) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    if (noise_batch)
      if (int rc = stage.up(c->bNoise, sizeof(double) * c->np, noise_batch + t0 * c->n, rowb, rowb, B)) return rc;
    launch_kbuild_square_batched(s, ps[0].kernel, c->bParams, B, c->x, c->n, c->np, noise_batch ? c->bNoise : c->noise,
                                 c->bA, c->ld, bs.sMat, (int)c->d, noise_batch ? c->np : 0);
    potrf_lower_batched(c, s, c->bA, c->np, c->ld, c->bInv, c->bInfo, bs);
    launch_residual_batched(s, c->y, mus ? c->bMu : nullptr, mus ? nullptr : c->bMu, c->bVec + 2 * c->np, c->n, c->np,
                            bs);
    trsv_forward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec + 2 * c->np, c->bVec, c->bInfo, bs);
    launch_lml_reduce(s, c->bVec, c->bA, c->ld, c->np, c->bRed, bs);
    double* alpha_dev = c->bVec + c->np;  // slot 1 of every problem's four work vectors
    trsv_backward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec, alpha_dev, c->bInfo, bs);
    trsm_identity_batched(s, c->bA, c->np, c->ld, c->bInv, c->bB2, bs);
    const GemmBatch syrk{B, bs.sMat, bs.sMat, bs.sMat};
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_lml_grad_batched(s, c->bParams, B, n_theta, c->x, c->n, c->np, c->bA, c->ld, bs.sMat, alpha_dev, alpha_dev,
                            bs.sVec, c->bGws, c->bGout);
    if (qdiag_out) {  // diag(alpha alpha^T - K^-1) per problem, into the noise buffer (consumed by the build above)
      launch_qdiag_batched(s, B, c->bA, c->ld, alpha_dev, c->bNoise, c->n, bs.sMat, bs.sVec, c->np);
      if (int rc = stage.down(qdiag_out + t0 * c->n, rowb, c->bNoise, sizeof(double) * c->np, rowb, B)) return rc;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bRed, c->bRed, sizeof(double) * 2 * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * W * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    if (alpha_out)
      if (int rc = stage.down(alpha_out + t0 * c->n, rowb, alpha_dev, sizeof(double) * bs.sVec, rowb, B)) return rc;
    if (int rc = stage.flush()) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    stage.finish();
    for (int b = 0; b < B; ++b) {
      const int inf = c->h_bInfo[b];
      INFOCHK(c, inf);
      lml[t0 + b] = -0.5 * c->h_bRed[2 * b] - c->h_bRed[2 * b + 1];
      for (int j = 0; j < n_theta; ++j) grad_theta[(t0 + b) * n_theta + j] = c->h_bGout[b * W + j];
      if (trace_q) trace_q[t0 + b] = c->h_bGout[b * W + n_theta];
      if (info) info[t0 + b] = inf;
    }
  }
  return GPMI_OK;
}

int gpmi_lml_grad_batch(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                        const double* mus, const double* mu_const, double* lml, double* grad_theta, double* trace_q,
                        double* alpha_out, int* info) {
  return lml_grad_batch_impl(c, kernel, T, thetas, n_theta, extra, mus, mu_const, nullptr, lml, grad_theta, trace_q,
                             alpha_out, nullptr, info);
}

int gpmi_lml_grad_batch_noise(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                              const double* mus, const double* mu_const, const double* noise_var, double* lml,
                              double* grad_theta, double* trace_q, double* alpha_out, double* qdiag_out, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, noise_var && qdiag_out, "noise_var / qdiag is NULL");
  ARGCHK(c, !c->ycov, "per-point noise hyper-parameters need diagonal data errors");
  return lml_grad_batch_impl(c, kernel, T, thetas, n_theta, extra, mus, mu_const, noise_var, lml, grad_theta, trace_q,
                             alpha_out, qdiag_out, info);
}

// Leave-one-out log-likelihood terms and gradient (regression.py:489-526) for T hyper-parameter vectors in lockstep: the
// batched form of gpmi_loo_grad - every launch carries the chunk in blockIdx.z.  What the reference's `multiprocessing.Pool`
// farms out start by start (regression.py:597-601) when the model selector is the cross-validation objective.
// noise_batch (T x n, host) / mdiag_out (T x n, host): the per-problem noise variances of HeteroscedasticNoise and the
// diagonal of M = K^-1 diag(c2) K^-1, which the gradient with respect to a point's own noise needs (dK = 2 s_i^2 e_i e_i^T:
// 2 s_i^2 (p_i alpha_i - M_ii)); both NULL for the plain form.  With them every call takes the lockstep path.
static int loo_grad_batch_impl(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                               const double* mus, const double* mu_const, const double* noise_batch, double* alpha_out,
                               double* ikdiag_out, double* p_out, double* mdiag_out, double* grad_theta, double* trace_q,
                               int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, thetas && alpha_out && ikdiag_out && p_out && grad_theta, "NULL argument");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  if (int rc = set_device(c)) return rc;
  const bool lockstep = (T >= 2 || c->lockstep_always || noise_batch) && c->np <= 4096 && !c->ycov;
  ARGCHK(c, lockstep || !noise_batch, "per-point noise in the batch: lockstep sizes only (n <= 4096, diagonal data errors)");
  if (!lockstep) {
    std::vector<double> mu_row((size_t)c->n);
    for (int64_t t = 0; t < T; ++t) {
      const double* mu_t = mus ? mus + t * c->n : mu_row.data();
      if (!mus) std::fill(mu_row.begin(), mu_row.end(), mu_const[t]);
      int inf = 0;
      const int rc = gpmi_loo_grad(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0, mu_t,
                                   alpha_out + t * c->n, ikdiag_out + t * c->n, p_out + t * c->n,
                                   grad_theta + t * n_theta, trace_q ? trace_q + t : nullptr, &inf);
      if (info) info[t] = inf;
      if (rc != GPMI_OK) return rc;
    }
    return GPMI_OK;
  }
  if (c->lanes.size() < 2)
    if (int rc = ensure_lanes(c, 2)) return rc;
  std::vector<KParams> ps((size_t)T);
  for (int64_t t = 0; t < T; ++t)
    if (int rc = make_params(c, kernel, thetas + t * n_theta, n_theta, extra ? extra[t] : 0.0, ps[(size_t)t])) return rc;
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "gpmi_loo_grad_batch: an asynchronous batch is pending on this handle (gpmi_lml_batch_wait first)");
  if (int rc = ensure_batch_ws(c, (int)(T < 64 ? (T < 2 ? 2 : T) : 64))) return rc;
  if (int rc = ensure_batch_grad_ws(c, c->bcap, n_theta)) return rc;
  // four more vectors per problem: diag(K^-1), c1, sqrt(c2) - later diag(M) -, p = K^-1 c1 (regression.py:505-513).
  // (the same stride as the work vectors': the fused contraction takes u = p and v = alpha with ONE stride)
  const int64_t sLoo = 4 * c->np;
  if (noise_batch && c->bNoise_cap < c->bgrad_cap) {
    if (c->bNoise) (void)hipFree(c->bNoise);
    c->bNoise = nullptr;
    c->bNoise_cap = 0;
    HIPCHK(c, hipMalloc(&c->bNoise, sizeof(double) * c->np * c->bgrad_cap));
    c->bNoise_cap = c->bgrad_cap;
  }
  if (c->bLoo_cap < c->bgrad_cap) {
    if (c->bLoo) (void)hipFree(c->bLoo);
    c->bLoo = nullptr;
    c->bLoo_cap = 0;
    HIPCHK(c, hipMalloc(&c->bLoo, sizeof(double) * sLoo * c->bgrad_cap));
    c->bLoo_cap = c->bgrad_cap;
  }
  hipStream_t s = c->lanes[1].stream;
  const int nt = (int)(c->np / GPMI_NB);
  const BatchShape shape0{1, c->np * c->ld, (c->np / GPMI_NB) * GPMI_NB * GPMI_NB, 4 * c->np};
  const int W = n_theta + 1;
  const size_t rowb = sizeof(double) * c->n;
  RowStage stage(c, s);  // strided rows travel between the caller's arrays and the device through pinned memory (api_internal.h)
  if (int rc = stage.reserve((size_t)c->bgrad_cap * (7 * (rowb + 256) + sizeof(KParams) + 512))) return rc;
  for (int64_t t0 = 0; t0 < T; t0 += c->bgrad_cap) {
    const int B = (int)((T - t0 < c->bgrad_cap) ? T - t0 : c->bgrad_cap);
    BatchShape bs = shape0;
    bs.count = B;
    if (int rc = stage.put_copy(c->bParams, ps.data() + t0, sizeof(KParams) * B)) return rc;  // (inputs leave from pinned memory)
    if (mus) {
      if (int rc = stage.put_copy(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n)) return rc;
    } else {
      if (int rc = stage.put_copy(c->bMu, mu_const + t0, sizeof(double) * B)) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    if (noise_batch)
      if (int rc = stage.up(c->bNoise, sizeof(double) * c->np, noise_batch + t0 * c->n, rowb, rowb, B)) return rc;
    launch_kbuild_square_batched(s, ps[0].kernel, c->bParams, B, c->x, c->n, c->np, noise_batch ? c->bNoise : c->noise,
                                 c->bA, c->ld, bs.sMat, (int)c->d, noise_batch ? c->np : 0);
    potrf_lower_batched(c, s, c->bA, c->np, c->ld, c->bInv, c->bInfo, bs);
    launch_residual_batched(s, c->y, mus ? c->bMu : nullptr, mus ? nullptr : c->bMu, c->bVec + 2 * c->np, c->n, c->np,
                            bs);
    trsv_forward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec + 2 * c->np, c->bVec, c->bInfo, bs);
    double* alpha_dev = c->bVec + c->np;  // slot 1 of every problem's four work vectors
    trsv_backward(c, s, c->bA, c->np, c->ld, c->bInv, c->bVec, alpha_dev, c->bInfo, bs);
    double* diag_dev = c->bLoo;
    double* c1_dev = c->bLoo + c->np;
    double* sc2_dev = c->bLoo + 2 * c->np;
    double* p_dev = c->bLoo + 3 * c->np;
    // L^-T, its row sums of squares = diag(K^-1), then K^-1 in full (both triangles)
    trsm_identity_batched(s, c->bA, c->np, c->ld, c->bInv, c->bB2, bs);
    launch_rows_sumsq(s, c->bB2, c->ld, c->np, c->np, 0.0, diag_dev, B, bs.sMat, sLoo, -1.0);
    const GemmBatch syrk{B, bs.sMat, bs.sMat, bs.sMat};
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 1, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_mirror_lower(s, c->bA, c->ld, c->np, B, bs.sMat);
    launch_loo_vectors(s, alpha_dev, diag_dev, c1_dev, sc2_dev, c->n, c->np, B, bs.sVec, sLoo);
    launch_rows_dot(s, c->bA, c->ld, c->np, c->np, c1_dev, p_dev, B, bs.sMat, sLoo);
    // M = K^-1 diag(c2) K^-1 = G G^T with G = K^-1 diag(sqrt c2); lower tiles, overwriting K^-1
    launch_scale_columns(s, c->bA, sc2_dev, c->bB2, c->ld, c->np, B, bs.sMat, sLoo);
    double* mdiag_dev = sc2_dev;  // (sqrt(c2) is spent once G = K^-1 diag(sqrt c2) exists)
    if (mdiag_out) launch_rows_sumsq(s, c->bB2, c->ld, c->np, c->np, 0.0, mdiag_dev, B, bs.sMat, sLoo, -1.0);  // M_ii = |row i of G|^2
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 0, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_lml_grad_batched(s, c->bParams, B, n_theta, c->x, c->n, c->np, c->bA, c->ld, bs.sMat, p_dev, alpha_dev, sLoo,
                            c->bGws, c->bGout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * W * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    if (int rc = stage.down(alpha_out + t0 * c->n, rowb, alpha_dev, sizeof(double) * bs.sVec, rowb, B)) return rc;
    if (int rc = stage.down(ikdiag_out + t0 * c->n, rowb, diag_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (int rc = stage.down(p_out + t0 * c->n, rowb, p_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (mdiag_out)
      if (int rc = stage.down(mdiag_out + t0 * c->n, rowb, mdiag_dev, sizeof(double) * sLoo, rowb, B)) return rc;
    if (int rc = stage.flush()) return rc;
    HIPCHK(c, hipStreamSynchronize(s));
    stage.finish();
    for (int b = 0; b < B; ++b) {
      const int inf = c->h_bInfo[b];
      INFOCHK(c, inf);
      // the contraction returns 1/2 sum Q o dK; the LOO gradient has no 1/2 (regression.py:513)
      for (int j = 0; j < n_theta; ++j) grad_theta[(t0 + b) * n_theta + j] = 2.0 * c->h_bGout[b * W + j];
      if (trace_q) trace_q[t0 + b] = c->h_bGout[b * W + n_theta];
      if (info) info[t0 + b] = inf;
    }
  }
  return GPMI_OK;
}

int gpmi_loo_grad_batch(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                        const double* mus, const double* mu_const, double* alpha_out, double* ikdiag_out, double* p_out,
                        double* grad_theta, double* trace_q, int* info) {
  return loo_grad_batch_impl(c, kernel, T, thetas, n_theta, extra, mus, mu_const, nullptr, alpha_out, ikdiag_out, p_out,
                             nullptr, grad_theta, trace_q, info);
}

int gpmi_loo_grad_batch_noise(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                              const double* mus, const double* mu_const, const double* noise_var, double* alpha_out,
                              double* ikdiag_out, double* p_out, double* mdiag_out, double* grad_theta, double* trace_q,
                              int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, noise_var && mdiag_out, "noise_var / mdiag is NULL");
  ARGCHK(c, !c->ycov, "per-point noise hyper-parameters need diagonal data errors");
  return loo_grad_batch_impl(c, kernel, T, thetas, n_theta, extra, mus, mu_const, noise_var, alpha_out, ikdiag_out, p_out,
                             mdiag_out, grad_theta, trace_q, info);
}

int gpmi_predict(gpmi_ctx* c, const double* pts, int64_t m, double* mu_out, double* var_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_predict needs a successful gpmi_fit");
  ARGCHK(c, c->fit_params.kernel >= 0 || c->mix_nk > 0,
         "gpmi_predict: the model was fitted with a caller-built covariance (gpmi_fit_dense) - use gpmi_predict_dense / gpmi_solve_rows");
  ARGCHK(c, pts && m > 0, "pts is NULL or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  const int64_t chunk = 2048;
  KParams p = c->fit_params;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t mp = round_up(mc, GPMI_NB);
    if (int rc = ensure_query_ws(c, mp)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pts, pts + m0 * c->d, sizeof(double) * mc * c->d,
                             hipMemcpyHostToDevice, s));
    {
      ProfScope ps(c, s, GPMI_PROF_KBUILD, 0.0, 8.0 * mp * c->np);
      launch_kbuild_cross(s, p, c->pts, mc, mp, c->x, c->n, c->np, c->Q, c->ld);
    }
    double* mu_dev = c->pvec;
    double* var_dev = c->pvec + mp;
    if (mu_out) launch_rows_dot(s, c->Q, c->ld, mp, c->np, c->alpha, mu_dev);
    if (var_out) {
      if (int rc = ensure_inv2(c, L, s)) return rc;
      trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);
      launch_rows_sumsq(s, c->Q2, c->ld, mp, c->np, p.a2, var_dev);  // K_qq[0,0] = a^2 (regression.py:210)
    }
    HIPCHK(c, hipGetLastError());
    if (mu_out)
      HIPCHK(c, hipMemcpyAsync(mu_out + m0, mu_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    if (var_out)
      HIPCHK(c, hipMemcpyAsync(var_out + m0, var_dev, sizeof(double) * mc, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  return GPMI_OK;
}

int gpmi_posterior(gpmi_ctx* c, const double* pts, int64_t m, double* mu_out, double* cov_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_posterior needs a successful gpmi_fit");
  ARGCHK(c, c->fit_params.kernel >= 0 || c->mix_nk > 0,
         "gpmi_posterior: the model was fitted with a caller-built covariance (gpmi_fit_dense) - use gpmi_predict_dense / gpmi_solve_rows");
  ARGCHK(c, pts && m > 0, "pts is NULL or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  KParams p = c->fit_params;
  const int64_t mp = round_up(m, GPMI_NB);
  if (int rc = ensure_query_ws(c, mp)) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pts, pts, sizeof(double) * m * c->d, hipMemcpyHostToDevice, s));
  launch_kbuild_cross(s, p, c->pts, m, mp, c->x, c->n, c->np, c->Q, c->ld);
  double* mu_dev = c->pvec;
  launch_rows_dot(s, c->Q, c->ld, mp, c->np, c->alpha, mu_dev);
  if (mu_out) HIPCHK(c, hipMemcpyAsync(mu_out, mu_dev, sizeof(double) * m, hipMemcpyDeviceToHost, s));
  if (cov_out) {
    double* Kqq = nullptr;
    const int64_t ldq = mp + 32;
    HIPCHK(c, hipMalloc(&Kqq, sizeof(double) * mp * ldq));
    if (int rc = ensure_inv2(c, L, s)) {
      (void)hipFree(Kqq);
      return rc;
    }
    trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q, mp, false, c->Q2, nullptr);
    launch_kbuild_cross(s, p, c->pts, m, mp, c->pts, m, mp, Kqq, ldq);  // no jitter (regression.py:441)
    // Sigma = K_qq - Q^T Q with Q = L^-1 K_qx^T, i.e. rows of c->Q2 dotted pairwise
    launch_gemm_nt(s, TILES_RECT, OP_SUB, Kqq, ldq, c->Q2, c->ld, c->Q2, c->ld, (int)(mp / GPMI_NB),
                   (int)(mp / GPMI_NB), (int)c->np);
    hipError_t e = hipMemcpy2DAsync(cov_out, sizeof(double) * m, Kqq, sizeof(double) * ldq,
                                    sizeof(double) * m, m, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(Kqq);
    HIPCHK(c, e);
  }
  HIPCHK(c, hipStreamSynchronize(s));
  return GPMI_OK;
}

int gpmi_spatial_derivatives(gpmi_ctx* c, const double* pts, int64_t m, double* dmu_out,
                             double* dvar_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_spatial_derivatives needs a successful gpmi_fit");
  ARGCHK(c, c->fit_params.kernel >= 0 || c->mix_nk > 0,
         "gpmi_spatial_derivatives: the model was fitted with a caller-built covariance (gpmi_fit_dense) - use gpmi_predict_dense / gpmi_solve_rows");
  ARGCHK(c, c->fit_params.kernel == GPMI_KERNEL_SE, "spatial derivatives: SquaredExponential only");
  ARGCHK(c, pts && m > 0 && dmu_out && dvar_out, "NULL argument or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  KParams p = c->fit_params;
  const int64_t chunk = 1024, d = c->d;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t mp = round_up(mc, GPMI_NB);
    if (int rc = ensure_query_ws(c, mp)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pts, pts + m0 * d, sizeof(double) * mc * d, hipMemcpyHostToDevice, s));
    launch_kbuild_cross(s, p, c->pts, mc, mp, c->x, c->n, c->np, c->Q, c->ld);
    launch_copy(s, c->Q, c->Q2, mp * c->ld);
    // Z = K^-1 k per row: forward then backward solve (regression.py:410)
    if (int rc = ensure_inv2(c, L, s)) return rc;
    if (int rc = ensure_trsm_panel(c, mp)) return rc;
    trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q2, mp, false, nullptr, c->trsm_panel);
    trsm_rows_backward(c, s, L.A, c->np, c->ld, L.invD, c->Q2, mp, L.inv2, c->trsm_panel);
    double* dmu_dev = c->pvec;
    double* dvar_dev = c->pvec + mp * d;
    launch_sd_reduce(s, p, c->x, c->n, c->pts, mc, c->Q, c->ld, c->alpha, 0, 1.0, dmu_dev);
    launch_sd_reduce(s, p, c->x, c->n, c->pts, mc, c->Q, c->ld, c->Q2, c->ld, -2.0, dvar_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dmu_out + m0 * d, dmu_dev, sizeof(double) * mc * d, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(dvar_out + m0 * d, dvar_dev, sizeof(double) * mc * d, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  return GPMI_OK;
}

int gpmi_gradient(gpmi_ctx* c, const double* pts, int64_t m, double* gmu_out, double* gcov_out) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted, "gpmi_gradient needs a successful gpmi_fit");
  ARGCHK(c, c->fit_params.kernel >= 0 || c->mix_nk > 0,
         "gpmi_gradient: the model was fitted with a caller-built covariance (gpmi_fit_dense) - use gpmi_predict_dense / gpmi_solve_rows");
  ARGCHK(c, c->fit_params.kernel == GPMI_KERNEL_SE, "gradient: SquaredExponential only");
  ARGCHK(c, pts && m > 0 && gmu_out && gcov_out, "NULL argument or m <= 0");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  hipStream_t s = L.stream;
  KParams p = c->fit_params;
  const int64_t d = c->d;
  int64_t chunk = 1024 / d;  // d right-hand sides per point
  if (chunk < 1) chunk = 1;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t rp = round_up(mc * d, GPMI_NB);
    if (int rc = ensure_query_ws(c, rp)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pts, pts + m0 * d, sizeof(double) * mc * d, hipMemcpyHostToDevice, s));
    launch_kbuild_cross(s, p, c->pts, mc, round_up(mc, GPMI_NB), c->x, c->n, c->np, c->Q, c->ld);
    double* gmu_dev = c->pvec;
    double* gcov_dev = c->pvec + rp;
    launch_sd_reduce(s, p, c->x, c->n, c->pts, mc, c->Q, c->ld, c->alpha, 0, 1.0, gmu_dev);
    launch_grad_rhs(s, p, c->x, c->n, c->np, c->pts, mc * d, rp, c->Q, c->ld, c->Q2);
    if (int rc = ensure_inv2(c, L, s)) return rc;
    trsm_rows_forward(c, s, L.A, c->np, c->ld, L.inv2, c->Q2, rp, false, c->Q, nullptr);  // c->Q is free again
    launch_grad_cov(s, p, c->Q, c->ld, c->np, mc, gcov_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(gmu_out + m0 * d, gmu_dev, sizeof(double) * mc * d, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(gcov_out + m0 * d * d, gcov_dev, sizeof(double) * mc * d * d,
                             hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  return GPMI_OK;
}

int gpmi_loo_diag(gpmi_ctx* c, double* ikdiag) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && ikdiag, "gpmi_loo_diag needs a successful gpmi_fit");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& F = c->lanes[0];
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  HIPCHK(c, hipStreamSynchronize(F.stream));
  // diag(K^-1)_a = sum_i (L^-1)_ia^2 = squared norm of row a of L^-T   (regression.py:460-462)
  if (int rc = enqueue_inverse_factor(c, L, F)) return rc;
  launch_rows_sumsq(L.stream, L.B2, c->ld, c->np, c->np, 0.0, L.vec);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(ikdiag, L.vec, sizeof(double) * c->n, hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  for (int64_t i = 0; i < c->n; ++i) ikdiag[i] = -ikdiag[i];  // rows_sumsq returns base - sum
  return GPMI_OK;
}

int gpmi_loo_terms(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                   const double* mu, double* alpha_out, double* ikdiag, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, mu && alpha_out && ikdiag, "mu / alpha / ikdiag is NULL");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  hipStream_t s = L.stream;
  double* mu_dev = L.vec + 3 * c->np;
  double* alpha_dev = L.vec + c->np;
  double* diag_dev = L.vec + 2 * c->np;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0, true, nullptr, false, nullptr, L.B2)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_rows_sumsq(s, L.B2, c->ld, c->np, c->np, 0.0, diag_dev);
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

int gpmi_loo_grad(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                  const double* mu, double* alpha_out, double* ikdiag, double* p_out,
                  double* grad_theta, double* trace_q, int* info) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, mu && alpha_out && ikdiag && p_out && grad_theta, "NULL argument");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  if (int rc = ensure_second_matrix(c, L)) return rc;
  const int64_t need = grad_ws_doubles(c->np, n_theta) + 4 * c->np;
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
  double* diag_dev = L.gws;              // 4 extra vectors live in front of the partial sums
  double* c1_dev = L.gws + c->np;
  double* sc2_dev = L.gws + 2 * c->np;
  double* p_dev = L.gws + 3 * c->np;
  double* partial = L.gws + 4 * c->np;
  double* gout = L.red + 16;
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0, true, nullptr, false, nullptr, L.B2)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  // K^-1 (full, both triangles) in A
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_rows_sumsq(s, L.B2, c->ld, c->np, c->np, 0.0, diag_dev);  // = -diag(K^-1)
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, true, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, nt, nt,
              (int)c->np);
  launch_mirror_lower(s, L.A, c->ld, c->np);
  // diag_dev holds -diag: flip sign inside the vector kernel by passing it through a scaled copy
  launch_negate(s, diag_dev, c->np);
  launch_loo_vectors(s, alpha_dev, diag_dev, c1_dev, sc2_dev, c->n, c->np);
  // p = K^-1 c1  (regression.py:512-513, 518-519 folded: c1^T K^-1 dK_j alpha = p^T dK_j alpha)
  launch_rows_dot(s, L.A, c->ld, c->np, c->np, c1_dev, p_dev);
  // M = K^-1 diag(c2) K^-1 = G G^T with G = K^-1 diag(sqrt c2); lower tiles, overwriting K^-1
  launch_scale_columns(s, L.A, sc2_dev, L.B2, c->ld, c->np);
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, false, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, nt, nt,
              (int)c->np);
  launch_lml_grad(s, p, n_theta, c->x, c->n, c->np, L.A, c->ld, p_dev, alpha_dev, partial, gout);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red + 16, gout, sizeof(double) * (n_theta + 1), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(alpha_out, alpha_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(ikdiag, diag_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(p_out, p_dev, sizeof(double) * c->n, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  // the contraction returns 1/2 sum Q o dK; the LOO gradient has no 1/2 (regression.py:513)
  for (int j = 0; j < n_theta; ++j) grad_theta[j] = 2.0 * L.h_red[16 + j];
  if (trace_q) *trace_q = L.h_red[16 + n_theta];
  INFOCHK(c, L.h_info[0]);
  if (info) *info = L.h_info[0];
  return GPMI_OK;
}

int gpmi_covariance(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra_diag,
                    int with_noise, double* K_host) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, extra_diag, p)) return rc;
  ARGCHK(c, K_host != nullptr, "K_host is NULL");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  double* noise = c->noise;
  double* zeros = nullptr;
  if (!with_noise) {
    HIPCHK(c, hipMalloc(&zeros, sizeof(double) * c->np));
    HIPCHK(c, hipMemsetAsync(zeros, 0, sizeof(double) * c->np, L.stream));
    noise = zeros;
  }
  launch_kbuild_square(L.stream, p, c->x, c->n, c->np, noise, L.A, c->ld, false);
  if (with_noise && c->ycov) launch_add_full(L.stream, L.A, c->ld, c->ycov, c->n);
  hipError_t e = hipMemcpy2DAsync(K_host, sizeof(double) * c->n, L.A, sizeof(double) * c->ld,
                                  sizeof(double) * c->n, c->n, hipMemcpyDeviceToHost, L.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(L.stream);
  if (zeros) (void)hipFree(zeros);
  HIPCHK(c, e);
  return GPMI_OK;
}

int gpmi_cross_covariance(gpmi_ctx* c, int kernel, const double* theta, int n_theta,
                          const double* pts, int64_t m, double* out) {
  if (!c) return GPMI_ERR_ARG;
  KParams p;
  if (int rc = make_params(c, kernel, theta, n_theta, 0.0, p)) return rc;
  ARGCHK(c, pts && out && m > 0, "pts / out is NULL or m <= 0");
  if (int rc = set_device(c)) return rc;
  hipStream_t s = c->lanes[0].stream;
  const int64_t chunk = 2048;
  for (int64_t m0 = 0; m0 < m; m0 += chunk) {
    const int64_t mc = (m - m0 < chunk) ? m - m0 : chunk;
    const int64_t mp = round_up(mc, GPMI_NB);
    if (int rc = ensure_query_ws(c, mp)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pts, pts + m0 * c->d, sizeof(double) * mc * c->d,
                             hipMemcpyHostToDevice, s));
    launch_kbuild_cross(s, p, c->pts, mc, mp, c->x, c->n, c->np, c->Q, c->ld);
    HIPCHK(c, hipMemcpy2DAsync(out + m0 * c->n, sizeof(double) * c->n, c->Q, sizeof(double) * c->ld,
                               sizeof(double) * c->n, mc, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  return GPMI_OK;
}

int gpmi_get_K(gpmi_ctx* c, double* K_host) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && K_host, "gpmi_get_K needs a successful gpmi_fit");
  ARGCHK(c, c->fit_params.kernel >= 0, "gpmi_get_K: the covariance of this fit was built by the caller (gpmi_fit_dense / gpmi_fit_mix)");
  if (int rc = set_device(c)) return rc;
  if (int rc = ensure_lanes(c, 2)) return rc;
  Lane& L = c->lanes[1];
  launch_kbuild_square(L.stream, c->fit_params, c->x, c->n, c->np, c->noise, L.A, c->ld, false);
  if (c->ycov) launch_add_full(L.stream, L.A, c->ld, c->ycov, c->n);
  HIPCHK(c, hipMemcpy2DAsync(K_host, sizeof(double) * c->n, L.A, sizeof(double) * c->ld,
                             sizeof(double) * c->n, c->n, hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  return GPMI_OK;
}

int gpmi_get_L(gpmi_ctx* c, double* L_host) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->fitted && L_host, "gpmi_get_L needs a successful gpmi_fit");
  if (int rc = set_device(c)) return rc;
  Lane& L = c->lanes[0];
  HIPCHK(c, hipMemcpy2DAsync(L_host, sizeof(double) * c->n, L.A, sizeof(double) * c->ld,
                             sizeof(double) * c->n, c->n, hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  for (int64_t i = 0; i < c->n; ++i)
    for (int64_t j = i + 1; j < c->n; ++j) L_host[i * c->n + j] = 0.0;  // numpy returns the upper triangle zeroed
  return GPMI_OK;
}

}  // extern "C"
