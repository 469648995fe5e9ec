// C-ABI of libgpmi (see include/gpmi.h): handle management, host<->device plumbing and the
// orchestration of the GP hot path (covariance build -> Cholesky -> solves -> reductions).
//
// This file: the handle - lanes / streams, workspaces, the shared K-build + factorise + forward-solve sequence -, the
// lifecycle entry points (create / destroy / data / streams / options) and the instrumentation (per-class events,
// in-kernel stamps, device-pointer entry points of the tools).  The entry points proper live beside it:
//   api_regression.hip  fit / LML / batches / gradients / predict / posterior / spatial gradients / leave-one-out
//   api_mix.hip         mixture covariance (ChangePoint), per-point noise (HeteroscedasticNoise)
//   api_linv.hip        linear inversion (GpLinearInverter)
//   api_dense.hip       plugin covariance functions (dense path), rank-one append
// api_internal.h declares what they share.
#include "api_internal.h"
#include <malloc.h>


thread_local std::string g_create_err;  // gpmi_create has no handle to keep its error text in


// A handle has TWO full-chip streams, those of its lanes 0 and 1; lane i >= 2 runs on the stream of lane i mod 2 (its
// buffers are its own).  Why: the HIP runtime multiplexes a process's streams onto 4 hardware queues (GPU_MAX_HW_QUEUES),
// and two streams on one queue run in order.  With a stream per lane a second handle alive in the process (a regressor
// beside the one being evaluated: 2 + 3 streams) put the two evaluation lanes of a likelihood sweep on one queue: config
// 3 lost 14 % (30.2 against 26.5 ms per evaluation at N = 16384).  Every pair the library runs side by side is a pair
// of neighbouring lanes (the half-batches of a lockstep chunk and the asynchronous slots: lanes 1 | 2; a sweep on two
// lanes: 1 | 2), lane 0's stream is idle while other lanes evaluate, and more than two evaluations at a time were never
// faster than two (§5).  Measured alternatives (round 4 A/B scripts, since removed; results in profiles/HISTORY.md): GPU_MAX_HW_QUEUES=8 cures the
// sweep but two processes on one device then time out in the flag-ordered kernels (bench.py --gpus 2 on one GPU);
// process-wide pooled streams cure it too but change the order in which a handle's queues are created, and the
// look-ahead of lane 1 (its stream and its CU-masked pair) lost 10 % (LML at N = 16384: 27.7 -> 31 ms).
int lane_streams(gpmi_ctx* c, Lane& L) {
  const size_t index = c->lanes.size() - 1;  // (the lane has just been appended)
  L.owns_stream = index < 2;
  if (!L.owns_stream) L.stream = c->lanes[index % 2].stream;
  if (L.owns_stream) HIPCHK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
  c->ncu = prop.multiProcessorCount;
  HIPCHK(c, hipEventCreateWithFlags(&L.ev_la, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&L.ev_panel, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&L.ev_main, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&L.ev_slice, hipEventDisableTiming));
  return GPMI_OK;
}


// CU-masked stream pair of the look-ahead: factor the next panel on pair_cus CUs (sp) while the trailing update
// runs on all the others (su).  Mask bits are dealt round-robin over the 8 XCDs (probed with tools/cumask_probe.hip),
// so the first 8 m bits are m CUs on every XCD.  The pair is created together with the lane's main stream, before any
// work is queued, and destroyed with it: created later (on first use, with kernels already in flight on the lane)
// hipStreamDestroy blocked forever on ROCm 7.2, and a pair that is never destroyed ends the process in a SIGSEGV
// inside the runtime's static destructors when rocprofv3 is attached (round-2 probe, profiles/HISTORY.md).  Only lane 0 (the
// fitted model) owns a pair; lane 1 (single evaluations) borrows it (lane_alloc below).
bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k) {
  (void)c;
  return L.sp[k] != nullptr && L.su[k] != nullptr;
}


int lane_masked_streams(gpmi_ctx* c, Lane& L) {
  const int ncu = c->ncu;
  if (ncu < 64 || ncu % 32 != 0) return GPMI_OK;
  for (int k = 0; k < GPMI_NPAIRS; ++k) {
    if (const char* e = std::getenv("GPMI_PANEL_CUS")) {  // tuning aid: CUs of the panel stream (a multiple of 8)
      const int v = std::atoi(e);
      if (v >= 8 && v <= ncu / 2 && v % 8 == 0) c->pair_cus[k] = v;
    }
    std::vector<uint32_t> panel((size_t)ncu / 32, 0u), upd((size_t)ncu / 32, 0xffffffffu);
    for (int i = 0; i < c->pair_cus[k]; ++i) {
      panel[(size_t)i / 32] |= 1u << (i % 32);
      upd[(size_t)i / 32] &= ~(1u << (i % 32));
    }
    hipError_t e1 = hipExtStreamCreateWithCUMask(&L.sp[k], (uint32_t)panel.size(), panel.data());
    hipError_t e2 = e1 == hipSuccess ? hipExtStreamCreateWithCUMask(&L.su[k], (uint32_t)upd.size(), upd.data()) : e1;
    if (e1 != hipSuccess || e2 != hipSuccess) {
      // (said once per process: such a lane factors in stream order on the full chip - same bits, slower)
      static bool told = false;
      if (!told) {
        told = true;
        std::fprintf(stderr, "[gpmi] a CU-masked stream pair could not be created (%s): this evaluation lane runs without "
                             "look-ahead and without the flag-ordered tail\n", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      }
      if (L.sp[k]) (void)hipStreamDestroy(L.sp[k]);
      if (L.su[k]) (void)hipStreamDestroy(L.su[k]);
      L.sp[k] = L.su[k] = nullptr;  // no look-ahead: everything on the full-chip stream
      (void)hipGetLastError();
    }
  }
  return GPMI_OK;
}



int lane_alloc(gpmi_ctx* c, Lane& L) {
  if (int rc = lane_streams(c, L)) return rc;
  // lanes[0] or lanes[1] (the lane has just been appended) of a problem large enough to ever enter the look-ahead
  // regime: destroying a CU-masked stream takes the runtime about a second, which small models (a GpOptimiser
  // builds a new regressor per added evaluation) should not pay
  if (c->lanes.size() == 1 && c->np >= 40 * GPMI_NB)
    if (int rc = lane_masked_streams(c, L)) return rc;
  // Lane 1 (single evaluations: marginal_likelihood, the gradients) borrows lane 0's pair.  The entry points of a handle
  // are synchronous, so the fitted lane never factorises while lane 1 does - and a pair of its own made lane 1's speed a
  // lottery of the order in which the process had created its queues: with any other handle created (and closed) before,
  // the chain launches on lane 1's own panel stream ran 40 % slower (LML at N = 8192: 7.7 against 5.3 ms, round 5,
  // tools/probe_lml.py), while lane 0's pair was fast in every order tried.  Two HSA queues per handle less as well.
  if (c->lanes.size() == 2 && c->np >= 40 * GPMI_NB) {
    for (int k = 0; k < GPMI_NPAIRS; ++k) {
      L.sp[k] = c->lanes[0].sp[k];
      L.su[k] = c->lanes[0].su[k];
    }
    L.owns_pair = false;
  }
  const int64_t nt = c->np / GPMI_NB;
  HIPCHK(c, hipMalloc(&L.A, sizeof(double) * c->np * c->ld));
  HIPCHK(c, hipMalloc(&L.invD, sizeof(double) * nt * GPMI_NB * GPMI_NB));
  // potrf_diag writes the block-lower part of an inverse only: the zeros above stay from here (round 4: zeroing them
  // in the kernel cost it 57 KB of stores per launch through one CU's memory path, as much as loading the block)
  ZERO_SYNC(c, L.invD, sizeof(double) * nt * GPMI_NB * GPMI_NB);
  HIPCHK(c, hipMalloc(&L.vec, sizeof(double) * 4 * c->np));
  HIPCHK(c, hipMalloc(&L.red, sizeof(double) * 2 * RED_SLOTS));
  HIPCHK(c, hipMalloc(&L.info, sizeof(int) * RED_SLOTS));
  HIPCHK(c, hipHostMalloc(&L.h_red, sizeof(double) * 2 * RED_SLOTS));
  HIPCHK(c, hipHostMalloc(&L.h_info, sizeof(int) * RED_SLOTS));
  return GPMI_OK;
}

#define DBG_FREE(msg) do { if (std::getenv("GPMI_DEBUG_FREE")) std::fprintf(stderr, "[free] %s\n", msg); } while (0)
void lane_free(Lane& L) {
  DBG_FREE("lane: sync main stream");
  if (L.stream) (void)hipStreamSynchronize(L.stream);
  DBG_FREE("lane: sync masked streams");
  if (!L.owns_pair)
    for (int k = 0; k < GPMI_NPAIRS; ++k) L.sp[k] = L.su[k] = nullptr;  // (lane 0's: synchronised and destroyed with it)
  for (int k = 0; k < GPMI_NPAIRS; ++k) {
    if (L.sp[k]) (void)hipStreamSynchronize(L.sp[k]);
    if (L.su[k]) (void)hipStreamSynchronize(L.su[k]);
  }
  potrf_flow_free(L);
  DBG_FREE("lane: destroy events");
  if (L.ev_la) (void)hipEventDestroy(L.ev_la);
  if (L.ev_panel) (void)hipEventDestroy(L.ev_panel);
  if (L.ev_join) (void)hipEventDestroy(L.ev_join);
  if (L.ev_main) (void)hipEventDestroy(L.ev_main);
  if (L.ev_slice) (void)hipEventDestroy(L.ev_slice);
  DBG_FREE("lane: destroy masked streams");
  // hipStreamDestroy on a CU-masked stream blocks forever on ROCm 7.2 when it follows the stream's last
  // synchronisation too closely (round-2 probe, profiles/HISTORY.md: 1 hang in 6 closes without the pause, 0 in 36 with 5 .. 300 ms; the round-1
  // library only got away with it because loading librccl happened to sit in between)
  if (L.sp[0] || L.su[0]) {
    static const int pause_ms = [] {
      const char* e = std::getenv("GPMI_DESTROY_PAUSE_MS");
      return e ? std::atoi(e) : 50;
    }();
    std::this_thread::sleep_for(std::chrono::milliseconds(pause_ms));
  }
  for (int k = 0; k < GPMI_NPAIRS; ++k) {
    if (L.sp[k]) (void)hipStreamDestroy(L.sp[k]);
    if (L.su[k]) (void)hipStreamDestroy(L.su[k]);
  }
  DBG_FREE("lane: free buffers");
  if (L.A) (void)hipFree(L.A);
  if (L.B2) (void)hipFree(L.B2);
  if (L.inv2) (void)hipFree(L.inv2);
  if (L.inv2_t) (void)hipFree(L.inv2_t);
  if (L.gws) (void)hipFree(L.gws);
  if (L.invD) (void)hipFree(L.invD);
  if (L.vec) (void)hipFree(L.vec);
  if (L.red) (void)hipFree(L.red);
  if (L.info) (void)hipFree(L.info);
  if (L.h_red) (void)hipHostFree(L.h_red);
  if (L.h_info) (void)hipHostFree(L.h_info);
  DBG_FREE("lane: destroy main stream");
  if (L.stream && L.owns_stream) (void)hipStreamDestroy(L.stream);
  DBG_FREE("lane: done");
  L = Lane();
}

int ensure_lanes(gpmi_ctx* c, size_t count) {
  while (c->lanes.size() < count) {
    c->lanes.emplace_back();
    int rc = lane_alloc(c, c->lanes.back());
    if (rc != GPMI_OK) {
      lane_free(c->lanes.back());
      c->lanes.pop_back();
      return rc;
    }
  }
  return GPMI_OK;
}

void linv_free(LinvState& S);

void free_data(gpmi_ctx* c) {
  linv_free(c->linv);
  for (double** q : {&c->mix_g, &c->mix_scratch, &c->mix_zero, &c->Q3}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  c->mix_nk = 0;
  c->q3_cap = 0;
  for (size_t i = c->lanes.size(); i-- > 0;) lane_free(c->lanes[i]);  // (lanes >= 2 borrow the streams of lanes 0, 1)
  c->lanes.clear();
  auto fr = [](double*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(c->x);
  fr(c->y);
  fr(c->noise);
  fr(c->ycov);
  fr(c->alpha);
  fr(c->Q);
  fr(c->Q2);
  fr(c->pts);
  fr(c->pvec);
  fr(c->bA);
  fr(c->bInv);
  fr(c->bVec);
  fr(c->bRed);
  fr(c->bMu);
  fr(c->bB2);  // (the gradient batches' buffers were left behind here until round 4: a handle given a second data set of
  fr(c->bGws);  //  another size kept using the first one's)
  fr(c->bGout);
  fr(c->bLoo);
  fr(c->bNoise);
  fr(c->bMixG);
  fr(c->bMixH);
  fr(c->bMixW);
  fr(c->bMixExtra);
  if (c->bMixP) (void)hipFree(c->bMixP);
  c->bMixP = nullptr;
  if (c->h_bGout) (void)hipHostFree(c->h_bGout);
  c->h_bGout = nullptr;
  c->bgrad_cap = c->bgrad_ntheta = c->bLoo_cap = c->bNoise_cap = c->bMix_cap = 0;
  if (c->bInfo) (void)hipFree(c->bInfo);
  c->bInfo = nullptr;
  if (c->bParams) (void)hipFree(c->bParams);
  c->bParams = nullptr;
  if (c->h_bRed) (void)hipHostFree(c->h_bRed);
  c->h_bRed = nullptr;
  if (c->h_bInfo) (void)hipHostFree(c->h_bInfo);
  c->h_bInfo = nullptr;
  for (int k = 0; k < 2; ++k) {
    if (c->h_bStage[k]) (void)hipHostFree(c->h_bStage[k]);
    c->h_bStage[k] = nullptr;
    c->h_bStage_bytes[k] = 0;
    c->bpend[k] = 0;
  }
  c->bcap = 0;
  c->mq_cap = 0;
  c->fitted = false;
  c->n = c->d = c->np = c->ld = 0;
}

int make_params(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra,
                KParams& p) {
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, kernel == GPMI_KERNEL_SE || kernel == GPMI_KERNEL_RQ, "unknown kernel id");
  const int off = (kernel == GPMI_KERNEL_SE) ? 1 : 2;
  ARGCHK(c, n_theta == c->d + off, "n_theta does not match the kernel and the data dimension");
  ARGCHK(c, theta != nullptr, "theta is NULL");
  std::memset(&p, 0, sizeof(p));
  p.kernel = kernel;
  p.d = (int)c->d;
  const double a = std::exp(theta[0]);
  p.a2 = a * a;                                             // (a**2), covariance.py:255
  p.kappa = (kernel == GPMI_KERNEL_RQ) ? std::exp(theta[1]) : 1.0;
  p.extra_diag = extra;
  for (int k = 0; k < p.d; ++k) {
    const double l = std::exp(theta[off + k]);
    p.inv_l2[k] = 1.0 / (l * l);
  }
  return GPMI_OK;
}

int set_device(gpmi_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  return GPMI_OK;
}


// dst (np x ld) = sum_m diag(g_m) K_m diag(g_m) + diag(noise + extra); every K_m carries its own a^2 1e-12
// jitter (covariance.py:546-559 builds the sub-kernels with build_covariance) and the identity padding
// survives through g_0 = 1 there
void build_mix_square(gpmi_ctx* c, hipStream_t s, const MixEval& mx, double* dst, bool lower_only) {
  for (int m = 0; m < mx.nk; ++m) {
    KParams pm = mx.p[m];
    pm.extra_diag = 0.0;
    launch_kbuild_square(s, pm, c->x, c->n, c->np, mx.zero, mx.scratch, c->ld, lower_only);
    launch_scale_add(s, dst, c->ld, mx.scratch, c->ld, mx.g + (int64_t)m * c->np, mx.g + (int64_t)m * c->np,
                     c->np, c->np, m > 0);
  }
  launch_add_diag_vec(s, dst, c->ld, c->noise, mx.extra, c->n);
}

// K(theta) + sig into lane.A (lower tiles), factorise, forward-solve the residual, reduce.
// Leaves: lane.A = L, lane.invD, vec[0:np] = v = L^-1 (y - mu), red[2*slot..] = {v.v, sum ln L_ii},
// info[slot].  `mu_dev` may be null (then mu_const is used).  `mix` != nullptr: mixture covariance.
int enqueue_factor_and_forward(gpmi_ctx* c, Lane& L, const KParams& p, const double* mu_dev,
                               double mu_const, int slot, bool allow_lookahead,
                               const MixEval* mix, bool prebuild_inv2, double* backward_out, double* early_identity) {
  hipStream_t s = L.stream;
  L.inv2_valid = false;
  potrf_pair_quiesce(L);  // (before this call's own work reaches the pair: the build below uses the update stream)
  HIPCHK(c, hipMemsetAsync(L.info + slot, 0, sizeof(int), s));
  if (mix) {
    build_mix_square(c, s, *mix, L.A, false);
  } else {
    // When the factorisation starts in its look-ahead regime the build is split: the first outer panel's columns on
    // this stream, the tiles to their right on the update stream - where the first trailing update will follow them in
    // stream order - so that the first panel (a 0.5 ms chain on an otherwise idle chip) is factored while they are
    // built.  (Not while the K-build class is being timed by events on this stream.)
    hipStream_t su = (!c->ycov && !((c->prof_mask >> GPMI_PROF_KBUILD) & 1))
                         ? potrf_first_update_stream(c, L, c->np, allow_lookahead)
                         : nullptr;
    static const bool no_split = std::getenv("GPMI_KBUILD_NO_SPLIT") != nullptr;
    if (su && !no_split) {
      HIPCHK(c, hipEventRecord(L.ev_join, s));  // the matrix is free once everything queued so far is through
      HIPCHK(c, hipStreamWaitEvent(su, L.ev_join, 0));
      launch_kbuild_square_part(s, p, c->x, c->n, c->np, c->noise, L.A, c->ld, 1, GPMI_OB);
      launch_kbuild_square_part(su, p, c->x, c->n, c->np, c->noise, L.A, c->ld, 2, GPMI_OB);
    } else {
      ProfScope ps(c, s, GPMI_PROF_KBUILD, 0.0, 4.0 * c->np * c->np);
      launch_kbuild_square(s, p, c->x, c->n, c->np, c->noise, L.A, c->ld, true);
    }
  }
  if (c->ycov) launch_add_full(s, L.A, c->ld, c->ycov, c->n);
  // the residual and the sweeps' sentinel fills do not depend on the factor: the factorisation enqueues them where the
  // lane's stream idles (GPMI_EARLY_FILL=0: behind it, as until round 6); `backward_out`: the caller's backward sweep
  // writes there next and passes prefilled = true
  static const bool early_fill = !std::getenv("GPMI_EARLY_FILL") || std::atoi(std::getenv("GPMI_EARLY_FILL")) != 0;
  L.early = Lane::EarlyWork();
  L.early.pending = true;
  L.early.y = c->y;
  L.early.mu = mu_dev;
  L.early.mu_const = mu_const;
  L.early.r = L.vec + 2 * c->np;
  L.early.n = c->n;
  L.early.np = c->np;
  L.early.fill[0] = L.vec;
  L.early.fill[1] = backward_out;
  L.early.ident = early_identity;  // (the caller computes the inverse factor next: its right-hand side, 70 us at N = 8192)
  L.early.ident_ld = c->ld;
  L.identity_ready = false;
  if (!early_fill) L.early.pending = false;
  potrf_lower(c, L, L.A, c->np, c->ld, L.invD, L.info + slot, allow_lookahead);
  if (!early_fill) L.early.pending = true;
  lane_run_early(L, s);  // (a no-op when the factorisation found a place for it)
  if (prebuild_inv2 && L.su[0] && c->np >= 4 * GPMI_OB) {
    // the fit's caller predicts next: the inverses of the 512 x 512 diagonal blocks (six small batched launches, 0.2 ms
    // on a stream of their own) are built on the lane's update stream beside the two triangular sweeps, which are
    // chain-latency bound and leave most of the chip idle; the caller orders its stream behind ev_main
    HIPCHK(c, hipEventRecord(L.ev_join, s));
    HIPCHK(c, hipStreamWaitEvent(L.su[0], L.ev_join, 0));
    if (int rc = ensure_inv2(c, L, L.su[0])) return rc;
    HIPCHK(c, hipEventRecord(L.ev_main, L.su[0]));
  }
  trsv_forward(c, s, L.A, c->np, c->ld, L.invD, L.vec + 2 * c->np, L.vec, L.info + slot, BatchShape(), true);
  if (prebuild_inv2 && L.inv2_valid && L.su[0]) {
    // (same caller: v . v and sum ln L_ii - one workgroup, 42 us - beside the backward sweep instead of in front of it)
    HIPCHK(c, hipEventRecord(L.ev_la, s));
    HIPCHK(c, hipStreamWaitEvent(L.su[0], L.ev_la, 0));
    launch_lml_reduce(L.su[0], L.vec, L.A, c->ld, c->np, L.red + 2 * slot);
    HIPCHK(c, hipEventRecord(L.ev_main, L.su[0]));
  } else {
    launch_lml_reduce(s, L.vec, L.A, c->ld, c->np, L.red + 2 * slot);
  }
  HIPCHK(c, hipGetLastError());
  return GPMI_OK;
}

int ensure_second_matrix(gpmi_ctx* c, Lane& L) {
  if (!L.B2) HIPCHK(c, hipMalloc(&L.B2, sizeof(double) * c->np * c->ld));
  return GPMI_OK;
}

// inverses of the 512-wide diagonal blocks of the factor held by lane F, built on stream s (which must
// be ordered after the factorisation); cached until the lane is factorised again
int ensure_inv2(gpmi_ctx* c, Lane& F, hipStream_t s) {
  if (F.inv2_valid) return GPMI_OK;
  const int64_t nob = (c->np / GPMI_NB + 3) / 4;
  if (!F.inv2) HIPCHK(c, hipMalloc(&F.inv2, sizeof(double) * nob * GPMI_OB * GPMI_OB));
  if (!F.inv2_t) HIPCHK(c, hipMalloc(&F.inv2_t, sizeof(double) * nob * 256 * 256));
  build_inv2(s, F.A, c->np, c->ld, F.invD, F.inv2, F.inv2_t);
  HIPCHK(c, hipGetLastError());
  F.inv2_valid = true;
  return GPMI_OK;
}

int ensure_trsm_panel(gpmi_ctx* c, int64_t rows) {
  if (rows <= c->trsm_panel_rows) return GPMI_OK;
  if (c->trsm_panel) (void)hipFree(c->trsm_panel);
  c->trsm_panel = nullptr;
  c->trsm_panel_rows = 0;
  HIPCHK(c, hipMalloc(&c->trsm_panel, sizeof(double) * rows * (GPMI_OB + 32)));
  c->trsm_panel_rows = rows;
  return GPMI_OK;
}

// L.B2 <- F^-T (row j = column j of the inverse of lane F's factor), by forward substitution on the identity
int enqueue_inverse_factor(gpmi_ctx* c, Lane& L, Lane& F) {
  if (int rc = ensure_inv2(c, F, L.stream)) return rc;
  if (int rc = ensure_trsm_panel(c, c->np)) return rc;
  if (!L.identity_ready) launch_set_identity(L.stream, L.B2, c->ld, c->np);  // (else: written beside the factorisation, Lane::EarlyWork)
  L.identity_ready = false;
  trsm_rows_forward(c, L.stream, F.A, c->np, c->ld, F.inv2, L.B2, c->np, true, nullptr, c->trsm_panel);
  return GPMI_OK;
}

// workspace for `want` small problems advancing in lockstep (capped by a 24 GiB budget and 512 matrices: round 6 - 6 GiB /
// 256 until then; one chunk of 512 evaluations at N = 2048 instead of three of 171 is 3.7 % faster, and two asynchronous
// slots of 256 carry config 5's 512 chains in two groups: 14 600 -> 15 450 evaluations/s through the tempering driver)
int ensure_batch_ws(gpmi_ctx* c, int want) {
  const int64_t per = c->np * c->ld * (int64_t)sizeof(double);
  // matrices of one lockstep chunk: GPMI_BATCH_GIB (GiB of them) and GPMI_BATCH_MAX (their number) bound the workspace
  static const int64_t budget_gib = [] {
    const char* e = std::getenv("GPMI_BATCH_GIB");
    const int v = e ? std::atoi(e) : 24;
    return (int64_t)(v >= 1 && v <= 200 ? v : 24);
  }();
  static const int max_chunk = [] {
    const char* e = std::getenv("GPMI_BATCH_MAX");
    const int v = e ? std::atoi(e) : 512;
    return v >= 2 && v <= 4096 ? v : 512;
  }();
  int cap = (int)((budget_gib << 30) / per);
  if (cap > max_chunk) cap = max_chunk;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want <= c->bcap) return GPMI_OK;
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "the batch workspace cannot grow while an asynchronous batch is pending (gpmi_lml_batch_wait first)");
  auto fr = [](double*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(c->bA);
  fr(c->bInv);
  fr(c->bVec);
  fr(c->bRed);
  fr(c->bMu);
  fr(c->bB2);
  fr(c->bGws);
  fr(c->bGout);
  fr(c->bLoo);
  fr(c->bNoise);
  fr(c->bMixG);
  fr(c->bMixH);
  fr(c->bMixW);
  fr(c->bMixExtra);
  if (c->bMixP) (void)hipFree(c->bMixP);
  c->bMixP = nullptr;
  c->bLoo_cap = c->bNoise_cap = c->bMix_cap = 0;
  if (c->h_bGout) (void)hipHostFree(c->h_bGout);
  c->h_bGout = nullptr;
  c->bgrad_cap = c->bgrad_ntheta = 0;
  if (c->bInfo) (void)hipFree(c->bInfo);
  if (c->bParams) (void)hipFree(c->bParams);
  if (c->h_bRed) (void)hipHostFree(c->h_bRed);
  if (c->h_bInfo) (void)hipHostFree(c->h_bInfo);
  c->bInfo = nullptr;
  c->bParams = nullptr;
  c->h_bRed = nullptr;
  c->h_bInfo = nullptr;
  c->bcap = 0;
  const int64_t nt = c->np / GPMI_NB;
  HIPCHK(c, hipMalloc(&c->bA, sizeof(double) * want * c->np * c->ld));
  HIPCHK(c, hipMalloc(&c->bInv, sizeof(double) * want * nt * GPMI_NB * GPMI_NB));
  ZERO_SYNC(c, c->bInv, sizeof(double) * want * nt * GPMI_NB * GPMI_NB);  // see lane_alloc
  HIPCHK(c, hipMalloc(&c->bVec, sizeof(double) * want * 4 * c->np));
  HIPCHK(c, hipMalloc(&c->bRed, sizeof(double) * 2 * want));
  HIPCHK(c, hipMalloc(&c->bMu, sizeof(double) * want * c->np));
  HIPCHK(c, hipMalloc(&c->bInfo, sizeof(int) * want));
  HIPCHK(c, hipMalloc(&c->bParams, sizeof(KParams) * want));
  HIPCHK(c, hipHostMalloc(&c->h_bRed, sizeof(double) * 2 * want));
  HIPCHK(c, hipHostMalloc(&c->h_bInfo, sizeof(int) * want));
  c->bcap = want;
  return GPMI_OK;
}

// second matrix, contraction partials and result slots for gradient batches of `want` problems (after ensure_batch_ws)
int ensure_batch_grad_ws(gpmi_ctx* c, int want, int n_theta) {
  if (want > c->bcap) want = c->bcap;
  if (want <= c->bgrad_cap && n_theta <= c->bgrad_ntheta) return GPMI_OK;
  auto fr = [](double*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(c->bB2);
  fr(c->bGws);
  fr(c->bGout);
  if (c->h_bGout) (void)hipHostFree(c->h_bGout);
  c->h_bGout = nullptr;
  c->bgrad_cap = c->bgrad_ntheta = 0;
  const int cap = c->bcap;
  HIPCHK(c, hipMalloc(&c->bB2, sizeof(double) * cap * c->np * c->ld));
  HIPCHK(c, hipMalloc(&c->bGws, sizeof(double) * cap * grad_ws_doubles(c->np, n_theta)));
  HIPCHK(c, hipMalloc(&c->bGout, sizeof(double) * cap * (n_theta + 1)));
  HIPCHK(c, hipHostMalloc(&c->h_bGout, sizeof(double) * cap * (n_theta + 1)));
  c->bgrad_cap = cap;
  c->bgrad_ntheta = n_theta;
  return GPMI_OK;
}

int ensure_query_ws(gpmi_ctx* c, int64_t mp) {
  if (mp <= c->mq_cap) return GPMI_OK;
  auto fr = [](double*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(c->Q);
  fr(c->Q2);
  fr(c->pts);
  fr(c->pvec);
  c->mq_cap = 0;
  HIPCHK(c, hipMalloc(&c->Q, sizeof(double) * mp * c->ld));
  HIPCHK(c, hipMalloc(&c->Q2, sizeof(double) * mp * c->ld));
  HIPCHK(c, hipMalloc(&c->pts, sizeof(double) * mp * c->d));
  HIPCHK(c, hipMalloc(&c->pvec, sizeof(double) * mp * (2 + 2 * c->d + c->d * c->d + GPMI_MAX_MIX)));
  c->mq_cap = mp;
  return GPMI_OK;
}

// ---- instrumentation ----------------------------------------------------------------
namespace {
__global__ void qdiag_batched_kernel(const double* __restrict__ iK, int64_t ld, const double* __restrict__ alpha,
                                     double* __restrict__ out, int64_t n, int64_t sMat, int64_t sVec, int64_t sOut) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, z = blockIdx.z;
  if (i < n) out[z * sOut + i] = alpha[z * sVec + i] * alpha[z * sVec + i] - iK[z * sMat + i * ld + i];
}

__global__ void negate_kernel(double* v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = -v[i];
}

__global__ void qdiag_kernel(const double* __restrict__ iK, int64_t ld, const double* __restrict__ alpha,
                             double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha[i] * alpha[i] - iK[i * ld + i];
}

__global__ void stamp_init_kernel(unsigned long long* pool, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots)
    for (int j = 0; j < GPMI_STAMP_WORDS; ++j) pool[(size_t)GPMI_STAMP_WORDS * i + j] = (j < 8) ? ~0ull : 0ull;
}
}  // namespace

void launch_negate(hipStream_t s, double* v, int64_t n) {
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n);
}
void launch_qdiag(hipStream_t s, const double* iK, int64_t ld, const double* alpha, double* out, int64_t n) {
  hipLaunchKernelGGL(qdiag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, iK, ld, alpha, out, n);
}
void launch_qdiag_batched(hipStream_t s, int batch, const double* iK, int64_t ld, const double* alpha, double* out, int64_t n,
                          int64_t sMat, int64_t sVec, int64_t sOut) {
  hipLaunchKernelGGL(qdiag_batched_kernel, dim3((unsigned)((n + 255) / 256), 1, (unsigned)batch), dim3(256), 0, s, iK, ld,
                     alpha, out, n, sMat, sVec, sOut);
}

unsigned long long* prof_stamp_slot(gpmi_ctx* c, double flops, double bytes, int klass) {
  if (!((c->prof_mask >> GPMI_PROF_SYRK) & 1) || !c->stamp_pool) return nullptr;
  if ((int)c->stamp_flops.size() >= GPMI_STAMP_SLOTS) return nullptr;
  c->stamp_flops.push_back(flops);
  c->stamp_bytes.push_back(bytes);
  c->stamp_class.push_back(klass);
  return c->stamp_pool + (size_t)GPMI_STAMP_WORDS * (c->stamp_flops.size() - 1);
}

ProfScope::ProfScope(gpmi_ctx* ctx, hipStream_t st, int klass, double flops, double bytes)
    : c(ctx), s(st), slot(nullptr) {
  if (!((c->prof_mask >> klass) & 1) || klass == GPMI_PROF_SYRK || klass == GPMI_PROF_SYRK_REST ||
      klass == GPMI_PROF_SYRK_SLICE || klass == GPMI_PROF_FLOW)
    return;
  if (c->prof_used == c->prof_slots.size()) {
    ProfSlot ns{};
    if (hipEventCreate(&ns.e0) != hipSuccess || hipEventCreate(&ns.e1) != hipSuccess) return;
    c->prof_slots.push_back(ns);
  }
  slot = &c->prof_slots[c->prof_used++];
  slot->klass = klass;
  slot->flops = flops;
  slot->bytes = bytes;
  (void)hipEventRecord(slot->e0, s);
}
ProfScope::~ProfScope() {
  if (slot) (void)hipEventRecord(slot->e1, s);
}
extern "C" {

int gpmi_version(void) { return GPMI_VERSION; }

int gpmi_device_count(int* count) {
  if (!count) return GPMI_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return GPMI_ERR_NODEVICE;
  }
  *count = n;
  return GPMI_OK;
}

int gpmi_device_pci_bus_id(int device, char* buf, int cap) {
  if (!buf || cap < 16) return GPMI_ERR_ARG;
  buf[0] = 0;
  if (hipDeviceGetPCIBusId(buf, cap, device) != hipSuccess) {
    (void)hipGetLastError();
    buf[0] = 0;
    return GPMI_ERR_NODEVICE;
  }
  buf[cap - 1] = 0;
  return GPMI_OK;
}

// The GPU queues of a process stall for 15 - 30 ms whenever glibc hands the top of the heap back to the kernel (brk
// shrink -> MMU notifier -> the driver stops the process's queues until the invalidation is through; ROCm 7.2, MI355X).
// A host loop that allocates and frees a few hundred KB per call (NumPy result arrays around every entry point) trims
// and re-grows the heap EVERY call once its peak exceeds M_TRIM_THRESHOLD (128 KiB by default, raised dynamically only
// if the application happens to free an mmapped block first - which made it a per-process lottery): every lockstep batch
// of the ChangePoint search at N = 2048 then took 28 ms instead of 4 (0.4 -> 2.6 s; profiles/r05_search.json,
// profiles/r06_search_regression.txt: MALLOC_TRIM_THRESHOLD_ alone cures it, munmap of mmapped blocks is harmless,
// AMD_DIRECT_DISPATCH=0 hides it).  The first handle of a process therefore keeps up to 1 GiB of freed heap top in the
// process (mallopt; a process-wide setting, like the MALLOC_TRIM_THRESHOLD_ environment variable).  GPMI_MALLOC_TRIM=keep
// leaves the allocator alone, and so does an application that has set MALLOC_TRIM_THRESHOLD_ / MALLOC_TOP_PAD_ itself.
static void keep_heap_top_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char* e = std::getenv("GPMI_MALLOC_TRIM");
    if (e && std::strcmp(e, "keep") == 0) return;
    if (std::getenv("MALLOC_TRIM_THRESHOLD_") || std::getenv("MALLOC_TOP_PAD_")) return;
    (void)mallopt(M_TRIM_THRESHOLD, 1 << 30);
    (void)mallopt(M_TOP_PAD, 16 << 20);  // grow in 16 MiB steps (growing does not stall anything; fewer brk calls)
  });
}

int gpmi_create(int device, gpmi_ctx** out) {
  if (!out) return GPMI_ERR_ARG;
  *out = nullptr;
  keep_heap_top_once();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_create_err = "no HIP device visible";
    return GPMI_ERR_NODEVICE;
  }
  if (device < 0 || device >= n) {
    g_create_err = "device index out of range";
    return GPMI_ERR_ARG;
  }
  gpmi_ctx* c = new gpmi_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipEventCreate(&c->t0) != hipSuccess ||
      hipEventCreate(&c->t1) != hipSuccess) {
    g_create_err = "cannot initialise the device";
    delete c;
    return GPMI_ERR_HIP;
  }
  *out = c;
  return GPMI_OK;
}

int gpmi_destroy(gpmi_ctx* c) {
  if (!c) return GPMI_OK;
  (void)hipSetDevice(c->device);
  // every stream of the handle synchronised once here and once more in lane_free: hipStreamDestroy on a CU-masked
  // stream that has only been synchronised once blocked forever on ROCm 7.2 (round-2 probe, profiles/HISTORY.md)
  (void)gpmi_sync(c);
  DBG_FREE("comm destroy");
  (void)gpmi_comm_destroy(c);
  DBG_FREE("free data");
  free_data(c);
  DBG_FREE("rest");
  for (auto& sl : c->prof_slots) {
    (void)hipEventDestroy(sl.e0);
    (void)hipEventDestroy(sl.e1);
  }
  if (c->stamp_pool) (void)hipFree(c->stamp_pool);
  if (c->dev_masked) (void)hipStreamDestroy(c->dev_masked);
  if (c->trsm_panel) (void)hipFree(c->trsm_panel);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->t0) (void)hipEventDestroy(c->t0);
  if (c->t1) (void)hipEventDestroy(c->t1);
  delete c;
  return GPMI_OK;
}

const char* gpmi_last_error(const gpmi_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int gpmi_sync(gpmi_ctx* c) {
  if (!c) return GPMI_ERR_ARG;
  if (set_device(c)) return GPMI_ERR_HIP;
  for (auto& L : c->lanes) {
    HIPCHK(c, hipStreamSynchronize(L.stream));
    for (int k = 0; k < GPMI_NPAIRS; ++k) {
      if (L.sp[k]) HIPCHK(c, hipStreamSynchronize(L.sp[k]));
      if (L.su[k]) HIPCHK(c, hipStreamSynchronize(L.su[k]));
    }
  }
  return GPMI_OK;
}

int gpmi_set_data(gpmi_ctx* c, const double* x, const double* y, const double* noise_var,
                  const double* y_cov, int64_t n, int64_t d) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, x && y && n > 0 && d > 0, "x, y must be non-NULL and n, d positive");
  ARGCHK(c, d <= GPMI_MAX_D, "more spatial dimensions than GPMI_MAX_D (64)");
  if (int rc = set_device(c)) return rc;
  free_data(c);
  c->n = n;
  c->d = d;
  c->np = round_up(n + c->reserve, GPMI_NB);
  c->ld = c->np + 32;  // keep rows 256-byte aligned but off a power-of-two pitch
  HIPCHK(c, hipMalloc(&c->x, sizeof(double) * c->np * d));
  HIPCHK(c, hipMalloc(&c->y, sizeof(double) * c->np));
  HIPCHK(c, hipMalloc(&c->noise, sizeof(double) * c->np));
  HIPCHK(c, hipMalloc(&c->alpha, sizeof(double) * c->np));
  ZERO_SYNC(c, c->x, sizeof(double) * c->np * d);
  ZERO_SYNC(c, c->y, sizeof(double) * c->np);
  ZERO_SYNC(c, c->noise, sizeof(double) * c->np);
  HIPCHK(c, hipMemcpy(c->x, x, sizeof(double) * n * d, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->y, y, sizeof(double) * n, hipMemcpyHostToDevice));
  if (y_cov) {
    HIPCHK(c, hipMalloc(&c->ycov, sizeof(double) * n * n));
    HIPCHK(c, hipMemcpy(c->ycov, y_cov, sizeof(double) * n * n, hipMemcpyHostToDevice));
  } else if (noise_var) {
    HIPCHK(c, hipMemcpy(c->noise, noise_var, sizeof(double) * n, hipMemcpyHostToDevice));
  }
  return ensure_lanes(c, 1);
}

int gpmi_set_streams(gpmi_ctx* c, int n_streams) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, c->n > 0, "gpmi_set_data has not been called");
  ARGCHK(c, n_streams >= 1 && n_streams <= 256, "n_streams out of range");
  ARGCHK(c, c->bpend[0] == 0 && c->bpend[1] == 0,
         "an asynchronous batch is pending on the lanes this call may release (gpmi_lml_batch_wait first)");
  if (int rc = set_device(c)) return rc;
  while ((int)c->lanes.size() > 1 + n_streams) {
    lane_free(c->lanes.back());
    c->lanes.pop_back();
  }
  return ensure_lanes(c, 1 + (size_t)n_streams);
}

int gpmi_set_option(gpmi_ctx* c, int option, int value) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, option == GPMI_OPT_LOCKSTEP_ALWAYS || option == GPMI_OPT_RESERVE_POINTS || option == GPMI_OPT_NO_FLOW,
         "unknown option");
  if (option == GPMI_OPT_LOCKSTEP_ALWAYS) c->lockstep_always = value != 0;
  if (option == GPMI_OPT_NO_FLOW) c->no_flow = value != 0;
  if (option == GPMI_OPT_RESERVE_POINTS) {
    ARGCHK(c, value >= 0, "reserve must be >= 0");
    c->reserve = value;
  }
  return GPMI_OK;
}
// ---- instrumentation ------------------------------------------------------------------
int gpmi_timer_start(gpmi_ctx* c) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, !c->lanes.empty(), "gpmi_set_data has not been called");
  if (int rc = gpmi_sync(c)) return rc;
  HIPCHK(c, hipEventRecord(c->t0, c->lanes[0].stream));
  return GPMI_OK;
}

int gpmi_timer_stop(gpmi_ctx* c, float* ms) {
  if (!c || !ms) return GPMI_ERR_ARG;
  ARGCHK(c, !c->lanes.empty(), "gpmi_set_data has not been called");
  if (int rc = set_device(c)) return rc;
  // the stop event is recorded on lane 0 after every other lane has drained into it
  for (size_t i = 1; i < c->lanes.size(); ++i) {
    hipEvent_t ev;
    HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(ev, c->lanes[i].stream));
    HIPCHK(c, hipStreamWaitEvent(c->lanes[0].stream, ev, 0));
    HIPCHK(c, hipEventDestroy(ev));
  }
  HIPCHK(c, hipEventRecord(c->t1, c->lanes[0].stream));
  HIPCHK(c, hipEventSynchronize(c->t1));
  HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
  return GPMI_OK;
}

static int dev_stream(gpmi_ctx* c, hipStream_t* s) {
  if (c->lanes.empty()) {
    // a bare handle (no data yet): give it a stream-only lane
    c->lanes.emplace_back();
    if (int rc = lane_streams(c, c->lanes[0])) return rc;
  }
  *s = c->lanes[0].stream;
  // tools only (tools/cu_scaling.sh): GPMI_DEV_CUS=<n> runs the device-pointer entry points on n CUs (n / 8 per XCD)
  if (const char* e = std::getenv("GPMI_DEV_CUS")) {
    if (!c->dev_masked) {
      const int n = std::atoi(e);
      std::vector<uint32_t> m((size_t)c->ncu / 32, 0u);
      for (int i = 0; i < n && i < c->ncu; ++i) m[(size_t)i / 32] |= 1u << (i % 32);
      HIPCHK(c, hipExtStreamCreateWithCUMask(&c->dev_masked, (uint32_t)m.size(), m.data()));
    }
    *s = c->dev_masked;
  }
  return GPMI_OK;
}

int gpmi_profile_enable(gpmi_ctx* c, int on) {
  if (!c) return GPMI_ERR_ARG;
  c->prof_mask = (on == 1) ? ((1u << GPMI_PROF_NCLASS) - 1u) : (unsigned)on >> 1;
  if (((c->prof_mask >> GPMI_PROF_SYRK) & 1) && !c->stamp_pool) {
    if (int rc = set_device(c)) return rc;
    HIPCHK(c, hipMalloc(&c->stamp_pool, sizeof(unsigned long long) * GPMI_STAMP_WORDS * GPMI_STAMP_SLOTS));
    // on the library's own stream: a kernel launched on the NULL stream left every later launch of the process
    // with a ~10 us dispatch gap to its predecessor (1.2 ms per fit at N = 16384)
    hipStream_t s0;
    if (int rc = dev_stream(c, &s0)) return rc;
    hipLaunchKernelGGL(stamp_init_kernel, dim3(GPMI_STAMP_SLOTS / 256), dim3(256), 0, s0, c->stamp_pool,
                       GPMI_STAMP_SLOTS);
    HIPCHK(c, hipStreamSynchronize(s0));
  }
  return GPMI_OK;
}

static int profile_collect(gpmi_ctx* c) {
  if (int rc = gpmi_sync(c)) return rc;
  for (size_t i = 0; i < c->prof_used; ++i) {
    ProfSlot& sl = c->prof_slots[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sl.e0, sl.e1) != hipSuccess) continue;
    c->prof_ms[sl.klass] += ms;
    c->prof_flops[sl.klass] += sl.flops;
    c->prof_bytes[sl.klass] += sl.bytes;
    c->prof_launches[sl.klass] += 1;
  }
  c->prof_used = 0;
  // device-stamped trailing-update launches: duration = (max end - min start) / 100 MHz
  const size_t ns = c->stamp_flops.size();
  if (ns && c->stamp_pool) {
    std::vector<unsigned long long> h((size_t)GPMI_STAMP_WORDS * ns);
    hipStream_t s0;
    if (int rc = dev_stream(c, &s0)) return rc;
    HIPCHK(c, hipMemcpyAsync(h.data(), c->stamp_pool, sizeof(unsigned long long) * GPMI_STAMP_WORDS * ns,
                             hipMemcpyDeviceToHost, s0));
    HIPCHK(c, hipStreamSynchronize(s0));
    for (size_t i = 0; i < ns; ++i) {
      // words 0..7: start of the launch's first eight workgroups; words 8..15: last end seen on each XCD
      const unsigned long long* w = h.data() + (size_t)GPMI_STAMP_WORDS * i;
      unsigned long long t0 = ~0ull, t1 = 0ull;
      for (int j = 0; j < 8; ++j) {
        if (w[j] < t0) t0 = w[j];
        if (w[8 + j] > t1) t1 = w[8 + j];
      }
      if (t1 <= t0) continue;  // launch never ran
      const int kl = c->stamp_class[i];
      if (kl >= GPMI_PROF_NCLASS) {  // GPMI_CHAIN_TRACE (potrf.hip): 0 potrf_diag, 1 panel TRSM, 2 inner update
        static unsigned long long origin = 0;
        if (!origin || kl == GPMI_PROF_NCLASS + 3) origin = t0;
        std::fprintf(stderr, "[chain] %d %.2f %.2f\n", kl - GPMI_PROF_NCLASS, (double)(t0 - origin) * 0.01,
                     (double)(t1 - origin) * 0.01);
        continue;
      }
      for (int j = 0; j < 8; ++j) {
        c->prof_clock_cycles += (double)(w[16 + j] >> 32);
        c->prof_clock_ticks += (double)(w[16 + j] & 0xffffffffull);
      }
      c->prof_ms[kl] += (double)(t1 - t0) * 1e-5;  // 10 ns ticks -> ms
      c->prof_flops[kl] += c->stamp_flops[i];
      c->prof_bytes[kl] += c->stamp_bytes[i];
      c->prof_launches[kl] += 1;
    }
    hipLaunchKernelGGL(stamp_init_kernel, dim3(GPMI_STAMP_SLOTS / 256), dim3(256), 0, s0, c->stamp_pool,
                       GPMI_STAMP_SLOTS);
    HIPCHK(c, hipStreamSynchronize(s0));
    c->stamp_flops.clear();
    c->stamp_bytes.clear();
    c->stamp_class.clear();
  }
  return GPMI_OK;
}

int gpmi_profile_read(gpmi_ctx* c, int klass, int64_t* launches, double* ms, double* flops,
                      double* bytes) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, klass >= 0 && klass < GPMI_PROF_NCLASS, "bad profile class");
  if (int rc = profile_collect(c)) return rc;
  if (launches) *launches = c->prof_launches[klass];
  if (ms) *ms = c->prof_ms[klass];
  if (flops) *flops = c->prof_flops[klass];
  if (bytes) *bytes = c->prof_bytes[klass];
  return GPMI_OK;
}

int gpmi_profile_clock(gpmi_ctx* c, double* ghz) {
  if (!c || !ghz) return GPMI_ERR_ARG;
  if (int rc = profile_collect(c)) return rc;
  *ghz = (c->prof_clock_ticks > 0.0) ? c->prof_clock_cycles / c->prof_clock_ticks * 0.1 : 0.0;
  return GPMI_OK;
}

int gpmi_profile_reset(gpmi_ctx* c) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = profile_collect(c)) return rc;
  for (int k = 0; k < GPMI_PROF_NCLASS; ++k) {
    c->prof_ms[k] = c->prof_flops[k] = c->prof_bytes[k] = 0.0;
    c->prof_launches[k] = 0;
  }
  c->prof_clock_cycles = c->prof_clock_ticks = 0.0;
  return GPMI_OK;
}

// ---- device-pointer entry points ---------------------------------------------------------
int gpmi_dev_alloc(gpmi_ctx* c, int64_t bytes, void** ptr) {
  if (!c || !ptr) return GPMI_ERR_ARG;
  if (int rc = set_device(c)) return rc;
  HIPCHK(c, hipMalloc(ptr, (size_t)bytes));
  return GPMI_OK;
}
int gpmi_dev_free(gpmi_ctx* c, void* ptr) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = set_device(c)) return rc;
  HIPCHK(c, hipFree(ptr));
  return GPMI_OK;
}
int gpmi_dev_upload(gpmi_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = set_device(c)) return rc;
  HIPCHK(c, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return GPMI_OK;
}
int gpmi_dev_download(gpmi_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c) return GPMI_ERR_ARG;
  if (int rc = set_device(c)) return rc;
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return GPMI_OK;
}

int gpmi_dev_potrf(gpmi_ctx* c, double* A, int64_t n, int64_t ld, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, A && n > 0 && n % GPMI_NB == 0 && ld >= n && ld % 2 == 0, "bad matrix shape");
  if (int rc = set_device(c)) return rc;
  hipStream_t s;
  if (int rc = dev_stream(c, &s)) return rc;
  double* invD = nullptr;
  int* dinfo = nullptr;
  HIPCHK(c, hipMalloc(&invD, sizeof(double) * (n / GPMI_NB) * GPMI_NB * GPMI_NB));
  ZERO_SYNC(c, invD, sizeof(double) * (n / GPMI_NB) * GPMI_NB * GPMI_NB);  // see lane_alloc
  HIPCHK(c, hipMalloc(&dinfo, sizeof(int)));
  HIPCHK(c, hipMemsetAsync(dinfo, 0, sizeof(int), s));
  if (n == GPMI_NB && std::getenv("GPMI_DIAG_STAMPS")) {
    // tools only: phase cycle counts of one potrf_diag launch
    // tools only: phase cycle counts and the per-wave timeline of one potrf_diag launch (potrf.hip: DIAG_TRACE_*)
    constexpr int TRACE_WORDS = 8 * 9 * 6;
    unsigned long long* dbg = nullptr;
    HIPCHK(c, hipMalloc(&dbg, (GPMI_STAMP_WORDS + TRACE_WORDS) * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(dbg, 0, (GPMI_STAMP_WORDS + TRACE_WORDS) * sizeof(unsigned long long), s));
    const unsigned long long magic = 0x7ACEull;
    HIPCHK(c, hipMemcpyAsync(dbg + 22, &magic, sizeof(magic), hipMemcpyHostToDevice, s));
    launch_potrf_diag(s, A, ld, invD, dinfo, 0, dbg);
    std::vector<unsigned long long> hw(GPMI_STAMP_WORDS + TRACE_WORDS);
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipMemcpy(hw.data(), dbg, hw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long* h = hw.data() + 16;
    std::fprintf(stderr, "[potrf_diag] %.2f us | wave 0 cycles: load %llu | waits for its two tiles %llu | sub-diagonal panel + trailing tile %llu | end %llu | factor16 x8 %llu\n",
                 (double)(hw[8] - hw[0]) * 0.01, h[0], h[1], h[2], h[3], h[4]);
    for (int w = 0; w < 8; ++w) {
      if (w == 4) continue;
      std::fprintf(stderr, "[potrf_diag] wave %d (%s):", w, w == 0 ? "chain: start, W published, tiles seen, next block ready" : w < 4 ? "factor: start, W seen, panel done, panels seen, updates done" : "inverse: start, W seen, row done, operands seen, sums done");
      for (int k = 0; k < 9; ++k) {
        std::fprintf(stderr, " |");
        for (int e = 0; e < 6; ++e) {
          const unsigned long long v = hw[GPMI_STAMP_WORDS + (w * 9 + k) * 6 + e];
          if (v) std::fprintf(stderr, " %llu", v);
        }
      }
      std::fprintf(stderr, "\n");
    }
    (void)hipFree(dbg);
  } else
  potrf_lower(c, c->lanes[0], A, n, ld, invD, dinfo);
  int h = 0;
  hipError_t e = hipMemcpyAsync(&h, dinfo, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(invD);
  (void)hipFree(dinfo);
  HIPCHK(c, e);
  if (info) *info = h;
  return GPMI_OK;
}

int gpmi_dev_gemm_nt(gpmi_ctx* c, double* C, int64_t ldc, const double* A, int64_t lda,
                     const double* B, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, m % GPMI_NB == 0 && n % GPMI_NB == 0 && k % 16 == 0 && m > 0 && n > 0 && k > 0,
         "m, n must be multiples of 128 and k of 16");
  if (int rc = set_device(c)) return rc;
  hipStream_t s;
  if (int rc = dev_stream(c, &s)) return rc;
  {
    const double tiles = lower ? (double)(n / GPMI_NB) * (n / GPMI_NB + 1) / 2.0 +
                                     (double)(m / GPMI_NB - n / GPMI_NB) * (n / GPMI_NB)
                               : (double)(m / GPMI_NB) * (n / GPMI_NB);
    unsigned long long* stamp =
        prof_stamp_slot(c, tiles * 2.0 * GPMI_NB * GPMI_NB * k, tiles * 16.0 * GPMI_NB * GPMI_NB);
    launch_gemm_nt(s, lower ? TILES_LOWER : TILES_RECT, OP_SUB, C, ldc, A, lda, B, ldb,
                   (int)(m / GPMI_NB), (int)(n / GPMI_NB), (int)k, stamp);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(s));
  return GPMI_OK;
}

}  // extern "C"
