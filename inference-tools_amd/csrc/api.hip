// C-ABI of libgpmi (see include/gpmi.h): handle management, host<->device plumbing and the
// orchestration of the GP hot path (covariance build -> Cholesky -> solves -> reductions).
//
// Layout of this file:
//   helpers (error macros, lanes / streams, workspaces, K-build + factorise + forward solve)
//   lifecycle, data upload, fit / LML / batched LML / LML gradient
//   predict, posterior, spatial gradients, leave-one-out, covariance downloads
//   instrumentation (per-class events, in-kernel stamps), device-pointer entry points (tools)
//   mixture covariance (ChangePoint), per-point noise (HeteroscedasticNoise), linear inversion (GpLinearInverter)
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gpmi_internal.h"

namespace {

thread_local std::string g_create_err;

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
      return (e__ == hipErrorOutOfMemory) ? GPMI_ERR_NOMEM : GPMI_ERR_HIP;                  \
    }                                                                                       \
  } while (0)

#define ARGCHK(ctx, cond, msg) \
  do {                         \
    if (!(cond)) {             \
      (ctx)->err = (msg);      \
      return GPMI_ERR_ARG;     \
    }                          \
  } while (0)

// a negative `info` is written by the flag-ordered kernels when a poll timed out: GPMI_INFO_FLOW_TIMEOUT by the tile-task
// factorisation (potrf_flow.hip) - the one case the stream-ordered schedule (GPMI_OPT_NO_FLOW) cures, marked "[flow-tail]"
// in the error text for the caller that wants to repeat the call -, GPMI_ERR_INTERNAL by the triangular sweeps
#define INFOCHK(ctx, inf)                                                                                     \
  do {                                                                                                        \
    if ((inf) < 0) {                                                                                          \
      (ctx)->err = (inf) == GPMI_INFO_FLOW_TIMEOUT                                                            \
                       ? "internal error: the tile-task factorisation timed out [flow-tail]"                  \
                       : "internal error: a flag-ordered triangular sweep timed out";                         \
      return GPMI_ERR_INTERNAL;                                                                               \
    }                                                                                                         \
  } while (0)

constexpr int RED_SLOTS = 8192;  // per-lane result slots for batched evaluations

// A handle has TWO full-chip streams, those of its lanes 0 and 1; lane i >= 2 runs on the stream of lane i mod 2 (its
// buffers are its own).  Why: the HIP runtime multiplexes a process's streams onto 4 hardware queues (GPU_MAX_HW_QUEUES),
// and two streams on one queue run in order.  With a stream per lane a second handle alive in the process (a regressor
// beside the one being evaluated: 2 + 3 streams) put the two evaluation lanes of a likelihood sweep on one queue: config
// 3 lost 14 % (30.2 against 26.5 ms per evaluation at N = 16384).  Every pair the library runs side by side is a pair
// of neighbouring lanes (the half-batches of a lockstep chunk and the asynchronous slots: lanes 1 | 2; a sweep on two
// lanes: 1 | 2), lane 0's stream is idle while other lanes evaluate, and more than two evaluations at a time were never
// faster than two (§5).  Measured alternatives (tools/scratch/ab_hwq.sh, ab_pool*.sh): GPU_MAX_HW_QUEUES=8 cures the
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

}  // namespace

// CU-masked stream pair of the look-ahead: factor the next panel on pair_cus CUs (sp) while the trailing update
// runs on all the others (su).  Mask bits are dealt round-robin over the 8 XCDs (probed with tools/cumask_probe.hip),
// so the first 8 m bits are m CUs on every XCD.  The pair is created together with the lane's main stream, before any
// work is queued, and destroyed with it: created later (on first use, with kernels already in flight on the lane)
// hipStreamDestroy blocked forever on ROCm 7.2, and a pair that is never destroyed ends the process in a SIGSEGV
// inside the runtime's static destructors when rocprofv3 is attached (tools/scratch/probe_exit.py).  Only the lanes that
// can factorise with look-ahead get one: lane 0 (the fitted model) and lane 1 (single evaluations).
bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k) {
  (void)c;
  return L.sp[k] != nullptr && L.su[k] != nullptr;
}

namespace {

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
    if (hipExtStreamCreateWithCUMask(&L.sp[k], (uint32_t)panel.size(), panel.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&L.su[k], (uint32_t)upd.size(), upd.data()) != hipSuccess) {
      if (L.sp[k]) (void)hipStreamDestroy(L.sp[k]);
      if (L.su[k]) (void)hipStreamDestroy(L.su[k]);
      L.sp[k] = L.su[k] = nullptr;  // no look-ahead: everything on the full-chip stream
      (void)hipGetLastError();
    }
  }
  return GPMI_OK;
}

}  // namespace

namespace {

int lane_alloc(gpmi_ctx* c, Lane& L) {
  if (int rc = lane_streams(c, L)) return rc;
  // lanes[0] or lanes[1] (the lane has just been appended) of a problem large enough to ever enter the look-ahead
  // regime: destroying a CU-masked stream takes the runtime about a second, which small models (a GpOptimiser
  // builds a new regressor per added evaluation) should not pay
  if (c->lanes.size() <= 2 && c->np >= 40 * GPMI_NB)
    if (int rc = lane_masked_streams(c, L)) return rc;
  const int64_t nt = c->np / GPMI_NB;
  HIPCHK(c, hipMalloc(&L.A, sizeof(double) * c->np * c->ld));
  HIPCHK(c, hipMalloc(&L.invD, sizeof(double) * nt * GPMI_NB * GPMI_NB));
  // potrf_diag writes the block-lower part of an inverse only: the zeros above stay from here (round 4: zeroing them
  // in the kernel cost it 57 KB of stores per launch through one CU's memory path, as much as loading the block)
  HIPCHK(c, hipMemset(L.invD, 0, sizeof(double) * nt * GPMI_NB * GPMI_NB));
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
  // synchronisation too closely (tools/scratch/probe_exit.py: 1 hang in 6 closes without the pause, 0 in 36 with 5 .. 300 ms; the round-1
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
  if (c->h_bGout) (void)hipHostFree(c->h_bGout);
  c->h_bGout = nullptr;
  c->bgrad_cap = c->bgrad_ntheta = c->bLoo_cap = c->bNoise_cap = 0;
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

// one evaluation of the mixture covariance K = sum_m diag(g_m) K_m diag(g_m) + extra I (+ data errors)
struct MixEval {
  int nk;
  const KParams* p;   // nk sub-kernels (their extra_diag is ignored)
  const double* g;    // nk x np device weights (row m: g_m; padding 1 for m = 0, else 0)
  double extra;       // WhiteNoise variance
  double* scratch;    // np x ld
  const double* zero; // np zeros
};

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
                               double mu_const, int slot, bool allow_lookahead = true,
                               const MixEval* mix = nullptr) {
  hipStream_t s = L.stream;
  L.inv2_valid = false;
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
  potrf_lower(c, L, L.A, c->np, c->ld, L.invD, L.info + slot, allow_lookahead);
  launch_residual(s, c->y, mu_dev, mu_const, L.vec + 2 * c->np, c->n, c->np);
  trsv_forward(c, s, L.A, c->np, c->ld, L.invD, L.vec + 2 * c->np, L.vec, L.info + slot);
  launch_lml_reduce(s, L.vec, L.A, c->ld, c->np, L.red + 2 * slot);
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
  launch_set_identity(L.stream, L.B2, c->ld, c->np);
  trsm_rows_forward(c, L.stream, F.A, c->np, c->ld, F.inv2, L.B2, c->np, true, nullptr, c->trsm_panel);
  return GPMI_OK;
}

// workspace for `want` small problems advancing in lockstep (capped by a 6 GiB budget)
int ensure_batch_ws(gpmi_ctx* c, int want) {
  const int64_t per = c->np * c->ld * (int64_t)sizeof(double);
  int cap = (int)((6LL << 30) / per);
  if (cap > 256) cap = 256;
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
  c->bLoo_cap = c->bNoise_cap = 0;
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
  HIPCHK(c, hipMemset(c->bInv, 0, sizeof(double) * want * nt * GPMI_NB * GPMI_NB));  // see lane_alloc
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

}  // namespace

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

__global__ void stamp_init_kernel(unsigned long long* pool, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots)
    for (int j = 0; j < GPMI_STAMP_WORDS; ++j) pool[(size_t)GPMI_STAMP_WORDS * i + j] = (j < 8) ? ~0ull : 0ull;
}
}  // namespace

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

int gpmi_create(int device, gpmi_ctx** out) {
  if (!out) return GPMI_ERR_ARG;
  *out = nullptr;
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
  // stream that has only been synchronised once blocked forever on ROCm 7.2 (tools/scratch/probe_exit.py)
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
  HIPCHK(c, hipMemset(c->x, 0, sizeof(double) * c->np * d));
  HIPCHK(c, hipMemset(c->y, 0, sizeof(double) * c->np));
  HIPCHK(c, hipMemset(c->noise, 0, sizeof(double) * c->np));
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
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0)) return rc;
  const auto h1 = std::chrono::steady_clock::now();
  // alpha = L^-T v
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, c->alpha, L.info);
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
    if (int rc = ensure_batch_ws(c, (int)(T < 256 ? (T < 2 ? 2 : T) : 256))) return rc;
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
  ARGCHK(c, T >= 1 && T <= 128, "T out of range (1 .. 128 per slot)");
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
    if (int rc = ensure_batch_ws(c, 256)) return rc;  // (a no-op once it is that large)
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
int ensure_gradient_lane(gpmi_ctx* c, int n_theta) {
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
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0)) return rc;
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
  for (int64_t t0 = 0; t0 < T; t0 += c->bgrad_cap) {
    const int B = (int)((T - t0 < c->bgrad_cap) ? T - t0 : c->bgrad_cap);
    BatchShape bs = shape0;
    bs.count = B;
    HIPCHK(c, hipMemcpyAsync(c->bParams, ps.data() + t0, sizeof(KParams) * B, hipMemcpyHostToDevice, s));
    if (mus)
      HIPCHK(c, hipMemcpyAsync(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n, hipMemcpyHostToDevice, s));
    else
      HIPCHK(c, hipMemcpyAsync(c->bMu, mu_const + t0, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    if (noise_batch)
      HIPCHK(c, hipMemcpy2DAsync(c->bNoise, sizeof(double) * c->np, noise_batch + t0 * c->n, sizeof(double) * c->n,
                                 sizeof(double) * c->n, B, hipMemcpyHostToDevice, s));
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
      hipLaunchKernelGGL(qdiag_batched_kernel, dim3((unsigned)((c->n + 255) / 256), 1, (unsigned)B), dim3(256), 0, s, c->bA,
                         c->ld, alpha_dev, c->bNoise, c->n, bs.sMat, bs.sVec, c->np);
      HIPCHK(c, hipMemcpy2DAsync(qdiag_out + t0 * c->n, sizeof(double) * c->n, c->bNoise, sizeof(double) * c->np,
                                 sizeof(double) * c->n, B, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bRed, c->bRed, sizeof(double) * 2 * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * W * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    if (alpha_out)
      HIPCHK(c, hipMemcpy2DAsync(alpha_out + t0 * c->n, sizeof(double) * c->n, alpha_dev, sizeof(double) * bs.sVec,
                                 sizeof(double) * c->n, B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
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
int gpmi_loo_grad_batch(gpmi_ctx* c, int kernel, int64_t T, const double* thetas, int n_theta, const double* extra,
                        const double* mus, const double* mu_const, double* alpha_out, double* ikdiag_out, double* p_out,
                        double* grad_theta, double* trace_q, int* info) {
  if (!c) return GPMI_ERR_ARG;
  ARGCHK(c, T >= 1 && T <= RED_SLOTS, "T out of range");
  ARGCHK(c, thetas && alpha_out && ikdiag_out && p_out && grad_theta, "NULL argument");
  ARGCHK(c, mus || mu_const, "one of mus / mu_const is required");
  if (int rc = set_device(c)) return rc;
  const bool lockstep = (T >= 2 || c->lockstep_always) && c->np <= 4096 && !c->ycov;
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
  // four more vectors per problem: diag(K^-1), c1, sqrt(c2), p = K^-1 c1 (regression.py:505-513)
  const int64_t sLoo = 4 * c->np;
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
  for (int64_t t0 = 0; t0 < T; t0 += c->bgrad_cap) {
    const int B = (int)((T - t0 < c->bgrad_cap) ? T - t0 : c->bgrad_cap);
    BatchShape bs = shape0;
    bs.count = B;
    HIPCHK(c, hipMemcpyAsync(c->bParams, ps.data() + t0, sizeof(KParams) * B, hipMemcpyHostToDevice, s));
    if (mus)
      HIPCHK(c, hipMemcpyAsync(c->bMu, mus + t0 * c->n, sizeof(double) * B * c->n, hipMemcpyHostToDevice, s));
    else
      HIPCHK(c, hipMemcpyAsync(c->bMu, mu_const + t0, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->bInfo, 0, sizeof(int) * B, s));
    launch_kbuild_square_batched(s, ps[0].kernel, c->bParams, B, c->x, c->n, c->np, c->noise, c->bA, c->ld, bs.sMat,
                                 (int)c->d);
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
    launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, 0, c->bA, c->ld, c->bB2, c->ld, c->bB2, c->ld, nt, nt, (int)c->np,
                nullptr, syrk);
    launch_lml_grad_batched(s, c->bParams, B, n_theta, c->x, c->n, c->np, c->bA, c->ld, bs.sMat, p_dev, alpha_dev, sLoo,
                            c->bGws, c->bGout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_bGout, c->bGout, sizeof(double) * W * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_bInfo, c->bInfo, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpy2DAsync(alpha_out + t0 * c->n, sizeof(double) * c->n, alpha_dev, sizeof(double) * bs.sVec,
                               sizeof(double) * c->n, B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpy2DAsync(ikdiag_out + t0 * c->n, sizeof(double) * c->n, diag_dev, sizeof(double) * sLoo,
                               sizeof(double) * c->n, B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpy2DAsync(p_out + t0 * c->n, sizeof(double) * c->n, p_dev, sizeof(double) * sLoo,
                               sizeof(double) * c->n, B, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
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
    trsm_rows_backward(c, s, L.A, c->np, c->ld, L.invD, c->Q2, mp);
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
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0)) return rc;
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
  if (int rc = enqueue_factor_and_forward(c, L, p, mu_dev, 0.0, 0)) return rc;
  trsv_backward(c, s, L.A, c->np, c->ld, L.invD, L.vec, alpha_dev, L.info);
  // K^-1 (full, both triangles) in A
  if (int rc = enqueue_inverse_factor(c, L, L)) return rc;
  launch_rows_sumsq(s, L.B2, c->ld, c->np, c->np, 0.0, diag_dev);  // = -diag(K^-1)
  launch_gemm(s, TILES_LOWER, OP_ASSIGN, false, true, L.A, c->ld, L.B2, c->ld, L.B2, c->ld, nt, nt,
              (int)c->np);
  launch_mirror_lower(s, L.A, c->ld, c->np);
  // diag_dev holds -diag: flip sign inside the vector kernel by passing it through a scaled copy
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((c->np + 255) / 256)), dim3(256), 0, s,
                     diag_dev, c->np);
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
  HIPCHK(c, hipMemset(invD, 0, sizeof(double) * (n / GPMI_NB) * GPMI_NB * GPMI_NB));  // see lane_alloc
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
    HIPCHK(c, hipMemset(c->mix_zero, 0, sizeof(double) * c->np));
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
                      const double* g_host, double extra_diag, const double* mu, double* lml,
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
  double* hdev = c->mix_scratch;   // nk x np row sums: the scratch matrix is free once K is factorised
  const MixEval mx{nk, ps, c->mix_g, extra_diag, c->mix_scratch, c->mix_zero};
  HIPCHK(c, hipMemcpyAsync(mu_dev, mu, sizeof(double) * c->n, hipMemcpyHostToDevice, s));
  if (int rc = enqueue_factor_and_forward(c, L, ps[0], mu_dev, 0.0, 0, true, &mx)) return rc;
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
    launch_mix_rowsum(s, L.A, L.B2, c->ld, alpha_dev, gm, hdev + (int64_t)m * c->np, c->n);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(L.h_red, L.red, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_red + 16, gout, sizeof(double) * goff, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(L.h_info, L.info, sizeof(int), hipMemcpyDeviceToHost, s));
  for (int m = 0; m < nk; ++m)
    HIPCHK(c, hipMemcpyAsync(hrows + (int64_t)m * c->n, hdev + (int64_t)m * c->np, sizeof(double) * c->n,
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
namespace {
__global__ void qdiag_kernel(const double* __restrict__ iK, int64_t ld, const double* __restrict__ alpha,
                             double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha[i] * alpha[i] - iK[i * ld + i];
}
}  // namespace

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
  hipLaunchKernelGGL(qdiag_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, L.stream, L.A, c->ld,
                     L.vec + c->np, out, c->n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(qdiag, out, sizeof(double) * c->n, hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  return GPMI_OK;
}

}  // extern "C"

// ---- Gaussian-process linear inversion --------------------------------------------------------------
namespace {

__global__ void linv_add_diag_kernel(double* __restrict__ J, int64_t ld, const double* __restrict__ sig2,
                                     int64_t mp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mp) J[i * ld + i] += sig2[i];
}

void linv_free(LinvState& S) {
  for (double** p : {&S.A, &S.At, &S.y, &S.sig2, &S.zero, &S.K, &S.T, &S.J, &S.J2, &S.Q, &S.X, &S.invD, &S.inv2,
                     &S.inv2_t, &S.panel, &S.vec, &S.gws}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  S = LinvState();
}

int linv_alloc(gpmi_ctx* c, double** p, int64_t doubles) {
  if (*p) return GPMI_OK;
  HIPCHK(c, hipMalloc(p, sizeof(double) * doubles));
  HIPCHK(c, hipMemset(*p, 0, sizeof(double) * doubles));  // (the inverses' upper 16-blocks rely on it, see lane_alloc)
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
  HIPCHK(c, hipMemset(S.A, 0, sizeof(double) * mp * ld));
  HIPCHK(c, hipMemset(S.At, 0, sizeof(double) * np * ldm));
  HIPCHK(c, hipMemset(S.y, 0, sizeof(double) * mp));
  HIPCHK(c, hipMemset(S.zero, 0, sizeof(double) * np));
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
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((c->np + 255) / 256)), dim3(256), 0, s, diag_dev, c->np);
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
