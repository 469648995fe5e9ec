// Blocked right-looking Cholesky factorisation (lower, in place, row-major) for gfx950.
//
// Replaces numpy.linalg.cholesky at regression.py:241 (fit), :537 (marginal_likelihood) and :555
// (marginal_likelihood_gradient).  Two-level blocking:
//   outer panels of OB columns  -> one trailing SYRK update with K = OB on the MFMA GEMM (the
//                                   compute-bound kernel: intensity OB/8 FLOP per byte of C traffic)
//   inner blocks of 128 columns -> potrf_diag (one workgroup: factor the 128 x 128 diagonal block
//                                   and invert it), panel TRSM as a product with the inverse
//                                   (MFMA), and the update of the rest of the outer panel (K = 128).
// A non-positive or non-finite pivot is reported LAPACK-style through `info` (first failing
// column + 1); the factorisation then continues with a unit pivot so that the launch sequence stays
// asynchronous — the host inspects `info` once at the end (regression.py:540-542 behaviour).
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "gpmi_internal.h"
#include "potrf_diag.h"

namespace {

constexpr int NB = GPMI_NB;
using potrf_diag::DIAG_THREADS;

// One workgroup (8 waves) per problem: L = chol(A_blk) in place (lower part of A), invD = L^-1 (dense 128 x 128, zero
// above the diagonal): potrf_diag.h.  blockIdx.z = problem of a lockstep batch.
__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(double* __restrict__ A, int64_t ld,
                                                         double* __restrict__ invD,
                                                         int* __restrict__ info, int col0,
                                                         unsigned long long* __restrict__ dbg,
                                                         int64_t strideA, int64_t strideInv, int* pub, int pub_val) {
  // (flag-ordered tail, potrf_flow.hip: this launch publishes what the chain launch before it produced)
  if (pub && threadIdx.x == 0 && blockIdx.z == 0)
    __hip_atomic_store(pub, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __shared__ potrf_diag::DiagShared sh;
  potrf_diag::potrf_diag_body<false>(A + (int64_t)blockIdx.z * strideA, ld, invD + (int64_t)blockIdx.z * strideInv,
                                     info + blockIdx.z, col0, dbg, sh, potrf_diag::DiagPub());
}

}  // namespace

namespace {
// test hook (tests/test_gpu_parity.py::test_suite_detects_a_1e11_fault): scales the factored diagonal block
__global__ void fault_scale_kernel(double* A, int64_t ld, double f, int64_t strideA) {
  A += (int64_t)blockIdx.z * strideA;
  const int r = blockIdx.x, c = threadIdx.x;
  if (c <= r) A[(int64_t)r * ld + c] *= f;
}
}  // namespace

// GPMI_FAULT_DIAG_EPS=<eps>: fault injection for the test suite's own sensitivity check, never set otherwise: scales the
// factored diagonal block (enqueued behind the launch that factored it)
void launch_potrf_diag_fault(hipStream_t s, double* Ablk, int64_t ld, const BatchShape& bs) {
  static const double fault = [] {
    const char* e = std::getenv("GPMI_FAULT_DIAG_EPS");
    return e ? std::atof(e) : 0.0;
  }();
  if (fault != 0.0)
    hipLaunchKernelGGL(fault_scale_kernel, dim3(NB, 1, (unsigned)bs.count), dim3(NB), 0, s, Ablk, ld, 1.0 + fault,
                       bs.sMat);
}

void launch_potrf_diag(hipStream_t s, double* Ablk, int64_t ld, double* invD, int* info, int col0,
                       unsigned long long* dbg, const BatchShape& bs, int* pub, int pub_val) {
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, (unsigned)bs.count), dim3(DIAG_THREADS), 0, s, Ablk, ld, invD,
                     info, col0, dbg, bs.sMat, bs.sInv, pub, pub_val);
  launch_potrf_diag_fault(s, Ablk, ld, bs);
}

namespace {

// factor outer panel [J, Je): inner right-looking steps on stream sp
void factor_panel(gpmi_ctx* c, hipStream_t sp, double* A, int64_t ld, double* invD, int* info, int nt,
                  int J, int Je, int ncu = 0) {
  GemmBatch on;  // one problem; the CU count of a masked panel stream steers the tile height of the TRSM
  on.ncu_hint = ncu;
  on.b_lower_tri = true;  // the panel TRSM's B is the inverse of the diagonal block
  // GPMI_CHAIN_TRACE=1 (tools/chain_trace.py, with the profile enabled): in-kernel wall-clock stamps of every launch
  // of the chain, printed by the next profile read
  static const bool trace = std::getenv("GPMI_CHAIN_TRACE") != nullptr;
  auto slot = [&](int tag) { return trace ? prof_stamp_slot(c, 0.0, 0.0, GPMI_PROF_NCLASS + tag) : nullptr; };
  for (int j = J; j < Je; ++j) {
    double* Ajj = A + (int64_t)j * NB * ld + (int64_t)j * NB;
    double* invDj = invD + (int64_t)j * NB * NB;
    const int below = nt - j - 1;
    {
      ProfScope ps(c, sp, GPMI_PROF_PANEL, (double)NB * NB * NB / 3.0 + 2.0 * below * NB * NB * NB,
                   8.0 * NB * NB * (2.0 + 2.0 * below));
      launch_potrf_diag(sp, Ajj, ld, invDj, info, j * NB, slot(0));
      if (below > 0) {
        // panel TRSM: A21 <- A21 * L11^-T  (in place: one tile column, see gemm_f64.hip)
        double* A21 = Ajj + (int64_t)NB * ld;
        launch_gemm_nt(sp, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDj, NB, below, 1, NB, slot(1), on);
      }
    }
    const int pc = Je - j - 1;  // remaining block columns of the outer panel
    if (below > 0 && pc > 0) {
      // inner update of the rest of the outer panel, rows below: tiles (ti >= tj, tj < pc)
      double* A21 = Ajj + (int64_t)NB * ld;
      double* C = A21 + NB;
      const double tiles = pc * (pc + 1) / 2.0 + (double)(below - pc) * pc;
      ProfScope ps(c, sp, GPMI_PROF_PANEL, tiles * 2.0 * NB * NB * NB, tiles * 16.0 * NB * NB);
      launch_gemm_nt(sp, TILES_LOWER, OP_SUB, C, ld, A21, ld, A21, ld, below, pc, NB, slot(2));
    }
  }
}

// Update of the tile columns [t0, t1) (rows t0 .. nt, lower tiles) by the factored tile columns [ka, ke):
// A[t0.., t0..t1) -= P P^T with P = A[t0.., ka..ke), K = (ke - ka) * 128.
// `slice` > 0: the last `slice` tiles of the logical tile list are left out (update_slice applies them on another
// stream)
void update_columns(gpmi_ctx* c, hipStream_t s, double* A, int64_t ld, int nt, int t0, int t1, int ka, int ke,
                    int ncu, int64_t slice = 0) {
  const int rows = nt - t0, cols = t1 - t0;
  const int kw = (ke - ka) * NB;
  if (rows <= 0 || cols <= 0 || kw <= 0) return;
  double* P = A + (int64_t)t0 * NB * ld + (int64_t)ka * NB;
  double* C = A + (int64_t)t0 * NB * ld + (int64_t)t0 * NB;
  const int64_t tiles = (int64_t)cols * (cols + 1) / 2 + (int64_t)(rows - cols) * cols - slice;
  // the tiles of a nearly empty last round (of `ncu` tiles) run as 64 x 64 tiles in a second launch
  const int64_t nfull = gemm_split_point(tiles, ncu, kw);
  // per-launch timing (bench roofline), every launch: the 128 x 128-tile kernel (launches with >= 384 tiles, their
  // full rounds) in class SYRK - the dominant kernel -, the 64 x 64-tile remainders and the launches with fewer
  // tiles (a different kernel in rocprof's tables) in class SYRK_REST
  static const int64_t BIG_MIN = [] {
    const char* e = std::getenv("GPMI_BIG_MIN");
    return (int64_t)(e ? std::atoi(e) : 384);
  }();
  const bool big = tiles >= BIG_MIN && kw > 128;
  int64_t nmain = big ? nfull : 0;
  if (big && nfull > 0 && nfull < tiles && gemm_mixed_launches()) nmain = tiles;  // ONE launch: full tiles + the rest in quarters
  unsigned long long* stamp = nullptr;
  unsigned long long* stamp_rest = nullptr;
  if (nmain > 0)
    stamp = prof_stamp_slot(c, (double)nmain * 2.0 * NB * NB * kw, (double)nmain * 16.0 * NB * NB + 8.0 * rows * NB * kw);
  if (tiles > nmain)
    stamp_rest = prof_stamp_slot(c, (double)(tiles - nmain) * 2.0 * NB * NB * kw, (double)(tiles - nmain) * 16.0 * NB * NB,
                                 GPMI_PROF_SYRK_REST);
  if (big)
    launch_gemm_nt_split(s, TILES_LOWER, OP_SUB, C, ld, P, ld, P, ld, rows, cols, kw, nfull, stamp, stamp_rest, tiles);
  else
    launch_gemm_nt_split(s, TILES_LOWER, OP_SUB, C, ld, P, ld, P, ld, rows, cols, kw, nfull, stamp_rest, nullptr, tiles);
}

// the last `slice` tiles of the same update (full 128 x 128 tiles, the dominant kernel; class SYRK_SLICE)
void update_slice(gpmi_ctx* c, hipStream_t s, double* A, int64_t ld, int nt, int t0, int t1, int ka, int ke,
                  int64_t slice) {
  const int rows = nt - t0, cols = t1 - t0;
  const int kw = (ke - ka) * NB;
  double* P = A + (int64_t)t0 * NB * ld + (int64_t)ka * NB;
  double* C = A + (int64_t)t0 * NB * ld + (int64_t)t0 * NB;
  const int64_t tiles = (int64_t)cols * (cols + 1) / 2 + (int64_t)(rows - cols) * cols;
  unsigned long long* stamp =
      prof_stamp_slot(c, (double)slice * 2.0 * NB * NB * kw, (double)slice * 16.0 * NB * NB + 8.0 * rows * NB * kw,
                      GPMI_PROF_SYRK_SLICE);
  launch_gemm_nt_range(s, TILES_LOWER, OP_SUB, C, ld, P, ld, P, ld, rows, cols, kw, tiles - slice, slice, stamp);
}

// How many tiles of a trailing update the panel stream takes over.  The panel chain of the next panel leaves the
// reserved CUs idle for the rest of the update (half of it while the trailing matrix is large); a slice of the
// update's last tiles (the far-right column strips, which neither the next look-ahead update nor the next panel
// touch) runs there behind the chain.  Cost model in microseconds, fitted to the measured timeline
// (profiles/r02_bench_timeline.txt): 64.4 us per tile and CU pair slot at K = 512 (0.29 us per tile on 224 CUs, 2.0 on
// 32), the chain 146 + 6.1 us per trailing tile row (re-fitted in round 4; 200 + 7.6 before); the slice takes GPMI_SLICE_PCT % (default 100; 0: none) of the
// balance point, rounded down to whole rounds of the reserved CUs.  Measured: 36.4 -> 35.9 ms per step at 100 and 130 %,
// 37.0 at 160 % (the next update then waits for the slice); the 32 extra CUs lower the clock of the other 224 from
// 2.364 to 2.352 GHz, and the chain's own launches wait behind them: the gain is a third of the idle time.
int64_t slice_tiles(int rem, int64_t tiles_la, int64_t tiles_main, int kw, int ncu_main, int ncu_panel) {
  static const int PCT = [] {
    const char* e = std::getenv("GPMI_SLICE_PCT");
    return e ? std::atoi(e) : 100;
  }();
  if (PCT <= 0 || kw != 4 * NB) return 0;
  const double tile_us = 64.4 * 2.0;  // one CU works on two tiles at a time
  const double per_main = tile_us / (2.0 * ncu_main), per_panel = tile_us / (2.0 * ncu_panel);
  // The panel chain of one outer panel on the reserved CUs, re-fitted to round 4's timeline (profiles/r04_bench_timeline.txt:
  // 880 us at 120 trailing tile rows = 4 x potrf_diag 20 + 4 x panel TRSM 0.56 us per row + inner updates 3.8 us per row +
  // launch gaps): 146 + 6.1 us per row (200 + 7.6 until round 3).  The slices grow by 30 % with it: 32.76 -> 32.53 ms per
  // step (medians of six alternating runs, round-3 A/B script, profiles/HISTORY.md); 45 % more is too much (32.86: the next update waits
  // for the slice).  GPMI_SLICE_CHAIN_US / GPMI_SLICE_CHAIN_SLOPE for A/B runs.
  static const double CHAIN0 = [] {
    const char* e = std::getenv("GPMI_SLICE_CHAIN_US");
    return e ? std::atof(e) : 146.0;
  }();
  static const double SLOPE = [] {
    const char* e = std::getenv("GPMI_SLICE_CHAIN_SLOPE");
    return e ? std::atof(e) : 6.1;
  }();
  const double chain = CHAIN0 + SLOPE * rem;
  const double su = 20.0 + (double)(tiles_la + tiles_main) * per_main;
  double x = (su - chain) / (per_main + per_panel) * PCT / 100.0;
  const int64_t round = 2 * ncu_panel;
  int64_t n = x > 0 ? (int64_t)(x / round) * round : 0;
  // never into the first column strip (8 tile columns): the next look-ahead update and panel work there
  const int64_t safe = tiles_main - (int64_t)8 * rem;
  if (n > safe) n = safe > 0 ? safe / round * round : 0;
  return n;
}

}  // namespace

bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k);  // api.hip

namespace {
constexpr int OUTER_TILES = 4;  // tile columns per outer panel
//   GPMI_LOOKAHEAD_MIN=<tile rows>  end of the look-ahead regime (0 disables it)
int lookahead_min() {
  static const int v = [] {
    const char* e = std::getenv("GPMI_LOOKAHEAD_MIN");
    const int x = e ? std::atoi(e) : 52;  // (60 until the chain step of the flag-ordered tail went from 57 to 33 us)
    return x > 0 ? x : (1 << 30);
  }();
  return v;
}
}  // namespace

// Lanes 0 and 1 share ONE CU-masked pair (api.hip: lane_alloc).  That is safe while every entry point joins the pair's
// work back into its lane and synchronises before it returns - which an early error return behind enqueued look-ahead
// work, or a caller driving lane 0 through gpmi_get_stream, does not do.  So nothing is assumed: a factorisation that
// finds the pair busy waits for it first (two stream queries per factorisation; never taken in a correct call sequence).
// Called BEFORE the caller enqueues anything of its own on the pair (the fit builds covariance tiles on the update
// stream beside the first panel: checked behind that build, the query would find it and wait for it - 0.13 ms per fit at
// N = 16384, measured); potrf_lower runs the check itself for callers that did not.
void potrf_pair_quiesce(Lane& lane) {
  for (int k = 0; k < GPMI_NPAIRS; ++k)
    for (hipStream_t q : {lane.sp[k], lane.su[k]})
      if (q && hipStreamQuery(q) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(q);
      }
  lane.pair_checked = true;
}

// The stream the first trailing update of potrf_lower will run on when it is not the lane's own (the look-ahead
// regime applies from the first panel on), else nullptr.  Work enqueued there beforehand precedes that update: the
// fit builds the covariance tiles to the right of the first panel on it while the first panel is being factored.
hipStream_t potrf_first_update_stream(gpmi_ctx* c, Lane& lane, int64_t np, bool allow_lookahead) {
  const int nt = (int)(np / NB);
  if (!allow_lookahead || nt - OUTER_TILES < lookahead_min() || !ensure_masked_pair(c, lane, 0)) return nullptr;
  return lane.su[0];
}

void potrf_lower(gpmi_ctx* c, Lane& lane, double* A, int64_t np, int64_t ld, double* invD, int* info,
                 bool allow_lookahead) {
  // Right-looking over outer panels of 512 columns (4 tile columns), software-pipelined: step p applies
  // the trailing update of panel p and factors panel p + 1.  While the trailing matrix is large the two
  // overlap on a pair of CU-masked streams (disjoint CUs: a 75 KiB potrf_diag workgroup never finds a slot
  // on a chip saturated by GEMM workgroups): the update stream first updates the columns of panel p + 1
  // ("la"), the panel stream then factors them on 32 CUs while the update stream applies the rest on the
  // other 224.  Below GPMI_LOOKAHEAD_MIN (52) trailing tile rows the update is shorter than the panel chain on its 32 CUs
  // and everything runs in order on the full-chip stream - and so does the very first panel (nothing to overlap
  // it with: 0.49 instead of 0.92 ms).
  // Measured and dropped: fewer CUs for the panel chain (sustained headline step with
  // GPMI_PANEL_CUS = 32 / 24 / 16 / 8: 32.2 / 34.2 / 38.6 / 55.4 ms: the chain becomes the longer of the two; the "equal
  // FLOP/s on 224, 240 and 256 CUs" of rounds 2-4 was the shader clock's ramp in short measurements, not a power
  // limit); trailing updates applied lazily with K = 1024 .. 2048 (in place the launches are already
  // split at round boundaries, which leaves +1 % for the larger K, and the narrower launches cost more).
  hipStream_t sf = lane.stream;
  const int nt = (int)(np / NB);
  const int OBT = OUTER_TILES;
  const int LOOKAHEAD_MIN = lookahead_min();
  if (!lane.pair_checked) potrf_pair_quiesce(lane);
  lane.pair_checked = false;
  const bool la_ok = allow_lookahead && nt - OBT >= LOOKAHEAD_MIN && ensure_masked_pair(c, lane, 0);
  // The chain-bound part - everything below LOOKAHEAD_MIN trailing tile rows, i.e. the whole of a matrix of N <= ~8000
  // and the tail of a larger one - runs as a flag-ordered tile-task launch beside the bare chain (potrf_flow.hip)
  // instead of in stream order: chain and updates then overlap instead of adding up.  Same tile bodies, same order of
  // summation: the factor is bit-identical either way (GPMI_FLOW=0 keeps the stream-ordered schedule).
  const bool flow_ok = allow_lookahead;
  if (flow_ok && !(la_ok && nt - OBT >= LOOKAHEAD_MIN) && potrf_flow_enabled(c, lane, nt) &&
      potrf_flow_tail(c, lane, A, ld, invD, info, nt, 0))
    return;
  auto follow = [](hipStream_t waiter, hipStream_t producer, hipEvent_t ev) {
    if (waiter != producer) (void)hipStreamWaitEvent(waiter, ev, 0);
  };
  auto tile0 = [&](int panel) { return panel * OBT < nt ? panel * OBT : nt; };  // first tile column of a panel
  const int NP = (nt + OBT - 1) / OBT;
  factor_panel(c, sf, A, ld, invD, info, nt, 0, tile0(1));
  hipStream_t panel_stream = sf;  // where the latest panel was factored (ev_panel recorded behind it)
  hipStream_t main_stream = sf;   // where the latest trailing update ran (ev_main recorded behind it)
  bool sliced = false;            // a slice of the latest update is in flight on the panel stream (ev_slice)
  (void)hipEventRecord(lane.ev_panel, sf);
  (void)hipEventRecord(lane.ev_main, sf);
  // from here to the join the lane's own stream idles when the masked pair takes over: the alpha phase's factor-independent
  // launches go there (gpmi_internal.h: Lane::EarlyWork)
  if (la_ok && NP > 1 && nt - tile0(1) >= LOOKAHEAD_MIN) lane_run_early(lane, sf);
  for (int p = 0; p + 1 < NP; ++p) {
    const int k0 = tile0(p), k1 = tile0(p + 1);  // tile columns of the panel being applied
    const int rem = nt - k1;                      // trailing tile rows
    const bool overlap = la_ok && rem >= LOOKAHEAD_MIN;
    hipStream_t su = overlap ? lane.su[0] : sf, sp = overlap ? lane.sp[0] : sf;
    const int ncu = overlap ? c->ncu - c->pair_cus[0] : c->ncu;
    follow(su, panel_stream, lane.ev_panel);
    follow(su, main_stream, lane.ev_main);
    int64_t slice = 0;
    if (overlap) {
      const int k2 = tile0(p + 2);
      update_columns(c, su, A, ld, nt, k1, k2, k0, k1, ncu);  // the columns of the next panel first
      (void)hipEventRecord(lane.ev_la, su);
      (void)hipStreamWaitEvent(sp, lane.ev_la, 0);
      const int w = k2 - k1, r2 = nt - k2;
      slice = slice_tiles(rem, (int64_t)w * (w + 1) / 2 + (int64_t)(rem - w) * w, (int64_t)r2 * (r2 + 1) / 2,
                          (k1 - k0) * NB, ncu, c->pair_cus[0]);
      // the previous slice wrote tiles of this update's region
      if (sliced) (void)hipStreamWaitEvent(su, lane.ev_slice, 0);
      update_columns(c, su, A, ld, nt, k2, nt, k0, k1, ncu, slice);
      (void)hipEventRecord(lane.ev_main, su);
    } else {
      if (sliced) (void)hipStreamWaitEvent(su, lane.ev_slice, 0);
      update_columns(c, su, A, ld, nt, k1, nt, k0, k1, ncu);
      // from here on the chain is the longer of the two: the rest as a flag-ordered tile-task launch beside the chain
      if (flow_ok && potrf_flow_enabled(c, lane, nt - k1) && potrf_flow_tail(c, lane, A, ld, invD, info, nt, k1)) return;
    }
    sliced = false;
    main_stream = su;
    factor_panel(c, sp, A, ld, invD, info, nt, k1, tile0(p + 2), overlap ? c->pair_cus[0] : 0);
    if (overlap) (void)hipEventRecord(lane.ev_panel, sp);
    panel_stream = sp;
    if (slice > 0) {
      // behind the panel chain on the reserved CUs; ordered after the previous update by ev_la (recorded behind it)
      update_slice(c, sp, A, ld, nt, tile0(p + 2), nt, k0, k1, slice);
      (void)hipEventRecord(lane.ev_slice, sp);
      sliced = true;
    }
  }
  follow(sf, panel_stream, lane.ev_panel);
  follow(sf, main_stream, lane.ev_main);
  if (sliced) (void)hipStreamWaitEvent(sf, lane.ev_slice, 0);
}

void potrf_lower_batched(gpmi_ctx* c, hipStream_t s, double* A, int64_t np, int64_t ld, double* invD,
                         int* info, const BatchShape& bs) {
  // Many small factorisations advance in lockstep: every launch carries all of them in blockIdx.z, so
  // a step that is latency-bound for one matrix fills the chip across the batch.  Two-level like the large-N driver:
  // inside an outer panel of GPMI_BATCH_OUTER (4) tile columns the K = 128 steps update the panel's own columns only;
  // the trailing matrix is touched once per outer panel, with K = 512.  (A batch of 32 matrices at N = 2048 is 1 GiB -
  // beyond L2 and Infinity Cache - so a right-looking update of the whole trailing matrix per 128 columns streamed
  // 320 MB per matrix through HBM at 16 FLOP per byte; per 512 columns it is 64 FLOP per byte and a fifth of the traffic.)
  (void)c;
  static const int OBT = [] {
    const char* e = std::getenv("GPMI_BATCH_OUTER");
    const int v = e ? std::atoi(e) : 4;
    return v > 0 ? v : 1;
  }();
  const int nt = (int)(np / NB);
  GemmBatch inplace{bs.count, bs.sMat, bs.sMat, bs.sInv};
  inplace.b_lower_tri = true;
  const GemmBatch upd{bs.count, bs.sMat, bs.sMat, bs.sMat};
  // Inside an outer panel the columns are brought up to date LEFT-looking (round 5): before column j is factored it
  // takes the panel's earlier columns in ONE launch with K = 128 (j - J), instead of every column pushing a K = 128 update
  // into the rest of the panel right behind its TRSM.  The batch is far beyond the caches, so these steps stream the
  // panel's tiles through HBM (~3 TB/s, tools/cfg5_trace.sh): the left-looking form touches a tile of the panel once per
  // column it belongs to, not once per column before it, at two to three times the K per pass.  Every element still takes
  // the panel's columns in ascending order on the ring kernels' accumulators: the same bits.  GPMI_BATCH_LEFT=0: the
  // right-looking form (A/B).
  static const bool left = [] {
    const char* e = std::getenv("GPMI_BATCH_LEFT");
    return !e || std::atoi(e) != 0;
  }();
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    for (int j = J; j < Je; ++j) {
      double* Ajj = A + (int64_t)j * NB * ld + (int64_t)j * NB;
      double* invDj = invD + (int64_t)j * NB * NB;
      const int below = nt - j - 1;
      if (left && j > J) {
        // tiles (i, j), i >= j: -= A(i, J..j-1) A(j, J..j-1)^T
        const double* P = A + (int64_t)j * NB * ld + (int64_t)J * NB;
        launch_gemm_nt(s, TILES_LOWER, OP_SUB, Ajj, ld, P, ld, P, ld, nt - j, 1, (j - J) * NB, nullptr, upd);
      }
      launch_potrf_diag(s, Ajj, ld, invDj, info, j * NB, nullptr, bs);
      if (below > 0) {
        double* A21 = Ajj + (int64_t)NB * ld;
        launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDj, NB, below, 1, NB, nullptr, inplace);
        const int pc = Je - j - 1;  // remaining tile columns of the outer panel
        if (!left && pc > 0)
          launch_gemm_nt(s, TILES_LOWER, OP_SUB, A21 + NB, ld, A21, ld, A21, ld, below, pc, NB, nullptr, upd);
      }
    }
    const int rest = nt - Je;
    if (rest > 0) {
      double* P = A + (int64_t)Je * NB * ld + (int64_t)J * NB;
      double* C = A + (int64_t)Je * NB * ld + (int64_t)Je * NB;
      launch_gemm_nt(s, TILES_LOWER, OP_SUB, C, ld, P, ld, P, ld, rest, rest, (Je - J) * NB, nullptr, upd);
    }
  }
}
