// Internal declarations shared by the HIP translation units of libgpmi.
// Public C-ABI: include/gpmi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

#include "gpmi.h"

constexpr int GPMI_NB = 128;     // tile / inner block size: every device matrix dimension is a multiple
constexpr int GPMI_MAX_D = 64;   // max spatial dimensions handled by the covariance kernels
constexpr int GPMI_INFO_FLOW_TIMEOUT = -6;  // `info` of a factorisation whose tile-task kernel gave up polling (api.hip: INFOCHK)
constexpr int GPMI_NPAIRS = 1;   // CU-masked stream pairs of the look-ahead (32 | 224 CUs)

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Hyper-parameters of one covariance evaluation, passed by value as a kernel argument.
struct KParams {
  int kernel;                 // GPMI_KERNEL_SE / GPMI_KERNEL_RQ
  int d;                      // spatial dimensions
  double a2;                  // exp(theta0)^2
  double kappa;               // RQ: exp(theta1)
  double extra_diag;          // WhiteNoise variance added to the diagonal
  double inv_l2[GPMI_MAX_D];  // 1 / l_k^2
};

struct ProfSlot {
  hipEvent_t e0, e1;
  int klass;
  double flops, bytes;
};

// One independent evaluation lane: a stream with its own n x n scratch matrix and vectors.
struct Lane {
  hipStream_t stream = nullptr;    // full-chip stream: covariance build, solves, small factorisations
  bool owns_stream = true;         // false (lanes 2..): the stream is lane 0's or lane 1's (api.hip: lane_streams)
  // look-ahead pairs (CU-masked, disjoint; lanes 0 and 1 only, created with the lane): pair k factors the next panel on
  // gpmi_ctx::pair_cus[k] CUs (sp) while the trailing update runs on all the others (su)
  hipStream_t sp[GPMI_NPAIRS] = {nullptr};
  hipStream_t su[GPMI_NPAIRS] = {nullptr};
  bool owns_pair = true;           // false (lane 1): the pair is lane 0's (api.hip: lane_alloc)
  // Work of the alpha phase that does not depend on the factor (the residual y - mu, the sentinel fills of the two sweeps'
  // outputs), enqueued on the lane's stream by the factorisation at the point where that stream has handed the work to the
  // masked pair and would otherwise idle until the join (potrf_lower / potrf_flow_tail call lane_run_early; whoever set it
  // calls it again behind the factorisation, where it is a no-op unless no such point came up): ~15 us per fit off the
  // critical path (round 6)
  struct EarlyWork {
    bool pending = false;
    const double* y = nullptr;
    const double* mu = nullptr;
    double mu_const = 0.0;
    double* r = nullptr;
    int64_t n = 0, np = 0;
    double* fill[2] = {nullptr, nullptr};
    double* ident = nullptr;  // np x np matrix (leading dimension ident_ld) to be set to the identity: the right-hand side
    int64_t ident_ld = 0;     // of the inverse factor the gradient paths compute next (enqueue_inverse_factor)
  } early;
  bool identity_ready = false;  // EarlyWork has written the identity into B2 for the inverse factor that follows
  bool pair_checked = false;       // potrf_pair_quiesce has run for the factorisation being enqueued (potrf.hip)
  hipEvent_t ev_la = nullptr, ev_panel = nullptr, ev_join = nullptr, ev_main = nullptr, ev_slice = nullptr;
  double* A = nullptr;      // np x ld scratch (K then L)
  double* invD = nullptr;   // (np/128) x 128 x 128 inverses of the diagonal blocks
  double* B2 = nullptr;     // second np x ld matrix (L^-T for the gradient / LOO paths), allocated lazily
  // inverses of the 512 x 512 diagonal blocks of L (many-right-hand-side solves), built lazily from invD
  double* inv2 = nullptr;    // ceil(nt / 4) x 512 x 512
  double* inv2_t = nullptr;  // scratch: ceil(nt / 4) x 256 x 256
  bool inv2_valid = false;
  double* gws = nullptr;    // gradient partial sums
  int64_t gws_doubles = 0;
  double* vec = nullptr;    // 4 x np work vectors
  double* red = nullptr;    // small reduction outputs (device)
  int* info = nullptr;      // device info word
  double* h_red = nullptr;  // pinned host mirror of red
  int* h_info = nullptr;    // pinned host mirror of info
  // flag-ordered factorisation of the chain-bound part (potrf_flow.hip): task lists of a tail of flow_m tile rows
  void* flow_tasks = nullptr;
  int* flow_off = nullptr;
  int* flow_flags = nullptr;
  int flow_m = 0, flow_nwg = 0, flow_nlists = 0;
  int64_t flow_ntasks = 0;
  double flow_flops_update = 0.0, flow_flops_trsm = 0.0;
};

// device state of the linear-inversion entry points (gpmi_linv_*): m data values, model matrix A (m x n)
struct LinvState {
  int64_t m = 0, mp = 0, ldm = 0;
  double* A = nullptr;     // mp x ld   model matrix, zero padded
  double* At = nullptr;    // np x ldm  its transpose
  double* y = nullptr;     // mp
  double* sig2 = nullptr;  // mp        y_err^2, 1 in the padding (keeps the padded J positive definite)
  double* zero = nullptr;  // np        zeros (the prior covariance carries no data noise)
  double* K = nullptr;     // np x ld   prior covariance (later A^T J^-1 A, K - X X^T)
  double* T = nullptr;     // mp x ld   A K (later J^-1 A)
  double* J = nullptr;     // mp x ldm  A K A^T + Sigma, then its factor L
  double* J2 = nullptr;    // mp x ldm  L^-T, then J^-1          (gradient only)
  double* Q = nullptr;     // np x ldm  K A^T                    (posterior only)
  double* X = nullptr;     // np x ldm  K A^T L^-T               (posterior only)
  double* invD = nullptr;  // 128-wide inverse diagonal blocks of L
  double* inv2 = nullptr;  // 512-wide
  double* inv2_t = nullptr;
  double* panel = nullptr; // mp x 544 (in-place many-right-hand-side solve)
  double* vec = nullptr;   // 8 x max(mp, np) work vectors
  double* gws = nullptr;
  int64_t gws_doubles = 0;
};

struct gpmi_ctx {
  int device = 0;
  int ncu = 256;      // compute units of the device
  int pair_cus[GPMI_NPAIRS] = {32};  // CUs of the panel stream of look-ahead pair k (the update stream has the rest)
  std::string err;
  // data
  int64_t n = 0, d = 0, np = 0, ld = 0;
  double* x = nullptr;      // np x d (rows >= n are zero)
  double* y = nullptr;      // np
  double* noise = nullptr;  // np diagonal data variances (zero padded)
  double* ycov = nullptr;   // n x n dense y covariance or nullptr
  // fitted state lives in lanes[0]
  std::vector<Lane> lanes;
  bool fitted = false;
  bool lockstep_always = false;  // GPMI_OPT_LOCKSTEP_ALWAYS
  bool no_flow = false;          // GPMI_OPT_NO_FLOW
  int64_t reserve = 0;           // GPMI_OPT_RESERVE_POINTS: extra rows of padding at the next gpmi_set_data
  KParams fit_params{};
  double* alpha = nullptr;  // np (device) — fitted alpha
  // prediction workspace
  double* Q = nullptr;      // mq_cap x ld
  double* Q2 = nullptr;     // second panel (spatial derivatives)
  int64_t mq_cap = 0;
  double* pts = nullptr;    // mq_cap x d
  double* pvec = nullptr;   // vectors of length mq_cap * (2 + 2 d)
  double* trsm_panel = nullptr;  // rows x 544 scratch of the in-place many-right-hand-side solve
  int64_t trsm_panel_rows = 0;
  // host staging (pinned)
  double* h_stage = nullptr;
  int64_t h_stage_bytes = 0;
  // batched small-problem workspace (gpmi_lml_batch for np <= 4096): `bcap` matrices side by side
  int bcap = 0;
  double* bA = nullptr;
  double* bInv = nullptr;
  double* bVec = nullptr;
  double* bRed = nullptr;
  double* bMu = nullptr;
  double* bB2 = nullptr;     // second matrix per problem (L^-T), gradient batches only
  double* bGws = nullptr;    // partial sums of the fused contraction
  double* bNoise = nullptr;  // gpmi_lml_grad_batch_noise: one noise-variance vector per problem
  int bNoise_cap = 0;
  // gpmi_lml_grad_batch_mix: per problem the window weights g_m and the row sums h_m (GPMI_MAX_MIX x np each), the
  // sub-kernels' parameters (GPMI_MAX_MIX arrays of bcap KParams) and the WhiteNoise variances
  double* bMixG = nullptr;
  double* bMixH = nullptr;
  double* bMixW = nullptr;  // the caller's row-sum weights (gpmi_*_grad_batch_mix: hw)
  KParams* bMixP = nullptr;
  double* bMixExtra = nullptr;
  int bMix_cap = 0;
  double* bLoo = nullptr;    // gpmi_loo_grad_batch: 4 vectors per problem (diag K^-1, c1, sqrt c2, p)
  int bLoo_cap = 0;
  double* bGout = nullptr;   // (n_theta + 1) results per problem
  double* h_bGout = nullptr;
  int bgrad_cap = 0, bgrad_ntheta = 0;
  int* bInfo = nullptr;
  KParams* bParams = nullptr;
  // asynchronous lockstep batches (gpmi_lml_batch_submit / _wait): two slots = the two halves of the workspace, on the
  // streams of lanes 1 and 2; evaluations pending per slot (0: free), pinned staging of their inputs
  int bpend[2] = {0, 0};
  char* h_bStage[2] = {nullptr, nullptr};
  int64_t h_bStage_bytes[2] = {0, 0};
  double* h_bRed = nullptr;
  int* h_bInfo = nullptr;
  LinvState linv;
  // fitted mixture model (gpmi_fit_mix): sub-kernel parameters and the training-point weights g (nk x np)
  int mix_nk = 0;
  KParams mix_p[4];
  double* mix_g = nullptr;        // nk x np (row m: g_m; padding 1 for m = 0, 0 otherwise)
  double* mix_scratch = nullptr;  // np x ld
  double* mix_zero = nullptr;     // np zeros
  double* Q3 = nullptr;           // third query panel (mq_cap x ld)
  int64_t q3_cap = 0;
  // RCCL result gather (comm.hip)
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  hipStream_t comm_stream = nullptr;
  double* comm_buf = nullptr;
  int64_t comm_buf_doubles = 0;
  hipStream_t dev_masked = nullptr;  // tools: CU-masked stream of the device-pointer entry points (GPMI_DEV_CUS)
  // instrumentation
  hipEvent_t t0 = nullptr, t1 = nullptr;
  unsigned prof_mask = 0;
  // device-stamped launches (trailing-update class): {min start, max end} per launch
  unsigned long long* stamp_pool = nullptr;
  std::vector<double> stamp_flops, stamp_bytes;
  std::vector<int> stamp_class;
  std::vector<ProfSlot> prof_slots;
  size_t prof_used = 0;
  double prof_clock_cycles = 0.0, prof_clock_ticks = 0.0;  // shader cycles / 10 ns ticks over stamped workgroups
  double prof_ms[GPMI_PROF_NCLASS] = {};
  double prof_flops[GPMI_PROF_NCLASS] = {};
  double prof_bytes[GPMI_PROF_NCLASS] = {};
  int64_t prof_launches[GPMI_PROF_NCLASS] = {};
};

// instrumentation helpers (api.hip)
constexpr int GPMI_STAMP_SLOTS = 16384;
constexpr int GPMI_STAMP_WORDS = 24;  // per launch: 8 start words + 8 end words (one per XCD) + 8 clock words, see gemm_f64.hip
// next stamp slot (GPMI_STAMP_WORDS words) for a trailing-update launch, or nullptr when that class is not profiled
unsigned long long* prof_stamp_slot(gpmi_ctx* c, double flops, double bytes, int klass = GPMI_PROF_SYRK);
struct ProfScope {
  gpmi_ctx* c;
  hipStream_t s;
  ProfSlot* slot;
  ProfScope(gpmi_ctx* ctx, hipStream_t st, int klass, double flops, double bytes);
  ~ProfScope();
};

// ---- kernel launchers -------------------------------------------------------------
// kbuild.hip
// square covariance of the np x np padded problem (identity in the padding), lower tiles only if lower_only
void launch_kbuild_square(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                          const double* noise, double* A, int64_t ld, bool lower_only);
void launch_kbuild_square_part(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                               const double* noise, double* A, int64_t ld, int part, int split_cols);
// cross covariance U (mp x d, mu valid rows) vs V (np x d, n valid rows): out mp x ld, zeros in padding
void launch_kbuild_cross(hipStream_t s, const KParams& p, const double* U, int64_t mu, int64_t mp,
                         const double* V, int64_t n, int64_t np, double* out, int64_t ld);
void launch_add_full(hipStream_t s, double* A, int64_t ld, const double* Y, int64_t n);

// grad.hip: fused contraction 1/2 sum (alpha alpha^T - K^-1) o dK/dtheta_j with dK recomputed from x.
// out[0..n_theta) = gradient, out[n_theta] = sum_i (alpha_i^2 - K^-1_ii).  iK holds K^-1 (lower tiles).
int64_t grad_ws_doubles(int64_t np, int n_theta);
// general form: Q_ab = 1/2 (u_a v_b + u_b v_a) - iK_ab  (LML gradient: u = v = alpha)
void launch_lml_grad(hipStream_t s, const KParams& p, int n_theta, const double* x, int64_t n,
                     int64_t np, const double* iK, int64_t ld, const double* u, const double* v,
                     double* ws, double* out);
void launch_lml_grad_batched(hipStream_t s, const KParams* pdev, int batch, int n_theta, const double* x, int64_t n,
                             int64_t np, const double* iK, int64_t ld, int64_t sK, const double* u, const double* v,
                             int64_t sV, double* ws, double* out);
void launch_mirror_lower(hipStream_t s, double* A, int64_t ld, int64_t np, int batch = 1, int64_t sMat = 0);
// (batch > 1: problem z works on A / G + z sMat and on vectors sVec (sAlpha, sLoo) apart)
void launch_scale_columns(hipStream_t s, const double* A, const double* sc, double* G, int64_t ld,
                          int64_t np, int batch = 1, int64_t sMat = 0, int64_t sVec = 0);
void launch_loo_vectors(hipStream_t s, const double* alpha, const double* ikdiag, double* c1,
                        double* sc2, int64_t n, int64_t np, int batch = 1, int64_t sAlpha = 0, int64_t sLoo = 0);

// flags a chain launch of the flag-ordered factorisation publishes / waits for at its start (potrf_flow.hip,
// gemm_tiles.h: flow_hook_enter)
struct FlowHook {
  int* pub = nullptr;         // *pub = pub_val at the start of the launch (first thread of workgroup 0)
  int pub_val = 0;
  const int* wait = nullptr;  // every workgroup: spin until *wait >= wait_val
  int wait_val = 0;
  int* abort = nullptr;       // non-zero: somebody timed out; stop waiting.  Set by a poll that times out itself
  int* info = nullptr;        // receives GPMI_ERR_INTERNAL on a time-out (if still zero)
  unsigned long long* wait_ticks = nullptr;  // diagnostics: 10 ns ticks spent waiting, summed over the workgroups
  unsigned long long* trace = nullptr;       // diagnostics: {launch start, wait over} (s_memrealtime), workgroup 0
};

// batch of independent equal-shape problems in one launch (blockIdx.z): strides in doubles
struct GemmBatch {
  int count = 1;
  int64_t sC = 0, sA = 0, sB = 0;
  // CUs the launch's stream may use (0: the whole chip): the choice of 32-row tiles ("while CUs are idle") scales with it
  int ncu_hint = 0;
  // B of a one-tile-column product with K = 128 is lower triangular (B[j][k] = 0 for k > j: the inverse of a diagonal
  // block): the kernel skips the zero part
  bool b_lower_tri = false;
  // lockstep launches whose values must not depend on how many problems share the launch: only tile shapes that sum in
  // the ring kernels' order (no 32-row register-staged tiles at K > 128, whose use depends on the batch size; round 6:
  // the inverse of a gradient batch of one differed from the same problem's inside a batch of six in the last bits)
  bool ring_order_only = false;
  FlowHook hook;
};
struct BatchShape {
  int count = 1;
  int64_t sMat = 0;   // between the np x ld matrices
  int64_t sInv = 0;   // between the invD arrays
  int64_t sVec = 0;   // between the work-vector sets
};

// mix.hip: pieces of the mixture covariance K = sum_m diag(g_m) K_m diag(g_m) (ChangePoint)
constexpr int GPMI_MAX_MIX = 4;
// (batch > 1: problem z of a lockstep batch works on matrices sD / sS / sA / sMat apart and on weight vectors sG apart)
void launch_scale_add(hipStream_t s, double* dst, int64_t ldd, const double* src, int64_t lds,
                      const double* gr, const double* gc, int64_t rows, int64_t cols, bool accumulate, int batch = 1,
                      int64_t sD = 0, int64_t sS = 0, int64_t sG = 0);
void launch_add_diag_vec(hipStream_t s, double* A, int64_t ld, const double* noise, double extra, int64_t n,
                         int batch = 1, int64_t sA = 0, const double* extras = nullptr);
void launch_vec_mul(hipStream_t s, const double* a, const double* b, double* out, int64_t n, int batch = 1,
                    int64_t sA = 0, int64_t sB = 0, int64_t sOut = 0);
// h_i = sum_j (1/2 (u_i alpha_j + alpha_i u_j) - iK_ij) Km_ij g_j  (iK, Km full n x n; u = nullptr: u = alpha, the LML form);
// pair != 0: the same pass also takes the weights at g + pair into h + pair
void launch_mix_rowsum(hipStream_t s, const double* iK, const double* Km, int64_t ld, const double* alpha,
                       const double* g, double* h, int64_t n, int batch = 1, int64_t sMat = 0, int64_t sAlpha = 0,
                       int64_t sG = 0, const double* u = nullptr, int64_t sU = 0, int64_t pair = 0);

// gemm_f64.hip  (all dims multiples of 128, k multiple of 16)
enum GemmTiles { TILES_RECT = 0, TILES_LOWER = 1 };
enum GemmOp { OP_SUB = 0, OP_ASSIGN = 1 };
// C(ntr*128 x ntc*128) op= A(rows x k) * B(cols x k)^T ; TILES_LOWER visits tiles ti >= tj only
void launch_gemm_nt(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc,
                    const double* A, int64_t lda, const double* B, int64_t ldb, int ntr, int ntc,
                    int k, unsigned long long* stamp = nullptr, const GemmBatch& bt = GemmBatch());

// balanced form of launch_gemm_nt for big launches (gemm_f64.hip): the first `nfull` = gemm_split_point(T, ncu, k)
// tiles run as 128 x 128 tiles, the tiles of the nearly empty last round as 64 x 64 tiles in a second launch;
// `stamp` times the first launch only.  nend >= 0: the product stops at logical tile `nend` - the tiles from there on
// are somebody else's (launch_gemm_nt_range on another stream)
int64_t gemm_split_point(int64_t T, int ncu, int k);
bool gemm_mixed_launches();  // GPMI_GEMM_MIXED: launch_gemm_nt_split puts the remainder's quarters into the launch of the full rounds
void launch_gemm_nt_split(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc, const double* A,
                          int64_t lda, const double* B, int64_t ldb, int ntr, int ntc, int k, int64_t nfull,
                          unsigned long long* stamp = nullptr, unsigned long long* stamp_rest = nullptr,
                          int64_t nend = -1);
void launch_gemm_nt_range(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc, const double* A,
                          int64_t lda, const double* B, int64_t ldb, int ntr, int ntc, int k, int64_t first,
                          int64_t count, unsigned long long* stamp = nullptr);

// general form: b_kmajor -> B is (k x cols) row-major; kskip = 1 (TILES_LOWER) -> contraction starts at
// the tile row's first column, kskip = 2 -> it ends with the tile column (B lower triangular)
void launch_gemm(hipStream_t s, GemmTiles tiles, GemmOp op, bool b_kmajor, int kskip, double* C,
                 int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int ntr,
                 int ntc, int k, unsigned long long* stamp = nullptr, const GemmBatch& bt = GemmBatch());

// potrf.hip
hipStream_t potrf_first_update_stream(gpmi_ctx* c, Lane& lane, int64_t np, bool allow_lookahead);
// waits for whatever an earlier (failed / foreign) call left on the lane's CU-masked pair; before the caller's own enqueues
void potrf_pair_quiesce(Lane& lane);
void launch_potrf_diag(hipStream_t s, double* Ablk, int64_t ld, double* invD, int* info, int col0,
                       unsigned long long* dbg = nullptr, const BatchShape& bs = BatchShape(), int* pub = nullptr,
                       int pub_val = 0);
void launch_potrf_diag_fault(hipStream_t s, double* Ablk, int64_t ld, const BatchShape& bs = BatchShape());  // test hook
// batched, in-order factorisation of bs.count matrices (small problems: no look-ahead)
void potrf_lower_batched(gpmi_ctx* c, hipStream_t s, double* A, int64_t np, int64_t ld, double* invD,
                         int* info, const BatchShape& bs);
void launch_kbuild_square_batched(hipStream_t s, int kernel, const KParams* pdev, int batch, const double* x,
                                  int64_t n, int64_t np, const double* noise, double* A, int64_t ld,
                                  int64_t stride, int d, int64_t noise_stride = 0);
// blocked right-looking Cholesky, in place, lower; invD receives the inverses of the diagonal blocks
// allow_lookahead = false keeps everything on the lane's full-chip stream (several lanes running
// concurrently already fill the chip, and their masked stream pairs would only fight for HW queues)
void potrf_lower(gpmi_ctx* c, Lane& lane, double* A, int64_t np, int64_t ld, double* invD,
                 int* info, bool allow_lookahead = true);

// potrf_flow.hip: the chain-bound part of the factorisation as one persistent tile-task launch beside the panel chain
bool potrf_flow_enabled(gpmi_ctx* c, Lane& lane, int m);
bool potrf_flow_tail(gpmi_ctx* c, Lane& lane, double* A, int64_t ld, double* invD, int* info, int nt, int t0);
void potrf_flow_free(Lane& lane);

// solve.hip
// forward substitution  L v = r : the solution goes to `out` (no aliasing; `r` is only read).
// `err` (device int, may be null) receives GPMI_ERR_INTERNAL if the sweep's polling ever times out.
// `prefilled`: `out` already holds the sentinel (lane_run_early), the sweep does not fill it again.
void trsv_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                  const double* invD, const double* r, double* out, int* err = nullptr,
                  const BatchShape& bs = BatchShape(), bool prefilled = false);
// backward substitution  L^T a = v : the solution goes to `out` (no aliasing; `r` is only read)
void trsv_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                   const double* invD, const double* r, double* out, int* err = nullptr,
                   const BatchShape& bs = BatchShape(), bool prefilled = false);
// the lane's pending EarlyWork, if any, on stream s (the lane's own stream)
void lane_run_early(Lane& lane, hipStream_t s);
// lockstep batch: Q_z <- L_z^-T (np <= 4096), every launch carries the batch
void trsm_identity_batched(hipStream_t s, const double* L, int64_t np, int64_t ld, const double* invD, double* Q,
                           const BatchShape& bs);
// inverses of the 512-wide diagonal blocks of L from the 128-wide ones: inv2 (slots of 512 x 512, ld 512),
// tmp: slots of 256 x 256
constexpr int GPMI_OB = 512;
void build_inv2(hipStream_t s, const double* L, int64_t np, int64_t ld, const double* invD, double* inv2,
                double* tmp);
// X = Q L^-T  (forward solve of mp right-hand sides stored as rows of Q, mp x np, ld).  Q is consumed.
// Qout != nullptr: X goes to Qout (same shape / ld).  Qout == nullptr: X replaces Q, staged through
// `panel` (mp x 544).  upper_rhs: Q is upper triangular (rows below the current block still zero).
void trsm_rows_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                       const double* inv2, double* Q, int64_t mp, bool upper_rhs, double* Qout,
                       double* panel);
void launch_set_identity(hipStream_t s, double* Q, int64_t ld, int64_t np, int batch = 1, int64_t sMat = 0);
// device-to-device vector copy as a kernel (a runtime D2D memcpy stalled the stream for tens of ms)
void launch_copy(hipStream_t s, const double* src, double* dst, int64_t n);
// `height` rows of `width` doubles: src rows `spitch` doubles apart -> dst rows `dpitch` doubles apart
void launch_copy_rows(hipStream_t s, const double* src, int64_t spitch, double* dst, int64_t dpitch, int64_t width,
                      int64_t height);
// r = y - mu (padded with zeros)
void launch_residual(hipStream_t s, const double* y, const double* mu, double mu_const, double* r,
                     int64_t n, int64_t np);
// batched: r_z = y - mu_consts[z]  (or mus + z * n when mus != nullptr), r_z at r + z * sVec
void launch_residual_batched(hipStream_t s, const double* y, const double* mus, const double* mu_consts,
                             double* r, int64_t n, int64_t np, const BatchShape& bs);
// red[0] = sum v^2, red[1] = sum log diag(L)
void launch_lml_reduce(hipStream_t s, const double* v, const double* L, int64_t ld, int64_t np,
                       double* red, const BatchShape& bs = BatchShape());
// out[m] = sum_n Q[m][n] * a[n]
void launch_rows_dot(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                     const double* a, double* out, int batch = 1, int64_t sQ = 0, int64_t sVec = 0);
// out[m] = sign (base - sum_n Q[m][n]^2)
void launch_rows_sumsq(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                       double base, double* out, int batch = 1, int64_t sQ = 0, int64_t sVec = 0, double sign = 1.0);

// Q (mp x np) <- Q L^-1   (backward solve of mp right-hand sides stored as rows)
// (inv2 / panel: the 512 x 512 inverse blocks and an mp x (512 + 32) scratch panel, as for trsm_rows_forward; without them
// the sweep runs over the 128-wide blocks with invD)
void trsm_rows_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                        const double* invD, double* Q, int64_t mp, const double* inv2 = nullptr, double* panel = nullptr);

// predgrad.hip
void launch_sd_reduce(hipStream_t s, const KParams& p, const double* x, int64_t n, const double* pts,
                      int64_t m, const double* Kq, int64_t ld, const double* W, int64_t ldw,
                      double scale, double* out);
void launch_grad_rhs(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                     const double* pts, int64_t rows_valid, int64_t rows_padded, const double* Kq,
                     int64_t ld, double* G);
void launch_grad_cov(hipStream_t s, const KParams& p, const double* G, int64_t ld, int64_t np,
                     int64_t m, double* cov);
