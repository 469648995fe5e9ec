// Flag-ordered ("dataflow") tile Cholesky for the part of the factorisation that is bound by the panel chain:
// the last tile rows of a large matrix and the whole of a matrix of N <= ~8000 (numpy.linalg.cholesky at
// regression.py:241, 537, 555).
//
// In stream order that part costs chain + updates: potrf_diag (29.5 us) -> panel TRSM -> inner update per 128 columns
// with the chip nearly idle, then a trailing update with the chain idle.  Here the two run side by side:
//
//   chain stream (the lane's CU-masked panel stream, ordinary launches in stream order, nothing but the critical path):
//       D(k)  = potrf_diag of tile (k, k)
//       Tc(k) = tile (k+1, k) <- tile (k+1, k) invD(k)^T              (4 workgroups of 32 rows)
//       Uc(k) = tile (k+1, k+1) -= L(k+1, k) L(k+1, k)^T               (3 workgroups of 64 x 64)
//   update stream (the other CUs): ONE persistent launch; every workgroup owns a fixed list of tile tasks and runs it in
//   order, starting a task when the flags of its inputs are set:
//       T(i, k), i >= k + 2    panel TRSM of tile (i, k), four 32-row slabs
//       U(i, j, k)             tile (i, j) -= L(i, k) L(j, k)^T for ONE column k (K = 128), four 64 x 64 sub-tiles;
//                              columns of the tile's own outer panel (and, near the diagonal, of the panel before it)
//       Z(i, j, q)             tile (i, j) -= L(i, 4q..4q+3) L(j, 4q..4q+3)^T  (K = 512, the throughput kernel), every
//                              earlier outer panel
//   i.e. the two-level blocking of potrf_lower, tile by tile, with the same kernels' tile bodies (gemm_tiles.h) and
//   the same order of summation for every element: the factor is bit-identical to the stream-ordered schedule's.
//
// Flags (ints, zeroed per factorisation): Ddone = number of diagonal blocks factored; Lcnt[i] = TRSM slabs finished
// in tile row i (4 per column, columns in order); F[i][j] = sub-updates applied to tile (i, j) (4 per column).
// A chain launch publishes the results of the launch before it at its own start (the kernel boundary has made them
// visible) and waits - normally not at all - for the flags of the tile it is about to touch.  A task kernel workgroup
// releases its stores (agent-scope release fence) before it bumps a flag and acquires after it has claimed a task.
//
// Scheduling is static: the tasks, sorted by a virtual time under which each task comes after all its inputs (T(i, k)
// at k, U(., ., k) at k + 1/2, Z(., ., q) right behind T(., 4q + 3)), are dealt round-robin to the workgroups - the
// tasks of the rows next to the chain (i - k <= 3) to the first FLOW_NEAR_WGS workgroups, which do nothing else, so
// that what the chain waits for never queues behind a K = 512 tile.  A workgroup polls only the two or three flags
// of ITS next task: no shared queue heads, no claims, nothing every workgroup re-reads.  (First version: dynamic
// in-order queues with a claim word per task, every idle workgroup re-reading ~300 flags at every publication:
// 60 ms for a 7 ms factorisation - the polling saturated the fabric; tools/sim/flow_sim.py prices the static deal
// at 3.7 ms against 3.6 for ideal queues.)
// Progress: every list is in virtual-time order, so the earliest unfinished task overall is the head of its list and
// all its inputs are complete: whoever holds that list finds it ready.  Every list has a holder that is resident: a
// list whose own workgroup has not started (two workgroups per CU of the update stream's mask is what 64 KiB of LDS
// and 256 VGPRs admit, but another process may hold those slots) is adopted after 50 us by a workgroup that is.
// The only cross-launch dependency is between the chain stream and the task kernel, which run on disjoint CU masks.
// Every poll is bounded: a time-out (a bug or a serialising profiler, never a wait) sets the abort word, every
// poller gives up, and the host reports GPMI_ERR_INTERNAL (info = GPMI_INFO_FLOW_TIMEOUT) instead of hanging the GPU.
// tools/sim/flow_sim.py holds the same task list as an executable model (NumPy replay in random admissible order
// + a discrete-event timing model that chose the queue layout).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <tuple>
#include <vector>

#include "gemm_tiles.h"
#include "potrf_diag.h"

using namespace gemm_tiles;

namespace {

constexpr int NB = GPMI_NB;
constexpr int OBT = 4;        // tile columns per outer panel (K = 512 chunks)
// tasks of the rows within FLOW_NEAR_D (3) tile rows of the column being applied go to the first FLOW_NEAR_WGS (32)
// workgroups (GPMI_FLOW_NEAR_D, GPMI_FLOW_NEAR_WGS: tuning aids)
int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
const int FLOW_NEAR_D = env_int("GPMI_FLOW_NEAR_D", 3);
const int FLOW_NEAR_WGS = (env_int("GPMI_FLOW_NEAR_WGS", 32) + 7) / 8 * 8;
constexpr int FT_T = 0, FT_U = 1, FT_Z = 2, FT_ZS = 3;
// (round 6) GPMI_FLOW_SPLIT=1: the K = 512 chunks of a workgroup in a SECOND list (list nwg + b of workgroup b), looked at
// only when the head of its first list (panel TRSMs, one-column updates) is not ready; GPMI_FLOW_QUARTER=<q0>: the chunks
// of the outer panels q >= q0 as four 64 x 64 sub-tile tasks FT_ZS (35 us instead of 125: what a short task can sit behind)
const int FLOW_SPLIT = env_int("GPMI_FLOW_SPLIT", 1);
const int FLOW_QUARTER = env_int("GPMI_FLOW_QUARTER", 0);
// GPMI_FLOW_URGENT=<d> (with GPMI_FLOW_SPLIT): the last d chunks of a tile - the ones its own outer panel waits for - in a list
// of their own per workgroup (list nwg + b), ordered by the panel that needs them (as late as possible: right in front
// of the tile's first single-column task) and looked at before the remaining chunks (list 2 nwg + b, as soon as possible:
// by outer panel q).  With every chunk in ONE list in q order, chunk q + 1 on the tiles the chain is about to reach sat
// behind all of chunk q on tiles it will not reach for another millisecond (tools/flow_curve.py: the bodies of chunk
// q + 1 began when the last body of chunk q ended, 200 - 250 us after their inputs were complete).
const int FLOW_URGENT = env_int("GPMI_FLOW_URGENT", 3);

struct FlowTask {
  uint8_t type, s, fadd, pad;
  uint16_t i, j, k, pad2;  // k: column (T, U) or outer panel q (Z)
};
static_assert(sizeof(FlowTask) == 12, "FlowTask layout");

// hot words on lines of their own
// FL_ROWCNT: 8 words (chain_fused_kernel, chain_column_kernel); FL_INVROW: 8 words on lines of their own (row blocks of
// the current inverse diagonal block, chain_column_kernel)
constexpr int FL_DDONE = 0, FL_ABORT = 32, FL_T0 = 48, FL_STATS = 64, FL_ROWCNT = 96, FL_INVROW = 128, FL_LCNT = 384;
__host__ __device__ inline int flow_f_off(int m) { return FL_LCNT + ((m + 31) / 32) * 32; }
__host__ __device__ inline int flow_owner_off(int m) { return flow_f_off(m) + ((m * m + 31) / 32) * 32; }  // one word per list
inline int flow_flag_ints(int m, int nwg) { return flow_owner_off(m) + nwg; }
constexpr int FLOW_MAX_LISTS = 1536;                 // lists a workgroup can hold (its own + adopted ones)
constexpr int FLOW_MAX_WGS = 512;                    // workgroups of the task launch (up to three lists each with GPMI_FLOW_SPLIT)
constexpr unsigned long long FLOW_GRACE_TICKS = 5000;  // 50 us: a list nobody has claimed by then is an orphan

struct FlowArgs {
  double* A;           // tile (0, 0) of the tail
  const double* invD;  // inverse of diagonal block 0 of the tail
  int64_t ld;
  int m;
  const FlowTask* tasks;  // the lists, workgroup after workgroup
  const int* off;         // list l: tasks[off[l] .. off[l + 1]); workgroup b owns list b (and list gridDim.x + b if nlists = 2 gridDim.x)
  int nlists;
  int* flags;
  int* info;
  unsigned long long* stamp;
  unsigned long long* stats;  // GPMI_FLOW_STATS: {waiting ticks, body ticks, polls, tasks} summed over the workgroups
  unsigned long long* trace;  // GPMI_FLOW_TRACE: per task {poll start, inputs seen, body done, published} (10 ns ticks)
  int fault;                  // test hook (GPMI_FLOW_FAULT=<n>): task n never sets its flag; -1: none
};

__device__ __forceinline__ bool flow_ready(const FlowTask& t, const int* __restrict__ fl, int m) {
  const int* Lcnt = fl + FL_LCNT;
  const int* F = fl + flow_f_off(m);
  if (t.type == FT_T) return flow_ld(fl + FL_DDONE) >= t.k + 1 && flow_ld(F + t.i * m + t.k) == 4 * t.k;
  if (t.type == FT_U)
    return flow_ld(Lcnt + t.i) >= 4 * (t.k + 1) && flow_ld(Lcnt + t.j) >= 4 * (t.k + 1) &&
           flow_ld(F + t.i * m + t.j) >= 4 * t.k;
  if (t.type == FT_ZS)  // a quarter of chunk k: its siblings may already have counted
    return flow_ld(Lcnt + t.i) >= 4 * OBT * (t.k + 1) && flow_ld(Lcnt + t.j) >= 4 * OBT * (t.k + 1) &&
           flow_ld(F + t.i * m + t.j) >= 4 * OBT * t.k;
  return flow_ld(Lcnt + t.i) >= 4 * OBT * (t.k + 1) && flow_ld(Lcnt + t.j) >= 4 * OBT * (t.k + 1) &&
         flow_ld(F + t.i * m + t.j) == 4 * OBT * t.k;
}

// The two small tile bodies are kept out of line: with all three inlined into one loop the register allocation ran out
// (256 VGPRs and 196 bytes of scratch per lane, spills inside the K = 512 body); as functions they keep their own
// and the K = 512 body, inlined, has the kernel's to itself.  (The LDS block travels as an address-space-3 pointer
// and the matrices as address-space-1 pointers so that the accesses stay ds_ / global_ instructions, not flat_.)
typedef __attribute__((address_space(3))) double lds_double_t;
typedef __attribute__((address_space(1))) double glb_double_t;  // (generic pointers would turn every access into flat_)
// PROTO (GPMI_FLOW_PROTO): how a tile changes hands between CUs.  0: write-through (sc1) stores, acquire on the
// consumer; 1: the same + an agent release in front of the flag; 2: sc1 stores and sc1 loads of every matrix byte;
// 3: non-temporal stores + agent release (whole-L2 write-back per task).
template <int PROTO>
__device__ __attribute__((noinline)) void flow_do_T(glb_double_t* C, const glb_double_t* invDk, int64_t ld,
                                                    lds_double_t* smem) {
  staged_tile<OP_ASSIGN, 0, 32, 128, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN, true>(
      (const double*)C, (const double*)invDk, (double*)C, ld, NB, ld, NB / BK, (double*)smem);
}
template <int PROTO>
__device__ __attribute__((noinline)) void flow_do_U(const glb_double_t* Ai, const glb_double_t* Bj, glb_double_t* C,
                                                    int64_t ld, lds_double_t* smem, int nk) {
  // (nk and the LDS block arrive in vector registers - the function is not inlined and has two callers -; the ring's
  // stage addresses must be scalar: they go through m0)
  lds_double_t* su = (lds_double_t*)(uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
  dma64_tile<OP_SUB, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN>(
      (const double*)Ai, (const double*)Bj, (double*)C, ld, ld, ld, __builtin_amdgcn_readfirstlane(nk), (double*)su);
}
// ---- the chain's two products per column, on tiles of 16 rows (round 4) ----------------------------------------------
// Tc(k): tile (k+1, k) <- tile * invD(k)^T and Uc(k): tile (k+1, k+1) -= X X^T are 128 x 128 x 128 products on the
// critical path.  The tile bodies of the throughput kernels put one wave on 4 - 8 MFMA tiles (a 32 x 128 TRSM slab: 8 per
// wave, 4.2 us of MFMA time on its SIMD; a 64 x 64 update tile: 4 per wave over K = 128, the same) - right when the
// launch has hundreds of workgroups, slow when it has four.  Here every wave owns ONE 16 x 16 tile of the result, takes
// its operands straight from L2 into registers (all requests in flight at once, no LDS, no barrier in the loop) and runs
// its 4 .. 32 MFMAs as one chain: 6.2 -> ~3 us and 6.7 -> ~3 us per column.
// The sums are those of the throughput bodies BIT FOR BIT, because a flow-ordered factorisation must equal the stream-
// ordered one, which runs these tiles inside launches of the generic kernels: the register-staged TRSM body
// (gemm_tiles::staged_tile, BTRI) feeds k = 16 s + 4 fk + q to MFMA step q of slab s and skips the slabs beyond the
// column block; the ring body of the update (gemm_tiles::dma64_tile) feeds k = 8 st + 2 fk to the first and
// 8 st + 2 fk + 1 to the second MFMA of stage st and negates A through the MFMA's BLGP field.  Same operands, same order.
struct ChainArgs {
  double* T;         // tile (k+1, k): 128 x 128, Tc's input and output, Uc's operand
  double* C;         // tile (k+1, k+1): Uc's target
  const double* invD;  // inverse of diagonal block k (row-major 128 x 128, lower triangular)
  int64_t ld;
  FlowHook hook;
};

// 8 workgroups x 8 waves: workgroup b owns the rows 16 b .. 16 b + 15 (in place: nobody else reads or writes them), wave c
// the column block c.
__global__ __launch_bounds__(512) void chain_trsm_kernel(ChainArgs g) {
  flow_hook_enter(g.hook);
  const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int fr = lane & 15, fk = lane >> 4;
  const double* arow = g.T + (int64_t)(16 * blockIdx.x + fr) * g.ld + 4 * fk;
  const double* brow = g.invD + (int64_t)(16 * c + fr) * NB + 4 * fk;
  d4_t a[NB / 16], b[NB / 16];
#pragma unroll
  for (int s = 0; s < NB / 16; ++s)
    if (s <= c) {  // column block c of a lower-triangular B: zero beyond slab c
      a[s] = *reinterpret_cast<const d4_t*>(arow + 16 * s);
      b[s] = *reinterpret_cast<const d4_t*>(brow + 16 * s);
    }
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < NB / 16; ++s)
    if (s <= c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][q], b[s][q], acc, 0, 0, 0);
    }
  // every wave of the workgroup has its operands (the MFMAs above waited for them) before anybody overwrites the rows
  __syncthreads();
  double* out = g.T + (int64_t)(16 * blockIdx.x + fk) * g.ld + 16 * c + fr;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(int64_t)4 * r * g.ld] = acc[r];
}

// 9 workgroups x 4 waves: the 36 tiles (i, j), j <= i, of 16 x 16 of the target's lower triangle (diagonal tiles in full)
__global__ __launch_bounds__(256) void chain_syrk_kernel(ChainArgs g) {
  flow_hook_enter(g.hook);
  const int lane = threadIdx.x & 63;
  const int t = __builtin_amdgcn_readfirstlane((int)(4 * blockIdx.x + (threadIdx.x >> 6)));
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  const int j = t - i * (i + 1) / 2;
  const int fr = lane & 15, fk = lane >> 4;
  const double* arow = g.T + (int64_t)(16 * i + fr) * g.ld + 2 * fk;
  const double* brow = g.T + (int64_t)(16 * j + fr) * g.ld + 2 * fk;
  double* cp = g.C + (int64_t)(16 * i + fk) * g.ld + 16 * j + fr;
  d4_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = cp[(int64_t)4 * r * g.ld];
  d2_t a[NB / 8], b[NB / 8];
#pragma unroll
  for (int st = 0; st < NB / 8; ++st) {
    a[st] = *reinterpret_cast<const d2_t*>(arow + 8 * st);
    b[st] = *reinterpret_cast<const d2_t*>(brow + 8 * st);
  }
#pragma unroll
  for (int st = 0; st < NB / 8; ++st)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st][h], b[st][h], acc, 0, 0, 1);  // C - A B^T
#pragma unroll
  for (int r = 0; r < 4; ++r) cp[(int64_t)4 * r * g.ld] = acc[r];
}

// Tc(k) and Uc(k) in ONE launch (GPMI_CHAIN_TILES=2, the default): the update tile (i, j) of 16 x 16 needs the rows i and j of
// X = T invD^T - two 16 x 128 strips, 64 MFMAs for the eight waves of a workgroup - so every workgroup computes the strips
// of ITS tile itself instead of waiting a kernel boundary (2.3 us) and an L2 round trip for somebody else's: workgroups
// 0 .. 27 the off-diagonal tiles (two strips each), 28 .. 31 two diagonal tiles each (two strips as well): 32 equal
// workgroups, one per CU of the panel stream.  Strips and update take their k in exactly the groups of
// chain_trsm_kernel / chain_syrk_kernel (bit-identical).  X replaces T in place: the diagonal-pair workgroup of a strip
// stores it once all eight workgroups that read the strip's rows of T have them in registers (a counter per strip,
// monotonic over the columns; only those four workgroups ever wait, so the others always finish and free their CUs).
struct ChainFusedArgs {
  double* T;
  double* C;
  const double* invD;
  int64_t ld;
  int* rowcnt;       // 8 counters: workgroups that have loaded strip r of T, summed over the columns so far
  int rowcnt_target; // 8 x (columns so far, this one included)
  const int* wait2;  // second flag: tile (k+1, k+1) has taken the columns before k
  int wait2_val;
  FlowHook hook;
};
constexpr int CF_PITCH = NB + 2;   // X strips, row-major (update operands: 16-byte reads at (row fr, 8 st + 2 fk))
constexpr int CF_APITCH = NB + 4;  // T strips, row-major (TRSM operands: 32-byte reads at (row fr, 16 s + 4 fk))

__global__ __launch_bounds__(512) void chain_fused_kernel(ChainFusedArgs g) {
  __shared__ __attribute__((aligned(16))) double Ts[2][16 * CF_APITCH];
  __shared__ __attribute__((aligned(16))) double Xs[2][16 * CF_PITCH];
  // the hook of a chain launch (gemm_tiles.h: flow_hook_enter) with BOTH flags in one poll: tile (k+1, k) and tile
  // (k+1, k+1) have taken the columns before k (normally long set; one L2 round trip for the two)
  {
    const FlowHook& h = g.hook;
    if (h.trace && blockIdx.x == 0 && threadIdx.x == 0) h.trace[0] = __builtin_amdgcn_s_memrealtime();
    if (h.pub && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(h.pub, h.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      int spins = 0;
      unsigned long long t0 = 0;
      if (h.wait_ticks) t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const int v1 = flow_ld(h.wait), v2 = flow_ld(g.wait2);
        if (v1 >= h.wait_val && v2 >= g.wait2_val) break;
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 63) == 0 && flow_ld(h.abort)) break;
        if (spins > FLOW_SPIN_LIMIT) {
          __hip_atomic_store(h.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (h.info) atomicCAS(h.info, 0, GPMI_INFO_FLOW_TIMEOUT);
          break;
        }
      }
      if (h.wait_ticks) atomicAdd(h.wait_ticks, __builtin_amdgcn_s_memrealtime() - t0);
      if (h.trace && blockIdx.x == 0) h.trace[1] = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  const int tid = threadIdx.x, lane = tid & 63, c = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int fr = lane & 15, fk = lane >> 4;
  const int w = (int)blockIdx.x;
  int r0, r1;  // the two strips of this workgroup: rows 16 r0 .., 16 r1 ..
  const bool diag_pair = w >= 28;
  if (diag_pair) {
    r0 = 2 * (w - 28);
    r1 = r0 + 1;
  } else {
    r0 = 1;
    while (r0 * (r0 + 1) / 2 <= w) ++r0;  // off-diagonal tiles (i, j), j < i, row by row: w = i (i - 1) / 2 + j
    r1 = w - r0 * (r0 - 1) / 2;
  }
  // ---- operands.  T: wave c fetches slab c (columns 16 c ..) of both strips, once per workgroup, and hands it on through
  // LDS.  invD: wave c computes column block c of strip 0 and column block 7 - c of strip 1 - row blocks c and 7 - c of the
  // lower-triangular invD, nine slabs and 36 MFMAs for every wave (one block of both strips per wave: 8 .. 64 MFMAs)
  const int c1 = 7 - c;
  const d4_t t0 = *reinterpret_cast<const d4_t*>(g.T + (int64_t)(16 * r0 + fr) * g.ld + 16 * c + 4 * fk);
  const d4_t t1 = *reinterpret_cast<const d4_t*>(g.T + (int64_t)(16 * r1 + fr) * g.ld + 16 * c + 4 * fk);
  const double* b0row = g.invD + (int64_t)(16 * c + fr) * NB + 4 * fk;
  const double* b1row = g.invD + (int64_t)(16 * c1 + fr) * NB + 4 * fk;
  d4_t b0[NB / 16], b1[NB / 16];
#pragma unroll
  for (int s = 0; s < NB / 16; ++s) {
    if (s <= c) b0[s] = *reinterpret_cast<const d4_t*>(b0row + 16 * s);
    if (s <= c1) b1[s] = *reinterpret_cast<const d4_t*>(b1row + 16 * s);
  }
  // the update tile(s) of this workgroup: one wave each; its C tile is requested now and arrives under the strips
  const int ntile = diag_pair ? 2 : 1;
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  double* cp = nullptr;
  if (c < ntile) {
    const int ti = diag_pair ? (c == 0 ? r0 : r1) : r0, tj = diag_pair ? ti : r1;
    cp = g.C + (int64_t)(16 * ti + fk) * g.ld + 16 * tj + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cp[(int64_t)4 * r * g.ld];
  }
  *reinterpret_cast<d4_t*>(&Ts[0][fr * CF_APITCH + 16 * c + 4 * fk]) = t0;
  *reinterpret_cast<d4_t*>(&Ts[1][fr * CF_APITCH + 16 * c + 4 * fk]) = t1;
  __syncthreads();  // (every wave has its slab of T in registers and in LDS)
  if (tid == 0) {
    __hip_atomic_fetch_add(g.rowcnt + r0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(g.rowcnt + r1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- the strips (chain_trsm_kernel's sums: k = 16 s + 4 fk + q at step q of slab s, slabs beyond the block skipped)
  d4_t x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < NB / 16; ++s) {
    if (s <= c) {
      const d4_t a = *reinterpret_cast<const d4_t*>(&Ts[0][fr * CF_APITCH + 16 * s + 4 * fk]);
#pragma unroll
      for (int q = 0; q < 4; ++q) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b0[s][q], x0, 0, 0, 0);
    }
    if (s <= c1) {
      const d4_t a = *reinterpret_cast<const d4_t*>(&Ts[1][fr * CF_APITCH + 16 * s + 4 * fk]);
#pragma unroll
      for (int q = 0; q < 4; ++q) x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1[s][q], x1, 0, 0, 0);
    }
  }
  // X strips to LDS in row-major form (the D layout of the MFMA: lane (fr, fk) holds rows fk + 4 r of its column block)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Xs[0][(fk + 4 * r) * CF_PITCH + 16 * c + fr] = x0[r];
    Xs[1][(fk + 4 * r) * CF_PITCH + 16 * c1 + fr] = x1[r];
  }
  __syncthreads();
  // ---- the update tile(s): chain_syrk_kernel's sums (k = 8 st + 2 fk, + 1; A negated by the MFMA)
  if (c < ntile) {
    const double* xa = Xs[diag_pair ? c : 0] + fr * CF_PITCH + 2 * fk;
    const double* xb = Xs[diag_pair ? c : 1] + fr * CF_PITCH + 2 * fk;
#pragma unroll
    for (int st = 0; st < NB / 8; ++st) {
      const d2_t av = *reinterpret_cast<const d2_t*>(xa + 8 * st);
      const d2_t bv = *reinterpret_cast<const d2_t*>(xb + 8 * st);
#pragma unroll
      for (int h = 0; h < 2; ++h) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h], bv[h], acc, 0, 0, 1);  // C - A B^T
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[(int64_t)4 * r * g.ld] = acc[r];
  }
  if (g.hook.trace && w == 0 && tid == 0) g.hook.trace[2] = g.hook.trace[3] = __builtin_amdgcn_s_memrealtime();
  if (!diag_pair || c < ntile) return;
  // ---- X in place of T (the diagonal-pair workgroups, the six waves without a tile, from LDS): strips r0 and r1, once
  // nobody needs their rows of T any more
  {
    int spins = 0;
    while (flow_ld(g.rowcnt + r0) < g.rowcnt_target || flow_ld(g.rowcnt + r1) < g.rowcnt_target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63) == 0 && flow_ld(g.hook.abort)) break;
      if (spins > FLOW_SPIN_LIMIT) {
        if (lane == 0) {
          __hip_atomic_store(g.hook.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (g.hook.info) atomicCAS(g.hook.info, 0, GPMI_INFO_FLOW_TIMEOUT);
        }
        break;
      }
    }
  }
  for (int blk = c - ntile; blk < 16; blk += 8 - ntile) {  // block = (strip, column block)
    const int st = blk >> 3, cb = blk & 7;
    const double* xs = Xs[st] + fk * CF_PITCH + 16 * cb + fr;
    double* o = g.T + (int64_t)(16 * (st ? r1 : r0) + fk) * g.ld + 16 * cb + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(int64_t)4 * r * g.ld] = xs[4 * r * CF_PITCH];
  }
}

// ---- D(k), Tc(k) and Uc(k) in ONE launch (round 5, GPMI_CHAIN_TILES=3, the default) ---------------------------------------
// Until round 4 a column of the chain was potrf_diag, a kernel boundary, and the fused Tc / Uc launch, which could not start
// before the whole inverse of the diagonal block existed: 18.5 + 2 + 13 us.  But row block c of the inverse is final after
// step c of potrf_diag's eight (the inverse is built right-looking), column block c of X = T invD^T needs row block c of the
// inverse only (X_c = sum_{s <= c} T_s invD[c][s]^T), and the update C -= X X^T takes the column blocks of X in order.
// So the two products run BESIDE potrf_diag, one row block behind it:
//   workgroup 0      potrf_diag (potrf_diag.h, PUB build): every byte of the inverse stored write-through, row block r
//                    counted in at flags[FL_INVROW + 32 r] by the four inverse waves once their stores are acknowledged
//   workgroups 1..28 the 36 lower 16 x 16 tiles of the update: 24 off-diagonal tiles, one per workgroup (two strips of X
//                    each), and 4 workgroups with the tiles (2p, 2p), (2p+1, 2p+1), (2p+1, 2p) (the same two strips).
//                    T and C are fetched at the start (under potrf_diag); wave c computes column block c of strip 0 when
//                    row block c arrives and column block 7 - c of strip 1 when row block 7 - c does; the tile waves add
//                    -X_c X_c^T for c = 0, 1, ... as the blocks land in LDS (per-block LDS counters, no workgroup barrier
//                    after the prologue).  The four pair workgroups also store X over T (strips 2p, 2p + 1) once every
//                    reader of those rows has them (FL_ROWCNT, seven readers per strip).
// Behind potrf_diag's last elimination there is left: the last row block of the inverse (one MFMA chain of 4 + its stores'
// acknowledgement), the hand-off, 32 MFMAs of the last column block, 4 of the update, the C store: ~6 us instead of 15.
// Same sums in the same order as chain_trsm_kernel / chain_syrk_kernel, i.e. as the generic tile bodies: bit-identical.
// Deadlock-free with any number of resident workgroups: workgroup 0 is dispatched first and waits for nobody in the
// launch; the others wait for workgroup 0 and - the pair workgroups, dispatched last - for workgroups before them.
struct ChainColArgs {
  double* Akk;          // tile (k, k); T = tile (k+1, k) and C = tile (k+1, k+1) follow from ld
  double* invD;         // inverse of diagonal block k
  int64_t ld;
  int* info;
  int col0;
  unsigned long long* dbg;  // potrf_diag's 24-word stamp slot (tools) or nullptr
  int* pub;                 // published at the start: Lcnt[k] = pub_val (the strip the launch before stored); nullptr for k = 0
  int pub_val;
  int* flags;               // the factorisation's flag block (FL_*)
  int k;                    // column of the tail
  const int* wait1;         // tile (k+1, k) has taken the columns before k
  const int* wait2;         // tile (k+1, k+1) likewise
  int wait_val;
  unsigned long long* wait_ticks;
  unsigned long long* trace;  // 4 words (worker 0): launch start, flags seen, last MFMA, end
};
constexpr int CC_WORKERS = 28;
constexpr int CC_READERS = 7;  // workgroups that load a strip of T: 6 off-diagonal tiles + the strip's pair workgroup

__global__ __launch_bounds__(512) void chain_column_kernel(ChainColArgs g) {
  __shared__ potrf_diag::DiagShared sh;
  int* fl = g.flags;
  if (blockIdx.x == 0) {
    if (g.pub && threadIdx.x == 0) __hip_atomic_store(g.pub, g.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    potrf_diag::DiagPub pub;
    pub.rows = fl + FL_INVROW;
    pub.base = potrf_diag::DIAG_INVERSE * g.k;
    pub.ddone = fl + FL_DDONE;
    pub.ddone_val = g.k + 1;
    potrf_diag::potrf_diag_body<true>(g.Akk, g.ld, g.invD, g.info, g.col0, g.dbg, sh, pub);
    return;
  }
  // ------------------------------------------------------------------------------------------------ a worker
  const int w = (int)blockIdx.x - 1;
  const int tid = threadIdx.x, lane = tid & 63, c = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int fr = lane & 15, fk = lane >> 4;
  double* Ts = sh.S;                            // [2][16 * CF_APITCH]  strips of T, row-major
  double* Xs = sh.S + 2 * 16 * CF_APITCH;       // [2][16 * CF_PITCH]   strips of X, row-major
  int* xs_cnt = reinterpret_cast<int*>(sh.Xl);  // [8] strips whose column block cb is in Xs (0 .. 2)
  static_assert(2 * 16 * (CF_APITCH + CF_PITCH) <= potrf_diag::S_DOUBLES, "the worker's strips fit the block image");
  double* T = g.Akk + (int64_t)NB * g.ld;
  double* C = T + NB;
  int* abortw = fl + FL_ABORT;
  auto give_up = [&]() {
    if (lane == 0) {
      __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g.info) atomicCAS(g.info, 0, GPMI_INFO_FLOW_TIMEOUT);
    }
  };
  if (tid < 8) xs_cnt[tid] = 0;
  if (g.trace && w == 0 && tid == 0) g.trace[0] = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {  // both tiles have taken the columns before k (normally long set; one round trip for the two)
    int spins = 0;
    unsigned long long t0 = 0;
    if (g.wait_ticks) t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const int v1 = flow_ld(g.wait1), v2 = flow_ld(g.wait2);
      if (v1 >= g.wait_val && v2 >= g.wait_val) break;
      __builtin_amdgcn_s_sleep(4);
      if ((++spins & 63) == 0 && flow_ld(abortw)) break;
      if (spins > FLOW_SPIN_LIMIT) {
        give_up();
        break;
      }
    }
    if (g.wait_ticks) atomicAdd(g.wait_ticks, __builtin_amdgcn_s_memrealtime() - t0);
    if (g.trace && w == 0) g.trace[1] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // every wave acquires for itself (one buffer_inv each): the loads of T and C below must not rest on wave 0's
  // invalidation being CU-wide, which the memory model does not promise
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // the two strips of this workgroup (rows 16 r0 .., 16 r1 ..) and its tiles
  const bool pair = w >= 24;
  int r0, r1;
  if (pair) {
    r0 = 2 * (w - 24);
    r1 = r0 + 1;
  } else {
    // the 24 off-diagonal tiles (i, j), j < i, that no pair workgroup owns, row by row
    int n = w, i = 1, j = 0;
    for (;;) {
      const bool pair_tile = (i & 1) && j == i - 1;
      if (!pair_tile) {
        if (n == 0) break;
        --n;
      }
      if (++j == i) {
        ++i;
        j = 0;
      }
    }
    r0 = i;
    r1 = j;
  }
  r0 = __builtin_amdgcn_readfirstlane(r0);
  r1 = __builtin_amdgcn_readfirstlane(r1);
  // T: wave c fetches slab c (columns 16 c ..) of both strips, once per workgroup, and hands it on through LDS
  const d4_t t0 = *reinterpret_cast<const d4_t*>(T + (int64_t)(16 * r0 + fr) * g.ld + 16 * c + 4 * fk);
  const d4_t t1 = *reinterpret_cast<const d4_t*>(T + (int64_t)(16 * r1 + fr) * g.ld + 16 * c + 4 * fk);
  // tile waves: 3 (off-diagonal tile (r0, r1); in a pair workgroup tile (r0, r0)), 4: (r1, r1), 5: (r1, r0) - waves whose own
  // column blocks are in the middle of the sequence, so that nothing of theirs sits behind the last row block
  const bool tile_wave = pair ? (c >= 3 && c <= 5) : c == 3;
  int ti = r0, tj = r1;  // strips of the tile's A and B operand
  if (pair) {
    ti = c == 3 ? r0 : r1;
    tj = c == 4 ? r1 : r0;
  }
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  double* cp = C + (int64_t)(16 * ti + fk) * g.ld + 16 * tj + fr;
  if (tile_wave) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cp[(int64_t)4 * r * g.ld];
  }
  *reinterpret_cast<d4_t*>(&Ts[fr * CF_APITCH + 16 * c + 4 * fk]) = t0;
  *reinterpret_cast<d4_t*>(&Ts[16 * CF_APITCH + fr * CF_APITCH + 16 * c + 4 * fk]) = t1;
  __syncthreads();  // (every wave has its slab of T in registers and in LDS; the LDS counters are zero)
  if (tid == 0) {
    __hip_atomic_fetch_add(fl + FL_ROWCNT + r0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(fl + FL_ROWCNT + r1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int row_target = potrf_diag::DIAG_INVERSE * (g.k + 1);
  const int readers_target = CC_READERS * (g.k + 1);
  const double* xa = Xs + (pair ? (ti == r0 ? 0 : 1) : 0) * 16 * CF_PITCH + fr * CF_PITCH + 2 * fk;
  const double* xb = Xs + (pair ? (tj == r0 ? 0 : 1) : 1) * 16 * CF_PITCH + fr * CF_PITCH + 2 * fk;
  bool readers_seen = false;
#pragma unroll
  for (int row = 0; row < NB / 16; ++row) {
    // ---- this wave's column block of a strip, if row block `row` of the inverse is its operand
    if (row == c || row == 7 - c) {
      const int strip = row == c ? 0 : 1;
      {
        int spins = 0;
        while (flow_ld(fl + FL_INVROW + 32 * row) < row_target) {
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 63) == 0 && flow_ld(abortw)) break;
          if (spins > FLOW_SPIN_LIMIT) {
            give_up();
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      const double* brow = g.invD + (int64_t)(16 * row + fr) * NB + 4 * fk;
      d4_t b[NB / 16];
#pragma unroll
      for (int s = 0; s <= row; ++s) b[s] = *reinterpret_cast<const d4_t*>(brow + 16 * s);
      // (chain_trsm_kernel's sums: k = 16 s + 4 fk + q at step q of slab s, slabs beyond the block skipped)
      d4_t x = {0.0, 0.0, 0.0, 0.0};
      const double* ts = Ts + strip * 16 * CF_APITCH + fr * CF_APITCH + 4 * fk;
#pragma unroll
      for (int s = 0; s <= row; ++s) {
        const d4_t a = *reinterpret_cast<const d4_t*>(ts + 16 * s);
#pragma unroll
        for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[s][q], x, 0, 0, 0);
      }
      // X block to LDS in row-major form (the D layout of the MFMA: lane (fr, fk) holds rows fk + 4 r of its column block)
      double* xo = Xs + strip * 16 * CF_PITCH + fk * CF_PITCH + 16 * row + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) xo[4 * r * CF_PITCH] = x[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(xs_cnt + row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (pair) {
        // X in place of T, once nobody needs the strip's rows of T any more (long the case: T is fetched under potrf_diag)
        if (!readers_seen) {
          int spins = 0;
          while (flow_ld(fl + FL_ROWCNT + r0) < readers_target || flow_ld(fl + FL_ROWCNT + r1) < readers_target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 63) == 0 && flow_ld(abortw)) break;
            if (spins > FLOW_SPIN_LIMIT) {
              give_up();
              break;
            }
          }
          readers_seen = true;
        }
        double* o = T + (int64_t)(16 * (strip ? r1 : r0) + fk) * g.ld + 16 * row + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(int64_t)4 * r * g.ld] = x[r];
      }
    }
    // ---- the tile waves: column block `row` of both strips (chain_syrk_kernel's sums: k = 8 st + 2 fk, + 1; A negated
    // by the MFMA), as soon as the two blocks are in LDS
    if (tile_wave) {
      int polls = 0;
      while (__hip_atomic_load(xs_cnt + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023) == 0 && flow_ld(abortw)) break;
        if (polls > (1 << 24)) {
          give_up();
          break;
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int st = 2 * row; st < 2 * row + 2; ++st) {
        const d2_t av = *reinterpret_cast<const d2_t*>(xa + 8 * st);
        const d2_t bv = *reinterpret_cast<const d2_t*>(xb + 8 * st);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h], bv[h], acc, 0, 0, 1);  // C - A B^T
      }
    }
  }
  if (tile_wave) {
    if (g.trace && w == 0) g.trace[2] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[(int64_t)4 * r * g.ld] = acc[r];
    if (g.trace && w == 0 && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g.trace[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

template <int PROTO>
__global__ __launch_bounds__(256, 2) void flow_task_kernel(FlowArgs a) {
  __shared__ double smem[DMA128_LDS_DOUBLES];
  __shared__ int sh[8];
  static_assert(staged_lds_doubles<0, 32, 128>() <= DMA128_LDS_DOUBLES, "LDS of the TRSM slab body");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* fl = a.flags;
  int* Lcnt = fl + FL_LCNT;
  int* F = fl + flow_f_off(a.m);
  if (a.stamp && tid == 0 && blockIdx.x < 8) a.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  unsigned long long scan_ticks = 0, body_ticks = 0;  // GPMI_FLOW_STATS (wave 0)
  int nscan = 0, ntask = 0;
  // Lists.  Workgroup b claims list b (CAS on the list's owner word) and normally runs just that.  A list nobody has
  // claimed 50 us after the launch began is an orphan - its workgroup is not resident: another launch (a second process
  // on this device) holds the slots - and is adopted by whichever workgroup finds it while it has nothing to do; a
  // workgroup holding several lists runs whichever head task is ready.  So the launch needs ONE resident workgroup to
  // finish, not all of them, and a workgroup that arrives after its list was adopted just leaves.
  __shared__ int s_own[FLOW_MAX_LISTS], s_cur[FLOW_MAX_LISTS];
  int* owner = fl + flow_owner_off(a.m);
  const int nlists = a.nlists;
  int nown = 0, last_slot = -1, n = 0;
  if (wave == 0) {
    if (lane == 0) {
      unsigned long long* t0p = reinterpret_cast<unsigned long long*>(fl + FL_T0);
      atomicCAS(t0p, 0ull, (unsigned long long)__builtin_amdgcn_s_memrealtime());
      // own lists, in priority order: a ready head of an earlier slot is taken first
      for (int l = (int)blockIdx.x; l < nlists; l += (int)gridDim.x)
        if (atomicCAS(owner + l, 0, (int)blockIdx.x + 1) == 0) {
          s_own[nown] = l;
          s_cur[nown] = a.off[l];
          ++nown;
        }
    }
    nown = __shfl(nown, 0, 64);
  }
  auto try_adopt = [&]() {  // wave 0
    const unsigned long long t0 = __hip_atomic_load(reinterpret_cast<unsigned long long*>(fl + FL_T0), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_s_memrealtime() - t0 < FLOW_GRACE_TICKS || nown >= FLOW_MAX_LISTS) return false;
    for (int r = 0; r < nlists; r += 64) {
      const int idx = ((int)blockIdx.x + 1 + r + lane) % nlists;
      const bool un = (r + lane < nlists) && flow_ld(owner + idx) == 0;
      const unsigned long long ub = __ballot(un);
      if (ub == 0ull) continue;
      const int sel = __ffsll((long long)ub) - 1;
      int ok = 0;
      if (lane == sel) {
        ok = atomicCAS(owner + idx, 0, (int)blockIdx.x + 1) == 0;
        if (ok) {
          s_own[nown] = idx;
          s_cur[nown] = a.off[idx];
        }
      }
      ok = __shfl(ok, sel, 64);
      if (ok) {
        ++nown;
        return true;
      }
    }
    return false;
  };
  auto any_unowned = [&]() {  // wave 0
    for (int r = 0; r < nlists; r += 64) {
      const bool un = (r + lane < nlists) && flow_ld(owner + r + lane) == 0;
      if (__ballot(un) != 0ull) return true;
    }
    return false;
  };
  for (;;) {
    if (wave == 0) {
      int got = 0;
      if (last_slot >= 0 && lane == 0) s_cur[last_slot] += 1;
      unsigned long long t_scan = 0;
      if (a.stats || a.trace) t_scan = __builtin_amdgcn_s_memrealtime();
      int spins = 0;
      for (;;) {
        // a ready head among the lists this workgroup holds (lane l looks at list slot l, 64 at a time)
        bool all_done = true;
        for (int base = 0; base < nown && got == 0; base += 64) {
          const int slot = base + lane;
          bool rdy = false;
          FlowTask mine{};
          int c = 0;
          if (slot < nown) {
            c = s_cur[slot];
            if (c < a.off[s_own[slot] + 1]) {
              all_done = false;
              mine = a.tasks[c];
              rdy = flow_ready(mine, fl, a.m);
            }
          }
          const unsigned long long rb = __ballot(rdy);
          if (rb != 0ull) {
            const int sel = __ffsll((long long)rb) - 1;
            const int w0 = __shfl(*reinterpret_cast<const int*>(&mine), sel, 64);
            const int w1 = __shfl(*(reinterpret_cast<const int*>(&mine) + 1), sel, 64);
            const int w2 = __shfl(*(reinterpret_cast<const int*>(&mine) + 2), sel, 64);
            n = __shfl(c, sel, 64);
            last_slot = base + sel;
            if (lane == 0) {
              sh[1] = w0;
              sh[2] = w1;
              sh[3] = w2;
            }
            got = 1;
          }
        }
        if (got != 0) break;
        if (__ballot(!all_done) == 0ull) {
          // every list held is finished: adopt an orphan, or leave once every list has an owner
          last_slot = -1;
          if (try_adopt()) continue;
          if (!any_unowned()) {
            got = -1;
            break;
          }
        } else if ((spins & 15) == 15) {
          try_adopt();  // the task waited for may belong to an orphan list
        }
        __builtin_amdgcn_s_sleep(8);
        ++spins;
        if ((spins & 63) == 0 && flow_ld(fl + FL_ABORT)) {
          got = -1;
          break;
        }
        if (spins > FLOW_SPIN_LIMIT) {
          if (lane == 0) {
            __hip_atomic_store(fl + FL_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.info) atomicCAS(a.info, 0, GPMI_INFO_FLOW_TIMEOUT);
          }
          got = -1;
          break;
        }
      }
      nscan += spins + 1;
      if (a.stats) scan_ticks += __builtin_amdgcn_s_memrealtime() - t_scan;
      if (got > 0 && a.trace && lane == 0) {
        a.trace[4 * (int64_t)n] = t_scan;
        a.trace[4 * (int64_t)n + 1] = __builtin_amdgcn_s_memrealtime();
      }
      if (lane == 0) {
        sh[0] = got;
        sh[4] = n;
      }
      if (got > 0) {
        // the inputs were written by other CUs: drop what this CU's L1 holds of them
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (sh[0] < 0) break;
    n = sh[4];
    FlowTask t;
    {
      int* w = reinterpret_cast<int*>(&t);
      w[0] = sh[1];
      w[1] = sh[2];
      w[2] = sh[3];
    }
    int* flag;
    const int64_t ld = a.ld;
    unsigned long long t_body = 0;
    if (a.stats) t_body = __builtin_amdgcn_s_memrealtime();
    if (t.type == FT_T) {
      // rows [32 s, 32 s + 32) of tile (i, k) <- the same rows times invD(k)^T, in place
      double* C = a.A + ((int64_t)t.i * NB + 32 * t.s) * ld + (int64_t)t.k * NB;
      flow_do_T<PROTO>((glb_double_t*)C, (const glb_double_t*)(a.invD + (int64_t)t.k * NB * NB), ld, (lds_double_t*)smem);
      flag = Lcnt + t.i;
    } else if (t.type == FT_U || t.type == FT_ZS) {
      // one column k (K = 128), or - FT_ZS - the four columns of outer panel k (K = 512), on a 64 x 64 sub-tile: the
      // ring body sums k in the order of the 128 x 128 body, so a chunk in quarters leaves the same bits
      const int r0 = 64 * (t.s >> 1), c0 = 64 * (t.s & 1);
      const int64_t kcol = t.type == FT_U ? (int64_t)t.k * NB : (int64_t)t.k * OBT * NB;
      double* C = a.A + ((int64_t)t.i * NB + r0) * ld + (int64_t)t.j * NB + c0;
      const double* Ai = a.A + ((int64_t)t.i * NB + r0) * ld + kcol;
      const double* Bj = a.A + ((int64_t)t.j * NB + c0) * ld + kcol;
      flow_do_U<PROTO>((const glb_double_t*)Ai, (const glb_double_t*)Bj, (glb_double_t*)C, ld, (lds_double_t*)smem,
                       t.type == FT_U ? NB / DMA_BK : OBT * NB / DMA_BK);
      flag = F + t.i * a.m + t.j;
    } else {
      double* C = a.A + (int64_t)t.i * NB * ld + (int64_t)t.j * NB;
      const double* Ai = a.A + (int64_t)t.i * NB * ld + (int64_t)t.k * OBT * NB;
      const double* Bj = a.A + (int64_t)t.j * NB * ld + (int64_t)t.k * OBT * NB;
      dma128_tile<OP_SUB, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN>(Ai, Bj, C, ld, ld, ld, OBT * NB / DMA_BK, smem);
      flag = F + t.i * a.m + t.j;
    }
    // publish: the tile was stored write-through (sc1); every wave's stores acknowledged, workgroup barrier, flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.trace && tid == 0) a.trace[4 * (int64_t)n + 2] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      if (PROTO == 1 || PROTO == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (n != a.fault) __hip_atomic_fetch_add(flag, (int)t.fadd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.trace) a.trace[4 * (int64_t)n + 3] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.stats) {
      body_ticks += __builtin_amdgcn_s_memrealtime() - t_body;
      ++ntask;
    }
  }
  if (a.stats && tid == 0) {
    atomicAdd(a.stats + 0, scan_ticks);
    atomicAdd(a.stats + 1, body_ticks);
    atomicAdd(a.stats + 2, (unsigned long long)nscan);
    atomicAdd(a.stats + 3, (unsigned long long)ntask);
  }
  if (a.stamp && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
}

// Task lists of a tail of m tile rows (tools/sim/flow_sim.py: build()).  Tile (i, j), P = j / 4: the columns of the
// outer panels before `lazy` are applied as K = 512 chunks, the columns [4 lazy, j) one at a time; lazy = P - 1 for
// the tiles within NEAR rows below their own panel's diagonal block (the K = 512 chunk of the panel just finished
// would sit on the chain: potrf_diag(4P) waits for it), lazy = P elsewhere.
const int FLOW_NEAR = env_int("GPMI_FLOW_NEAR", 4);
inline int flow_lazy_panels(int i, int j) {
  const int P = j / OBT;
  const int e = (i < OBT * P + OBT + FLOW_NEAR) ? P - 1 : P;
  return e > 0 ? e : 0;
}

struct FlowLists {
  std::vector<FlowTask> tasks;
  std::vector<int> off;
  double flops_update = 0.0, flops_trsm = 0.0;
};

FlowLists flow_build(int m, int nwg) {
  FlowLists out;
  typedef std::tuple<int, int, int, int> Key;  // (4 x virtual time, i, j, s)
  std::vector<std::pair<Key, FlowTask>> near, rest;
  for (int k = 0; k < m; ++k)
    for (int i = k + 2; i < m; ++i) {
      auto& dst = (i - k <= FLOW_NEAR_D) ? near : rest;
      for (int s = 0; s < 4; ++s) {
        FlowTask t{(uint8_t)FT_T, (uint8_t)s, 1, 0, (uint16_t)i, 0, (uint16_t)k, 0};
        dst.push_back({Key(4 * k, i, 0, s), t});
      }
      out.flops_trsm += (double)NB * NB * NB;
      for (int j = k + 1; j <= i; ++j) {
        if (OBT * flow_lazy_panels(i, j) > k) continue;
        // a diagonal tile keeps only its lower triangle current: the upper-right 64 x 64 sub-tile is skipped (as the
        // lower-tile launches of potrf_lower do) and sub-tile 0 counts for two
        for (int s = 0; s < 4; ++s) {
          if (i == j && s == 1) continue;
          FlowTask t{(uint8_t)FT_U, (uint8_t)s, (uint8_t)((i == j && s == 0) ? 2 : 1), 0, (uint16_t)i, (uint16_t)j,
                     (uint16_t)k, 0};
          dst.push_back({Key(4 * k + 2, i, j, s), t});
        }
        out.flops_update += (i == j ? 0.75 : 1.0) * 2.0 * NB * NB * NB;
      }
    }
  auto by_key = [](const std::pair<Key, FlowTask>& x, const std::pair<Key, FlowTask>& y) { return x.first < y.first; };
  std::sort(near.begin(), near.end(), by_key);
  std::sort(rest.begin(), rest.end(), by_key);
  const int nn = (nwg > 2 * FLOW_NEAR_WGS) ? FLOW_NEAR_WGS : 0;  // a small launch: one deal for everything
  std::vector<std::vector<std::pair<Key, FlowTask>>> lists((size_t)nwg);
  if (nn == 0) {
    rest.insert(rest.end(), near.begin(), near.end());
    std::sort(rest.begin(), rest.end(), by_key);
    near.clear();
  }
  for (size_t n = 0; n < near.size(); ++n) lists[n % (size_t)nn].push_back(near[n]);
  for (size_t n = 0; n < rest.size(); ++n) lists[(size_t)nn + n % (size_t)(nwg - nn)].push_back(rest[n]);
  // The K = 512 tiles: chunk after chunk, rows from the top (the rows the chain reaches first), dealt round-robin like
  // the rest.  (Dealt for L2 locality instead - column strips of 8, blocks of 64 consecutive tiles to the workgroups
  // of one XCD, as the launch-per-product kernels order their tiles - the tiles were no faster, 117 us with two on a CU
  // either way, and the chain waited 0.8 ms longer for the rows near the diagonal: 7.0 against 6.1 ms at N = 8192.)
  {
    std::vector<std::pair<Key, FlowTask>> zt, zu;  // the chunks; the urgent ones (GPMI_FLOW_URGENT)
    const bool split = FLOW_SPLIT && nn > 0;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j <= i; ++j) {
        const int lazy = flow_lazy_panels(i, j);
        for (int q = 0; q < lazy; ++q) {
          const bool urgent = split && FLOW_URGENT > 0 && lazy - q <= FLOW_URGENT;
          auto& dst = urgent ? zu : zt;
          // as soon as possible: right behind T(., 4 q + 3); as late as possible: right in front of the tile's first
          // single-column task T / U(., ., 4 lazy) - both positions respect every dependency of the chunk
          const int key = urgent ? 4 * (OBT * lazy) - 1 : 4 * (OBT * q + OBT - 1) + 1;
          const int major = urgent ? q * 4096 + i : i;
          if (q >= FLOW_QUARTER) {
            for (int s = 0; s < 4; ++s) {
              if (i == j && s == 1) continue;  // (as the one-column tasks: the lower triangle of a diagonal tile)
              FlowTask t{(uint8_t)FT_ZS, (uint8_t)s, (uint8_t)((i == j && s == 0) ? 2 * OBT : OBT), 0, (uint16_t)i,
                         (uint16_t)j, (uint16_t)q, 0};
              dst.push_back({Key(key, major, j, s), t});
            }
            out.flops_update += (i == j ? 0.75 : 1.0) * 2.0 * NB * NB * NB * OBT;
          } else {
            FlowTask t{(uint8_t)FT_Z, 0, (uint8_t)(4 * OBT), 0, (uint16_t)i, (uint16_t)j, (uint16_t)q, 0};
            dst.push_back({Key(key, major, j, 0), t});
            out.flops_update += 2.0 * NB * NB * NB * OBT;
          }
        }
      }
    std::sort(zt.begin(), zt.end(), by_key);
    std::sort(zu.begin(), zu.end(), by_key);
    if (split) {
      // further lists: list c nwg + b belongs to workgroup b (c = 1: urgent chunks, if any; then the rest); the near
      // workgroups take no chunks, as before
      const size_t classes = zu.empty() ? 2 : 3;
      lists.resize(classes * (size_t)nwg);
      const size_t c0 = (size_t)nn, cw = (size_t)(nwg - nn);  // the workgroups that take chunks
      for (size_t n = 0; n < zu.size(); ++n) lists[(size_t)nwg + c0 + n % cw].push_back(zu[n]);
      for (size_t n = 0; n < zt.size(); ++n) lists[(classes - 1) * (size_t)nwg + c0 + n % cw].push_back(zt[n]);
    } else {
      for (size_t n = 0; n < zt.size(); ++n) lists[(size_t)nn + (n + rest.size()) % (size_t)(nwg - nn)].push_back(zt[n]);
    }
  }
  for (auto& l : lists) std::stable_sort(l.begin(), l.end(), by_key);
  out.off.push_back(0);
  for (auto& l : lists) {
    for (auto& e : l) out.tasks.push_back(e.second);
    out.off.push_back((int)out.tasks.size());
  }
  return out;
}

// GPMI_FLOW_LISTS=<file prefix> (experiments, tools/sim/flow_sched.py --write): task lists built outside the library for
// a tail of m tile rows, read from <prefix>_m<m>.bin = int32 {m, nwg, ntasks}, int32 off[nwg + 1], FlowTask[ntasks].  The
// same tasks as flow_build's in another deal and order (checked: a permutation); a list order that is not admissible ends
// in the kernel's bounded polls, not in a hang.
void flow_lists_override(int m, int nwg, FlowLists& fl) {
  static const char* prefix = std::getenv("GPMI_FLOW_LISTS");
  if (!prefix) return;
  char path[1024];
  std::snprintf(path, sizeof(path), "%s_m%d.bin", prefix, m);
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return;
  int32_t hdr[3] = {0, 0, 0};
  std::vector<int> off;
  std::vector<FlowTask> tasks;
  bool ok = std::fread(hdr, sizeof(hdr), 1, fp) == 1 && hdr[0] == m && hdr[1] == nwg &&
            hdr[2] == (int32_t)fl.tasks.size() && fl.off.size() == (size_t)nwg + 1;  // (not with GPMI_FLOW_SPLIT)
  if (ok) {
    off.resize((size_t)nwg + 1);
    tasks.resize((size_t)hdr[2]);
    ok = std::fread(off.data(), sizeof(int), off.size(), fp) == off.size() &&
         std::fread(tasks.data(), sizeof(FlowTask), tasks.size(), fp) == tasks.size() && off[0] == 0 &&
         off[(size_t)nwg] == hdr[2];
    // every list a range inside the task array: the kernel indexes tasks[off[w] .. off[w + 1]) unchecked
    for (size_t w = 0; ok && w < (size_t)nwg; ++w) ok = off[w] >= 0 && off[w] <= off[w + 1] && off[w + 1] <= hdr[2];
  }
  std::fclose(fp);
  if (ok) {  // a permutation of the library's own tasks
    auto key = [](const FlowTask& t) {
      return std::make_tuple((int)t.type, (int)t.i, (int)t.j, (int)t.k, (int)t.s, (int)t.fadd);
    };
    std::vector<std::tuple<int, int, int, int, int, int>> a, b;
    for (auto& t : fl.tasks) a.push_back(key(t));
    for (auto& t : tasks) b.push_back(key(t));
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    ok = a == b;
  }
  if (!ok) {
    std::fprintf(stderr, "[flow] %s does not hold the task lists of m = %d, %d workgroups: ignored\n", path, m, nwg);
    return;
  }
  fl.tasks.swap(tasks);
  fl.off.swap(off);
  static bool told = false;
  if (!told) std::fprintf(stderr, "[flow] task lists from %s\n", path);
  told = true;
}

}  // namespace

// Host-only view of the task lists (no device call): what tests/test_flow_cpu.py replays with NumPy tile operations.
// out: 8 ints per task {type (0 T, 1 U, 2 Z), i, j, k (column, or outer panel for Z), s, fadd, owner, 0}, list after list.
extern "C" int gpmi_flow_task_lists(int m, int nwg, int64_t cap, int32_t* out, int64_t* ntasks) {
  if (m < 1 || nwg < 1 || !ntasks) return GPMI_ERR_ARG;
  FlowLists fl = flow_build(m, nwg);
  flow_lists_override(m, nwg, fl);  // (GPMI_FLOW_LISTS: what the device would be given)
  *ntasks = (int64_t)fl.tasks.size();
  if (!out) return GPMI_OK;
  if (cap < *ntasks) return GPMI_ERR_ARG;
  for (int w = 0; w + 1 < (int)fl.off.size(); ++w)  // (w: the list; with GPMI_FLOW_SPLIT list nwg + b is workgroup b's second)
    for (int n = fl.off[(size_t)w]; n < fl.off[(size_t)w + 1]; ++n) {
      const FlowTask& t = fl.tasks[(size_t)n];
      int32_t* o = out + 8 * (int64_t)n;
      o[0] = t.type;
      o[1] = t.i;
      o[2] = t.j;
      o[3] = t.k;
      o[4] = t.s;
      o[5] = t.fadd;
      o[6] = w;
      o[7] = 0;
    }
  return GPMI_OK;
}

bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k);  // api.hip

// One flag-ordered factorisation per device at a time, process-wide: a workgroup of the task kernel may wait for a task
// that belongs to a workgroup of the same launch which is not resident yet; with the CUs to itself every workgroup
// becomes resident, but two such launches on the same CUs (two contexts, two threads) could each hold the slots the
// other one's missing workgroups need.  A second factorisation that arrives while the event of the previous one has not
// completed takes the stream-ordered schedule instead.
namespace {
std::mutex g_flow_mu;
hipEvent_t g_flow_ev[64] = {nullptr};
// state of a device's gate: FREE, ENQUEUING (a caller is between flow_gate_try and flow_gate_leave: the event does not
// describe its launch yet, so nobody may query it), RECORDED (the event marks the end of the launch in flight)
enum FlowGate { GATE_FREE = 0, GATE_ENQUEUING, GATE_RECORDED };
int g_flow_state[64] = {GATE_FREE};

bool flow_gate_try(int device) {
  if (device < 0 || device >= 64) return false;
  std::lock_guard<std::mutex> lk(g_flow_mu);
  if (g_flow_state[device] == GATE_ENQUEUING) return false;
  if (g_flow_state[device] == GATE_RECORDED) {
    if (hipEventQuery(g_flow_ev[device]) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    g_flow_state[device] = GATE_FREE;
  }
  if (!g_flow_ev[device] && hipEventCreateWithFlags(&g_flow_ev[device], hipEventDisableTiming) != hipSuccess) {
    g_flow_ev[device] = nullptr;
    (void)hipGetLastError();
    return false;
  }
  g_flow_state[device] = GATE_ENQUEUING;  // until flow_gate_leave has recorded the event behind the launch
  return true;
}
// `launched` = false: nothing was enqueued (set-up failed), the gate is free again
void flow_gate_leave(int device, hipStream_t s, bool launched = true) {
  std::lock_guard<std::mutex> lk(g_flow_mu);
  if (launched && hipEventRecord(g_flow_ev[device], s) == hipSuccess) {
    g_flow_state[device] = GATE_RECORDED;
  } else {
    (void)hipGetLastError();
    g_flow_state[device] = GATE_FREE;
  }
}
}  // namespace

// GPMI_FLOW=0 keeps the stream-ordered schedule everywhere; GPMI_FLOW_MIN=<tile rows> (default 8) is the smallest
// tail worth a task kernel
bool potrf_flow_enabled(gpmi_ctx* c, Lane& lane, int m) {
  static const bool on = [] {
    const char* e = std::getenv("GPMI_FLOW");
    return !e || std::atoi(e) != 0;
  }();
  static const int min_rows = [] {
    const char* e = std::getenv("GPMI_FLOW_MIN");
    return e ? std::atoi(e) : 8;
  }();
  // the task kernel keeps a 64 KiB operand ring + 4 KiB of list state in static LDS and counts on two workgroups per CU
  // (gfx950: 160 KiB per CU); a part with less takes the stream-ordered schedule
  static const bool lds_ok = [&] {
    int per_block = 0;
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    return per_block >= 72 * 1024;
  }();
  return on && lds_ok && !c->no_flow && m >= min_rows && m < 4096 && ensure_masked_pair(c, lane, 0);
}

void potrf_flow_free(Lane& lane) {
  if (lane.flow_tasks) (void)hipFree(lane.flow_tasks);
  if (lane.flow_off) (void)hipFree(lane.flow_off);
  if (lane.flow_flags) (void)hipFree(lane.flow_flags);
  lane.flow_tasks = nullptr;
  lane.flow_off = nullptr;
  lane.flow_flags = nullptr;
  lane.flow_m = 0;
  lane.flow_nwg = 0;
}

// Factor the trailing tile rows [t0, nt) of A (every update by the columns before t0 applied) in place.  The caller
// has ordered lane.stream behind whatever produced that state; on return lane.stream is ordered behind the factor.
// Returns false (nothing enqueued) if the task lists cannot be set up.
bool potrf_flow_tail(gpmi_ctx* c, Lane& lane, double* A, int64_t ld, double* invD, int* info, int nt, int t0) {
  const int m = nt - t0;
  hipStream_t sf = lane.stream, sp = lane.sp[0], su = lane.su[0];
  const int ncu_u = c->ncu - c->pair_cus[0];
  static const int wgs_per_cu = [] {
    const char* e = std::getenv("GPMI_FLOW_WGS");
    const int v = e ? std::atoi(e) : 2;
    return (v == 1 || v == 2) ? v : 2;
  }();
  // (a lone resident workgroup must be able to adopt every list: never more lists than FLOW_MAX_LISTS)
  const int nwg = wgs_per_cu * ncu_u < FLOW_MAX_WGS ? wgs_per_cu * ncu_u : FLOW_MAX_WGS;
  if (!flow_gate_try(c->device)) return false;
  if (lane.flow_m != m || lane.flow_nwg != nwg) {
    potrf_flow_free(lane);
    FlowLists fl = flow_build(m, nwg);
    flow_lists_override(m, nwg, fl);
    const size_t total = fl.tasks.size();
    if (hipMalloc(&lane.flow_tasks, sizeof(FlowTask) * (total ? total : 1)) != hipSuccess ||
        hipMalloc(&lane.flow_off, sizeof(int) * fl.off.size()) != hipSuccess ||
        hipMalloc(&lane.flow_flags, sizeof(int) * flow_flag_ints(m, (int)fl.off.size() - 1)) != hipSuccess ||
        (total && hipMemcpy(lane.flow_tasks, fl.tasks.data(), sizeof(FlowTask) * total, hipMemcpyHostToDevice) !=
                      hipSuccess) ||
        hipMemcpy(lane.flow_off, fl.off.data(), sizeof(int) * fl.off.size(), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      potrf_flow_free(lane);
      flow_gate_leave(c->device, sf, false);
      return false;
    }
    lane.flow_flops_update = fl.flops_update;
    lane.flow_flops_trsm = fl.flops_trsm;
    lane.flow_ntasks = (int64_t)total;
    lane.flow_m = m;
    lane.flow_nwg = nwg;
    lane.flow_nlists = (int)fl.off.size() - 1;
  }
  const int nlists = lane.flow_nlists;
  int* fl = lane.flow_flags;
  (void)hipMemsetAsync(fl, 0, sizeof(int) * flow_flag_ints(m, nlists), sf);
  (void)hipEventRecord(lane.ev_join, sf);
  (void)hipStreamWaitEvent(sp, lane.ev_join, 0);
  (void)hipStreamWaitEvent(su, lane.ev_join, 0);

  double* A0 = A + (int64_t)t0 * NB * ld + (int64_t)t0 * NB;
  double* invD0 = invD + (int64_t)t0 * NB * NB;
  FlowArgs fa{};
  fa.A = A0;
  fa.invD = invD0;
  fa.ld = ld;
  fa.m = m;
  fa.tasks = static_cast<const FlowTask*>(lane.flow_tasks);
  fa.off = lane.flow_off;
  fa.nlists = nlists;
  fa.flags = fl;
  fa.info = info;
  // one stamped "launch" of its own class for the bench: FLOPs of every U and Z task over the launch's whole duration
  fa.stamp = prof_stamp_slot(c, lane.flow_flops_update, 0.0, GPMI_PROF_FLOW);
  // GPMI_FLOW_TRACE=<file>: time stamps of every task and chain launch, written after a (synchronous) factorisation:
  // tools/flow_trace.py reads them
  static const char* trace_path = std::getenv("GPMI_FLOW_TRACE");
  const int64_t ntasks = lane.flow_ntasks;
  const int64_t trace_words = 4 * ntasks + (int64_t)m * (GPMI_STAMP_WORDS + 4);
  unsigned long long* trace = nullptr;
  if (trace_path && hipMalloc(&trace, sizeof(unsigned long long) * trace_words) == hipSuccess)
    (void)hipMemsetAsync(trace, 0, sizeof(unsigned long long) * trace_words, sf);
  fa.trace = trace;
  unsigned long long* ctrace = trace ? trace + 4 * ntasks : nullptr;  // per column: potrf_diag's 24 words, Tc 2, Uc 2
  static const bool want_stats = std::getenv("GPMI_FLOW_STATS") != nullptr;
  unsigned long long* stats = want_stats ? reinterpret_cast<unsigned long long*>(fl + FL_STATS) : nullptr;
  fa.stats = stats;
  // GPMI_FLOW_FAULT=<n>: the suite's own check that a lost flag ends in GPMI_ERR_INTERNAL, not in a hung GPU
  static const int fault = env_int("GPMI_FLOW_FAULT", -1);
  fa.fault = fault;
  if (trace) {  // the chain and the task kernel start behind the trace buffer's memset too
    (void)hipEventRecord(lane.ev_join, sf);
    (void)hipStreamWaitEvent(sp, lane.ev_join, 0);
    (void)hipStreamWaitEvent(su, lane.ev_join, 0);
  }
  lane_run_early(lane, sf);  // (sf idles until the join: the alpha phase's factor-independent launches, Lane::EarlyWork)
  static const int proto = [] {
    const char* e = std::getenv("GPMI_FLOW_PROTO");
    return e ? std::atoi(e) : 0;
  }();
  const dim3 grid((unsigned)nwg);
  if (proto == 1) hipLaunchKernelGGL(flow_task_kernel<1>, grid, dim3(256), 0, su, fa);
  else if (proto == 2) hipLaunchKernelGGL(flow_task_kernel<2>, grid, dim3(256), 0, su, fa);
  else if (proto == 3) hipLaunchKernelGGL(flow_task_kernel<3>, grid, dim3(256), 0, su, fa);
  else hipLaunchKernelGGL(flow_task_kernel<0>, grid, dim3(256), 0, su, fa);

  // the chain
  int* Lcnt = fl + FL_LCNT;
  int* F = fl + flow_f_off(m);
  {
    ProfScope ps(c, sp, GPMI_PROF_PANEL, (double)m * NB * NB * NB / 3.0 + lane.flow_flops_trsm, 0.0);
    for (int k = 0; k < m; ++k) {
      double* Akk = A0 + (int64_t)k * NB * ld + (int64_t)k * NB;
      double* invDk = invD0 + (int64_t)k * NB * NB;
      unsigned long long* ck = ctrace ? ctrace + (int64_t)k * (GPMI_STAMP_WORDS + 4) : nullptr;
      // GPMI_CHAIN_TILES=0: the generic kernels (32-row TRSM slabs, 64 x 64 update tiles) as until round 3; 1: the two
      // 16 x 16-tile kernels; 2: both products in one launch behind potrf_diag (round 4); 3 (default, round 5): potrf_diag
      // and both products in ONE launch, the products one row block of the inverse behind the factorisation - same bits
      static const int chain_mode = env_int("GPMI_CHAIN_TILES", 3);
      if (chain_mode == 3) {
        ChainColArgs ca{};
        ca.Akk = Akk;
        ca.invD = invDk;
        ca.ld = ld;
        ca.info = info;
        ca.col0 = (t0 + k) * NB;
        ca.dbg = ck;
        ca.pub = k > 0 ? Lcnt + k : nullptr;  // (the strip X of column k - 1 was stored by the launch before this one)
        ca.pub_val = 4 * k;
        ca.flags = fl;
        ca.k = k;
        ca.wait1 = F + (k + 1) * m + k;
        ca.wait2 = F + (k + 1) * m + (k + 1);
        ca.wait_val = 4 * k;
        ca.wait_ticks = stats ? stats + 4 : nullptr;
        ca.trace = ck ? ck + GPMI_STAMP_WORDS : nullptr;
        const bool has_next = k + 1 < m;
        if (!has_next) ca.wait1 = ca.wait2 = nullptr;
        hipLaunchKernelGGL(chain_column_kernel, dim3(has_next ? 1 + CC_WORKERS : 1), dim3(512), 0, sp, ca);
        launch_potrf_diag_fault(sp, Akk, ld);
        continue;
      }
      // (fused: the strip X of column k - 1 was stored by the launch before this one; potrf_diag publishes it)
      const bool pub_here = chain_mode == 2 && k > 0;
      launch_potrf_diag(sp, Akk, ld, invDk, info, (t0 + k) * NB, ck, BatchShape(), pub_here ? Lcnt + k : nullptr, 4 * k);
      if (k + 1 < m) {
        double* A21 = Akk + (int64_t)NB * ld;
        const bool chain_tiles = chain_mode != 0;
        GemmBatch tc;
        tc.ncu_hint = c->pair_cus[0];
        tc.b_lower_tri = true;
        tc.hook.pub = fl + FL_DDONE;
        tc.hook.pub_val = k + 1;
        tc.hook.wait = F + (k + 1) * m + k;
        tc.hook.wait_val = 4 * k;
        tc.hook.abort = fl + FL_ABORT;
        tc.hook.info = info;
        tc.hook.wait_ticks = stats ? stats + 4 : nullptr;
        tc.hook.trace = ck ? ck + GPMI_STAMP_WORDS : nullptr;
        GemmBatch uc;
        uc.hook.pub = Lcnt + k + 1;
        uc.hook.pub_val = 4 * (k + 1);
        uc.hook.wait = F + (k + 1) * m + (k + 1);
        uc.hook.wait_val = 4 * k;
        uc.hook.abort = fl + FL_ABORT;
        uc.hook.info = info;
        uc.hook.wait_ticks = stats ? stats + 5 : nullptr;
        uc.hook.trace = ck ? ck + GPMI_STAMP_WORDS + 2 : nullptr;
        if (chain_mode == 2) {
          ChainFusedArgs fa2{A21, A21 + NB, invDk, ld, fl + FL_ROWCNT, 8 * (k + 1), uc.hook.wait, uc.hook.wait_val, tc.hook};
          hipLaunchKernelGGL(chain_fused_kernel, dim3(32), dim3(512), 0, sp, fa2);
        } else if (chain_tiles) {
          ChainArgs ta{A21, A21 + NB, invDk, ld, tc.hook};
          hipLaunchKernelGGL(chain_trsm_kernel, dim3(NB / 16), dim3(512), 0, sp, ta);
          ChainArgs ua{A21, A21 + NB, invDk, ld, uc.hook};
          hipLaunchKernelGGL(chain_syrk_kernel, dim3(9), dim3(256), 0, sp, ua);
        } else {
          launch_gemm_nt(sp, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDk, NB, 1, 1, NB, nullptr, tc);
          launch_gemm_nt(sp, TILES_LOWER, OP_SUB, A21 + NB, ld, A21, ld, A21, ld, 1, 1, NB, nullptr, uc);
        }
      }
    }
  }
  (void)hipEventRecord(lane.ev_panel, sp);
  (void)hipEventRecord(lane.ev_main, su);
  (void)hipStreamWaitEvent(sf, lane.ev_panel, 0);
  (void)hipStreamWaitEvent(sf, lane.ev_main, 0);
  flow_gate_leave(c->device, sf);
  if (trace) {  // diagnostics only: synchronous
    (void)hipStreamSynchronize(sf);
    std::vector<unsigned long long> h((size_t)trace_words);
    std::vector<FlowTask> ht((size_t)ntasks);
    std::vector<int> ho((size_t)nlists + 1);
    (void)hipMemcpy(h.data(), trace, sizeof(unsigned long long) * trace_words, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ht.data(), lane.flow_tasks, sizeof(FlowTask) * ntasks, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ho.data(), lane.flow_off, sizeof(int) * (nlists + 1), hipMemcpyDeviceToHost);
    (void)hipFree(trace);
    if (FILE* fp = std::fopen(trace_path, "wb")) {
      const int64_t hdr[4] = {m, nlists, ntasks, GPMI_STAMP_WORDS + 4};
      std::fwrite(hdr, sizeof(hdr), 1, fp);
      std::fwrite(ho.data(), sizeof(int), ho.size(), fp);
      std::fwrite(ht.data(), sizeof(FlowTask), ht.size(), fp);
      std::fwrite(h.data(), sizeof(unsigned long long), h.size(), fp);
      std::fclose(fp);
    }
  }
  if (stats) {  // diagnostics only: synchronous
    unsigned long long h[8] = {0};
    (void)hipStreamSynchronize(sf);
    (void)hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost);
    std::fprintf(stderr,
                 "[flow] m=%d tasks=%llu polls=%llu | per workgroup: waiting %.1f us, in tasks %.1f us | chain waited: "
                 "Tc %.1f us, Uc %.1f us\n",
                 m, h[3], h[2], h[0] * 0.01 / grid.x, h[1] * 0.01 / grid.x,
                 h[4] * 0.01 / (env_int("GPMI_CHAIN_TILES", 3) == 3 ? (double)CC_WORKERS : 4.0), h[5] * 0.01 / 3.0);
  }
  return true;
}
