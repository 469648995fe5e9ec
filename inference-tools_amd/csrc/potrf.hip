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

namespace {

constexpr int NB = GPMI_NB;
constexpr int BS = 16;      // base block = one MFMA tile
constexpr int NBLK = NB / BS;
constexpr int WP = BS + 1;  // pitch of the 16 x 16 inverse diagonal blocks

__device__ inline double rcp_newton(double p) {
  double y = __builtin_amdgcn_rcp(p);
  double e = fma(-p, y, 1.0);
  y = fma(y, e, y);
  e = fma(-p, y, 1.0);
  return fma(y, e, y);
}

// LDS image of the 128 x 128 block: only the block-lower part is kept (block row ib holds
// (ib + 1) * 16 columns), each row padded by one double so that row and column walks are
// conflict-free: 74,752 bytes.  (Round 1 kept the kernel's LDS below 80 KiB so that it could share a CU with a GEMM
// workgroup; since the panel chain has CUs of its own - CU-masked streams - the kernel also keeps the inverse in LDS,
// 153 KiB in all: one workgroup per CU.)
constexpr int S_DOUBLES = 16 * (16 * 36 + 8);
// value of lane C of the lane's own 16-lane row (DPP row_newbcast on the 64-bit pair: VALU only, no LDS round trip)
template <int C>
__device__ inline double row_bcast(double v) {
  long long x = __builtin_bit_cast(long long, v);
  x = __builtin_amdgcn_mov_dpp(x, 0x150 + C, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}

// State of the 16 x 16 elimination ("column per lane", round 4).  Lane (g = lane >> 4, k = lane & 15) holds ALL 16
// rows of column k of the block - x[0..15], the four 16-lane groups redundantly - and the entries E[k][4 q + g]
// (q = 0..3) of row k of the accumulated row operations E (A = M D M^T, E -> M^-1).  Elimination step C is then
//   x[i] += bcast_C(x[i]) * nt     i > C     nt  = -(x[C] / p_C): the lane's own pivot-row element - no cross-lane
//   e[q] += bcast_C(e[q]) * nte    4q+g <= C nte = nt in the rows below the pivot (lanes k > C), 0 elsewhere
// with bcast_C = DPP row_newbcast:C folded into the instruction (v_fmac_f64_dpp): ONE instruction per row and step, no
// LDS round trip, no wave-wide shuffle.  (Round 1-3 spread a column over the four lane groups and moved the pivot
// row between them with ds_bpermute: 380 cycles per step; this form: ~120, bound by the issue rate of fp64 vector
// instructions of one wave - 6.6 cycles each, tools/probes/factor16_probe.hip - not by the dependency chain.)
struct Elim16 {
  double x[16], e[4];
  double p, ip;  // current pivot and its reciprocal (uniform over a 16-lane row)
  double myp;    // lane (., k): pivot k
  int k, g;
};

#include "factor16_steps.h"  // ElimStepAsm<C>: one asm block per step (tools/gen_factor16.py)

template <int C>
struct ElimStep {
  static __device__ __forceinline__ void run(Elim16& s) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C + 1 < BS) {
      const double nt = -(s.x[C] * s.ip);
      const double nte = (s.k > C) ? nt : 0.0;
      double pn, ipn;
      ElimStepAsm<C>::run(s, nt, nte, pn, ipn);
      s.p = pn;
      s.ip = ipn;
      ElimStep<C + 1>::run(s);
    }
  }
};

// 1 / sqrt(p) to 1 ulp: v_rsq_f64 (2^-23) and one cubic step, y (1 + e/2 + 3 e^2 / 8) with e = 1 - p y^2
__device__ inline double rsqrt_refined(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double e = fma(-(p * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// Scratch of the elimination in LDS, one per parity of the block index: T receives the block in the MFMA D layout and
// hands it back as whole columns; afterwards the same words carry the raw columns of U (and rs the 1 / sqrt(p_i)) to
// the wave that writes L to global memory one step later.
constexpr int TP = 18;  // row pitch of T (16-byte aligned rows for ds_read_b128)
struct ElimScratch {
  double T[BS * TP];
  double rs[BS];
};

// One wave: factor the symmetric 16 x 16 diagonal block `kb` (both triangles valid) and invert the factor.  Gaussian
// elimination without square roots on the critical path (ElimStep); the same row operations applied to the identity
// give M^-1; then L[k][i] = U[i][k] / sqrt(p_i) and W = L^-1 = D^-1/2 M^-1.
// `blk`: the block itself, lane (g = lane >> 4, k = lane & 15) element j = entry (4 j + g, k) - the D layout of the
// MFMA that produced it.  Returns W in the A-operand layout of the next product: w[q] = W[k][4 q + g]; W also goes
// to Wl (LDS, pitch WP); the raw U and 1 / sqrt(p) go to `sc` for flush_diag.
__device__ inline d4_t factor16(ElimScratch& sc, double* Wl, int kb, int* info, int col0, int lane, const d4_t& blk) {
  const int k = lane & 15, g = lane >> 4;
  Elim16 s;
#pragma unroll
  for (int j = 0; j < 4; ++j) sc.T[(4 * j + g) * TP + k] = blk[j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its own LDS writes are visible to it in order
#pragma unroll
  for (int m = 0; m < BS / 2; ++m) {
    const d2_t v = *reinterpret_cast<const d2_t*>(&sc.T[k * TP + 2 * m]);  // column k = row k (symmetric)
    s.x[2 * m] = v[0];
    s.x[2 * m + 1] = v[1];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) s.e[q] = (4 * q + g == k) ? 1.0 : 0.0;
  s.k = k;
  s.g = g;
  s.myp = 1.0;
  s.p = row_bcast<0>(s.x[0]);
  s.ip = rcp_newton(s.p);
  ElimStep<0>::run(s);
  // Pivots are examined once, behind the chain: lane k holds p_k.  A non-positive or non-finite pivot is reported
  // (LAPACK-style) and the block's values are then whatever the arithmetic gave - the factorisation is void.
  const bool badp = !(s.myp > 0.0) || !(s.myp < 1.79e308);
  const unsigned long long bad = __ballot(badp) & 0xffffull;
  const double rs = rsqrt_refined(badp ? 1.0 : s.myp);
  d4_t w;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    w[q] = s.e[q] * rs;  // W[k][4 q + g] (zero above the diagonal)
    Wl[k * WP + 4 * q + g] = w[q];
  }
  if (g == 0) {
#pragma unroll
    for (int m = 0; m < BS / 2; ++m) *reinterpret_cast<d2_t*>(&sc.T[k * TP + 2 * m]) = d2_t{s.x[2 * m], s.x[2 * m + 1]};
    sc.rs[k] = rs;
  }
  if (bad && lane == 0 && *info == 0) *info = col0 + kb * BS + __builtin_ctzll(bad) + 1;
  return w;
}

// The diagonal 16 x 16 block of L and of the inverse, from LDS to global memory (one wave, one step behind factor16):
// L[k][i] = U[i][k] / sqrt(p_i) for i <= k, W in full.
__device__ inline void flush_diag(const ElimScratch& sc, const double* Wl, double* __restrict__ invD,
                                  double* __restrict__ A, int64_t ld, int kb, int lane) {
  const int base = kb * BS;
  const int k = lane >> 2, i0 = (lane & 3) * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = i0 + t;
    if (i <= k) A[(int64_t)(base + k) * ld + base + i] = sc.T[k * TP + i] * sc.rs[i];
    invD[(base + k) * NB + base + i] = Wl[k * WP + i];
  }
}

// One workgroup (8 waves): L = chol(A_blk) in place (lower part of A), invD = L^-1 (dense 128 x 128, zero above
// the diagonal).  Blocked by 16; everything except the 16 x 16 eliminations runs as 16 x 16 x 16 products on
// v_mfma_f64_16x16x4_f64 - 72 cycles each on a SIMD (tools/probes/factor16_probe.hip), 840 of them: the MFMA time of
// one CU is a third of the kernel, so who computes what, when, decides the kernel as much as the chain does.
//
// Round 4 - three roles, ordered by six LDS counters (data flow; the one barrier is at the very start):
//   wave 0 (SIMD 0) - the chain: factor16(kb) -> W_kb published | sub-diagonal tile
//       (kb+1, kb) = A W^T and the trailing product of tile (kb+1, kb+1), in registers -> factor16(kb+1) ...
//   waves 1-3 (one per SIMD 1-3, priority 2) - the factor: every other tile (i, j) of the matrix has ONE owner for the
//       whole kernel (dealt round-robin in column-major order: any step's active tiles are a contiguous range of that
//       order, every step is balanced to within one tile) and lives in its owner's registers from its load (global
//       memory -> registers, one step before its first use: the fetch path of one CU, ~10 B / clk, needs 6600 cycles
//       for the 66 KB of the block - spread over the kernel instead of in front of it) to its last use, TRANSPOSED in the
//       MFMA D layout (accT[r] = tile[fr][fk + 4 r]) - at once the B operand of the panel step P^T = W tile^T.  Step kb of
//       an owner: A - panel step of its tiles of column kb (P to LDS: the operand of everybody's updates, and to global
//       memory); E - tile -= P_i,c P_j,c^T, lazily: a tile next used at step je receives the columns 0 .. je - 2 in one
//       chain of MFMAs at step je - 2 and column je - 1 at step je - 1 (13 - 17 products per step instead of 33, 25,
//       18, ...).  A tile leaves the registers once: after A, or - the two tiles wave 0 takes over next - after E.
//   waves 4-7 (one per SIMD, priority 0) - the inverse, right-looking, tiles owned the same way: B - row block kb,
//       X[kb][jb] = -W T[kb][jb]; D - T[i][jb] += L[i][kb] X[kb][jb] for the rows below.  More than half of the kernel's
//       MFMAs and on nobody's critical path until the last row.  (Wave 4 shares SIMD 0 with the chain: at the lowest
//       priority its MFMAs cost the chain's vector instructions little, and a quarter of the inverse leaves the three
//       SIMDs the factor waves need - 21.2 -> 18.5 us against wave 4 idle, GPMI_DIAG_INVERSE=3.)
// The slot loops are unrolled over compile-time tile codes (tile_code / slot_code below: a slot's accumulator is a named
// register, which tile it holds one of three or four constants picked by the wave's index), and a product's operands
// are read from LDS while the MFMAs of the product before it run (two operand sets in turn, pinned with
// sched_barrier: left alone, the compiler put every product's LDS reads right in front of its MFMAs, ~1000 cycles per
// product against ~300).  (A sequence of products generated on the fly by scalar code, the accumulator picked by a
// scalar switch over the slot, was built first: PHI webs over all slots, 256 VGPRs with spills, 73 us.)
// Counters (monotonic, LDS atomics behind the writer's own LDS traffic; nothing written to global memory is read
// again in this kernel, so no fence ever waits for a store acknowledgement): w_done (W_kb published), sub_ready (wave
// 0's tile (kb+1, kb)), panel_cnt (panel tiles, cumulative over the columns), hand_cnt (tiles handed to wave 0),
// xrow_cnt (finished tiles of the inverse, cumulative over its row blocks), flush_cnt (elimination scratch consumed).
#ifdef GPMI_DIAG_NOINV  // experiment (tools/build_variant.sh): the factor without the inverse's MFMAs beside it
#define GPMI_DIAG_NOINV_COND &&kb > 100
#define GPMI_DIAG_NOINV_SKIP true
#else
#define GPMI_DIAG_NOINV_COND
#define GPMI_DIAG_NOINV_SKIP false
#endif
constexpr int DIAG_THREADS = 512;
#ifndef GPMI_DIAG_INVERSE
#define GPMI_DIAG_INVERSE 4  // 4: wave 4 (the chain's SIMD mate) is a fourth owner of the inverse's tiles; 3: it leaves at once (A/B builds)
#endif
constexpr int DIAG_FACTOR = 3, DIAG_INVERSE = GPMI_DIAG_INVERSE;  // owner waves of the matrix tiles / of the inverse's tiles
constexpr int NE_TILES = NBLK * (NBLK + 1) / 2 - 3;  // tiles (i, j), j <= i, without (0,0), (1,0), (1,1): wave 0's from the start
constexpr int ND_TILES = NBLK * (NBLK - 1) / 2;      // tiles of the inverse below the diagonal
constexpr int E_SLOTS = (NE_TILES + DIAG_FACTOR - 1) / DIAG_FACTOR;
constexpr int D_SLOTS = (ND_TILES + DIAG_INVERSE - 1) / DIAG_INVERSE;
#ifdef GPMI_DIAG_TRACE
constexpr int ES_BUFS = 2;
constexpr int DIAG_TRACE_EV = 6;
constexpr unsigned long long DIAG_TRACE_MAGIC = 0x7ACEull;
#else
constexpr int ES_BUFS = 4;
#endif

// The tiles in the order in which they are dealt: column-major over the lower triangle - the matrix's tiles without
// (0,0), (1,0), (1,1): (2,0) .. (7,0), (2,1) .. (7,1), (2,2) .. - and the inverse's below the diagonal: (1,0) .. (7,0),
// (2,1) ..   tile_code(n) = row << 4 | column of the n-th tile, 0xff past the end; evaluated at compile time (the slot
// index is a template parameter, the owner's index picks one of three constants): a table in memory cost every launch a
// round of dependent loads before the first instruction of real work.
constexpr int tile_code(int n, bool inverse) {
  for (int j = 0; j < NBLK; ++j)
    for (int i = inverse ? j + 1 : (j < 2 ? 2 : j); i < NBLK; ++i)
      if (n-- == 0) return i << 4 | j;
  return 0xff;
}
template <int S, bool INV>
__device__ __forceinline__ int slot_code(int wb) {
  // the inverse's tiles are dealt from the other end (evens out the slot counts of a SIMD's two waves)
  constexpr int NW = INV ? DIAG_INVERSE : DIAG_FACTOR;
  constexpr int c0 = tile_code((INV ? NW - 1 : 0) + NW * S, INV), c1 = tile_code((INV ? NW - 2 : 1) + NW * S, INV),
                c2 = tile_code((INV ? NW - 3 : 2) + NW * S, INV);
  // (no fourth alternative unless there is a fourth owner: with an "empty" code among a slot's possible values the
  // compiler stops specialising the slot loops - 256 VGPRs, spills, 28 instead of 21 us)
  if constexpr (NW == 3) {
    return wb == 0 ? c0 : (wb == 1 ? c1 : c2);
  } else {
    constexpr int c3 = tile_code((INV ? NW - 4 : 3) + NW * S, INV);
    return wb == 0 ? c0 : (wb == 1 ? c1 : (wb == 2 ? c2 : c3));
  }
}
template <bool INV, int... S>
__device__ __forceinline__ void slot_codes(int wb, int* code, std::integer_sequence<int, S...>) {
  ((code[S] = slot_code<S, INV>(wb)), ...);
}
static_assert(DIAG_FACTOR == 3 && (DIAG_INVERSE == 3 || DIAG_INVERSE == 4), "slot_code deals to three or four owners");
// panel tiles (rows >= c + 2) of the columns 0 .. kb; tiles of the inverse's row blocks 0 .. kb
__device__ inline int panels_through(int kb) { return (kb + 1) * (NBLK - 2) - kb * (kb + 1) / 2; }
__device__ inline int xtiles_through(int kb) { return kb * (kb + 1) / 2; }
// S: first double of block row i, and its row pitch
__device__ inline int sbase(int i) { return 16 * (8 * i * (i + 1) + i); }
__device__ inline int spitch(int i) { return 16 * (i + 1) + 1; }

__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(double* __restrict__ A, int64_t ld,
                                                         double* __restrict__ invD,
                                                         int* __restrict__ info, int col0,
                                                         unsigned long long* __restrict__ dbg,
                                                         int64_t strideA, int64_t strideInv, int* pub, int pub_val) {
  // (flag-ordered tail, potrf_flow.hip: this launch publishes what the chain launch before it produced)
  if (pub && threadIdx.x == 0 && blockIdx.z == 0)
    __hip_atomic_store(pub, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  A += (int64_t)blockIdx.z * strideA;
  invD += (int64_t)blockIdx.z * strideInv;
  info += blockIdx.z;
  // dbg != nullptr (tools only; a 24-word stamp slot): word 0 / word 8 = wall clock (s_memrealtime) at the first
  // instruction / behind the last store, words 16..21 = cycle counts of the phases, accumulated by wave 0
  if (dbg && threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_prev = 0, acc_t[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int slot) {
    if (dbg) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      acc_t[slot] += t - t_prev;
      t_prev = t;
    }
  };
  if (dbg) t_prev = __builtin_amdgcn_s_memtime();
  // LDS image of the block's lower 16 x 16 tiles (row pitch odd: row and column walks conflict-free): the panel tiles
  // (final L) as MFMA operands, and the two tiles per step that change hands
  __shared__ double S[S_DOUBLES];
  // the inverse as it grows: 16 x 16 block (k2, jb), jb <= k2, at Xl[k2 (k2 + 1) / 2 + jb] (row pitch 17): finished
  // row blocks are the operands of the later ones' sums
  __shared__ double Xl[NBLK * (NBLK + 1) / 2][BS * WP];
  // factor16's scratch, by block index modulo 4: wave 0 is never held up by the wave that writes a block's L out
  __shared__ __attribute__((aligned(16))) ElimScratch Es[ES_BUFS];
  __shared__ int w_done, sub_ready, panel_cnt, hand_cnt, xrow_cnt, flush_cnt;
#ifdef GPMI_DIAG_TRACE
  // tools only (a build of its own, tools/build_variant.sh trace -DGPMI_DIAG_TRACE: the table takes LDS that the regular
  // build gives to the elimination's scratch; dbg[22] == DIAG_TRACE_MAGIC: a buffer of 8 x 9 x DIAG_TRACE_EV more words
  // follows the 24): per wave and step, the clock at up to DIAG_TRACE_EV points (tools/diag_stamps.py prints the timeline)
  __shared__ unsigned int trace_t[8][NBLK + 1][DIAG_TRACE_EV];
  const bool tracing = dbg && dbg[22] == DIAG_TRACE_MAGIC;
  const unsigned long long trace_t0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
  auto ev = [&](int step, int e) {
    if (tracing && (threadIdx.x & 63) == 0)
      trace_t[threadIdx.x >> 6][step][e] = (unsigned int)(__builtin_amdgcn_s_memtime() - trace_t0);
  };
  if (tracing)
    for (int i = threadIdx.x; i < 8 * (NBLK + 1) * DIAG_TRACE_EV; i += DIAG_THREADS) (&trace_t[0][0][0])[i] = 0;
#else
  auto ev = [](int, int) {};
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: roles and tile indices stay in SGPRs
  const int fr = lane & 15, fk = lane >> 4;
  if (tid == 0) {
    w_done = 0;
    sub_ready = 0;
    panel_cnt = 0;
    hand_cnt = 0;
    xrow_cnt = 0;
    flush_cnt = 0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the counters are zero for everybody
  if (DIAG_INVERSE == 3 && wave == 4) return;  // the chain has SIMD 0 to itself

  // (every wave of the workgroup is resident, so a counter always arrives; the bound - ~0.1 s - only keeps a bug from
  // hanging the GPU: the factorisation is then wrong and says so through info)
  auto wait_for = [&](int* counter, int target) {
    int polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++polls > (1 << 21)) {
        if (lane == 0) *info = col0 + 1;
        break;
      }
    }
    asm volatile("" ::: "memory");
  };
  auto signal = [&](int* counter, int add) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes are done: in order behind them
    if (lane == 0) __hip_atomic_fetch_add(counter, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  if (wave == 0) {
    // ---------------------------------------------------------------------------------------------- the chain
    __builtin_amdgcn_s_setprio(3);
    d4_t blk, t, b;  // diagonal block kb / the next one / the sub-diagonal tile (kb+1, kb), from the lower triangle
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + fk;  // entry (i, fr)
      blk[r] = A[(int64_t)(i > fr ? i : fr) * ld + (i > fr ? fr : i)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // its first operands come straight from global memory, under the first elimination
      const int i = fk + 4 * r;
      t[r] = A[(int64_t)(BS + (i > fr ? i : fr)) * ld + BS + (i > fr ? fr : i)];
      b[r] = A[(int64_t)(BS + fr) * ld + fk + 4 * r];
    }
    lap(0);
#pragma nounroll
    for (int kb = 0; kb < NBLK; ++kb) {
      if (kb >= ES_BUFS) wait_for(&flush_cnt, kb - ES_BUFS + 1);  // the scratch of step kb - ES_BUFS has been written out
      ev(kb, 0);
      const d4_t w = factor16(Es[kb % ES_BUFS], Xl[kb * (kb + 1) / 2 + kb], kb, info, col0, lane, blk);
      signal(&w_done, 1);
      ev(kb, 1);
      lap(4);
      if (kb + 1 == NBLK) break;
      // the sub-diagonal tile P = A[kb+1][kb] W^T and the trailing product of tile (kb+1, kb+1), without an LDS round trip
      // between them: P is computed transposed (P^T = W A^T), which makes its D registers at once the A and the B operand
      // of the trailing product (P[fr][fk + 4 q] = pt[q]); W comes in registers from factor16 (w[q] = W[fr][fk + 4 q]:
      // its A-operand layout) and the result stays in registers for the next factor16.
      const int ib = kb + 1;
      const int rb = sbase(ib) + fr * spitch(ib) + kb * BS + fk;
      if (kb > 0) {
        wait_for(&hand_cnt, 2 * kb);  // both tiles carry every column before kb
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = fk + 4 * r;  // entry (i, fr) of the symmetric tile: from the lower triangle
          t[r] = S[sbase(ib) + (i > fr ? i : fr) * spitch(ib) + ib * BS + (i > fr ? fr : i)];
          b[r] = S[rb + 4 * r];
        }
      }
      ev(kb, 2);
      lap(1);
      d4_t pt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(w[q], b[q], pt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) S[rb + 4 * r] = pt[r];  // (an inverse wave copies it to global memory)
      signal(&sub_ready, 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(pt[q], pt[q], t, 0, 0, 1);  // BLGP 1: -A B + C
      blk = t;
      ev(kb, 3);
      lap(2);
    }
  } else if (wave <= 3) {
    // ---------------------------------------------------------------------------------------------- the factor
    __builtin_amdgcn_s_setprio(2);
    const int wb = wave - 1;
    // The tiles of this wave, in registers for the whole kernel (declared per role: the three roles' registers overlap
    // instead of adding up): slot s = tile (ti, tj) of the matrix, transposed; je = the step of its next use by somebody
    // else - its panel step, or, for the tiles (i, i), (i, i - 1) that receive column i - 1 from wave 0 itself, i - 1.
    d4_t acc[E_SLOTS];
    int code[E_SLOTS], ti[E_SLOTS], tj[E_SLOTS], je[E_SLOTS];
    slot_codes<false>(wb, code, std::make_integer_sequence<int, E_SLOTS>{});
#pragma unroll
    for (int s = 0; s < E_SLOTS; ++s) {
      ti[s] = tj[s] = -1;
      je[s] = 100;
      if (code[s] != 0xff) {
        ti[s] = code[s] >> 4;
        tj[s] = code[s] & 15;
        je[s] = ti[s] - tj[s] <= 1 ? ti[s] - 1 : tj[s];
      }
      acc[s] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    // tiles first used at step `step` (their first update is at step max(je - 2, 0)): global memory -> registers,
    // transposed (accT[r] = tile[fr][fk + 4 r]; the upper half of a diagonal tile from its mirror image).  Nothing waits for
    // the data before the first use; requested one step ahead, it is there by then.
    auto fetch_tiles = [&](int step, bool also_next) {
#pragma unroll
      for (int s = 0; s < E_SLOTS; ++s) {
        const int fu = je[s] >= 2 ? je[s] - 2 : 0;
        if (ti[s] >= 0 && (fu == step || (also_next && fu == step + 1))) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = fr, col = fk + 4 * r;
            if (ti[s] == tj[s] && col > row) {
              row = fk + 4 * r;
              col = fr;
            }
            acc[s][r] = A[(int64_t)(ti[s] * BS + row) * ld + tj[s] * BS + col];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    fetch_tiles(0, true);  // steps 0 and 1
#pragma nounroll
    for (int kb = 0; kb < NBLK; ++kb) {
      ev(kb, 0);
      wait_for(&w_done, kb + 1);
      ev(kb, 1);
      const double* W = Xl[kb * (kb + 1) / 2 + kb];
      // A: panel step of the tiles of column kb (rows kb + 2 ..): P^T = W tile^T; P to LDS and to global memory
      double wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = W[fr * WP + fk + 4 * q];  // A operand: W[fr][fk + 4 q]
      int npanel = 0;
#pragma unroll
      for (int s = 0; s < E_SLOTS; ++s)
        if (tj[s] == kb && ti[s] >= kb + 2) {
          d4_t pt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q], acc[s][q], pt, 0, 0, 0);
          const int rb = sbase(ti[s]) + fr * spitch(ti[s]) + kb * BS + fk;
#pragma unroll
          for (int r = 0; r < 4; ++r) S[rb + 4 * r] = pt[r];  // (an inverse wave copies it to global memory)
          ++npanel;
          __builtin_amdgcn_sched_barrier(0);
        }
      if (npanel) signal(&panel_cnt, npanel);
      ev(kb, 2);
      if (kb + 1 == NBLK) break;
      wait_for(&panel_cnt, panels_through(kb));
      wait_for(&sub_ready, kb + 1);
      ev(kb, 3);
      // E: tile (i, j) -= P_i,c P_j,c^T, transposed: accT -= P_j,c P_i,c^T (the MFMA's BLGP field negates A).  First the
      // tiles that are used next at step kb + 1 (je == kb + 1: column kb is their last; the two that change hands go to
      // LDS at once - wave 0 is waiting for them), then the tiles with je == kb + 2: columns 0 .. kb in one chain on the
      // accumulator, two columns per round on alternating operand registers, the next column's operands requested
      // before the MFMAs of the current one.
      auto column = [&](int s, int ra, int rb, int c, double (&a)[4], double (&b)[4]) {
        (void)s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = S[ra + c * BS + 4 * q];
          b[q] = S[rb + c * BS + 4 * q];
        }
      };
#pragma unroll
      for (int s = 0; s < E_SLOTS; ++s)
        if (je[s] == kb + 1) {
          const int i = ti[s], j = tj[s];
          const int ra = sbase(j) + fr * spitch(j) + fk, rb = sbase(i) + fr * spitch(i) + fk;
          double a[4], b[4];
          column(s, ra, rb, kb, a, b);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[s], 0, 0, 1);
          if (i - j <= 1) {
            const int rs = sbase(i) + fr * spitch(i) + j * BS + fk;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (i != j || fk + 4 * r <= fr) S[rs + 4 * r] = acc[s][r];
            signal(&hand_cnt, 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int s = 0; s < E_SLOTS; ++s)
        if (je[s] == kb + 2) {
          const int i = ti[s], j = tj[s];
          const int ra = sbase(j) + fr * spitch(j) + fk, rb = sbase(i) + fr * spitch(i) + fk;
          double a0[4], b0[4], a1[4], b1[4];
          column(s, ra, rb, 0, a0, b0);
          for (int c = 0; c <= kb; c += 2) {
            column(s, ra, rb, c + 1 <= kb ? c + 1 : c, a1, b1);  // (an odd last round re-reads its own column: no branch)
            __builtin_amdgcn_sched_barrier(0);  // the requests above stay above the MFMAs below
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc[s], 0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 > kb) break;
            column(s, ra, rb, c + 2 <= kb ? c + 2 : c + 1, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], acc[s], 0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      // The tiles first used at the next step but one.  Requested here, they arrive while the wave waits for the chain;
      // this wave issues no other vector-memory instruction (the stores of what it computes are the inverse waves'
      // job), so the s_waitcnt vmcnt(0) in front of a tile's first use waits for nothing else - in particular for no
      // store acknowledgement.
      fetch_tiles(kb + 2, false);
      ev(kb, 4);
    }
  } else {
    // ---------------------------------------------------------------------------------------------- the inverse
    __builtin_amdgcn_s_setprio(0);
    const int wb = wave == 4 ? 3 : wave - 5;
    d4_t acc[D_SLOTS];  // slot s: the sum T[ti][tj]
    int code[D_SLOTS], ti[D_SLOTS], tj[D_SLOTS];
    slot_codes<true>(wb, code, std::make_integer_sequence<int, D_SLOTS>{});
#pragma unroll
    for (int s = 0; s < D_SLOTS; ++s) {
      ti[s] = tj[s] = -1;
      if (code[s] != 0xff) {
        ti[s] = code[s] >> 4;
        tj[s] = code[s] & 15;
      }
      acc[s] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    // (the strictly-upper 16-blocks of the inverse are zero since the buffer's allocation: api.hip, lane_alloc)
#pragma nounroll
    for (int kb = 0; kb < NBLK; ++kb) {
      ev(kb, 0);
      wait_for(&w_done, kb + 1);
      ev(kb, 1);
      const double* W = Xl[kb * (kb + 1) / 2 + kb];
      // B: row block kb of the inverse, X[kb][jb] = -W T[kb][jb] (T in the D layout is the B operand)
      double wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = W[fr * WP + fk + 4 * q];
      int nx = 0;
#pragma unroll
      for (int s = 0; s < D_SLOTS; ++s)
        if (ti[s] == kb) {
          ++nx;
          if (GPMI_DIAG_NOINV_SKIP) continue;
          d4_t X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) X = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q], acc[s][q], X, 0, 0, 1);  // -W T
          double* Xt = Xl[kb * (kb + 1) / 2 + tj[s]];
          double* dst = invD + (kb * BS + fk) * NB + tj[s] * BS + fr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Xt[(fk + 4 * r) * WP + fr] = X[r];
            dst[4 * r * NB] = X[r];
          }
        }
      if (nx) signal(&xrow_cnt, nx);
      // the diagonal 16-blocks of L and of the inverse that the elimination of step kb left in LDS
      if (wb == kb % DIAG_INVERSE) {
        flush_diag(Es[kb % ES_BUFS], W, invD, A, ld, kb, lane);
        signal(&flush_cnt, 1);
      }
      ev(kb, 2);
      if (kb + 1 == NBLK) break;
      wait_for(&xrow_cnt, xtiles_through(kb));
      wait_for(&panel_cnt, panels_through(kb));
      wait_for(&sub_ready, kb + 1);
      ev(kb, 3);
      // column kb of L is final in LDS: to global memory (these waves never load, so nothing of theirs ever waits for
      // a store acknowledgement)
      for (int i = kb + 1 + wb; i < NBLK; i += DIAG_INVERSE) {
        const int rs = sbase(i) + fr * spitch(i) + kb * BS + fk;
        double* dst = A + (int64_t)(i * BS + fr) * ld + kb * BS + fk;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[4 * r] = S[rs + 4 * r];
      }
      // D: T[i][jb] += L[i][kb] X[kb][jb] for the row blocks i below, jb <= kb
#pragma unroll
      for (int s = 0; s < D_SLOTS; ++s)
        if (tj[s] >= 0 && tj[s] <= kb && kb < ti[s] GPMI_DIAG_NOINV_COND) {
          const double* X = Xl[kb * (kb + 1) / 2 + tj[s]];
          const int ra = sbase(ti[s]) + fr * spitch(ti[s]) + kb * BS + fk;
          double a[4], b[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            a[q] = S[ra + 4 * q];
            b[q] = X[(fk + 4 * q) * WP + fr];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[s], 0, 0, 0);
        }
      ev(kb, 4);
    }
  }
  ev(NBLK, 0);
  lap(3);
  if (dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
#ifdef GPMI_DIAG_TRACE
    if (tracing)
      for (int i = tid; i < 8 * (NBLK + 1) * DIAG_TRACE_EV; i += DIAG_THREADS - 64) dbg[24 + i] = (&trace_t[0][0][0])[i];
#endif
    if (tid == 0) {
      for (int i = 0; i < 6; ++i) dbg[16 + i] = acc_t[i];
      dbg[8] = __builtin_amdgcn_s_memrealtime();
    }
  }
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

void launch_potrf_diag(hipStream_t s, double* Ablk, int64_t ld, double* invD, int* info, int col0,
                       unsigned long long* dbg, const BatchShape& bs, int* pub, int pub_val) {
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, (unsigned)bs.count), dim3(DIAG_THREADS), 0, s, Ablk, ld, invD,
                     info, col0, dbg, bs.sMat, bs.sInv, pub, pub_val);
  // GPMI_FAULT_DIAG_EPS=<eps>: fault injection for the test suite's own sensitivity check, never set otherwise
  static const double fault = [] {
    const char* e = std::getenv("GPMI_FAULT_DIAG_EPS");
    return e ? std::atof(e) : 0.0;
  }();
  if (fault != 0.0)
    hipLaunchKernelGGL(fault_scale_kernel, dim3(NB, 1, (unsigned)bs.count), dim3(NB), 0, s, Ablk, ld, 1.0 + fault,
                       bs.sMat);
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
  const int64_t nmain = big ? nfull : 0;
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
// 2.364 to 2.352 GHz (the update runs at the chip's power limit), which is why the gain is a third of the idle time.
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
  // step (medians of six alternating runs, tools/scratch/knobs4.sh); 45 % more is too much (32.86: the next update waits
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
  // Measured and dropped: a second pair with 16 | 240 CUs for the early panels (the
  // update is bound by the chip's power budget: 224, 240 and 256 CUs deliver the same FLOP/s, in-kernel clock
  // 2.05 / 1.97 GHz); trailing updates applied lazily with K = 1024 .. 2048 (in place the launches are already
  // split at round boundaries, which leaves +1 % for the larger K, and the narrower launches cost more).
  hipStream_t sf = lane.stream;
  const int nt = (int)(np / NB);
  const int OBT = OUTER_TILES;
  const int LOOKAHEAD_MIN = lookahead_min();
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
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    for (int j = J; j < Je; ++j) {
      double* Ajj = A + (int64_t)j * NB * ld + (int64_t)j * NB;
      double* invDj = invD + (int64_t)j * NB * NB;
      const int below = nt - j - 1;
      launch_potrf_diag(s, Ajj, ld, invDj, info, j * NB, nullptr, bs);
      if (below > 0) {
        double* A21 = Ajj + (int64_t)NB * ld;
        launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDj, NB, below, 1, NB, nullptr, inplace);
        const int pc = Je - j - 1;  // remaining tile columns of the outer panel
        if (pc > 0) launch_gemm_nt(s, TILES_LOWER, OP_SUB, A21 + NB, ld, A21, ld, A21, ld, below, pc, NB, nullptr, upd);
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
