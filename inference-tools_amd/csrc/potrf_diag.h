// potrf_diag: Cholesky factor and inverse of one 128 x 128 diagonal block by one workgroup of 8 waves (gfx950).
// The body is shared by potrf_diag_kernel (potrf.hip: the stream-ordered and lockstep factorisations) and by the chain
// launch of the flag-ordered tail (potrf_flow.hip: chain_column_kernel, whose workgroup 0 runs it beside the workgroups that
// apply the inverse to the next tile row).  Replaces the diagonal-block work inside numpy.linalg.cholesky's dpotrf at
// regression.py:241, 537, 555.
#pragma once
#include <type_traits>
#include <utility>

#include "gpmi_internal.h"

namespace potrf_diag {

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

// A store of the inverse.  PUB (the chain launch of the flag-ordered tail, potrf_flow.hip: chain_column_kernel): other
// CUs read the inverse while this kernel is still running, so every byte of it goes out write-through at agent scope
// (sc1) and the storing wave counts the row block in once the stores are acknowledged (DiagPub below).
template <bool PUB>
__device__ __forceinline__ void store_inv(double* p, double v) {
  if constexpr (PUB)
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else
    *p = v;
}

// The diagonal 16 x 16 block of L and of the inverse, from LDS to global memory (one wave, one step behind factor16):
// L[k][i] = U[i][k] / sqrt(p_i) for i <= k, W in full.
template <bool PUB>
__device__ inline void flush_diag(const ElimScratch& sc, const double* Wl, double* __restrict__ invD,
                                  double* __restrict__ A, int64_t ld, int kb, int lane) {
  const int base = kb * BS;
  const int k = lane >> 2, i0 = (lane & 3) * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = i0 + t;
    if (i <= k) A[(int64_t)(base + k) * ld + base + i] = sc.T[k * TP + i] * sc.rs[i];
    store_inv<PUB>(&invD[(base + k) * NB + base + i], Wl[k * WP + i]);
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

// Row-block publication of the inverse (PUB builds of the body): the four inverse waves each add 1 to rows[32 r] once
// their stores of row block r (and of everything before it) are acknowledged; a reader on another CU that sees
// base + DIAG_INVERSE there may load row block r after an agent-scope acquire.  The wave that completes row block 7
// also stores ddone_val to *ddone (the whole inverse, for the tile tasks' panel TRSMs).
struct DiagPub {
  int* rows = nullptr;
  int base = 0;
  int* ddone = nullptr;
  int ddone_val = 0;
};

// LDS of the kernel (one object, so that a kernel with a second role can lay its own scratch over it)
struct __attribute__((aligned(16))) DiagShared {
  // image of the block's lower 16 x 16 tiles (row pitch odd: row and column walks conflict-free): the panel tiles
  // (final L) as MFMA operands, and the two tiles per step that change hands
  double S[S_DOUBLES];
  // the inverse as it grows: 16 x 16 block (k2, jb), jb <= k2, at Xl[k2 (k2 + 1) / 2 + jb] (row pitch 17): finished
  // row blocks are the operands of the later ones' sums
  double Xl[NBLK * (NBLK + 1) / 2][BS * WP];
  // factor16's scratch, by block index modulo 4: wave 0 is never held up by the wave that writes a block's L out
  ElimScratch Es[ES_BUFS];
  int w_done, sub_ready, panel_cnt, hand_cnt, xrow_cnt, flush_cnt;
#ifdef GPMI_DIAG_TRACE
  // tools only (a build of its own, tools/build_variant.sh trace -DGPMI_DIAG_TRACE: the table takes LDS that the regular
  // build gives to the elimination's scratch; dbg[22] == DIAG_TRACE_MAGIC: a buffer of 8 x 9 x DIAG_TRACE_EV more words
  // follows the 24): per wave and step, the clock at up to DIAG_TRACE_EV points (tools/diag_stamps.py prints the timeline)
  unsigned int trace_t[8][NBLK + 1][DIAG_TRACE_EV];
#endif
};

// The whole workgroup (DIAG_THREADS threads) calls this; A / invD / info are the problem's own (batch offsets applied).
template <bool PUB>
__device__ __forceinline__ void potrf_diag_body(double* __restrict__ A, int64_t ld, double* __restrict__ invD,
                                                int* __restrict__ info, int col0, unsigned long long* __restrict__ dbg,
                                                DiagShared& sh, const DiagPub& pub) {
  double(&S)[S_DOUBLES] = sh.S;
  double(&Xl)[NBLK * (NBLK + 1) / 2][BS * WP] = sh.Xl;
  ElimScratch(&Es)[ES_BUFS] = sh.Es;
  int &w_done = sh.w_done, &sub_ready = sh.sub_ready, &panel_cnt = sh.panel_cnt, &hand_cnt = sh.hand_cnt,
      &xrow_cnt = sh.xrow_cnt, &flush_cnt = sh.flush_cnt;
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
#ifdef GPMI_DIAG_TRACE
  unsigned int(&trace_t)[8][NBLK + 1][DIAG_TRACE_EV] = sh.trace_t;
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
    // PUB: row block r of the inverse is handed to the other CUs by every inverse wave once its stores up to there are
    // acknowledged - rows 0 .. 6 one step late (at the top of the next step: the stores left a step's MFMAs ago, the wait
    // costs nothing), row 7 at once
    auto publish_row = [&](int r) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const int old = __hip_atomic_fetch_add(pub.rows + 32 * r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == NBLK - 1 && pub.ddone && old == pub.base + DIAG_INVERSE - 1)
          __hip_atomic_store(pub.ddone, pub.ddone_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
#pragma nounroll
    for (int kb = 0; kb < NBLK; ++kb) {
      ev(kb, 0);
      if constexpr (PUB)
        if (kb > 0) publish_row(kb - 1);
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
          for (int r = 0; r < 4; ++r) Xt[(fk + 4 * r) * WP + fr] = X[r];
          // (PUB: the stores are inline assembly, which the hazard recognizer does not see as a reader of the MFMA's
          // result registers - the wait states an XDL write needs before a VMEM read of it are spelt out)
          if constexpr (PUB) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
          for (int r = 0; r < 4; ++r) store_inv<PUB>(&dst[4 * r * NB], X[r]);
        }
      if (nx) signal(&xrow_cnt, nx);
      // the diagonal 16-blocks of L and of the inverse that the elimination of step kb left in LDS
      if (wb == kb % DIAG_INVERSE) {
        flush_diag<PUB>(Es[kb % ES_BUFS], W, invD, A, ld, kb, lane);
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
    if constexpr (PUB) publish_row(NBLK - 1);
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

}  // namespace potrf_diag
