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
#include <vector>

#include "gpmi_internal.h"

namespace {

constexpr int NB = GPMI_NB;
constexpr int BS = 16;      // base block = one MFMA tile
constexpr int NBLK = NB / BS;
constexpr int WP = BS + 1;  // pitch of the 16 x 16 inverse diagonal blocks

// value of `v` in lane `src` (compile-time constant) as a wave-uniform scalar
__device__ inline double lane_bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

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
__device__ inline int prow(int r) {
  const int ib = r >> 4;
  return 16 * (8 * ib * (ib + 1) + ib) + (r & 15) * ((ib + 1) * 16 + 1);
}

// value of lane C of the lane's own 16-lane row (DPP row_newbcast: VALU only, no LDS round trip)
template <int C>
__device__ inline double row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + C, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + C, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// State of the 16 x 16 elimination: lane (g = lane >> 4, k = lane & 15) holds column k of the rows
// i = 4 j + g (j = 0..3) of the block (a) and of the accumulated row operations (e).
struct Elim16 {
  double a[4], e[4];
  double p, ip;        // current pivot and its reciprocal (wave-uniform)
  double arow, erow;   // pivot row elements of this lane's column
  double myp;          // lane (., k): pivot k
  int badcol;
  int k, g;
};

// Elimination step C: row_i -= (A[i][C] / p_C) row_C for the rows i > C.  Three things keep the dependency
// chain of a step down to one LDS round trip:
//  * the multiplier A[i][C] of a lane's row comes from lane C of its own 16-lane row through DPP;
//  * the pivot and its reciprocal are wave-uniform scalars kept one step ahead: p_{C+1} is formed from three
//    lane reads at the start of step C exactly as the owning lane forms it, so the Newton chain of the
//    reciprocal overlaps the row operations;
//  * row C + 1 is updated first and sent on its way to the other lane groups (ds_bpermute) before the
//    remaining rows of step C are updated.
template <int C>
struct ElimStep {
  static __device__ inline void run(Elim16& s) {
    constexpr int jc = C >> 2, gc = C & 3;
    constexpr int j1 = (C + 1) >> 2, g1 = (C + 1) & 3;  // slot / lane group of row C + 1
    double pn = 1.0, ipn = 1.0;
    if (C + 1 < BS) {
      const double s1 = lane_bcast(s.a[j1 & 3], g1 * 16 + C);             // A[C+1][C]
      const double s2 = lane_bcast(s.a[j1 & 3], g1 * 16 + ((C + 1) & 15));  // A[C+1][C+1]
      const double s3 = lane_bcast(s.a[jc], gc * 16 + ((C + 1) & 15));      // A[C][C+1]
      const double m1 = s1 * s.ip;
      pn = fma(-m1, s3, s2);
      if (!(pn > 0.0) || !(pn < 1.79e308)) {  // wave-uniform
        if (s.badcol < 0) s.badcol = C + 1;
        pn = 1.0;
      }
      ipn = rcp_newton(pn);
    }
    if (s.k == C) s.myp = s.p;
    auto update = [&](int j) {
      double m = row_bcast<C>(s.a[j]) * s.ip;  // A[i][C] / p_C for this lane's row i = 4 j + g
      if (j == jc && s.g <= gc) m = 0.0;       // row i <= C: untouched
      s.a[j] = fma(-m, s.arow, s.a[j]);
      s.e[j] = fma(-m, s.erow, s.e[j]);
    };
    double arow_n = 0.0, erow_n = 0.0;
    if (C + 1 < BS) {
      update(j1 & 3);
      arow_n = __shfl(s.a[j1 & 3], g1 * 16 + s.k, 64);
      erow_n = __shfl(s.e[j1 & 3], g1 * 16 + s.k, 64);
    }
#pragma unroll
    for (int j = jc; j < 4; ++j)
      if (!(C + 1 < BS && j == (j1 & 3))) update(j);
    s.arow = arow_n;
    s.erow = erow_n;
    s.p = pn;
    s.ip = ipn;
    ElimStep<C + 1>::run(s);
  }
};
template <>
struct ElimStep<BS> {
  static __device__ inline void run(Elim16&) {}
};

// One wave: factor the symmetric 16 x 16 diagonal block `kb` of S (both triangles valid) and
// invert the factor.  Gaussian elimination without square roots on the critical path (ElimStep); the
// same row operations applied to the identity give M^-1 (A = M D M^T); then
// L[k][i] = U[i][k] / sqrt(p_i) and W = L^-1 = D^-1/2 M^-1 (to LDS for the panel phase and to the
// diagonal block of invD in global memory); L itself also goes straight to the matrix in global memory.
// `blk`: the block itself, lane (g = lane >> 4, k = lane & 15) element j = entry (4 j + g, k) - the D layout of the MFMA
// that produced it (potrf_diag_kernel keeps it in registers from the trailing product to the elimination).
__device__ inline void factor16(double* S, double* Wl, double* __restrict__ invD, double* __restrict__ A,
                                int64_t ld, int kb, int* info, int col0, int lane, const d4_t& blk) {
  const int k = lane & 15, g = lane >> 4;
  const int base = kb * BS;
  Elim16 s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * j + g;
    s.a[j] = blk[j];
    s.e[j] = (i == k) ? 1.0 : 0.0;
  }
  s.k = k;
  s.g = g;
  s.myp = 1.0;
  s.badcol = -1;
  s.p = lane_bcast(s.a[0], 0);  // A[0][0]
  if (!(s.p > 0.0) || !(s.p < 1.79e308)) {
    s.badcol = 0;
    s.p = 1.0;
  }
  s.ip = rcp_newton(s.p);
  s.arow = __shfl(s.a[0], k, 64);  // row 0
  s.erow = __shfl(s.e[0], k, 64);
  ElimStep<0>::run(s);
  const double myp = s.myp;
  const int badcol = s.badcol;
  const double* a = s.a;
  const double* e = s.e;
  const double rs = 1.0 / sqrt(myp);  // lane (., k): 1 / sqrt(p_k)
  const int rowk = prow(base + k);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * j + g;
    const double rsi = __shfl(rs, i, 64);
    if (i <= k) {
      const double lki = a[j] * rsi;  // L[k][i] = U[i][k] / sqrt(p_i)
      S[rowk + base + i] = lki;
      A[(int64_t)(base + k) * ld + base + i] = lki;
    }
    const double w = e[j] * rsi;                    // W[i][k] (zero above the diagonal)
    Wl[i * WP + k] = w;
    invD[(base + i) * NB + base + k] = w;
  }
  if (badcol >= 0 && lane == 0 && *info == 0) *info = col0 + base + badcol + 1;
}

// One workgroup (8 waves): L = chol(A_blk) in place (lower part of A), invD = L^-1 (dense
// 128 x 128, zero above the diagonal).  Blocked by 16 inside LDS; everything except the 16 x 16
// eliminations runs as 16 x 16 x 16 products on v_mfma_f64_16x16x4_f64.  The eliminations (factor16,
// one wave, ~2.7 us each) are the critical path, so wave 0 does nothing but that chain:
//   wave 0, step kb:   factor16(kb) | barrier | panel tile (kb+1, kb), trailing tile (kb+1, kb+1)
//   waves 1-7, step kb (one step behind, "bulk(kb - 1)"):  the other panel tiles of column kb - 1, row
//       block kb - 1 of the inverse (X[kb-1][jb] = -W sum_k L[kb-1][k] X[k][jb], earlier X rows read back
//       from invD through L2), the other trailing tiles of step kb - 1 | barrier
// One barrier per step; inside bulk() the seven waves order themselves through two LDS counters
// (panel tiles of the column finished / wave 0's sub-diagonal tile finished).  L and the inverse go to
// global memory tile by tile as they are produced.
constexpr int DIAG_THREADS = 512;  // wave 0: elimination chain; waves 1-7: everything else
constexpr int DIAG_BULK = DIAG_THREADS / 64 - 1;

__global__ __launch_bounds__(DIAG_THREADS) void potrf_diag_kernel(double* __restrict__ A, int64_t ld,
                                                         double* __restrict__ invD,
                                                         int* __restrict__ info, int col0,
                                                         unsigned long long* __restrict__ dbg,
                                                         int64_t strideA, int64_t strideInv) {
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
  __shared__ double S[S_DOUBLES];
  // the inverse as it grows: 16 x 16 block (k2, jb), jb <= k2, at Xl[k2 (k2 + 1) / 2 + jb] (row pitch 17).  Later row
  // blocks are built from the earlier ones; keeping them here (78 KiB) instead of reading them back through L2 takes
  // the store fence - the wait for the acknowledgement of every global store of a step - out of the waves' steps
  __shared__ double Xl[NBLK * (NBLK + 1) / 2][BS * WP];
  __shared__ int sub_ready;   // wave 0: sub-diagonal tiles (k + 1, k) finished for k < sub_ready
  __shared__ int panel_done;  // waves 1-7: 7 (k + 1) once every one of them has finished its panel tiles of column k
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  if (tid == 0) {
    sub_ready = 0;
    panel_done = 0;
  }
  // Wave 0 starts the elimination chain at once: it fetches the first 16 x 16 block straight into the registers of
  // factor16 (both triangles from the lower one).  Waves 1-7 meanwhile bring the block-lower part of the 128 x 128
  // block into LDS (coalesced 16-byte loads, all of a thread's loads in flight before its first LDS store) -
  // everything except block (0, 0), which wave 0 writes itself; the barrier of step 0 is the first point where
  // anybody reads what somebody else loaded.  (Wave 0 running its sub-diagonal step of step 0 ahead of that barrier, on
  // operands of its own, was measured: 30.9 instead of 29.5 us - the other waves' step 0, the heaviest, then starts
  // later and wave 0 waits for it at the next barrier.)
  d4_t blk = {0.0, 0.0, 0.0, 0.0};  // wave 0: diagonal block kb, in factor16's layout
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * j + fk;  // entry (i, fr), from the lower triangle
      blk[j] = A[(int64_t)(i > fr ? i : fr) * ld + (i > fr ? fr : i)];
    }
  } else {
    // the strictly-upper 16-blocks of the inverse are zero: in row r the contiguous columns from 16 ((r >> 4) + 1) on,
    // one predicated 16-byte store per row and lane (fire-and-forget, acknowledged while the loads are in flight)
#pragma unroll
    for (int i = 0; i < (NB - BS + DIAG_BULK - 1) / DIAG_BULK; ++i) {
      const int r = (wave - 1) + DIAG_BULK * i;
      const int c = ((r >> 4) + 1) * BS + 2 * lane;
      if (r < NB - BS && c < NB) *reinterpret_cast<d2_t*>(invD + r * NB + c) = d2_t{0.0, 0.0};
    }
    // rows 16 .. 127, one 1 KiB row per wave and load, dealt round-robin to the seven waves (16 rows each); only the
    // lower triangle is stored - the first readers of a diagonal 16-block (sub_chain, trailing_tile) take its upper
    // entries from the mirror position
    const int cc = lane * 2;
    d2_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = BS + (wave - 1) + DIAG_BULK * i;
      v[i] = (cc <= r) ? *reinterpret_cast<const d2_t*>(A + (int64_t)r * ld + cc) : d2_t{0.0, 0.0};
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = BS + (wave - 1) + DIAG_BULK * i;
      const int pr = prow(r) + cc;
      if (cc <= r) S[pr] = v[i][0];
      if (cc + 1 <= r) S[pr + 1] = v[i][1];
    }
  }
  lap(0);

  // A[ib][kb] <- A[ib][kb] * W_kb^T, to LDS and to the matrix in global memory
  auto panel_tile = [&](int kb, int ib) {
    const double* W = Xl[kb * (kb + 1) / 2 + kb];
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    const int ra = prow(ib * BS + fr) + kb * BS + fk;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ra + 4 * q], W[fr * WP + fk + 4 * q], acc, 0, 0, 0);  // B[k][j] = W[j][k]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[prow(ib * BS + fk + 4 * r) + kb * BS + fr] = acc[r];
      A[(int64_t)(ib * BS + fk + 4 * r) * ld + kb * BS + fr] = acc[r];
    }
  };
  // A[ib][jb] -= P_ib P_jb^T  (P = column block kb after the panel step); diagonal tiles in full (symmetric)
  auto trailing_tile = [&](int kb, int ib, int jb) {
    d4_t acc;
    int rc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = fk + 4 * r;
      rc[r] = prow(ib * BS + i) + jb * BS + fr;
      // a diagonal tile is symmetric and only its lower triangle is kept up to date
      acc[r] = S[(ib == jb && i < fr) ? prow(ib * BS + fr) + jb * BS + i : rc[r]];
    }
    const int ra = prow(ib * BS + fr) + kb * BS + fk, rb = prow(jb * BS + fr) + kb * BS + fk;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[ra + 4 * q], S[rb + 4 * q], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ib != jb || fk + 4 * r >= fr) S[rc[r]] = acc[r];
  };
  auto wait_for = [&](int* counter, int target) {
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto signal = [&](int* counter, int add) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(counter, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // wave 0, step kb: the sub-diagonal tile P = A[kb+1][kb] W^T and the trailing product of tile (kb+1, kb+1), without an
  // LDS round trip between them: P is computed transposed (P^T = W A^T), which makes its D registers at once the A and
  // the B operand of the trailing product (P[fr][fk + 4 q] = pt[q]); the result lands in factor16's layout and stays
  // in registers.  (Same products, same summation order as panel_tile / trailing_tile.)
  auto sub_chain_regs = [&](int kb, const d4_t& b, d4_t t) -> d4_t {
    const double* W = Xl[kb * (kb + 1) / 2 + kb];
    const int ib = kb + 1;
    d4_t pt = {0.0, 0.0, 0.0, 0.0};
    const int rb = prow(ib * BS + fr) + kb * BS + fk;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pt = __builtin_amdgcn_mfma_f64_16x16x4f64(W[fr * WP + fk + 4 * q], b[q], pt, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      S[rb + 4 * r] = pt[r];
      A[(int64_t)(ib * BS + fr) * ld + kb * BS + fk + 4 * r] = pt[r];
    }
    signal(&sub_ready, 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(-pt[q], pt[q], t, 0, 0, 0);
    return t;
  };
  auto sub_chain = [&](int kb) -> d4_t {
    const int ib = kb + 1;
    d4_t t, b;
    const int rb = prow(ib * BS + fr) + kb * BS + fk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = fk + 4 * r;  // entry (i, fr) of the symmetric tile: from the lower triangle
      t[r] = S[prow(ib * BS + (i > fr ? i : fr)) + ib * BS + (i > fr ? fr : i)];
      b[r] = S[rb + 4 * r];
    }
    return sub_chain_regs(kb, b, t);
  };
  // everything of step kb that is not on the elimination chain (waves 1-7)
  auto bulk = [&](int kb) {
    const int wb = wave - 1;
    const int base = kb * BS;
    const int jb = wb;  // this wave's column of the inverse row block (tiles jb < kb)
    if (kb + 2 + wb < NBLK) panel_tile(kb, kb + 2 + wb);
    signal(&panel_done, 1);
    if (jb < kb) {
      double bx[NBLK - 1][4];
#pragma unroll
      for (int kk = 0; kk < NBLK - 1; ++kk) {
        const int k2 = jb + kk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bx[kk][q] = (k2 < kb) ? Xl[k2 * (k2 + 1) / 2 + jb][(fk + 4 * q) * WP + fr] : 0.0;  // X[k2][jb]
      }
      const double* W = Xl[kb * (kb + 1) / 2 + kb];
      double wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = -W[fr * WP + fk + 4 * q];  // A operand: -W[i = fr][k = fk + 4 q]
      const int ra = prow(base + fr) + fk;
      d4_t T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < NBLK - 1; ++kk) {
        const int k2 = jb + kk;
        if (k2 < kb) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ra + k2 * BS + 4 * q], bx[kk][q], T, 0, 0, 0);
        }
      }
      // the D layout of T (row = fk + 4 r) is exactly the B-operand layout of k-step q = r
      d4_t X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) X = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q], T[q], X, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Xl[kb * (kb + 1) / 2 + jb][(fk + 4 * r) * WP + fr] = X[r];
        invD[(base + fk + 4 * r) * NB + jb * BS + fr] = X[r];
      }
    }
    if (kb + 1 < NBLK) {
      // trailing tiles (ib, jb), kb < jb <= ib, except (kb + 1, kb + 1) which wave 0 keeps on its chain; they
      // read the panel tiles of the other waves and wave 0's tile (kb + 1, kb)
      wait_for(&panel_done, DIAG_BULK * (kb + 1));
      wait_for(&sub_ready, kb + 1);
      const int m = NBLK - 1 - kb;
      const int ntile = m * (m + 1) / 2;
      for (int t = 1 + wb; t < ntile; t += DIAG_BULK) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        const int j = t - i * (i + 1) / 2;
        trailing_tile(kb, kb + 1 + i, kb + 1 + j);
      }
    }
  };

  for (int kb = 0; kb < NBLK; ++kb) {
    // The barrier of a step orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): nothing a wave writes to global
    // memory is read again inside this kernel (L and the inverse go out tile by tile; the inverse is also kept in Xl).
    if (wave == 0) {
      factor16(S, Xl[kb * (kb + 1) / 2 + kb], invD, A, ld, kb, info, col0, lane, blk);
      lap(4);
    } else if (kb > 0) {
      bulk(kb - 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    lap(1);
    if (wave == 0 && kb + 1 < NBLK) {
      blk = sub_chain(kb);
      lap(2);
    }
  }
  if (wave > 0) bulk(NBLK - 1);
  lap(3);
  if (dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
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
                       unsigned long long* dbg, const BatchShape& bs) {
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, 1, (unsigned)bs.count), dim3(DIAG_THREADS), 0, s, Ablk, ld, invD,
                     info, col0, dbg, bs.sMat, bs.sInv);
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
// 32), the chain 200 + 7.6 us per trailing tile row; the slice takes GPMI_SLICE_PCT % (default 100; 0: none) of the
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
  const double chain = 200.0 + 7.6 * rem;
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
    const int x = e ? std::atoi(e) : 60;
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
  // other 224.  Below GPMI_LOOKAHEAD_MIN (60) trailing tile rows the update is shorter than the panel chain on its 32 CUs
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
  const GemmBatch inplace{bs.count, bs.sMat, bs.sMat, bs.sInv};
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
