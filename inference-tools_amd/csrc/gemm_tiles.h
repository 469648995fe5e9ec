// Workgroup-tile bodies of the fp64 MFMA GEMM (gfx950), shared by the launch-per-product kernels of gemm_f64.hip and
// the persistent tile-task kernel of potrf_flow.hip.  One body = one C tile of one 256-thread workgroup:
//   staged_tile<OP, BKN, BM, BN>  register-staged operands, LDS double buffer (any tile shape, k-major B, in place)
//   dma128_tile<OP>               128 x 128, operands by LDS-DMA into a 4-stage ring
//   dma64_tile<OP>                64 x 64 on the same ring
// See gemm_f64.hip for the layouts, the MFMA operand maps and the measurements behind each choice.  A body takes
// resolved operand pointers (first element of the tile's rows at the first k of the contraction), the leading
// dimensions, the number of k slabs and the workgroup's LDS block; it leaves its last global stores in flight.
#pragma once
#include "gpmi_internal.h"

// C tiles are read once and written once per launch: those accesses are marked non-temporal so that they do not
// displace the operand panels (re-read by every tile of a strip) from the XCD's L2 (+0.5 % on the trailing update;
// -DGPMI_C_PLAIN builds the plain form for comparison)
#ifndef GPMI_C_PLAIN
#define GPMI_C_LOAD(p) __builtin_nontemporal_load(p)
#define GPMI_C_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define GPMI_C_LOAD(p) (*(p))
#define GPMI_C_STORE(v, p) (*(p) = (v))
#endif

namespace gemm_tiles {

// C-store flavour of a tile body.  CST_NT: non-temporal (launch-per-product kernels: the kernel boundary publishes the
// tile).  CST_SC1: agent-scope write-through stores (tile tasks of potrf_flow.hip: the tile is handed to other CUs
// inside the launch - with every byte stored sc1, each wave's `s_waitcnt vmcnt(0)` and a workgroup barrier in front of
// the flag, no release fence is needed; a release would write back every dirty line of the XCD's L2 per task).
constexpr int CST_NT = 0, CST_SC1 = 1;
// The sc1 stores are inline assembly (no builtin stores 16 bytes with a scope), and the compiler's hazard recognizer
// does not see an asm statement as a reader of MFMA results still in flight (XDL write -> VMEM read needs up to 19 wait
// states that the hardware does not interlock): the first stores of an epilogue wrote half-finished accumulators.
// An epilogue of asm stores therefore starts with the wait states itself.
template <int CST>
__device__ __forceinline__ void c_store_begin() {
  if (CST == CST_SC1) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}
template <int CST>
__device__ __forceinline__ void c_store(d2_t v, d2_t* p) {
  if (CST == CST_SC1)
    // (+ the wait states a store of more than 8 bytes needs before its data registers may be overwritten: the compiler
    // does not keep that hazard around an asm statement and re-used them in the next instruction)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else
    GPMI_C_STORE(v, p);
}
template <int CST>
__device__ __forceinline__ void c_store(double v, double* p) {
  if (CST == CST_SC1)
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else
    GPMI_C_STORE(v, p);
}

// Load flavour of a tile body (potrf_flow.hip's protocol experiments): LD_SC1 makes every load of matrix data an
// agent-scope (sc1) load - C tile, staged operands, LDS-DMA operands.
constexpr int LD_PLAIN = 0, LD_SC1 = 1;
template <int LD>
__device__ __forceinline__ d2_t c_load2(const d2_t* p) {
  if (LD == LD_SC1) {
    const double* q = reinterpret_cast<const double*>(p);
    return d2_t{__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
  }
  return GPMI_C_LOAD(p);
}
template <int LD>
__device__ __forceinline__ double c_load1(const double* p) {
  if (LD == LD_SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return GPMI_C_LOAD(p);
}
template <int LD>
__device__ __forceinline__ d2_t op_load2(const d2_t* p) {
  if (LD == LD_SC1) {
    const double* q = reinterpret_cast<const double*>(p);
    return d2_t{__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
  }
  return *p;
}

constexpr int BK = 16;
constexpr int LDS_STRIDE = 18;  // doubles per staged row (16 + 2 pad)
// operand slabs requested ahead of the MFMAs by the tiles smaller than 128 x 128 (see staged_tile)
#ifndef GPMI_SMALL_PF
#define GPMI_SMALL_PF 4
#endif

template <int BKN, int BM, int BN>
constexpr int staged_lds_doubles() {
  return 2 * (BM * LDS_STRIDE + (BKN ? BK * (BN + 16) : BN * LDS_STRIDE));
}

// BKN = 0: B is (cols x k), K-contiguous ("NT");  BKN = 1: B is (k x cols), row-major ("NN").
// BM x BN is the workgroup tile (128 or 64 each): the small tiles serve launches with few 128-tiles
// or K = 128 (panel TRSM, inner panel updates, solves with few right-hand sides), which are bound
// by the time of ONE tile rather than by throughput.
// Ag: row 0 of the tile's A rows at the first k; Bg: row 0 of the tile's B rows at the first k (BKN: first k row, first
// column of the tile); Cg: the tile's first element; nk: 16-deep slabs.
// BTRI (round 4; BKN = 0, BN = 128 = the whole contraction, OP_ASSIGN): B is LOWER TRIANGULAR, B[j][k] = 0 for k > j - the
// panel TRSM as a product with the inverse of the diagonal block (regression.py:241's dpotrf does a triangular solve
// there).  Column block c (16 columns) of the result needs the slabs kt <= c only: the MFMAs of the others are skipped -
// the very same sums, bit for bit (the skipped terms are products with exact zeros) - and a wave takes the column
// blocks wc, wc + 2, wc + 4, wc + 6 instead of four neighbouring ones, so that its share of what is left (16 of 32
// slab steps for wc = 0, 20 for wc = 1) is even: 44 % fewer MFMAs, 37 % less MFMA time per workgroup.
template <int OP, int BKN, int BM, int BN, int CST = CST_NT, int LD = LD_PLAIN, bool BTRI = false>
__device__ __forceinline__ void staged_tile(const double* __restrict__ Ag, const double* __restrict__ Bg, double* Cg,
                                            int64_t lda, int64_t ldb, int64_t ldc, int nk, double* smem) {
  static_assert(!BTRI || (BKN == 0 && BN == 128 && OP == OP_ASSIGN), "BTRI: the in-place panel TRSM only");
  constexpr int CSTEP = BTRI ? 32 : 16;  // columns between a wave's consecutive MFMA tiles
  constexpr int TM = BM / 32, TN = BN / 32;  // 16 x 16 MFMA tiles per wave (2 x 2 waves)
  constexpr int LDS_STRIDE_KN = BN + 16;
  constexpr int A_DOUBLES = BM * LDS_STRIDE;
  constexpr int B_DOUBLES = BKN ? BK * LDS_STRIDE_KN : BN * LDS_STRIDE;
  constexpr int BUF_DOUBLES = A_DOUBLES + B_DOUBLES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // global -> register staging: 4 x 16-byte chunks per operand per thread (8 threads cover a row)
  const int lrow = tid >> 3, lkc = (tid & 7) * 2;
  // swizzled position of this thread's 16-byte piece: rows 4..11 (mod 16) swap their 32-byte chunk pairs
  // (lrow + 32 i has the same row & 15 for every i)
  const int lkc_sw = lkc ^ ((((lrow & 15) >= 4) && ((lrow & 15) < 12)) ? 4 : 0);
  // k-major B: BN / 2 chunks per k-row, 512 / BN k-rows per pass
  const int nrow = tid / (BN / 2), nnc = (tid % (BN / 2)) * 2;
  // The small tiles serve launches that are bound by the time of ONE workgroup: a K loop of 8 .. 32 slabs, each a
  // round trip to L2 or beyond (0.7 us) with a few MFMAs behind it.  They keep PF slabs in flight in a ring of staging
  // registers (slot = slab % PF; the loop is unrolled PF times so that the slots are static) instead of one; the
  // 128 x 128 tile (throughput-bound, 128 accumulator registers) keeps the single slab.
  constexpr int PF = (BM * BN < 128 * 128) ? GPMI_SMALL_PF : 1;
  d2_t ra[PF][TM], rb[PF][TN];
  // per-lane byte offsets (32-bit: a tile spans < 2^32 bytes) against uniform slab bases: the loads take the
  // SGPR-base + VGPR-offset form, no 64-bit address arithmetic per request
  unsigned voa[TM], vob[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) voa[i] = (unsigned)(((int64_t)(lrow + 32 * i) * lda + lkc) * 8);
#pragma unroll
  for (int i = 0; i < TN; ++i)
    vob[i] = BKN ? (unsigned)(((int64_t)(nrow + (512 / BN) * i) * ldb + nnc) * 8)
                 : (unsigned)(((int64_t)(lrow + 32 * i) * ldb + lkc) * 8);
  auto gload = [&](int k0, int slot) {
    const char* ab = reinterpret_cast<const char*>(Ag + k0);
    const char* bb = reinterpret_cast<const char*>(BKN ? Bg + (int64_t)k0 * ldb : Bg + k0);
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[slot][i] = op_load2<LD>(reinterpret_cast<const d2_t*>(ab + voa[i]));
#pragma unroll
    for (int i = 0; i < TN; ++i) rb[slot][i] = op_load2<LD>(reinterpret_cast<const d2_t*>(bb + vob[i]));
  };
  auto sstore = [&](int buf, int slot) {
    double* sa = smem + buf * BUF_DOUBLES;
    double* sb = sa + A_DOUBLES;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      *reinterpret_cast<d2_t*>(sa + (lrow + 32 * i) * LDS_STRIDE + lkc_sw) = (OP == OP_SUB) ? -ra[slot][i] : ra[slot][i];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (BKN)
        *reinterpret_cast<d2_t*>(sb + (nrow + (512 / BN) * i) * LDS_STRIDE_KN + nnc) = rb[slot][i];
      else
        *reinterpret_cast<d2_t*>(sb + (lrow + 32 * i) * LDS_STRIDE + lkc_sw) = rb[slot][i];
    }
  };

  const int fr = lane & 15, fk = lane >> 4;
  const int wcol = BTRI ? wc * 16 : wc * (BN / 2);  // first column of the wave's first MFMA tile
  Cg += (int64_t)(wr * (BM / 2)) * ldc + wcol;
  const int sw = (fr >= 4 && fr < 12) ? 1 : 0;
  const int a_off = (wr * (BM / 2) + fr) * LDS_STRIDE + ((fk ^ sw) << 2);
  const int b_off = BKN ? 4 * fk * LDS_STRIDE_KN + wc * (BN / 2) + fr
                        : (wcol + fr) * LDS_STRIDE + ((fk ^ sw) << 2);

  // OP_SUB: the accumulators start from the C tile and the A operand is negated on its way into LDS,
  // so C - A B^T comes out of the MFMA chain itself and the epilogue is stores only.  The first operand slab is
  // requested BEFORE the 64 C loads and the first 16-deep step is peeled off the loop: its MFMAs wait for
  // their own accumulator tile only (vmcnt counts in order, the slab is older, the next slab's prefetch newer),
  // so the C tile streams in under the first step instead of in front of it.  After that step nothing is
  // pending on an accumulator register, which keeps every vmcnt wait out of the loop proper (with a pending
  // C load at loop entry the compiler puts s_waitcnt vmcnt(0) in front of the MFMAs of the last
  // accumulators INSIDE the loop, i.e. a wait for the prefetch just issued, in every step).
  // requests beyond the last slab repeat it (an unconditional request keeps the loop free of branches around loads,
  // which is what lets the compiler count the loads in flight instead of waiting for all of them)
  const int klast = (nk - 1) * BK;
  auto gload_clamped = [&](int kt, int slot) { gload(kt < nk ? kt * BK : klast, slot); };
  // PF == 1 (128 x 128 tiles): slab 0 is requested BEFORE the C loads and the first step is peeled, see above.
  // PF > 1: the C tile (16 loads per lane at most) goes first, so that the wait for slab 0 covers it and the loop
  // starts in the state it has at its back edge (PF - 1 slabs in flight, nothing else).
  if (PF == 1) gload(0, 0);
  d4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (OP == OP_SUB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = c_load1<LD>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + j * CSTEP + fr]);
      } else {
        acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
      }
    }
  if (PF > 1) {
    gload(0, 0);
#pragma unroll
    for (int p = 1; p < PF; ++p) gload_clamped(p, p);
  }
  sstore(0, 0);
  __syncthreads();
  // step kt: slab kt is in LDS buffer kt & 1, slabs kt + 1 .. kt + PF - 1 are in flight or in their slots; slot
  // kt % PF (slab kt went to LDS at the end of step kt - 1) takes the request for slab kt + PF
  auto kstep = [&](int kt, int slot_free, int slot_next) {
    const int cur = kt & 1;
    if (PF == 1) {
      if (kt + 1 < nk) gload((kt + 1) * BK, 0);
    } else {
      gload_clamped(kt + PF, slot_free);
    }
    const double* sa = smem + cur * BUF_DOUBLES;
    const double* sb = sa + A_DOUBLES;
    // lane (fr, fk) supplies k = 4 fk + q of the slab to MFMA step q (same map for A and B)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      d2_t a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t)
        a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * LDS_STRIDE + 2 * h);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        if (BKN)
          b[t] = d2_t{sb[b_off + (2 * h) * LDS_STRIDE_KN + t * 16],
                      sb[b_off + (2 * h + 1) * LDS_STRIDE_KN + t * 16]};
        else
          b[t] = *reinterpret_cast<const d2_t*>(sb + b_off + t * CSTEP * LDS_STRIDE + 2 * h);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (BTRI && kt > 2 * j + wc) continue;  // column block 2 j + wc of a lower-triangular B: zero beyond slab 2 j + wc
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
      }
    }
    // PF > 1: unconditional (the last step stages a repeat of the last slab, which nobody reads)
    if (PF > 1 || kt + 1 < nk) sstore(cur ^ 1, slot_next);
    __syncthreads();
  };
  if (PF == 1) {
    kstep(0, 0, 0);
    for (int kt = 1; kt < nk; ++kt) kstep(kt, 0, 0);
  } else {
    // groups of PF steps with static slots; the only branches are exits
    for (int kb = 0; kb < nk; kb += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (kb + u >= nk) break;
        kstep(kb + u, u, (1 + u) % PF);
      }
    }
  }

  // epilogue: stores only; each instruction covers 4 rows x 128 contiguous bytes
  c_store_begin<CST>();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_store<CST>(acc[i][j][r], &Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + j * CSTEP + fr]);
}

// ---- 128 x 128 tiles with K-contiguous operands: operand ring fed by LDS-DMA ---------------------------------------------------------------
// Same 128 x 128 tile, C -= A B^T, but the operand slabs go from global memory straight into a ring of DMA_STAGES
// 8-deep LDS stages with global_load_lds_dwordx4 (no staging registers, no LDS store instructions), three stages
// ahead of the MFMAs instead of one 16-deep slab.  The loads and their waits are inline assembly: the compiler
// orders every ds_read behind `s_waitcnt vmcnt(0)` when it knows of an LDS-DMA in flight, which would drain the ring
// at every barrier.  Piece (row r, k pair q) of a stage lives in 16-byte slot 4 r + ((q + 2 (r >> 2)) & 3): conflict-
// free ds_read_b128 fragments (lane (fr, fk) reads pair fk of row fr: the four 4-lane quads of a read group land on
// four different slot positions).  The A fragments are negated after the LDS read (the DMA path cannot negate on the way in).
constexpr int DMA_BK = 8;
#ifndef GPMI_DMA_STAGES
#define GPMI_DMA_STAGES 4
#endif
constexpr int DMA_STAGES = GPMI_DMA_STAGES;
constexpr int DMA_OP_DOUBLES = 128 * DMA_BK;  // one operand of one stage: 8 KiB
constexpr int DMA128_LDS_DOUBLES = DMA_STAGES * 2 * DMA_OP_DOUBLES;
constexpr int DMA64_LDS_DOUBLES = DMA_STAGES * 2 * 64 * DMA_BK;

__device__ inline void dma_wait(int newer) {
  // wait until at most `newer` younger vector-memory operations of this wave are outstanding
  if (newer >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (newer >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (newer >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// nk: 8-deep stages
template <int OP, int CST = CST_NT, int LD = LD_PLAIN>
__device__ __forceinline__ void dma128_tile(const double* __restrict__ Ag, const double* __restrict__ Bg, double* Cg,
                                            int64_t lda, int64_t ldb, int64_t ldc, int nk, double* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  // this thread's two pieces per operand and stage: p = i * 256 + tid -> row p >> 2, slot p & 3
  const int row0 = tid >> 2, slot = tid & 3;
  const int q0 = (slot - 2 * ((row0 >> 2) & 3)) & 3;          // rows row0 and row0 + 64 have the same (row >> 2) & 3
  const double* a_src0 = Ag + (int64_t)row0 * lda + 2 * q0;
  const double* a_src1 = a_src0 + (int64_t)64 * lda;
  // B rows (= columns of C) are permuted on their way into LDS: LDS row 16 t + fr of a wave column block holds
  // column 2 fr + t (t < 2) or 32 + 2 fr + t - 2 of that block, so that MFMA tiles (j, j + 1) of a lane are two
  // ADJACENT columns of C: the C tile is read and written with 16-byte accesses, 256 contiguous bytes per row and
  // instruction, half as many instructions as the 8-byte form
  auto bperm = [](int R) {
    const int t = (R >> 4) & 3, f = R & 15;
    return (R & 64) + ((t & 2) << 4) + 2 * f + (t & 1);
  };
  const double* b_src0 = Bg + (int64_t)bperm(row0) * ldb + 2 * q0;
  const double* b_src1 = Bg + (int64_t)bperm(row0 + 64) * ldb + 2 * q0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;  // LDS byte address of the ring (address space 3 pointers are offsets)
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
  auto issue = [&](int st, int k0) {
    const unsigned base = lds0 + (unsigned)st * (2 * DMA_OP_DOUBLES * 8) + wave_off;
    const double* p0 = a_src0 + k0;
    const double* p1 = a_src1 + k0;
    const double* p2 = b_src0 + k0;
    const double* p3 = b_src1 + k0;
    if (LD == LD_SC1)
      asm volatile(
          "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1\n\t"
          "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\t"
          "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\t"
          "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off sc1"
          :
          : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(base), "s"(base + 256 * 16), "s"(base + DMA_OP_DOUBLES * 8),
            "s"(base + DMA_OP_DOUBLES * 8 + 256 * 16)
          : "memory");
    else
    asm volatile(
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"
        :
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(base), "s"(base + 256 * 16), "s"(base + DMA_OP_DOUBLES * 8),
          "s"(base + DMA_OP_DOUBLES * 8 + 256 * 16)
        : "memory");
  };
  for (int st = 0; st < DMA_STAGES - 1 && st < nk; ++st) issue(st, st * DMA_BK);

  Cg += (int64_t)(wr * 64) * ldc + wc * 64;
  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2_t cv = d2_t{0.0, 0.0};
        if (OP == OP_SUB)
          cv = c_load2<LD>(reinterpret_cast<const d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + jp * 32 + 2 * fr]));
        acc[i][2 * jp][r] = cv[0];
        acc[i][2 * jp + 1][r] = cv[1];
      }

  const int rslot = (fk + 2 * (fr >> 2)) & 3;
  const int a_off = ((wr * 64 + fr) * 4 + rslot) * 2;                      // doubles
  const int b_off = DMA_OP_DOUBLES + ((wc * 64 + fr) * 4 + rslot) * 2;
  auto stage = [&](int kt) {
    const int ahead = nk - 1 - kt;  // stages issued after this one
    // the first stages also have the 64 C loads behind them: any vmcnt <= 63 covers the stage (in-order return)
    dma_wait(4 * (ahead < DMA_STAGES - 2 ? ahead : DMA_STAGES - 2));
    __syncthreads();
#ifdef GPMI_DMA_ISSUE_EARLY
    if (kt + DMA_STAGES - 1 < nk) issue((kt + DMA_STAGES - 1) % DMA_STAGES, (kt + DMA_STAGES - 1) * DMA_BK);
#endif
    const double* sa = smem + (kt % DMA_STAGES) * 2 * DMA_OP_DOUBLES;
    d2_t a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * 4 * 2);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const d2_t*>(sa + b_off + t * 16 * 4 * 2);
#ifndef GPMI_DMA_ISSUE_EARLY
    if (kt + DMA_STAGES - 1 < nk) issue((kt + DMA_STAGES - 1) % DMA_STAGES, (kt + DMA_STAGES - 1) * DMA_BK);
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][h], b[j][h], acc[i][j], 0, 0, OP == OP_SUB ? 1 : 0);  // f64 MFMA: the BLGP field is neg[A, B, C]: C - A B^T
  };
  // first stage peeled off the loop: its MFMAs wait for their own accumulator tile only, the C tile streams in
  // under them instead of in front of the loop
  stage(0);
  int kt = 1;
  for (; kt + 3 < nk; kt += 4) {  // four stages per trip: the ring slot of a stage is a compile-time offset from kt's
    stage(kt);
    stage(kt + 1);
    stage(kt + 2);
    stage(kt + 3);
  }
  for (; kt < nk; ++kt) stage(kt);
  c_store_begin<CST>();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        c_store<CST>((d2_t{acc[i][2 * jp][r], acc[i][2 * jp + 1][r]}),
                     reinterpret_cast<d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + jp * 32 + 2 * fr]));
}

// ---- 256 x 128 tiles on the same ring: 8 waves (4 x 2 of 64 x 64), one workgroup per CU -----------------------------------------------
// The tile VERDICT r03 asked to be tried: the B slab of a stage is shared by four wave rows instead of two (operand
// bytes per flop x 0.75), at the price of one barrier domain per CU (the two 128 x 128 workgroups of a CU run their
// stages in anti-phase, one's barrier under the other's MFMAs).  Three 16-byte pieces per thread and stage (two of A's
// 256 rows, one of B's 128), 24 KiB per stage.  skip_top: the wave rows 0-1 (the upper 128 rows of the tile) take part
// in the ring but neither compute nor store (the tile on the diagonal of a lower-triangular product).
constexpr int DMA256_STAGE_DOUBLES = 3 * DMA_OP_DOUBLES;
constexpr int DMA256_LDS_DOUBLES = DMA_STAGES * DMA256_STAGE_DOUBLES;

template <int OP, int CST = CST_NT, int LD = LD_PLAIN>
__device__ __forceinline__ void dma256_tile(const double* __restrict__ Ag, const double* __restrict__ Bg, double* Cg,
                                            int64_t lda, int64_t ldb, int64_t ldc, int nk, double* smem, bool skip_top) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const int row0 = tid >> 2, slot = tid & 3;                   // 0 .. 127
  const int q0 = (slot - 2 * ((row0 >> 2) & 3)) & 3;
  const double* a_src0 = Ag + (int64_t)row0 * lda + 2 * q0;
  const double* a_src1 = a_src0 + (int64_t)128 * lda;
  auto bperm = [](int R) {
    const int t = (R >> 4) & 3, f = R & 15;
    return (R & 64) + ((t & 2) << 4) + 2 * f + (t & 1);
  };
  const double* b_src0 = Bg + (int64_t)bperm(row0) * ldb + 2 * q0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
  auto issue = [&](int st, int k0) {
    const unsigned base = lds0 + (unsigned)st * (DMA256_STAGE_DOUBLES * 8) + wave_off;
    const double* p0 = a_src0 + k0;
    const double* p1 = a_src1 + k0;
    const double* p2 = b_src0 + k0;
    asm volatile(
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off"
        :
        : "v"(p0), "v"(p1), "v"(p2), "s"(base), "s"(base + DMA_OP_DOUBLES * 8), "s"(base + 2 * DMA_OP_DOUBLES * 8)
        : "memory");
  };
  for (int st = 0; st < DMA_STAGES - 1 && st < nk; ++st) issue(st, st * DMA_BK);

  const bool idle = skip_top && wr < 2;  // wave-uniform
  Cg += (int64_t)(wr * 64) * ldc + wc * 64;
  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2_t cv = d2_t{0.0, 0.0};
        if (OP == OP_SUB && !idle)
          cv = c_load2<LD>(reinterpret_cast<const d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + jp * 32 + 2 * fr]));
        acc[i][2 * jp][r] = cv[0];
        acc[i][2 * jp + 1][r] = cv[1];
      }

  const int rslot = (fk + 2 * (fr >> 2)) & 3;
  const int a_off = ((wr * 64 + fr) * 4 + rslot) * 2;
  const int b_off = 2 * DMA_OP_DOUBLES + ((wc * 64 + fr) * 4 + rslot) * 2;
  auto wait3 = [](int newer) {
    if (newer >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (newer >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto stage = [&](int kt) {
    const int ahead = nk - 1 - kt;
    wait3(3 * (ahead < DMA_STAGES - 2 ? ahead : DMA_STAGES - 2));
    __syncthreads();
    const double* sa = smem + (kt % DMA_STAGES) * DMA256_STAGE_DOUBLES;
    d2_t a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * 4 * 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const d2_t*>(sa + b_off + t * 16 * 4 * 2);
    if (kt + DMA_STAGES - 1 < nk) issue((kt + DMA_STAGES - 1) % DMA_STAGES, (kt + DMA_STAGES - 1) * DMA_BK);
    if (!idle) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][h], b[j][h], acc[i][j], 0, 0, OP == OP_SUB ? 1 : 0);
    }
  };
  stage(0);
  int kt = 1;
  for (; kt + 3 < nk; kt += 4) {
    stage(kt);
    stage(kt + 1);
    stage(kt + 2);
    stage(kt + 3);
  }
  for (; kt < nk; ++kt) stage(kt);
  if (idle) return;
  c_store_begin<CST>();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        c_store<CST>((d2_t{acc[i][2 * jp][r], acc[i][2 * jp + 1][r]}),
                     reinterpret_cast<d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + jp * 32 + 2 * fr]));
}

// ---- 64 x 64 tiles on the same ring ---------------------------------------------------------------------------------
// The remainders of split launches, the narrow look-ahead updates (fewer than 384 tiles) and the tail's outer updates
// run 64 x 64 tiles; with the register-staged kernel they reached 40-47 TFLOP/s at K = 512.  Same scheme as
// dma128_tile at half the edge: one 16-byte piece per operand, stage and thread (64 rows x 4 pieces), 2 x 2 waves
// of 32 x 32 (2 x 2 MFMA tiles), 8 KiB per stage.  The MFMAs take the k of a stage in the same groups as the 128 x 128
// ring kernel ({0,2,4,6}, {1,3,5,7}).
template <int OP, int CST = CST_NT, int LD = LD_PLAIN>
__device__ __forceinline__ void dma64_tile(const double* __restrict__ Ag, const double* __restrict__ Bg, double* Cg,
                                           int64_t lda, int64_t ldb, int64_t ldc, int nk, double* smem) {
  constexpr int OPD = 64 * DMA_BK;  // doubles of one operand of one stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  // this thread's piece per operand and stage: row tid >> 2, slot tid & 3 (see dma128_tile for the slot swizzle)
  const int row0 = tid >> 2, slot = tid & 3;
  const int q0 = (slot - 2 * ((row0 >> 2) & 3)) & 3;
  const double* a_src = Ag + (int64_t)row0 * lda + 2 * q0;
  // LDS row 16 t + f of a wave's 32-column block holds column 2 f + t: the two MFMA tiles of a lane are adjacent columns
  const int bcol = (row0 & 32) + 2 * (row0 & 15) + ((row0 >> 4) & 1);
  const double* b_src = Bg + (int64_t)bcol * ldb + 2 * q0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
  auto issue = [&](int st, int k0) {
    const unsigned base = lds0 + (unsigned)st * (2 * OPD * 8) + wave_off;
    const double* p0 = a_src + k0;
    const double* p1 = b_src + k0;
    if (LD == LD_SC1)
      asm volatile(
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1"
          :
          : "v"(p0), "v"(p1), "s"(base), "s"(base + OPD * 8)
          : "memory");
    else
    asm volatile(
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
        :
        : "v"(p0), "v"(p1), "s"(base), "s"(base + OPD * 8)
        : "memory");
  };
  for (int st = 0; st < DMA_STAGES - 1 && st < nk; ++st) issue(st, st * DMA_BK);

  Cg += (int64_t)(wr * 32) * ldc + wc * 32;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d2_t cv = d2_t{0.0, 0.0};
      if (OP == OP_SUB)
        cv = c_load2<LD>(reinterpret_cast<const d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + 2 * fr]));
      acc[i][0][r] = cv[0];
      acc[i][1][r] = cv[1];
    }
  const int rslot = (fk + 2 * (fr >> 2)) & 3;
  const int a_off = ((wr * 32 + fr) * 4 + rslot) * 2;  // doubles
  const int b_off = OPD + ((wc * 32 + fr) * 4 + rslot) * 2;
  auto wait2 = [](int newer) {  // at most `newer` younger vector-memory operations outstanding (2 per stage)
    if (newer >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (newer >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (newer >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto stage = [&](int kt) {
    const int ahead = nk - 1 - kt;
    wait2(2 * (ahead < DMA_STAGES - 2 ? ahead : DMA_STAGES - 2));
    __syncthreads();
    const double* sa = smem + (kt % DMA_STAGES) * 2 * OPD;
    d2_t a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * 4 * 2);
      b[t] = *reinterpret_cast<const d2_t*>(sa + b_off + t * 16 * 4 * 2);
    }
    if (kt + DMA_STAGES - 1 < nk) issue((kt + DMA_STAGES - 1) % DMA_STAGES, (kt + DMA_STAGES - 1) * DMA_BK);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][h], b[j][h], acc[i][j], 0, 0, OP == OP_SUB ? 1 : 0);
  };
  stage(0);
  int kt = 1;
  for (; kt + 3 < nk; kt += 4) {
    stage(kt);
    stage(kt + 1);
    stage(kt + 2);
    stage(kt + 3);
  }
  for (; kt < nk; ++kt) stage(kt);
  c_store_begin<CST>();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      c_store<CST>((d2_t{acc[i][0][r], acc[i][1][r]}),
                   reinterpret_cast<d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * ldc + 2 * fr]));
}

// ---- flag hooks of the flag-ordered factorisation (potrf_flow.hip) ------------------------------------------------
// A launch on the chain stream publishes what the launch BEFORE it in that stream produced (the kernel boundary has
// made those stores visible device-wide) and waits for the tile-task kernel to have brought its own C tile up to date.
constexpr int FLOW_SPIN_LIMIT = 1 << 20;  // polls of ~1 us (an s_sleep and a few L2-bypassing loads)

__device__ __forceinline__ int flow_ld(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by every thread of the workgroup at the top of a chain kernel
__device__ __forceinline__ void flow_hook_enter(const FlowHook& h) {
  if (h.trace && blockIdx.x == 0 && threadIdx.x == 0) h.trace[0] = __builtin_amdgcn_s_memrealtime();
  if (h.pub && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    __hip_atomic_store(h.pub, h.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (h.wait) {
    if (threadIdx.x == 0) {
      int spins = 0;
      unsigned long long t0 = 0;
      if (h.wait_ticks) t0 = __builtin_amdgcn_s_memrealtime();
      while (flow_ld(h.wait) < h.wait_val) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 63) == 0 && flow_ld(h.abort)) break;
        if (spins > FLOW_SPIN_LIMIT) {
          __hip_atomic_store(h.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (h.info) atomicCAS(h.info, 0, GPMI_INFO_FLOW_TIMEOUT);  // the chain launches of the tile-task factorisation
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
}

}  // namespace gemm_tiles
