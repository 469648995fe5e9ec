// fp64 MFMA GEMM for gfx950:  C (op)= A * B^T  with both operands K-contiguous ("NT").
//
// This is the one dense contraction of the GP path: the trailing SYRK/GEMM update of the blocked
// Cholesky factorisation (numpy.linalg.cholesky at regression.py:241,537,555), the panel TRSM done as
// a product with the inverted diagonal block, and the triangular solves with many right-hand sides
// of the batched predict (regression.py:213, 447).
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA tiles of
// v_mfma_f64_16x16x4_f64, 128 accumulator VGPRs), BK = 16, LDS double-buffered (72 KiB -> 2
// workgroups per CU = 2 waves per SIMD).  LDS rows are 16 doubles + 16 bytes of padding (144 B = 9
// slots of 16 B) and lane (row fr, fk) owns the four consecutive k = 4 fk .. 4 fk + 3 of a 16-deep
// slab, fetched as two ds_read_b128.  The 32-byte chunk c of row r is stored at chunk position
// c ^ (4 <= (r & 15) <= 11): with the 9-slot pitch every 16-lane group of a ds_read_b128 then hits 16
// distinct slots (conflict-free at 256 B/clk; the compiler's ds_read2_b64 form of the naive layout
// ran at half rate with 40 % conflict cycles).
//
// v_mfma_f64_16x16x4_f64 operand maps (verified by tools/mfma_probe.hip on MI355X):
//   A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
//   D: 4 values per lane, D[row = (l >> 4) + 4 r][col = l & 15]
// It issues every 64 cycles per SIMD: 256 CUs x 4 SIMDs x 2048 FLOP / 64 clk x 2.4 GHz = 78.6 TFLOP/s.
#include <cstdlib>

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

namespace {

constexpr int BK = 16;
constexpr int LDS_STRIDE = 18;  // doubles per staged row (16 + 2 pad)
// operand slabs requested ahead of the MFMAs by the tiles smaller than 128 x 128 (see gemm_nt_kernel)
#ifndef GPMI_SMALL_PF
#define GPMI_SMALL_PF 4
#endif

struct GemmArgs {
  double* C;
  const double* A;
  const double* B;
  int64_t ldc, lda, ldb;
  int ntr, ntc, k;
  // 1 (TILES_LOWER): contraction starts at k = ti * BM (operands are zero before it);
  // 2: contraction ends at k = (tj + 1) * BN (B is lower triangular: B[j][k] = 0 for k > j)
  int kskip;
  // optional wall-clock stamps of this launch (s_memrealtime, 100 MHz; 16 words, see the kernel): per-launch
  // durations for the roofline without HIP events in the stream (event records between the look-ahead
  // streams slowed the factorisation 2x)
  unsigned long long* stamp;
  int64_t sC, sA, sB;  // batch strides (doubles): problem blockIdx.z works on C + z sC, A + z sA, B + z sB
  // split > 0 (64 x 64 kernels only): this launch covers the 128 x 128 tiles [tile_base, ..) of the
  // (ntr / 2) x (ntc / 2) tile grid, four workgroups per tile (the tail of a launch whose other tiles run as
  // full 128 x 128 tiles, see launch_gemm_nt_split).  split == 0: the launch covers the tiles [tile_base, tile_base +
  // gridDim.x) of the logical tile list (launch_gemm_nt_range; 0 for whole products)
  int split, tile_base;
};

// B stored k-major: 16 rows of BN + 16 pad (rows 16 doubles apart mod 32)

// linear workgroup id -> (ti, tj), with an XCD-aware remap: workgroups b and b + 8 run on the same
// XCD (round-robin dispatch), so each XCD is handed a contiguous chunk of the logical tile list and
// neighbouring tiles (which share operand panels) hit the same L2.  Speed only, never correctness.
__device__ inline int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Logical tile order.  TILES_RECT: row-major.  TILES_LOWER (tiles ti >= tj, tj < ntc): column strips
// of 8 tile columns, each strip walked row by row, so that any 64 consecutive tiles — what one XCD
// (32 CUs x 2 workgroups) holds at a time — form an 8 x 8 block sharing 8 + 8 operand panels in
// that XCD's L2 instead of streaming every B panel from the Infinity Cache.
template <int TILES>
__device__ inline void tile_of(int id, int ntr, int ntc, int& ti, int& tj) {
  if (TILES == TILES_RECT) {
    if (ntr <= 16 && ntc > ntr) {
      // few tile rows, many columns (the many-right-hand-side solves: 8 x 124 tiles): column-major, so that the
      // contiguous chunk of an XCD is a set of COLUMNS - its B panels (rows of L) are fetched once by that XCD alone,
      // the few A panels are shared by all; row-major, every XCD streamed all of L's panel through its L2
      tj = id / ntr;
      ti = id - tj * ntr;
    } else {
      ti = id / ntc;
      tj = id - ti * ntc;
    }
  } else {
    int c0 = 0;
    for (;;) {
      const int w = (ntc - c0 < 8) ? ntc - c0 : 8;         // strip width
      const int tri = w * (w + 1) / 2;                      // rows c0 .. c0 + w - 1 (triangular top)
      const int cnt = tri + (ntr - c0 - w) * w;
      if (id < cnt) {
        if (id < tri) {
          int r = 0;
          while ((r + 1) * (r + 2) / 2 <= id) ++r;
          ti = c0 + r;
          tj = c0 + id - r * (r + 1) / 2;
        } else {
          const int rem = id - tri;
          ti = c0 + w + rem / w;
          tj = c0 + rem % w;
        }
        return;
      }
      id -= cnt;
      c0 += w;
    }
  }
}

// BKN = 0: B is (cols x k), K-contiguous ("NT");  BKN = 1: B is (k x cols), row-major ("NN").
// BM x BN is the workgroup tile (128 or 64 each): the small tiles serve launches with few 128-tiles
// or K = 128 (panel TRSM, inner panel updates, solves with few right-hand sides), which are bound
// by the time of ONE tile rather than by throughput.
template <int TILES, int OP, int BKN, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  constexpr int TM = BM / 32, TN = BN / 32;  // 16 x 16 MFMA tiles per wave (2 x 2 waves)
  constexpr int LDS_STRIDE_KN = BN + 16;
  constexpr int A_DOUBLES = BM * LDS_STRIDE;
  constexpr int B_DOUBLES = BKN ? BK * LDS_STRIDE_KN : BN * LDS_STRIDE;
  constexpr int BUF_DOUBLES = A_DOUBLES + B_DOUBLES;
  __shared__ double smem[2 * BUF_DOUBLES];  // [buffer][A | B]
  int ti, tj;
  // k-skipped launches have tiles of very different length (128 (ntr - ti) k-steps): deal them out
  // round-robin over the XCDs instead of in contiguous chunks, or the XCD holding the long tiles ends last
  const int wid = g.kskip ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  if (g.split) {
    int bi, bj;
    tile_of<TILES>(g.tile_base + (wid >> 2), g.ntr >> 1, g.ntc >> 1, bi, bj);
    ti = 2 * bi + ((wid >> 1) & 1);
    tj = 2 * bj + (wid & 1);
  } else {
    tile_of<TILES>(g.tile_base + wid, g.ntr, g.ntc, ti, tj);
  }

  const int tid = threadIdx.x;
  // per-launch timing without atomics (16 words per launch): the first eight workgroups store their start
  // time in words 0..7, every workgroup stores its end time in word 8 + XCC id.  Workgroups of one XCD share
  // an L2, so the word keeps the value of whoever finished last there; the host takes min / max.  (Device-scope
  // atomicMin / atomicMax on one address from ~8000 workgroups cost 1 ms per step, and an atomic issued
  // at the start sat in front of the first operand loads in the wave's in-order memory queue.)
  const bool stamp_first = g.stamp && tid == 0 && blockIdx.x < 8 && blockIdx.z == 0;
  if (stamp_first) g.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  // eight workgroups from the middle of the launch (the chip fully loaded, the clock settled) time their own
  // lifetime in shader cycles and in wall time: the clock the launch runs at
  const unsigned mid = gridDim.x >> 1;
  const bool stamp_clock = g.stamp && tid == 0 && blockIdx.x >= mid && blockIdx.x < mid + 8 && blockIdx.z == 0;
  unsigned long long c_start = 0, r_start = 0;
  if (stamp_clock) {
    r_start = __builtin_amdgcn_s_memrealtime();
    c_start = __builtin_amdgcn_s_memtime();
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int kbeg = (g.kskip == 1) ? ti * BM : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * BN < g.k) ? (tj + 1) * BN : g.k;
  const int64_t bz = blockIdx.z;
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * BM * g.lda + kbeg;
  const double* __restrict__ Bg = BKN ? g.B + bz * g.sB + (int64_t)kbeg * g.ldb + (int64_t)tj * BN
                                      : g.B + bz * g.sB + (int64_t)tj * BN * g.ldb + kbeg;

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
  for (int i = 0; i < TM; ++i) voa[i] = (unsigned)(((int64_t)(lrow + 32 * i) * g.lda + lkc) * 8);
#pragma unroll
  for (int i = 0; i < TN; ++i)
    vob[i] = BKN ? (unsigned)(((int64_t)(nrow + (512 / BN) * i) * g.ldb + nnc) * 8)
                 : (unsigned)(((int64_t)(lrow + 32 * i) * g.ldb + lkc) * 8);
  auto gload = [&](int k0, int slot) {
    const char* ab = reinterpret_cast<const char*>(Ag + k0);
    const char* bb = reinterpret_cast<const char*>(BKN ? Bg + (int64_t)k0 * g.ldb : Bg + k0);
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[slot][i] = *reinterpret_cast<const d2_t*>(ab + voa[i]);
#pragma unroll
    for (int i = 0; i < TN; ++i) rb[slot][i] = *reinterpret_cast<const d2_t*>(bb + vob[i]);
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
  double* Cg = g.C + bz * g.sC + ((int64_t)ti * BM + wr * (BM / 2)) * g.ldc + (int64_t)tj * BN + wc * (BN / 2);
  const int sw = (fr >= 4 && fr < 12) ? 1 : 0;
  const int a_off = (wr * (BM / 2) + fr) * LDS_STRIDE + ((fk ^ sw) << 2);
  const int b_off = BKN ? 4 * fk * LDS_STRIDE_KN + wc * (BN / 2) + fr
                        : (wc * (BN / 2) + fr) * LDS_STRIDE + ((fk ^ sw) << 2);

  // OP_SUB: the accumulators start from the C tile and the A operand is negated on its way into LDS,
  // so C - A B^T comes out of the MFMA chain itself and the epilogue is stores only.  The first operand slab is
  // requested BEFORE the 64 C loads and the first 16-deep step is peeled off the loop: its MFMAs wait for
  // their own accumulator tile only (vmcnt counts in order, the slab is older, the next slab's prefetch newer),
  // so the C tile streams in under the first step instead of in front of it.  After that step nothing is
  // pending on an accumulator register, which keeps every vmcnt wait out of the loop proper (with a pending
  // C load at loop entry the compiler puts s_waitcnt vmcnt(0) in front of the MFMAs of the last
  // accumulators INSIDE the loop, i.e. a wait for the prefetch just issued, in every step).
  const int nk = (kend - kbeg) / BK;
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
        for (int r = 0; r < 4; ++r) acc[i][j][r] = GPMI_C_LOAD(&Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + j * 16 + fr]);
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
          b[t] = *reinterpret_cast<const d2_t*>(sb + b_off + t * 16 * LDS_STRIDE + 2 * h);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
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

  // End stamp (only the workgroups that can be the launch's last: dispatch is in order and tiles are uniform,
  // so the last one to finish is among the last two rounds of 512), taken behind the epilogue stores.
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 1024 >= gridDim.x;

  // epilogue: stores only; each instruction covers 4 rows x 128 contiguous bytes
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) GPMI_C_STORE(acc[i][j][r], &Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + j * 16 + fr]);
  if (stamp_end) {
    // after this wave's own epilogue stores have been acknowledged: the stamp then dates the end of the workgroup as
    // rocprofv3 sees it (the first version took the time before the stores and under-stated a launch by 2-3 %)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
  // cycles (s_memtime) in the high word, 10 ns ticks (s_memrealtime) in the low word -> clock = cycles / ticks x 100 MHz
  if (stamp_clock)
    g.stamp[16 + blockIdx.x - mid] = ((__builtin_amdgcn_s_memtime() - c_start) << 32) |
                                     ((__builtin_amdgcn_s_memrealtime() - r_start) & 0xffffffffull);
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

__device__ inline void dma_wait(int newer) {
  // wait until at most `newer` younger vector-memory operations of this wave are outstanding
  if (newer >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (newer >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (newer >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifndef GPMI_DMA_WGS
#define GPMI_DMA_WGS 2
#endif
template <int TILES, int OP>
__global__ __launch_bounds__(256, GPMI_DMA_WGS) void gemm_dma_kernel(GemmArgs g) {
  __shared__ double smem[DMA_STAGES * 2 * DMA_OP_DOUBLES];
  int ti, tj;
  // k-skipped launches have tiles of very different length: dealt round-robin over the XCDs (see gemm_nt_kernel)
  tile_of<TILES>(g.tile_base + (g.kskip ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x)), g.ntr, g.ntc, ti, tj);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stamp_first = g.stamp && tid == 0 && blockIdx.x < 8 && blockIdx.z == 0;
  if (stamp_first) g.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const unsigned mid = gridDim.x >> 1;
  const bool stamp_clock = g.stamp && tid == 0 && blockIdx.x >= mid && blockIdx.x < mid + 8;
  unsigned long long c_start = 0, r_start = 0;
  if (stamp_clock) {
    r_start = __builtin_amdgcn_s_memrealtime();
    c_start = __builtin_amdgcn_s_memtime();
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const int kbeg = (g.kskip == 1) ? ti * 128 : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * 128 < g.k) ? (tj + 1) * 128 : g.k;
  const int64_t bz = blockIdx.z;  // batch (lockstep factorisations): problem z works on C + z sC, A + z sA, B + z sB
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * 128 * g.lda + kbeg;
  const double* __restrict__ Bg = g.B + bz * g.sB + (int64_t)tj * 128 * g.ldb + kbeg;
  // this thread's two pieces per operand and stage: p = i * 256 + tid -> row p >> 2, slot p & 3
  const int row0 = tid >> 2, slot = tid & 3;
  const int q0 = (slot - 2 * ((row0 >> 2) & 3)) & 3;          // rows row0 and row0 + 64 have the same (row >> 2) & 3
  const double* a_src0 = Ag + (int64_t)row0 * g.lda + 2 * q0;
  const double* a_src1 = a_src0 + (int64_t)64 * g.lda;
  // B rows (= columns of C) are permuted on their way into LDS: LDS row 16 t + fr of a wave column block holds
  // column 2 fr + t (t < 2) or 32 + 2 fr + t - 2 of that block, so that MFMA tiles (j, j + 1) of a lane are two
  // ADJACENT columns of C: the C tile is read and written with 16-byte accesses, 256 contiguous bytes per row and
  // instruction, half as many instructions as the 8-byte form
  auto bperm = [](int R) {
    const int t = (R >> 4) & 3, f = R & 15;
    return (R & 64) + ((t & 2) << 4) + 2 * f + (t & 1);
  };
  const double* b_src0 = Bg + (int64_t)bperm(row0) * g.ldb + 2 * q0;
  const double* b_src1 = Bg + (int64_t)bperm(row0 + 64) * g.ldb + 2 * q0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;  // LDS byte address of the ring (address space 3 pointers are offsets)
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
  auto issue = [&](int st, int k0) {
    const unsigned base = lds0 + (unsigned)st * (2 * DMA_OP_DOUBLES * 8) + wave_off;
    const double* p0 = a_src0 + k0;
    const double* p1 = a_src1 + k0;
    const double* p2 = b_src0 + k0;
    const double* p3 = b_src1 + k0;
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
  const int nk = (kend - kbeg) / DMA_BK;
  for (int st = 0; st < DMA_STAGES - 1 && st < nk; ++st) issue(st, st * DMA_BK);

  double* Cg = g.C + bz * g.sC + ((int64_t)ti * 128 + wr * 64) * g.ldc + (int64_t)tj * 128 + wc * 64;
  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2_t cv = d2_t{0.0, 0.0};
        if (OP == OP_SUB)
          cv = GPMI_C_LOAD(reinterpret_cast<const d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + jp * 32 + 2 * fr]));
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
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 1024 >= gridDim.x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        GPMI_C_STORE((d2_t{acc[i][2 * jp][r], acc[i][2 * jp + 1][r]}),
                     reinterpret_cast<d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + jp * 32 + 2 * fr]));
  if (stamp_end) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
  if (stamp_clock)
    g.stamp[16 + blockIdx.x - mid] = ((__builtin_amdgcn_s_memtime() - c_start) << 32) |
                                     ((__builtin_amdgcn_s_memrealtime() - r_start) & 0xffffffffull);
}


// ---- 64 x 64 tiles on the same ring ---------------------------------------------------------------------------------
// The remainders of split launches, the narrow look-ahead updates (fewer than 384 tiles) and the tail's outer updates
// run 64 x 64 tiles; with the register-staged kernel they reached 40-47 TFLOP/s at K = 512.  Same scheme as
// gemm_dma_kernel at half the edge: one 16-byte piece per operand, stage and thread (64 rows x 4 pieces), 2 x 2 waves
// of 32 x 32 (2 x 2 MFMA tiles), 8 KiB per stage.  The MFMAs take the k of a stage in the same groups as the 128 x 128
// ring kernel ({0,2,4,6}, {1,3,5,7}).
template <int TILES, int OP>
__global__ __launch_bounds__(256, 4) void gemm_dma64_kernel(GemmArgs g) {
  constexpr int OPD = 64 * DMA_BK;  // doubles of one operand of one stage
  __shared__ double smem[DMA_STAGES * 2 * OPD];
  int ti, tj;
  const int wid = g.kskip ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  if (g.split) {
    int bi, bj;
    tile_of<TILES>(g.tile_base + (wid >> 2), g.ntr >> 1, g.ntc >> 1, bi, bj);
    ti = 2 * bi + ((wid >> 1) & 1);
    tj = 2 * bj + (wid & 1);
  } else {
    tile_of<TILES>(g.tile_base + wid, g.ntr, g.ntc, ti, tj);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stamp_first = g.stamp && tid == 0 && blockIdx.x < 8;
  if (stamp_first) g.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const int kbeg = (g.kskip == 1) ? ti * 64 : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * 64 < g.k) ? (tj + 1) * 64 : g.k;
  const int64_t bz = blockIdx.z;
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * 64 * g.lda + kbeg;
  const double* __restrict__ Bg = g.B + bz * g.sB + (int64_t)tj * 64 * g.ldb + kbeg;
  // this thread's piece per operand and stage: row tid >> 2, slot tid & 3 (see gemm_dma_kernel for the slot swizzle)
  const int row0 = tid >> 2, slot = tid & 3;
  const int q0 = (slot - 2 * ((row0 >> 2) & 3)) & 3;
  const double* a_src = Ag + (int64_t)row0 * g.lda + 2 * q0;
  // LDS row 16 t + f of a wave's 32-column block holds column 2 f + t: the two MFMA tiles of a lane are adjacent columns
  const int bcol = (row0 & 32) + 2 * (row0 & 15) + ((row0 >> 4) & 1);
  const double* b_src = Bg + (int64_t)bcol * g.ldb + 2 * q0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);
  auto issue = [&](int st, int k0) {
    const unsigned base = lds0 + (unsigned)st * (2 * OPD * 8) + wave_off;
    const double* p0 = a_src + k0;
    const double* p1 = b_src + k0;
    asm volatile(
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
        :
        : "v"(p0), "v"(p1), "s"(base), "s"(base + OPD * 8)
        : "memory");
  };
  const int nk = (kend - kbeg) / DMA_BK;
  for (int st = 0; st < DMA_STAGES - 1 && st < nk; ++st) issue(st, st * DMA_BK);

  double* Cg = g.C + bz * g.sC + ((int64_t)ti * 64 + wr * 32) * g.ldc + (int64_t)tj * 64 + wc * 32;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d2_t cv = d2_t{0.0, 0.0};
      if (OP == OP_SUB)
        cv = GPMI_C_LOAD(reinterpret_cast<const d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + 2 * fr]));
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
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 2048 >= gridDim.x;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      GPMI_C_STORE((d2_t{acc[i][0][r], acc[i][1][r]}),
                   reinterpret_cast<d2_t*>(&Cg[(int64_t)(i * 16 + fk + 4 * r) * g.ldc + 2 * fr]));
  if (stamp_end) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

namespace {
// part: 0 = the whole product; 1 = only the first `nfull` 128 x 128 tiles (as 128 x 128 tiles); 2 = only the
// tiles [nfull, nend) (nend < 0: to the last), as 64 x 64 tiles; 3 = the tiles [nfull, nend) as 128 x 128 tiles
void launch_gemm_part(hipStream_t s, GemmTiles tiles, GemmOp op, bool b_kmajor, int kskip, double* C,
                      int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int ntr,
                      int ntc, int k, unsigned long long* stamp, const GemmBatch& bt, int part, int64_t nfull,
                      int64_t nend = -1) {
  // ntr, ntc are in units of 128 rows / columns
  if (ntr <= 0 || ntc <= 0 || k <= 0) return;
  if (tiles == TILES_LOWER && ntc > ntr) ntc = ntr;
  const int64_t big = (tiles == TILES_RECT) ? (int64_t)ntr * ntc
                                            : (int64_t)ntc * (ntc + 1) / 2 + (int64_t)(ntr - ntc) * ntc;
  // tile shape: full 128 x 128 tiles when there is enough work to fill the chip at K >= 256,
  // 64-row tiles for one-tile-column products (in place: a workgroup must own whole rows),
  // 64 x 64 tiles for the remaining short / small launches
  int bm = 128, bn = 128;
  static const int64_t BIG_MIN = [] {
    const char* e = std::getenv("GPMI_BIG_MIN");
    return (int64_t)(e ? std::atoi(e) : 384);
  }();
  static const int64_t SMALL_M32_MAX = [] {
    const char* e = std::getenv("GPMI_M32_MAX");
    return (int64_t)(e ? std::atoi(e) : 192);
  }();
  // thresholds of the 32-row tiles, scaled to the CUs of a masked stream (the panel chain on its 32 CUs is throughput-
  // bound from 24 workgroups on: halving the tiles there only doubles the operand traffic - 60 vs 45 us per panel TRSM)
  const int64_t m32_max = bt.ncu_hint > 0 ? SMALL_M32_MAX * bt.ncu_hint / 256 : SMALL_M32_MAX;
  const bool small = part == 0 && ((k <= 128) || (big * bt.count < BIG_MIN));
  if (part == 2) {
    bm = 64;
    bn = 64;
  } else if (small && kskip != 1) {
    if (ntc == 1 && tiles == TILES_RECT && op == OP_ASSIGN) {
      // the panel TRSM: one 128-column strip, bound by the MFMA time of ONE workgroup (6.8 us for 64 rows at
      // K = 128): 32-row tiles while that still leaves CUs idle
      bm = (!b_kmajor && (int64_t)ntr * 2 * bt.count <= m32_max) ? 32 : 64;
    } else if (!b_kmajor || op == OP_SUB) {
      bm = 64;
      bn = 64;
      // fewer 64 x 64 tiles than three quarters of the CUs, and each of them long (K > 128: the products with the
      // 512 x 512 inverse blocks on the chain of the many-right-hand-side solves, 128 workgroups x up to 13 us of
      // MFMA time on one CU each): 32-row tiles put the same work on twice as many CUs
      if (tiles == TILES_RECT && !b_kmajor && k > 128 && big * 4 * bt.count <= m32_max) bm = 32;
    }
  }
  GemmArgs g{C, A, B, ldc, lda, ldb, ntr * (128 / bm), ntc * (128 / bn), k, kskip, stamp,
             bt.sC, bt.sA, bt.sB, part == 2 ? 1 : 0, part >= 2 ? (int)nfull : 0};
  int64_t nwg;
  if (tiles == TILES_RECT)
    nwg = (int64_t)g.ntr * g.ntc;
  else
    nwg = (int64_t)g.ntc * (g.ntc + 1) / 2 + (int64_t)(g.ntr - g.ntc) * g.ntc;
  if (part == 1) nwg = nfull;
  if (nend < 0 || nend > big) nend = big;
  if (part == 2) nwg = 4 * (nend - nfull);
  if (part == 3) nwg = nend - nfull;
  if (nwg <= 0) return;
  dim3 grid((unsigned)nwg, 1, (unsigned)bt.count), block(256);
  // full 128 x 128 tiles with K-contiguous operands take the LDS-DMA ring kernel (GPMI_GEMM_NO_DMA=1: the
  // register-staged kernel everywhere, for A/B timing).  The ring kernels feed k = {0,2,4,6} / {1,3,5,7} of a stage to
  // their two MFMAs, the register-staged kernels k = {q, 4+q, 8+q, 12+q}: the sums differ in the last bit.  Lockstep
  // batches (values must not depend on the batch size, which decides between 128 x 128 and 64 x 64 tiles) stay
  // consistent because BOTH tile sizes of their C -= A B^T launches are ring kernels (this one and gemm_dma64_kernel)
  // and their in-place TRSM is register-staged at either tile height.
  static const bool no_dma = std::getenv("GPMI_GEMM_NO_DMA") != nullptr;
  if (!no_dma && bm == 128 && bn == 128 && !b_kmajor && part != 2 && k % 128 == 0) {
    if (tiles == TILES_RECT) {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma_kernel<TILES_RECT, OP_SUB>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma_kernel<TILES_RECT, OP_ASSIGN>), grid, block, 0, s, g);
    } else {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma_kernel<TILES_LOWER, OP_SUB>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma_kernel<TILES_LOWER, OP_ASSIGN>), grid, block, 0, s, g);
    }
    return;
  }
  // 64 x 64 tiles with K-contiguous operands: the ring kernel at half the edge (GPMI_GEMM_NO_DMA64=1: register-staged)
  static const bool no_dma64 = std::getenv("GPMI_GEMM_NO_DMA64") != nullptr;
  static const int dma64_min_k = [] {
    const char* e = std::getenv("GPMI_DMA64_MIN_K");
    return e ? std::atoi(e) : 128;  // also the K = 128 inner updates of the panel chain: 34.9 -> 34.45 ms per step
  }();
  if (!no_dma && !no_dma64 && bm == 64 && bn == 64 && !b_kmajor && k % 64 == 0 && k >= dma64_min_k) {
    if (tiles == TILES_RECT) {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma64_kernel<TILES_RECT, OP_SUB>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma64_kernel<TILES_RECT, OP_ASSIGN>), grid, block, 0, s, g);
    } else {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma64_kernel<TILES_LOWER, OP_SUB>), grid, block, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma64_kernel<TILES_LOWER, OP_ASSIGN>), grid, block, 0, s, g);
    }
    return;
  }
#define GPMI_LAUNCH(T, O, B, M, N) \
  hipLaunchKernelGGL((gemm_nt_kernel<T, O, B, M, N>), grid, block, 0, s, g)
  if (bm == 128) {
    if (tiles == TILES_RECT) {
      if (op == OP_SUB) {
        if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_SUB, 1, 128, 128); else GPMI_LAUNCH(TILES_RECT, OP_SUB, 0, 128, 128);
      } else {
        if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 1, 128, 128); else GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 128, 128);
      }
    } else {
      if (op == OP_SUB) GPMI_LAUNCH(TILES_LOWER, OP_SUB, 0, 128, 128); else GPMI_LAUNCH(TILES_LOWER, OP_ASSIGN, 0, 128, 128);
    }
  } else if (bn == 128 && bm == 32) {
    GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 32, 128);
  } else if (bn == 128) {
    if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 1, 64, 128); else GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 64, 128);
  } else if (bm == 32) {
    if (op == OP_SUB) GPMI_LAUNCH(TILES_RECT, OP_SUB, 0, 32, 64); else GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 32, 64);
  } else {
    if (tiles == TILES_RECT) {
      if (op == OP_SUB) {
        if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_SUB, 1, 64, 64); else GPMI_LAUNCH(TILES_RECT, OP_SUB, 0, 64, 64);
      } else {
        GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 64, 64);
      }
    } else {
      if (op == OP_SUB) GPMI_LAUNCH(TILES_LOWER, OP_SUB, 0, 64, 64); else GPMI_LAUNCH(TILES_LOWER, OP_ASSIGN, 0, 64, 64);
    }
  }
#undef GPMI_LAUNCH
}
}  // namespace

void launch_gemm(hipStream_t s, GemmTiles tiles, GemmOp op, bool b_kmajor, int kskip, double* C,
                 int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int ntr,
                 int ntc, int k, unsigned long long* stamp, const GemmBatch& bt) {
  launch_gemm_part(s, tiles, op, b_kmajor, kskip, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, bt, 0, 0);
}

// A launch of T equal tiles on `ncu` CUs runs in ceil(T / ncu) rounds of one tile time (the two workgroups
// of a CU share its MFMA pipes, so a CU with one tile left is as slow as a CU with two): a last round that
// is nearly empty wastes up to 70 us x ncu CUs.  When that round would be less than 70 % full (55 % before the
// 64 x 64 tiles ran on the LDS-DMA ring) its tiles run as 64 x 64 tiles instead (four times as many workgroups, spread over all CUs), in a second launch.
int64_t gemm_split_point(int64_t T, int ncu, int k) {
  static const int64_t BIG_MIN = [] {
    const char* e = std::getenv("GPMI_BIG_MIN");
    return (int64_t)(e ? std::atoi(e) : 384);
  }();
  static const int64_t SPLIT_PCT = [] {
    const char* e = std::getenv("GPMI_SPLIT_PCT");
    return (int64_t)(e ? std::atoi(e) : 70);
  }();
  if (k <= 128 || T < BIG_MIN || ncu <= 0) return T;  // these launches use the small tiles throughout
  const int64_t rem = T % ncu;
  if (rem == 0 || rem * 100 > (int64_t)ncu * SPLIT_PCT) return T;
  return T - rem;
}

void launch_gemm_nt_split(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc, const double* A,
                          int64_t lda, const double* B, int64_t ldb, int ntr, int ntc, int k, int64_t nfull,
                          unsigned long long* stamp, unsigned long long* stamp_rest, int64_t nend) {
  if (tiles == TILES_LOWER && ntc > ntr) ntc = ntr;
  const int64_t T = (tiles == TILES_RECT) ? (int64_t)ntr * ntc
                                          : (int64_t)ntc * (ntc + 1) / 2 + (int64_t)(ntr - ntc) * ntc;
  const GemmBatch one{};
  if (nend < 0 || nend > T) nend = T;
  if (nfull >= T) {
    launch_gemm_part(s, tiles, op, false, 0, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, one, 0, 0);
    return;
  }
  launch_gemm_part(s, tiles, op, false, 0, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, one, 1, nfull);
  if (nend > nfull)
    launch_gemm_part(s, tiles, op, false, 0, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp_rest, one, 2, nfull, nend);
}

// the tiles [first, first + count) of the logical tile list as full 128 x 128 tiles (a slice of a product whose other
// tiles another stream computes)
void launch_gemm_nt_range(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc, const double* A,
                          int64_t lda, const double* B, int64_t ldb, int ntr, int ntc, int k, int64_t first,
                          int64_t count, unsigned long long* stamp) {
  if (count <= 0) return;
  const GemmBatch one{};
  launch_gemm_part(s, tiles, op, false, 0, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, one, 3, first, first + count);
}

void launch_gemm_nt(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc,
                    const double* A, int64_t lda, const double* B, int64_t ldb, int ntr, int ntc,
                    int k, unsigned long long* stamp, const GemmBatch& bt) {
  launch_gemm(s, tiles, op, false, false, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, bt);
}
