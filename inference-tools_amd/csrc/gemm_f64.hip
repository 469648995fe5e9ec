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

#include "gemm_tiles.h"

namespace {

using namespace gemm_tiles;

struct GemmArgs {
  double* C;
  const double* A;
  const double* B;
  int64_t ldc, lda, ldb;
  int ntr, ntc, k;
  // 1 (TILES_LOWER): contraction starts at k = ti * BM (operands are zero before it);
  // 2: contraction ends at k = (tj + 1) * BN (B is lower triangular: B[j][k] = 0 for k > j)
  // 3: one tile column of 128 = the whole contraction with a lower-triangular B (the panel TRSM as a product with the
  //    inverse diagonal block): the zero part is skipped inside the tile (gemm_tiles::staged_tile, BTRI)
  // 4 (register-staged kernels, k-major B): contraction starts at k = tj * BN (B[k][j] = 0 for k < j: a lower-triangular
  //    B applied from the right, untransposed - the backward row solve's products with the 512 x 512 inverse blocks)
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
  FlowHook hook;  // potrf_flow.hip: flags published / awaited at the start of a chain launch
  // > 0 (lockstep batches of a multiple of 8 problems): a one-dimensional launch of 8 x zlocal_tiles x (batch / 8)
  // workgroups in which problem z runs entirely on XCD z % 8 (workgroup n -> XCD n % 8), its zlocal_tiles tiles in their
  // logical order: a problem's operand panels are fetched into ONE L2 instead of all eight
  int zlocal_tiles;
  // > 0 (gemm_dma_kernel only, round 6): a MIXED launch - workgroups [0, mixed_full) take the 128 x 128 tiles
  // [tile_base, tile_base + mixed_full), the workgroups behind them the tiles from tile_base + mixed_full on as four
  // 64 x 64 quarters each: the tail of a trailing update in the SAME launch as its full rounds, so that the quarters fill
  // the CUs the last round's stragglers leave free instead of running behind a kernel boundary on a nearly idle chip
  int mixed_full;
};

// (tile index within the launch's tile list, problem) of this workgroup
struct WgId {
  int wid;
  int64_t bz;
};
__device__ inline int xcd_remap(int b, int nwg);
__device__ inline WgId wg_id(int kskip, int zlocal_tiles) {
  if (zlocal_tiles > 0) {
    const int n = (int)blockIdx.x, m = n >> 3;
    return {m % zlocal_tiles, (int64_t)(m / zlocal_tiles) * 8 + (n & 7)};
  }
  return {kskip ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x), (int64_t)blockIdx.z};
}

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

// register-staged tiles (gemm_tiles::staged_tile): any tile shape, k-major B, in-place products
template <int TILES, int OP, int BKN, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ double smem[staged_lds_doubles<BKN, BM, BN>()];  // [buffer][A | B]
  flow_hook_enter(g.hook);
  int ti, tj;
  // k-skipped launches have tiles of very different length (128 (ntr - ti) k-steps): deal them out
  // round-robin over the XCDs instead of in contiguous chunks, or the XCD holding the long tiles ends last
  const WgId me = wg_id(g.kskip, g.zlocal_tiles);
  const int wid = me.wid;
  if (g.split) {
    int bi, bj;
    tile_of<TILES>(g.tile_base + (wid >> 2), g.ntr >> 1, g.ntc >> 1, bi, bj);
    ti = 2 * bi + ((wid >> 1) & 1);
    tj = 2 * bj + (wid & 1);
  } else {
    tile_of<TILES>(g.tile_base + wid, g.ntr, g.ntc, ti, tj);
    // Products with a lower-triangular B (kskip 2: the contraction of tile column tj ends at (tj + 1) BN) have tile columns of
    // very different length.  The second half of the tile rows takes them in mirrored order, so that the two workgroups a
    // CU holds - one from each half of a launch of 2 x #CUs tiles - carry a long and a short contraction.
    if (TILES == TILES_RECT && (g.kskip == 2 || g.kskip == 4) && 2 * ti >= g.ntr) tj = g.ntc - 1 - tj;
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
  const int kbeg = (g.kskip == 1) ? ti * BM : (g.kskip == 4) ? tj * BN : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * BN < g.k) ? (tj + 1) * BN : g.k;
  const int64_t bz = me.bz;
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * BM * g.lda + kbeg;
  const double* __restrict__ Bg = BKN ? g.B + bz * g.sB + (int64_t)kbeg * g.ldb + (int64_t)tj * BN
                                      : g.B + bz * g.sB + (int64_t)tj * BN * g.ldb + kbeg;
  double* Cg = g.C + bz * g.sC + (int64_t)ti * BM * g.ldc + (int64_t)tj * BN;
  if constexpr (OP == OP_ASSIGN && BKN == 0 && BN == 128 && BM <= 64) {
    if (g.kskip == 3) {
      staged_tile<OP, BKN, BM, BN, CST_NT, LD_PLAIN, true>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, (kend - kbeg) / BK, smem);
    } else {
      staged_tile<OP, BKN, BM, BN>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, (kend - kbeg) / BK, smem);
    }
  } else {
    staged_tile<OP, BKN, BM, BN>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, (kend - kbeg) / BK, smem);
  }

  // End stamp (only the workgroups that can be the launch's last: dispatch is in order and tiles are uniform,
  // so the last one to finish is among the last two rounds of 512), taken behind the epilogue stores.
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 1024 >= gridDim.x;
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


// ---- 128 x 128 and 64 x 64 tiles with K-contiguous operands on the LDS-DMA ring (gemm_tiles::dma128_tile, dma64_tile)
#ifndef GPMI_DMA_WGS
#define GPMI_DMA_WGS 2
#endif
// The 64 x 64 body of a mixed launch, out of line: inlined beside the 128 x 128 body it would share its register
// allocation (the flag-ordered task kernel keeps its small bodies out of line for the same reason).  The LDS block
// travels as an address-space-3 pointer and the matrices as address-space-1 pointers (ds_ / global_ instructions, not
// flat_); nk and the LDS base arrive in vector registers and are made scalar again for the ring's m0 operands.
typedef __attribute__((address_space(3))) double gemm_lds_double_t;
typedef __attribute__((address_space(1))) double gemm_glb_double_t;
template <int OP>
__device__ __attribute__((noinline)) void dma64_quarter(const gemm_glb_double_t* Ag, const gemm_glb_double_t* Bg,
                                                        gemm_glb_double_t* Cg, int64_t lda, int64_t ldb, int64_t ldc, int nk,
                                                        gemm_lds_double_t* smem) {
  gemm_lds_double_t* su = (gemm_lds_double_t*)(uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
  dma64_tile<OP>((const double*)Ag, (const double*)Bg, (double*)Cg, lda, ldb, ldc, __builtin_amdgcn_readfirstlane(nk),
                 (double*)su);
}

template <int TILES, int OP>
__global__ __launch_bounds__(256, GPMI_DMA_WGS) void gemm_dma_kernel(GemmArgs g) {
  __shared__ double smem[DMA128_LDS_DOUBLES];
  flow_hook_enter(g.hook);
  int ti, tj;
  if (g.mixed_full > 0 && (int)blockIdx.x >= g.mixed_full) {
    // a quarter of one of the launch's last tiles (dispatched behind every full tile: it runs where the last round frees a CU)
    const int w = xcd_remap((int)blockIdx.x - g.mixed_full, (int)gridDim.x - g.mixed_full);
    tile_of<TILES>(g.tile_base + g.mixed_full + (w >> 2), g.ntr, g.ntc, ti, tj);
    const int r0 = 64 * ((w >> 1) & 1), c0 = 64 * (w & 1);
    const double* Aq = g.A + ((int64_t)ti * 128 + r0) * g.lda;
    const double* Bq = g.B + ((int64_t)tj * 128 + c0) * g.ldb;
    double* Cq = g.C + ((int64_t)ti * 128 + r0) * g.ldc + (int64_t)tj * 128 + c0;
    dma64_quarter<OP>((const gemm_glb_double_t*)Aq, (const gemm_glb_double_t*)Bq, (gemm_glb_double_t*)Cq, g.lda, g.ldb, g.ldc,
                      g.k / DMA_BK, (gemm_lds_double_t*)smem);
    if (g.stamp && threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  // k-skipped launches have tiles of very different length: dealt round-robin over the XCDs (see gemm_nt_kernel)
  WgId me = wg_id(g.kskip, g.zlocal_tiles);
  if (g.mixed_full > 0) me.wid = xcd_remap((int)blockIdx.x, g.mixed_full);
  tile_of<TILES>(g.tile_base + me.wid, g.ntr, g.ntc, ti, tj);
  const int tid = threadIdx.x;
  const bool stamp_first = g.stamp && tid == 0 && blockIdx.x < 8 && blockIdx.z == 0;
  if (stamp_first) g.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const unsigned mid = gridDim.x >> 1;
  const bool stamp_clock = g.stamp && tid == 0 && blockIdx.x >= mid && blockIdx.x < mid + 8;
  unsigned long long c_start = 0, r_start = 0;
  if (stamp_clock) {
    r_start = __builtin_amdgcn_s_memrealtime();
    c_start = __builtin_amdgcn_s_memtime();
  }
  const int kbeg = (g.kskip == 1) ? ti * 128 : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * 128 < g.k) ? (tj + 1) * 128 : g.k;
  const int64_t bz = me.bz;  // batch (lockstep factorisations): problem z works on C + z sC, A + z sA, B + z sB
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * 128 * g.lda + kbeg;
  const double* __restrict__ Bg = g.B + bz * g.sB + (int64_t)tj * 128 * g.ldb + kbeg;
  double* Cg = g.C + bz * g.sC + (int64_t)ti * 128 * g.ldc + (int64_t)tj * 128;
  dma128_tile<OP>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, (kend - kbeg) / DMA_BK, smem);
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 1024 >= gridDim.x;
  if (stamp_end) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
  if (stamp_clock)
    g.stamp[16 + blockIdx.x - mid] = ((__builtin_amdgcn_s_memtime() - c_start) << 32) |
                                     ((__builtin_amdgcn_s_memrealtime() - r_start) & 0xffffffffull);
}

// 256 x 128 tiles (gemm_tiles::dma256_tile), experiment GPMI_GEMM_256=1: whole products with even tile counts only
template <int TILES, int OP>
__global__ __launch_bounds__(512, 1) void gemm_dma256_kernel(GemmArgs g) {
  __shared__ double smem[DMA256_LDS_DOUBLES];
  int ti, tj;  // ti: 256-row tile, tj: 128-column tile
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int nr2 = g.ntr >> 1;
  if (TILES == TILES_RECT) {
    ti = id / g.ntc;
    tj = id - ti * g.ntc;
  } else {
    const int h = g.ntc >> 1, tri = h * (h + 1);
    if (id < tri) {
      int r = (int)((sqrtf(4.0f * id + 1.0f) - 1.0f) * 0.5f);
      while (r * (r + 1) > id) --r;
      while ((r + 1) * (r + 2) <= id) ++r;
      ti = r;
      tj = id - r * (r + 1);
    } else {
      ti = h + (id - tri) / g.ntc;
      tj = (id - tri) % g.ntc;
    }
  }
  (void)nr2;
  const int64_t bz = blockIdx.z;
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * 256 * g.lda;
  const double* __restrict__ Bg = g.B + bz * g.sB + (int64_t)tj * 128 * g.ldb;
  double* Cg = g.C + bz * g.sC + (int64_t)ti * 256 * g.ldc + (int64_t)tj * 128;
  dma256_tile<OP>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, g.k / DMA_BK, smem, TILES == TILES_LOWER && tj == 2 * ti + 1);
}

template <int TILES, int OP>
__global__ __launch_bounds__(256, 4) void gemm_dma64_kernel(GemmArgs g) {
  __shared__ double smem[DMA64_LDS_DOUBLES];
  flow_hook_enter(g.hook);
  int ti, tj;
  const WgId me = wg_id(g.kskip, g.zlocal_tiles);
  const int wid = me.wid;
  if (g.split) {
    int bi, bj;
    tile_of<TILES>(g.tile_base + (wid >> 2), g.ntr >> 1, g.ntc >> 1, bi, bj);
    ti = 2 * bi + ((wid >> 1) & 1);
    tj = 2 * bj + (wid & 1);
  } else {
    tile_of<TILES>(g.tile_base + wid, g.ntr, g.ntc, ti, tj);
  }
  const int tid = threadIdx.x;
  const bool stamp_first = g.stamp && tid == 0 && blockIdx.x < 8;
  if (stamp_first) g.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const int kbeg = (g.kskip == 1) ? ti * 64 : 0;
  const int kend = (g.kskip == 2 && (tj + 1) * 64 < g.k) ? (tj + 1) * 64 : g.k;
  const int64_t bz = me.bz;
  const double* __restrict__ Ag = g.A + bz * g.sA + (int64_t)ti * 64 * g.lda + kbeg;
  const double* __restrict__ Bg = g.B + bz * g.sB + (int64_t)tj * 64 * g.ldb + kbeg;
  double* Cg = g.C + bz * g.sC + (int64_t)ti * 64 * g.ldc + (int64_t)tj * 64;
  dma64_tile<OP>(Ag, Bg, Cg, g.lda, g.ldb, g.ldc, (kend - kbeg) / DMA_BK, smem);
  const bool stamp_end = g.stamp && tid == 0 && blockIdx.x + 2048 >= gridDim.x;
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
      if (tiles == TILES_RECT && !b_kmajor && k > 128 && big * 4 * bt.count <= m32_max && !bt.ring_order_only) bm = 32;
      // ... and with a triangular B (the contraction of a tile column ends at its diagonal) 32 x 32 tiles: twice the
      // workgroups again, the long columns' MFMA time halved, long and short columns paired on a CU (gemm_nt_kernel).
      // GPMI_CHAIN_BN32=0: 32 x 64 tiles (rounds 5-6a)
      static const bool bn32 = [] {
        const char* e = std::getenv("GPMI_CHAIN_BN32");
        return !e || std::atoi(e) != 0;
      }();
      if (bn32 && bm == 32 && kskip == 2 && op == OP_ASSIGN && big * 8 * bt.count <= 2 * 256) bn = 32;
    }
  }
  // the backward row solve's product with an inverse block (k-major triangular B, kskip 4): 32 x 32 tiles while that puts
  // at most two workgroups on a CU (long and short contractions paired, see gemm_nt_kernel), 64 x 128 beyond
  if (part == 0 && kskip == 4 && b_kmajor && op == OP_ASSIGN && tiles == TILES_RECT) {
    if (big * 16 * bt.count <= 2 * 256) {
      bm = 32;
      bn = 32;
    } else {
      bm = 64;
      bn = 128;
    }
  }
  // the panel TRSM with the caller's word that B (the inverse of a diagonal block) is lower triangular
  if (bt.b_lower_tri && kskip == 0 && bn == 128 && bm <= 64 && ntc == 1 && k == 128 && !b_kmajor && op == OP_ASSIGN) kskip = 3;
  GemmArgs g{C, A, B, ldc, lda, ldb, ntr * (128 / bm), ntc * (128 / bn), k, kskip, stamp,
             bt.sC, bt.sA, bt.sB, part == 2 ? 1 : 0, (part == 2 || part == 3) ? (int)nfull : 0, bt.hook, 0, 0};
  int64_t nwg;
  if (tiles == TILES_RECT)
    nwg = (int64_t)g.ntr * g.ntc;
  else
    nwg = (int64_t)g.ntc * (g.ntc + 1) / 2 + (int64_t)(g.ntr - g.ntc) * g.ntc;
  if (part == 1) nwg = nfull;
  if (nend < 0 || nend > big) nend = big;
  if (part == 2) nwg = 4 * (nend - nfull);
  if (part == 3) nwg = nend - nfull;
  if (part == 4) {  // mixed: nfull whole tiles + the tiles [nfull, nend) in quarters, one launch (gemm_dma_kernel)
    g.mixed_full = (int)nfull;
    nwg = nfull + 4 * (nend - nfull);
  }
  if (nwg <= 0) return;
  dim3 grid((unsigned)nwg, 1, (unsigned)bt.count), block(256);
  // lockstep batches of a multiple of 8 problems: every problem on ONE XCD (GemmArgs::zlocal_tiles; GPMI_BATCH_XCD=0: off)
  static const bool batch_xcd = [] {
    const char* e = std::getenv("GPMI_BATCH_XCD");
    return !e || std::atoi(e) != 0;
  }();
  if (batch_xcd && bt.count >= 8 && bt.count % 8 == 0 && !stamp && nwg * bt.count < (int64_t)1 << 30) {
    g.zlocal_tiles = (int)nwg;
    grid = dim3((unsigned)(nwg * bt.count), 1, 1);
  }
  // full 128 x 128 tiles with K-contiguous operands take the LDS-DMA ring kernel (GPMI_GEMM_NO_DMA=1: the
  // register-staged kernel everywhere, for A/B timing).  The ring kernels feed k = {0,2,4,6} / {1,3,5,7} of a stage to
  // their two MFMAs, the register-staged kernels k = {q, 4+q, 8+q, 12+q}: the sums differ in the last bit.  Lockstep
  // batches (values must not depend on the batch size, which decides between 128 x 128 and 64 x 64 tiles) stay
  // consistent because BOTH tile sizes of their C -= A B^T launches are ring kernels (this one and gemm_dma64_kernel)
  // and their in-place TRSM is register-staged at either tile height.
  static const bool no_dma = std::getenv("GPMI_GEMM_NO_DMA") != nullptr;
  static const bool tall = std::getenv("GPMI_GEMM_256") != nullptr;
  if (tall && bm == 128 && bn == 128 && !b_kmajor && part == 0 && kskip == 0 && k % 128 == 0 && ntr % 2 == 0 && ntc % 2 == 0 && !stamp) {
    const int h = ntc / 2;
    const int64_t n256 = tiles == TILES_RECT ? (int64_t)(ntr / 2) * ntc : (int64_t)h * (h + 1) + (int64_t)(ntr / 2 - h) * ntc;
    dim3 grid2((unsigned)n256, 1, (unsigned)bt.count), block2(512);
    if (tiles == TILES_RECT) {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma256_kernel<TILES_RECT, OP_SUB>), grid2, block2, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma256_kernel<TILES_RECT, OP_ASSIGN>), grid2, block2, 0, s, g);
    } else {
      if (op == OP_SUB) hipLaunchKernelGGL((gemm_dma256_kernel<TILES_LOWER, OP_SUB>), grid2, block2, 0, s, g);
      else hipLaunchKernelGGL((gemm_dma256_kernel<TILES_LOWER, OP_ASSIGN>), grid2, block2, 0, s, g);
    }
    return;
  }
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
  } else if (bm == 32 && bn == 32) {
    if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 1, 32, 32); else GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0, 32, 32);
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

bool gemm_mixed_launches() {
  static const bool mixed = [] {
    const char* e = std::getenv("GPMI_GEMM_MIXED");  // (0: the remainder in a launch of its own, as until round 5)
    return !e || std::atoi(e) != 0;
  }();
  return mixed;
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
  // round 6: the remainder's quarters in the same launch as the full rounds (one stamp slot for the whole launch): headline
  // 31.96 - 32.04 -> 31.80 - 31.88 ms per step, same bits (GPMI_GEMM_MIXED=0: two launches)
  if (gemm_mixed_launches() && nfull > 0 && nend > nfull && k % 128 == 0 && k > 128) {
    launch_gemm_part(s, tiles, op, false, 0, C, ldc, A, lda, B, ldb, ntr, ntc, k, stamp, one, 4, nfull, nend);
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
