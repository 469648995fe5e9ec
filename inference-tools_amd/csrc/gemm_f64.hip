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
#include "gpmi_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_STRIDE = 18;                   // doubles per staged row (16 + 2 pad)
constexpr int TILE_DOUBLES = BM * LDS_STRIDE;    // one operand tile in LDS

struct GemmArgs {
  double* C;
  const double* A;
  const double* B;
  int64_t ldc, lda, ldb;
  int ntr, ntc, k;
  int kskip;  // TILES_LOWER only: contraction starts at k = ti * 128 (operands are zero before it)
};

constexpr int LDS_STRIDE_KN = 144;  // B stored k-major: 16 rows of 128 + 16 pad (rows 16 doubles apart mod 32)

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
    ti = id / ntc;
    tj = id - ti * ntc;
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
template <int TILES, int OP, int BKN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ double smem[2 * 2 * TILE_DOUBLES];  // [buffer][A|B][128][18]
  int ti, tj;
  tile_of<TILES>(xcd_remap(blockIdx.x, gridDim.x), g.ntr, g.ntc, ti, tj);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int kbeg = g.kskip ? ti * BM : 0;
  const double* __restrict__ Ag = g.A + (int64_t)ti * BM * g.lda + kbeg;
  const double* __restrict__ Bg =
      BKN ? g.B + (int64_t)kbeg * g.ldb + (int64_t)tj * BN : g.B + (int64_t)tj * BN * g.ldb + kbeg;

  // global -> register staging: 4 x 16-byte chunks per operand per thread (8 threads cover a row)
  const int lrow = tid >> 3, lkc = (tid & 7) * 2;
  // swizzled position of this thread's 16-byte piece: rows 4..11 (mod 16) swap their 32-byte chunk pairs
  // (lrow + 32 i has the same row & 15 for every i)
  const int lkc_sw = lkc ^ ((((lrow & 15) >= 4) && ((lrow & 15) < 12)) ? 4 : 0);
  const int nrow = tid >> 6, nnc = (tid & 63) * 2;  // k-major B: 64 chunks per k-row
  d2_t ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 32 * i;
      ra[i] = *reinterpret_cast<const d2_t*>(Ag + (int64_t)row * g.lda + k0 + lkc);
      if (BKN)
        rb[i] = *reinterpret_cast<const d2_t*>(Bg + (int64_t)(k0 + nrow + 4 * i) * g.ldb + nnc);
      else
        rb[i] = *reinterpret_cast<const d2_t*>(Bg + (int64_t)row * g.ldb + k0 + lkc);
    }
  };
  auto sstore = [&](int buf) {
    double* sa = smem + buf * 2 * TILE_DOUBLES;
    double* sb = sa + TILE_DOUBLES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<d2_t*>(sa + row * LDS_STRIDE + lkc_sw) = ra[i];
      if (BKN)
        *reinterpret_cast<d2_t*>(sb + (nrow + 4 * i) * LDS_STRIDE_KN + nnc) = rb[i];
      else
        *reinterpret_cast<d2_t*>(sb + row * LDS_STRIDE + lkc_sw) = rb[i];
    }
  };

  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

  const int fr = lane & 15, fk = lane >> 4;
  const int sw = (fr >= 4 && fr < 12) ? 1 : 0;
  const int a_off = (wr * 64 + fr) * LDS_STRIDE + ((fk ^ sw) << 2);
  const int b_off =
      BKN ? 4 * fk * LDS_STRIDE_KN + wc * 64 + fr : (wc * 64 + fr) * LDS_STRIDE + ((fk ^ sw) << 2);

  gload(0);
  sstore(0);
  __syncthreads();
  const int nk = (g.k - kbeg) / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const double* sa = smem + cur * 2 * TILE_DOUBLES;
    const double* sb = sa + TILE_DOUBLES;
    // lane (fr, fk) supplies k = 4 fk + q of the slab to MFMA step q (same map for A and B)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      d2_t a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * LDS_STRIDE + 2 * h);
        if (BKN)
          b[t] = d2_t{sb[b_off + (2 * h) * LDS_STRIDE_KN + t * 16],
                      sb[b_off + (2 * h + 1) * LDS_STRIDE_KN + t * 16]};
        else
          b[t] = *reinterpret_cast<const d2_t*>(sb + b_off + t * 16 * LDS_STRIDE + 2 * h);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: each access covers 4 rows x 128 contiguous bytes.  The 16 loads of one row-tile are
  // issued together before any store (a load-subtract-store chain per element serialises on memory
  // latency: 64 round trips per tile, measured 27 % of a workgroup's lifetime at K = 512).
  double* Cg = g.C + ((int64_t)ti * BM + wr * 64) * g.ldc + (int64_t)tj * BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double* rowp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rowp[r] = Cg + (int64_t)(i * 16 + fk + 4 * r) * g.ldc + fr;
    if (OP == OP_SUB) {
      double cv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[j][r] = rowp[r][j * 16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp[r][j * 16] = cv[j][r] - acc[i][j][r];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp[r][j * 16] = acc[i][j][r];
    }
  }
}

}  // namespace

void launch_gemm(hipStream_t s, GemmTiles tiles, GemmOp op, bool b_kmajor, bool kskip, double* C,
                 int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int ntr,
                 int ntc, int k) {
  if (ntr <= 0 || ntc <= 0 || k <= 0) return;
  GemmArgs g{C, A, B, ldc, lda, ldb, ntr, ntc, k, kskip ? 1 : 0};
  int nwg;
  if (tiles == TILES_RECT) {
    nwg = ntr * ntc;
  } else {
    if (ntc > ntr) ntc = g.ntc = ntr;
    nwg = ntc * (ntc + 1) / 2 + (ntr - ntc) * ntc;
  }
  dim3 grid((unsigned)nwg), block(256);
#define GPMI_LAUNCH(T, O, B) hipLaunchKernelGGL((gemm_nt_kernel<T, O, B>), grid, block, 0, s, g)
  if (tiles == TILES_RECT) {
    if (op == OP_SUB) {
      if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_SUB, 1); else GPMI_LAUNCH(TILES_RECT, OP_SUB, 0);
    } else {
      if (b_kmajor) GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 1); else GPMI_LAUNCH(TILES_RECT, OP_ASSIGN, 0);
    }
  } else {
    if (op == OP_SUB) GPMI_LAUNCH(TILES_LOWER, OP_SUB, 0); else GPMI_LAUNCH(TILES_LOWER, OP_ASSIGN, 0);
  }
#undef GPMI_LAUNCH
}

void launch_gemm_nt(hipStream_t s, GemmTiles tiles, GemmOp op, double* C, int64_t ldc,
                    const double* A, int64_t lda, const double* B, int64_t ldb, int ntr, int ntc,
                    int k) {
  launch_gemm(s, tiles, op, false, false, C, ldc, A, lda, B, ldb, ntr, ntc, k);
}
