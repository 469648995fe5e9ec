// fp64 MFMA GEMM for gfx950:  C (op)= A * B^T  with both operands K-contiguous ("NT").
//
// This is the one dense contraction of the GP path: the trailing SYRK/GEMM update of the blocked
// Cholesky factorisation (numpy.linalg.cholesky at regression.py:241,537,555), the panel TRSM done as
// a product with the inverted diagonal block, and the triangular solves with many right-hand sides
// of the batched predict (regression.py:213, 447).
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 MFMA tiles of
// v_mfma_f64_16x16x4_f64, 128 accumulator VGPRs), BK = 16, LDS double-buffered (72 KiB -> 2
// workgroups per CU = 2 waves per SIMD).  LDS rows are padded to 18 doubles: the ds_read_b64 of a
// fragment (16 rows x 2 k per 32-lane half) then touches 32 distinct bank pairs.
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

template <int TILES>
__device__ inline void tile_of(int id, int ntr, int ntc, int& ti, int& tj) {
  if (TILES == TILES_RECT) {
    ti = id / ntc;
    tj = id - ti * ntc;
  } else {
    // rows ti < ntc hold ti + 1 tiles (triangle), rows ti >= ntc hold ntc tiles
    const int tri = ntc * (ntc + 1) / 2;
    if (id < tri) {
      int t = (int)((sqrt(8.0 * id + 1.0) - 1.0) * 0.5);
      while ((t + 1) * (t + 2) / 2 <= id) ++t;
      while (t * (t + 1) / 2 > id) --t;
      ti = t;
      tj = id - t * (t + 1) / 2;
    } else {
      const int rem = id - tri;
      ti = ntc + rem / ntc;
      tj = rem % ntc;
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
      *reinterpret_cast<d2_t*>(sa + row * LDS_STRIDE + lkc) = ra[i];
      if (BKN)
        *reinterpret_cast<d2_t*>(sb + (nrow + 4 * i) * LDS_STRIDE_KN + nnc) = rb[i];
      else
        *reinterpret_cast<d2_t*>(sb + row * LDS_STRIDE + lkc) = rb[i];
    }
  };

  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

  const int fr = lane & 15, fk = lane >> 4;
  const int a_off = (wr * 64 + fr) * LDS_STRIDE + fk;
  const int b_off = BKN ? fk * LDS_STRIDE_KN + wc * 64 + fr : (wc * 64 + fr) * LDS_STRIDE + fk;

  gload(0);
  sstore(0);
  __syncthreads();
  const int nk = (g.k - kbeg) / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const double* sa = smem + cur * 2 * TILE_DOUBLES;
    const double* sb = sa + TILE_DOUBLES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = sa[a_off + t * 16 * LDS_STRIDE + q * 4];
        b[t] = BKN ? sb[b_off + q * 4 * LDS_STRIDE_KN + t * 16] : sb[b_off + t * 16 * LDS_STRIDE + q * 4];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: each store instruction covers 4 rows x 128 contiguous bytes
  double* Cg = g.C + ((int64_t)ti * BM + wr * 64) * g.ldc + (int64_t)tj * BN + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* p = Cg + (int64_t)(i * 16 + fk + 4 * r) * g.ldc + j * 16 + fr;
        if (OP == OP_SUB)
          *p = *p - acc[i][j][r];
        else
          *p = acc[i][j][r];
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
