// Triangular solves and reductions of the GP path (gfx950).
//
// Replaces scipy.linalg.solve_triangular at regression.py:213, 242-244, 410, 447, 538 and the
// reductions of regression.py:214, 539.  All solves use the inverted 128 x 128 diagonal blocks
// produced by potrf_diag, so every step is a matrix-vector / matrix-matrix product:
//   single right-hand side  -> HBM-bound GEMV sweeps (L is read exactly once per solve)
//   many right-hand sides    -> the fp64 MFMA GEMM (right-hand sides stored as rows, "NT" form)
#include "gpmi_internal.h"

namespace {

constexpr int NB = GPMI_NB;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- fused single-right-hand-side sweeps -------------------------------------------------------
// One launch per 128-block (instead of a diagonal-apply launch plus a panel launch): every
// workgroup recomputes the 128-vector of the block itself (a 128 x 128 GEMV out of L2, ~1 us) and
// then applies it to its own 64 rows / columns of L, so the only serialisation left is the kernel
// boundary.  Small workgroup slices (64 KiB of L each) keep every CU under its ~10 B/clk load limit.

// forward, block k:  v = invD_k r_k ;  block 0 of the grid stores v to vout ;  r[rows below] -= L[rows, k] v
// (r_k itself is left untouched: other workgroups of the launch are still reading it)
__global__ __launch_bounds__(256) void trsv_fwd_step_kernel(const double* __restrict__ Lp, int64_t ld,
                                                            const double* __restrict__ invD,
                                                            const double* __restrict__ rk,
                                                            double* __restrict__ vout,
                                                            double* __restrict__ rbelow, int64_t rows,
                                                            int64_t sMat, int64_t sInv, int64_t sVec) {
  Lp += (int64_t)blockIdx.z * sMat;
  invD += (int64_t)blockIdx.z * sInv;
  rk += (int64_t)blockIdx.z * sVec;
  vout += (int64_t)blockIdx.z * sVec;
  rbelow += (int64_t)blockIdx.z * sVec;
  __shared__ double rin[NB];
  __shared__ double v[NB];
  const int tid = threadIdx.x;
  if (tid < NB) rin[tid] = rk[tid];
  __syncthreads();
  {
    // v[row] = sum_c invD[row][c] rin[c]: two threads per row, 64 columns each (invD is lower triangular)
    const int row = tid >> 1, half = tid & 1;
    const double* p = invD + row * NB + half * 64;
    double s = 0.0;
#pragma unroll 8
    for (int c = 0; c < 64; c += 2) {
      const d2_t a = *reinterpret_cast<const d2_t*>(p + c);
      s = fma(a[0], rin[half * 64 + c], s);
      s = fma(a[1], rin[half * 64 + c + 1], s);
    }
    s += __shfl_xor(s, 1, 64);
    if (half == 0) v[row] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < NB) vout[tid] = v[tid];
  // rows below: 64 rows per workgroup, 4 threads per row (32 columns each)
  const int64_t row = (int64_t)blockIdx.x * 64 + (tid >> 2);
  const int part = tid & 3;
  double s = 0.0;
  if (row < rows) {
    const double* p = Lp + row * ld + part * 32;
#pragma unroll 8
    for (int c = 0; c < 32; c += 2) {
      const d2_t a = *reinterpret_cast<const d2_t*>(p + c);
      s = fma(a[0], v[part * 32 + c], s);
      s = fma(a[1], v[part * 32 + c + 1], s);
    }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (row < rows && part == 0) rbelow[row] -= s;
}

// backward, block k:  a = invD_k^T r_k ;  block 0 stores a to aout ;  r[j] -= sum_i L[k-block row i][j] a[i], j < cols
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(const double* __restrict__ Lr, int64_t ld,
                                                            const double* __restrict__ invD,
                                                            const double* __restrict__ rk,
                                                            double* __restrict__ aout,
                                                            double* __restrict__ r, int64_t cols) {
  __shared__ double rin[NB];
  __shared__ double part[2 * NB];
  __shared__ double a[NB];
  __shared__ double red[4 * 64];
  const int tid = threadIdx.x;
  if (tid < NB) rin[tid] = rk[tid];
  __syncthreads();
  {
    // a[c] = sum_i invD[i][c] rin[i]: thread (half, c) sums 64 rows; coalesced along c
    const int c = tid & 127, half = tid >> 7;
    double s = 0.0;
#pragma unroll 8
    for (int i = half * 64; i < half * 64 + 64; ++i) s = fma(invD[i * NB + c], rin[i], s);
    part[half * NB + c] = s;
  }
  __syncthreads();
  if (tid < NB) a[tid] = part[tid] + part[NB + tid];
  __syncthreads();
  if (blockIdx.x == 0 && tid < NB) aout[tid] = a[tid];
  // columns left of the block: 64 columns per workgroup, 4 row groups of 32 rows
  const int64_t j = (int64_t)blockIdx.x * 64 + (tid & 63);
  const int rg = tid >> 6;
  double s = 0.0;
  if (j < cols) {
#pragma unroll 8
    for (int i = rg * 32; i < rg * 32 + 32; ++i) s = fma(Lr[(int64_t)i * ld + j], a[i], s);
  }
  red[rg * 64 + (tid & 63)] = s;
  __syncthreads();
  if (tid < 64 && j < cols) r[j] -= red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
}

__global__ void set_identity_kernel(double* __restrict__ Q, int64_t ld, int64_t np) {
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t i = blockIdx.y;
  if (j < np)
    *reinterpret_cast<d2_t*>(Q + i * ld + j) = d2_t{j == i ? 1.0 : 0.0, j + 1 == i ? 1.0 : 0.0};
}

__global__ void copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void residual_batched_kernel(const double* __restrict__ y, const double* __restrict__ mus,
                                        const double* __restrict__ mu_consts, double* __restrict__ r,
                                        int64_t n, int64_t np, int64_t sVec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int64_t z = blockIdx.z;
  const double m = mus ? mus[z * n + (i < n ? i : 0)] : mu_consts[z];
  r[z * sVec + i] = (i < n) ? y[i] - m : 0.0;
}

__global__ void residual_kernel(const double* __restrict__ y, const double* __restrict__ mu,
                                double mu_const, double* __restrict__ r, int64_t n, int64_t np) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  r[i] = (i < n) ? y[i] - (mu ? mu[i] : mu_const) : 0.0;
}

// deterministic two-value reduction: red[0] = sum v^2, red[1] = sum ln L_ii  (fixed tree order)
__global__ __launch_bounds__(1024) void lml_reduce_kernel(const double* __restrict__ v,
                                                          const double* __restrict__ L, int64_t ld,
                                                          int64_t np, double* __restrict__ red,
                                                          int64_t sMat, int64_t sVec) {
  v += (int64_t)blockIdx.z * sVec;
  L += (int64_t)blockIdx.z * sMat;
  red += 2 * blockIdx.z;
  __shared__ double s0[16], s1[16];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < np; i += 1024) {
    const double vi = v[i];
    a = fma(vi, vi, a);
    b += log(L[i * ld + i]);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s0[wave] = a;
    s1[wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) {
      ta += s0[w];
      tb += s1[w];
    }
    red[0] = ta;
    red[1] = tb;
  }
}

// one wave per row: out[m] = sum_n Q[m][n] * a[n]
__global__ __launch_bounds__(256) void rows_dot_kernel(const double* __restrict__ Q, int64_t ld,
                                                       int64_t mp, int64_t np,
                                                       const double* __restrict__ a,
                                                       double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= mp) return;
  const double* q = Q + row * ld;
  double s = 0.0;
  for (int64_t j = lane * 2; j < np; j += 128) {
    const d2_t qv = *reinterpret_cast<const d2_t*>(q + j);
    const d2_t av = *reinterpret_cast<const d2_t*>(a + j);
    s = fma(qv[0], av[0], s);
    s = fma(qv[1], av[1], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double* __restrict__ Q, int64_t ld,
                                                         int64_t mp, int64_t np, double base,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= mp) return;
  const double* q = Q + row * ld;
  double s = 0.0;
  for (int64_t j = lane * 2; j < np; j += 128) {
    const d2_t qv = *reinterpret_cast<const d2_t*>(q + j);
    s = fma(qv[0], qv[0], s);
    s = fma(qv[1], qv[1], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = base - s;
}

}  // namespace

void trsv_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                  const double* invD, double* r, double* out, const BatchShape& bs) {
  const int nt = (int)(np / NB);
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)np * np, 4.0 * np * np);
  for (int k = 0; k < nt; ++k) {
    const int64_t rows = np - (int64_t)(k + 1) * NB;
    const unsigned blocks = rows > 0 ? (unsigned)((rows + 63) / 64) : 1u;
    hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3(blocks, 1, (unsigned)bs.count), dim3(256), 0, s,
                       L + (int64_t)(k + 1) * NB * ld + (int64_t)k * NB, ld, invD + (int64_t)k * NB * NB,
                       r + (int64_t)k * NB, out + (int64_t)k * NB, r + (int64_t)(k + 1) * NB, rows,
                       bs.sMat, bs.sInv, bs.sVec);
  }
}

void trsv_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                   const double* invD, double* r, double* out) {
  const int nt = (int)(np / NB);
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)np * np, 4.0 * np * np);
  for (int k = nt - 1; k >= 0; --k) {
    const int64_t cols = (int64_t)k * NB;
    const unsigned blocks = cols > 0 ? (unsigned)((cols + 63) / 64) : 1u;
    hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(blocks), dim3(256), 0, s, L + (int64_t)k * NB * ld, ld,
                       invD + (int64_t)k * NB * NB, r + (int64_t)k * NB, out + (int64_t)k * NB, r, cols);
  }
}

void trsm_rows_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                       const double* invD, double* Q, int64_t mp, bool upper_rhs) {
  // Two-level right-looking sweep, like the factorisation: inside an outer block of 512 columns the
  // 128-wide steps only touch the block (K = 128, few tiles, latency bound), then ONE update with
  // K = 512 carries the block's contribution to all remaining columns (throughput bound, 4x less
  // traffic on Q than 128-wide updates).
  const int nt = (int)(np / NB), mt_all = (int)(mp / NB);
  const int OBT = 4;
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)mp * np * np, 4.0 * np * np);
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    // upper_rhs: Q is upper triangular (e.g. the identity): rows below block Je - 1 are still zero here
    const int mt = upper_rhs ? (Je < mt_all ? Je : mt_all) : mt_all;
    for (int k = J; k < Je; ++k) {
      double* Qk = Q + (int64_t)k * NB;
      // Q[:, k] <- Q[:, k] * invD_k^T   (in place, one tile column)
      launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, Qk, ld, Qk, ld, invD + (int64_t)k * NB * NB, NB, mt, 1, NB);
      const int rem = Je - k - 1;
      if (rem > 0)  // Q[:, k+1:Je] -= Q[:, k] * L[k+1:Je, k]^T
        launch_gemm_nt(s, TILES_RECT, OP_SUB, Qk + NB, ld, Qk, ld,
                       L + (int64_t)(k + 1) * NB * ld + (int64_t)k * NB, ld, mt, rem, NB);
    }
    const int rest = nt - Je;
    if (rest > 0)  // Q[:, Je:] -= Q[:, J:Je] * L[Je:, J:Je]^T
      launch_gemm_nt(s, TILES_RECT, OP_SUB, Q + (int64_t)Je * NB, ld, Q + (int64_t)J * NB, ld,
                     L + (int64_t)Je * NB * ld + (int64_t)J * NB, ld, mt, rest, (Je - J) * NB);
  }
}

void trsm_rows_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                        const double* invD, double* Q, int64_t mp) {
  const int nt = (int)(np / NB), mt = (int)(mp / NB);
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)mp * np * np, 4.0 * np * np);
  for (int k = nt - 1; k >= 0; --k) {
    double* Qk = Q + (int64_t)k * NB;
    // Q[:, k] <- Q[:, k] * invD_k          (B = invD_k is k-major here)
    launch_gemm(s, TILES_RECT, OP_ASSIGN, true, false, Qk, ld, Qk, ld, invD + (int64_t)k * NB * NB,
                NB, mt, 1, NB);
    // Q[:, 0:k] -= Q[:, k] * L[k, 0:k]     (B = block row k of L, k-major)
    if (k > 0)
      launch_gemm(s, TILES_RECT, OP_SUB, true, false, Q, ld, Qk, ld, L + (int64_t)k * NB * ld, ld, mt,
                  k, NB);
  }
}

void launch_set_identity(hipStream_t s, double* Q, int64_t ld, int64_t np) {
  dim3 grid((unsigned)((np / 2 + 255) / 256), (unsigned)np);
  hipLaunchKernelGGL(set_identity_kernel, grid, dim3(256), 0, s, Q, ld, np);
}

void launch_copy(hipStream_t s, const double* src, double* dst, int64_t n) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}

void launch_residual(hipStream_t s, const double* y, const double* mu, double mu_const, double* r,
                     int64_t n, int64_t np) {
  hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, y, mu,
                     mu_const, r, n, np);
}

void launch_lml_reduce(hipStream_t s, const double* v, const double* L, int64_t ld, int64_t np,
                       double* red, const BatchShape& bs) {
  hipLaunchKernelGGL(lml_reduce_kernel, dim3(1, 1, (unsigned)bs.count), dim3(1024), 0, s, v, L, ld, np, red,
                     bs.sMat, bs.sVec);
}

void launch_residual_batched(hipStream_t s, const double* y, const double* mus, const double* mu_consts,
                             double* r, int64_t n, int64_t np, const BatchShape& bs) {
  hipLaunchKernelGGL(residual_batched_kernel, dim3((unsigned)((np + 255) / 256), 1, (unsigned)bs.count),
                     dim3(256), 0, s, y, mus, mu_consts, r, n, np, bs.sVec);
}

void launch_rows_dot(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                     const double* a, double* out) {
  hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((mp + 3) / 4)), dim3(256), 0, s, Q, ld, mp, np,
                     a, out);
}

void launch_rows_sumsq(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                       double base, double* out) {
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3((unsigned)((mp + 3) / 4)), dim3(256), 0, s, Q, ld, mp,
                     np, base, out);
}
