// Spatial-gradient kernels of the GP predictor (SquaredExponential only, as in the reference:
// RationalQuadratic has no gradient_terms, covariance.py:38-44).
//
// Replace the per-point loops of GpRegressor.gradient (regression.py:351-385) and
// GpRegressor.spatial_derivatives (regression.py:387-419) by batched kernels over M query points:
//   A_in = (x_n,i - q_i) / l_i^2         (gradient_terms, covariance.py:257-266)
//   dmu_i   = sum_n A_in k_n alpha_n                       dvar_i = -2 sum_n A_in k_n (K^-1 k)_n
//   cov_ij  = R_j - sum_n Q_in Q_jn,  Q = L^-1 (A o k)^T,  R_j = (a / l_j)^2
#include "gpmi_internal.h"

namespace {

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ inline double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out[m][i] = scale * sum_n (x[n][i] - q[m][i]) il2[i] * Kq[m][n] * W[m][n]   (W row-strided by ldw; ldw = 0 -> vector)
__global__ __launch_bounds__(256) void sd_reduce_kernel(KParams p, const double* __restrict__ x,
                                                        int64_t n, const double* __restrict__ pts,
                                                        const double* __restrict__ Kq, int64_t ld,
                                                        const double* __restrict__ W, int64_t ldw,
                                                        double scale, double* __restrict__ out) {
  __shared__ double red[4];
  const int64_t m = blockIdx.x;
  const double* kq = Kq + m * ld;
  const double* w = W + m * ldw;
  for (int i = 0; i < p.d; ++i) {
    const double qi = pts[m * p.d + i];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) acc = fma((x[j * p.d + i] - qi) * kq[j], w[j], acc);
    const double v = block_sum(acc, red);
    if (threadIdx.x == 0) out[m * p.d + i] = scale * p.inv_l2[i] * v;
  }
}

// G[(m * d + i)][n] = (x[n][i] - q[m][i]) il2[i] Kq[m][n]; rows >= m_valid * d and columns >= n are zero
__global__ __launch_bounds__(256) void grad_rhs_kernel(KParams p, const double* __restrict__ x,
                                                       int64_t n, int64_t np,
                                                       const double* __restrict__ pts, int64_t rows_valid,
                                                       const double* __restrict__ Kq, int64_t ld,
                                                       double* __restrict__ G) {
  const int64_t row = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= np) return;
  double v = 0.0;
  if (row < rows_valid && j < n) {
    const int64_t m = row / p.d;
    const int i = (int)(row - m * p.d);
    v = (x[j * p.d + i] - pts[m * p.d + i]) * p.inv_l2[i] * Kq[m * ld + j];
  }
  G[row * ld + j] = v;
}

// cov[m][i][j] = R[j] - sum_n G[(m d + i)][n] G[(m d + j)][n]
__global__ __launch_bounds__(256) void grad_cov_kernel(KParams p, const double* __restrict__ G,
                                                       int64_t ld, int64_t np,
                                                       double* __restrict__ cov) {
  __shared__ double red[4];
  const int64_t m = blockIdx.x;
  const int d = p.d;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      const double* gi = G + (m * d + i) * ld;
      const double* gj = G + (m * d + j) * ld;
      double acc = 0.0;
      for (int64_t t = threadIdx.x; t < np; t += 256) acc = fma(gi[t], gj[t], acc);
      const double v = block_sum(acc, red);
      if (threadIdx.x == 0) {
        cov[(m * d + i) * d + j] = p.a2 * p.inv_l2[j] - v;  // R - Q^T Q broadcasts R over rows (regression.py:380)
        cov[(m * d + j) * d + i] = p.a2 * p.inv_l2[i] - v;
      }
    }
}

}  // namespace

void launch_sd_reduce(hipStream_t s, const KParams& p, const double* x, int64_t n, const double* pts,
                      int64_t m, const double* Kq, int64_t ld, const double* W, int64_t ldw,
                      double scale, double* out) {
  hipLaunchKernelGGL(sd_reduce_kernel, dim3((unsigned)m), dim3(256), 0, s, p, x, n, pts, Kq, ld, W,
                     ldw, scale, out);
}

void launch_grad_rhs(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                     const double* pts, int64_t rows_valid, int64_t rows_padded, const double* Kq,
                     int64_t ld, double* G) {
  dim3 grid((unsigned)((np + 255) / 256), (unsigned)rows_padded);
  hipLaunchKernelGGL(grad_rhs_kernel, grid, dim3(256), 0, s, p, x, n, np, pts, rows_valid, Kq, ld, G);
}

void launch_grad_cov(hipStream_t s, const KParams& p, const double* G, int64_t ld, int64_t np,
                     int64_t m, double* cov) {
  hipLaunchKernelGGL(grad_cov_kernel, dim3((unsigned)m), dim3(256), 0, s, p, G, ld, np, cov);
}
