// Fused log-marginal-likelihood gradient contraction (gfx950).
//
// Replaces the dense gradient matrices of covariance_and_gradients (covariance.py:268-276, 350-365:
// d+1 / d+2 N x N arrays) and the reductions of regression.py:565-566
//     grad_j = 1/2 sum_ab (alpha_a alpha_b - K^-1_ab) dK_j[a][b]
// and, with u = K^-1 c1, v = alpha, M = K^-1 diag(c2) K^-1, the leave-one-out gradient of
// regression.py:508-514 (the same contraction: sum_ab dK_j[a][b] (sym(u v^T) - M)_ab)
// by one pass over the lower triangle of K^-1: every 64 x 64 tile recomputes K and dK_j from the
// staged point panels, weights them with Q_ab = alpha_a alpha_b - K^-1_ab (off-diagonal elements
// counted twice) and reduces to one partial per tile and parameter; a second kernel sums the
// partials in a fixed order (bit-reproducible).  HBM-read bound: 8 bytes per element of K^-1.
#include "gpmi_internal.h"

namespace {

constexpr int KT = 64;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum (256 threads), result valid in thread 0
__device__ inline double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// batch (blockIdx.z, lockstep evaluations of gpmi_lml_grad_batch): problem z takes pdev[z], iK + z sK, u / v + z sV,
// ws + z sW
__device__ __forceinline__ void lml_grad_body(const KParams& p, int n_theta, const double* __restrict__ x, int64_t n,
                                              const double* __restrict__ iK, int64_t ld,
                                              const double* __restrict__ uvec, const double* __restrict__ vvec,
                                              double* __restrict__ ws);

__global__ __launch_bounds__(256) void lml_grad_kernel(KParams p, int n_theta,
                                                       const double* __restrict__ x, int64_t n,
                                                       const double* __restrict__ iK, int64_t ld,
                                                       const double* __restrict__ uvec,
                                                       const double* __restrict__ vvec,
                                                       double* __restrict__ ws) {
  lml_grad_body(p, n_theta, x, n, iK, ld, uvec, vvec, ws);
}

__global__ __launch_bounds__(256) void lml_grad_batched_kernel(const KParams* __restrict__ pdev, int n_theta,
                                                               const double* __restrict__ x, int64_t n,
                                                               const double* __restrict__ iK, int64_t ld,
                                                               const double* __restrict__ uvec,
                                                               const double* __restrict__ vvec, double* __restrict__ ws,
                                                               int64_t sK, int64_t sV, int64_t sW) {
  const int64_t z = blockIdx.z;
  lml_grad_body(pdev[z], n_theta, x, n, iK + z * sK, ld, uvec + z * sV, vvec + z * sV, ws + z * sW);
}

__device__ __forceinline__ void lml_grad_body(const KParams& p, int n_theta, const double* __restrict__ x, int64_t n,
                                              const double* __restrict__ iK, int64_t ld,
                                              const double* __restrict__ uvec, const double* __restrict__ vvec,
                                              double* __restrict__ ws) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  __shared__ double su[GPMI_MAX_D * KT];
  __shared__ double sv[GPMI_MAX_D * KT];
  __shared__ double red[4];
  const int tid = threadIdx.x, d = p.d;
  const int64_t i0 = (int64_t)ti * KT, j0 = (int64_t)tj * KT;
  for (int idx = tid; idx < KT * d; idx += 256) {
    int pt = idx / d, k = idx - pt * d;
    int64_t gi = i0 + pt, gj = j0 + pt;
    su[k * KT + pt] = (gi < n) ? x[gi * d + k] : 0.0;
    sv[k * KT + pt] = (gj < n) ? x[gj * d + k] : 0.0;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  double s[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s[r][c] = 0.0;
  for (int k = 0; k < d; ++k) {
    const double il2 = p.inv_l2[k];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double dx = su[k * KT + ty * 4 + r] - sv[k * KT + tx * 4 + c];
        s[r][c] = fma(0.5 * dx * dx, il2, s[r][c]);
      }
  }
  // wk = 1/2 * multiplicity * Q_ab * (factor multiplying dx_k^2 / l_k^2 in dK_{scale k})
  double wk[4][4];
  double g_amp = 0.0, g_shape = 0.0, tq = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t ga = i0 + ty * 4 + r;
    const double ua = (ga < n) ? uvec[ga] : 0.0;
    const double va = (ga < n) ? vvec[ga] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t gb = j0 + tx * 4 + c;
      double w = 0.0;
      if (ga < n && gb <= ga) {
        const double q = 0.5 * (ua * vvec[gb] + va * uvec[gb]) - iK[ga * ld + gb];
        w = (ga == gb) ? 0.5 * q : q;
        if (ga == gb) tq += q;
      }
      const double z = s[r][c];
      if (p.kernel == GPMI_KERNEL_SE) {
        double K = p.a2 * exp(-z);
        if (ga == gb) K = p.a2 * (exp(-z) + 1e-12);
        g_amp = fma(w, 2.0 * K, g_amp);   // dK/d ln a = 2 K            (covariance.py:273)
        wk[r][c] = w * K;                 // dK/d ln l_k = (dx_k^2 / l_k^2) K   (covariance.py:275)
      } else {
        const double F = 1.0 + z / p.kappa;
        const double lnF = log(F);
        double K = p.a2 * exp(-p.kappa * lnF);  // covariance.py:356-360
        if (ga == gb) K = p.a2 * (exp(-p.kappa * lnF) + 1e-12);
        g_amp = fma(w, 2.0 * K, g_amp);
        g_shape = fma(w, -K * (lnF * p.kappa - z / F), g_shape);  // covariance.py:361
        wk[r][c] = w * K / F;             // G (distances_k / l_k^2) = (K / F) dx_k^2 / l_k^2   (covariance.py:362-364)
      }
    }
  }
  const int64_t tile = (int64_t)ti * (ti + 1) / 2 + tj;
  double* out = ws + tile * (n_theta + 1);
  const int off = (p.kernel == GPMI_KERNEL_SE) ? 1 : 2;
  double v = block_sum(g_amp, red);
  if (tid == 0) out[0] = v;
  if (off == 2) {
    v = block_sum(g_shape, red);
    if (tid == 0) out[1] = v;
  }
  v = block_sum(tq, red);
  if (tid == 0) out[n_theta] = v;
  for (int k = 0; k < d; ++k) {
    const double il2 = p.inv_l2[k];
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double dx = su[k * KT + ty * 4 + r] - sv[k * KT + tx * 4 + c];
        acc = fma(wk[r][c], dx * dx * il2, acc);
      }
    v = block_sum(acc, red);
    if (tid == 0) out[off + k] = v;
  }
}

// out[j] = sum over tiles of ws[tile][j], fixed order: thread-strided partial sums, then a tree
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* __restrict__ ws,
                                                          int64_t ntiles, int width,
                                                          double* __restrict__ out, int64_t sW) {
  __shared__ double red[4];
  const int j = blockIdx.x;
  ws += (int64_t)blockIdx.z * sW;        // batch: partials of problem z, results at out + z * width
  out += (int64_t)blockIdx.z * width;
  double acc = 0.0;
  for (int64_t t = threadIdx.x; t < ntiles; t += 256) acc += ws[t * width + j];
  const double v = block_sum(acc, red);
  if (threadIdx.x == 0) out[j] = v;
}

// A[j][i] = A[i][j] for j > i (tile-wise through LDS so that both sides are coalesced)
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int64_t ld, int64_t sA) {
  __shared__ double t[64][65];
  A += (int64_t)blockIdx.z * sA;
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) t[r][tx] = A[((int64_t)ti * 64 + r) * ld + (int64_t)tj * 64 + tx];
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t row = (int64_t)tj * 64 + r, col = (int64_t)ti * 64 + tx;
    if (col > row) A[row * ld + col] = t[tx][r];
  }
}

// G[b][a] = A[b][a] * s[a]
__global__ void scale_columns_kernel(const double* __restrict__ A, const double* __restrict__ s,
                                     double* __restrict__ G, int64_t ld, int64_t np, int64_t sMat, int64_t sVec) {
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t i = blockIdx.y;
  if (j >= np) return;
  A += (int64_t)blockIdx.z * sMat;
  G += (int64_t)blockIdx.z * sMat;
  s += (int64_t)blockIdx.z * sVec;
  const d2_t a = *reinterpret_cast<const d2_t*>(A + i * ld + j);
  *reinterpret_cast<d2_t*>(G + i * ld + j) = d2_t{a[0] * s[j], a[1] * s[j + 1]};
}

// leave-one-out vectors (regression.py:505-510): var = 1 / diag(K^-1), c1 = alpha var,
// sqrt(c2) with c2 = 1/2 var (1 + var alpha^2); zero in the padding
__global__ void loo_vectors_kernel(const double* __restrict__ alpha, const double* __restrict__ ikdiag,
                                   double* __restrict__ c1, double* __restrict__ sc2, int64_t n,
                                   int64_t np, int64_t sAlpha, int64_t sLoo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  alpha += (int64_t)blockIdx.z * sAlpha;
  ikdiag += (int64_t)blockIdx.z * sLoo;
  c1 += (int64_t)blockIdx.z * sLoo;
  sc2 += (int64_t)blockIdx.z * sLoo;
  double a = 0.0, b = 0.0;
  if (i < n) {
    const double var = 1.0 / ikdiag[i];
    a = alpha[i] * var;
    b = sqrt(0.5 * var * (1.0 + var * alpha[i] * alpha[i]));
  }
  c1[i] = a;
  sc2[i] = b;
}

}  // namespace

void launch_mirror_lower(hipStream_t s, double* A, int64_t ld, int64_t np, int batch, int64_t sMat) {
  dim3 grid((unsigned)(np / 64), (unsigned)(np / 64), (unsigned)batch);
  hipLaunchKernelGGL(mirror_lower_kernel, grid, dim3(256), 0, s, A, ld, sMat);
}

void launch_scale_columns(hipStream_t s, const double* A, const double* sc, double* G, int64_t ld,
                          int64_t np, int batch, int64_t sMat, int64_t sVec) {
  dim3 grid((unsigned)((np / 2 + 255) / 256), (unsigned)np, (unsigned)batch);
  hipLaunchKernelGGL(scale_columns_kernel, grid, dim3(256), 0, s, A, sc, G, ld, np, sMat, sVec);
}

void launch_loo_vectors(hipStream_t s, const double* alpha, const double* ikdiag, double* c1,
                        double* sc2, int64_t n, int64_t np, int batch, int64_t sAlpha, int64_t sLoo) {
  hipLaunchKernelGGL(loo_vectors_kernel, dim3((unsigned)((np + 255) / 256), 1, (unsigned)batch), dim3(256), 0, s, alpha,
                     ikdiag, c1, sc2, n, np, sAlpha, sLoo);
}

int64_t grad_ws_doubles(int64_t np, int n_theta) {
  const int64_t t = np / KT;
  return t * (t + 1) / 2 * (n_theta + 1);
}

void launch_lml_grad(hipStream_t s, const KParams& p, int n_theta, const double* x, int64_t n,
                     int64_t np, const double* iK, int64_t ld, const double* u, const double* v,
                     double* ws, double* out) {
  const int64_t t = np / KT;
  dim3 grid((unsigned)t, (unsigned)t);
  hipLaunchKernelGGL(lml_grad_kernel, grid, dim3(256), 0, s, p, n_theta, x, n, iK, ld, u, v, ws);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(n_theta + 1)), dim3(256), 0, s, ws,
                     t * (t + 1) / 2, n_theta + 1, out, (int64_t)0);
}

// B lockstep evaluations: out[z * (n_theta + 1) ..] = gradient and trace of problem z (likelihood gradient: u = v =
// alpha_z; leave-one-out gradient: u = p_z, v = alpha_z, both with stride sV)
void launch_lml_grad_batched(hipStream_t s, const KParams* pdev, int batch, int n_theta, const double* x, int64_t n,
                             int64_t np, const double* iK, int64_t ld, int64_t sK, const double* u, const double* v,
                             int64_t sV, double* ws, double* out) {
  const int64_t t = np / KT;
  const int64_t sW = grad_ws_doubles(np, n_theta);
  dim3 grid((unsigned)t, (unsigned)t, (unsigned)batch);
  hipLaunchKernelGGL(lml_grad_batched_kernel, grid, dim3(256), 0, s, pdev, n_theta, x, n, iK, ld, u, v, ws, sK, sV, sW);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(n_theta + 1), 1, (unsigned)batch), dim3(256), 0, s, ws,
                     t * (t + 1) / 2, n_theta + 1, out, sW);
}
