// Element-wise helpers of the mixture covariance  K = sum_m diag(g_m) K_m diag(g_m)  (ChangePoint,
// covariance.py:371-606: g_m are products of logistic windows along one axis, evaluated on the host in
// O(N)).  Every K_m is produced by the ordinary covariance-build kernel into a scratch matrix; these kernels
// fold it into the sum and form the extra reductions the change-point gradients need.  All HBM bound.
#include "gpmi_internal.h"

namespace {

// dst = (or +=) gr_i gc_j src_ij over rows x cols (both multiples of 128), 2 doubles per thread
// (batched: problem blockIdx.z works on dst + z sD, src + z sS and the weight vectors gr / gc + z sG)
__global__ void scale_add_kernel(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src,
                                 int64_t lds, const double* __restrict__ gr, const double* __restrict__ gc,
                                 int64_t cols, int accumulate, int64_t sD, int64_t sS, int64_t sG) {
  dst += (int64_t)blockIdx.z * sD;
  src += (int64_t)blockIdx.z * sS;
  gr += (int64_t)blockIdx.z * sG;
  gc += (int64_t)blockIdx.z * sG;
  const int64_t i = blockIdx.y;
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (j >= cols) return;
  const double r = gr[i];
  const d2_t s = *reinterpret_cast<const d2_t*>(src + i * lds + j);
  d2_t v = d2_t{r * gc[j] * s[0], r * gc[j + 1] * s[1]};
  if (accumulate) {
    const d2_t o = *reinterpret_cast<const d2_t*>(dst + i * ldd + j);
    v = d2_t{o[0] + v[0], o[1] + v[1]};
  }
  *reinterpret_cast<d2_t*>(dst + i * ldd + j) = v;
}

// (batched: problem blockIdx.z works on A + z sA and adds extras[z] when `extras` is given)
__global__ void add_diag_vec_kernel(double* __restrict__ A, int64_t ld, const double* __restrict__ noise,
                                    double extra, int64_t n, int64_t sA, const double* __restrict__ extras) {
  A += (int64_t)blockIdx.z * sA;
  if (extras) extra = extras[blockIdx.z];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * ld + i] += noise[i] + extra;
}

__global__ void vec_mul_kernel(const double* __restrict__ a, const double* __restrict__ b,
                               double* __restrict__ out, int64_t n, int64_t sA, int64_t sB, int64_t sOut) {
  a += (int64_t)blockIdx.z * sA;
  b += (int64_t)blockIdx.z * sB;
  out += (int64_t)blockIdx.z * sOut;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

// one wave per row:  h_i = sum_j (1/2 (u_i v_j + v_i u_j) - M_ij) Km_ij g_j   (fixed order: bit-reproducible).
// LML gradient: u = v = alpha, M = K^-1 (1/2 (x + x) = x exactly: the same bits as alpha_i alpha_j); leave-one-out
// gradient: u = p, v = alpha, M = K^-1 diag(c2) K^-1 (regression.py:509-514)
__global__ __launch_bounds__(256) void mix_rowsum_kernel(const double* __restrict__ iK,
                                                         const double* __restrict__ Km, int64_t ld,
                                                         const double* __restrict__ u, const double* __restrict__ v,
                                                         const double* __restrict__ g,
                                                         double* __restrict__ h, int64_t n, int64_t sMat,
                                                         int64_t sU, int64_t sV, int64_t sG, int64_t pair) {
  iK += (int64_t)blockIdx.z * sMat;
  Km += (int64_t)blockIdx.z * sMat;
  u += (int64_t)blockIdx.z * sU;
  v += (int64_t)blockIdx.z * sV;
  g += (int64_t)blockIdx.z * sG;
  h += (int64_t)blockIdx.z * sG;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double ui = u[row], vi = v[row];
  const double* q = iK + row * ld;
  const double* k = Km + row * ld;
  // pair != 0: a second weight vector `pair` doubles behind g, its row sums `pair` doubles behind h - the two window
  // factors of a ChangePoint sub-kernel from ONE pass over iK and Km (round 6; each sum in the order it has alone)
  const double* g2 = g + pair;
  double s = 0.0, s2 = 0.0;
  for (int64_t j = lane * 2; j < n; j += 128) {
    const d2_t qv = *reinterpret_cast<const d2_t*>(q + j);
    const d2_t kv = *reinterpret_cast<const d2_t*>(k + j);
    const double t0 = (0.5 * (ui * v[j] + vi * u[j]) - qv[0]) * kv[0];
    s = fma(t0, g[j], s);
    if (pair) s2 = fma(t0, g2[j], s2);
    if (j + 1 < n) {
      const double t1 = (0.5 * (ui * v[j + 1] + vi * u[j + 1]) - qv[1]) * kv[1];
      s = fma(t1, g[j + 1], s);
      if (pair) s2 = fma(t1, g2[j + 1], s2);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  if (lane == 0) {
    h[row] = s;
    if (pair) h[row + pair] = s2;
  }
}

}  // namespace

void launch_scale_add(hipStream_t s, double* dst, int64_t ldd, const double* src, int64_t lds,
                      const double* gr, const double* gc, int64_t rows, int64_t cols, bool accumulate, int batch,
                      int64_t sD, int64_t sS, int64_t sG) {
  dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)rows, (unsigned)batch);
  hipLaunchKernelGGL(scale_add_kernel, grid, dim3(256), 0, s, dst, ldd, src, lds, gr, gc, cols,
                     accumulate ? 1 : 0, sD, sS, sG);
}

void launch_add_diag_vec(hipStream_t s, double* A, int64_t ld, const double* noise, double extra, int64_t n, int batch,
                         int64_t sA, const double* extras) {
  hipLaunchKernelGGL(add_diag_vec_kernel, dim3((unsigned)((n + 255) / 256), 1, (unsigned)batch), dim3(256), 0, s, A, ld,
                     noise, extra, n, sA, extras);
}

void launch_vec_mul(hipStream_t s, const double* a, const double* b, double* out, int64_t n, int batch, int64_t sA,
                    int64_t sB, int64_t sOut) {
  hipLaunchKernelGGL(vec_mul_kernel, dim3((unsigned)((n + 255) / 256), 1, (unsigned)batch), dim3(256), 0, s, a, b, out, n,
                     sA, sB, sOut);
}

void launch_mix_rowsum(hipStream_t s, const double* iK, const double* Km, int64_t ld, const double* alpha,
                       const double* g, double* h, int64_t n, int batch, int64_t sMat, int64_t sAlpha, int64_t sG,
                       const double* u, int64_t sU, int64_t pair) {
  if (!u) {  // the LML form: u = v = alpha
    u = alpha;
    sU = sAlpha;
  }
  hipLaunchKernelGGL(mix_rowsum_kernel, dim3((unsigned)((n + 3) / 4), 1, (unsigned)batch), dim3(256), 0, s, iK, Km, ld,
                     u, alpha, g, h, n, sMat, sU, sAlpha, sG, pair);
}
