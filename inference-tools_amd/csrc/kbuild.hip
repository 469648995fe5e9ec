// Covariance-matrix build for SquaredExponential / RationalQuadratic on gfx950.
//
// Replaces the reference's N x N x d broadcast tensors (covariance.py:218-219, 254, 315-316, 347)
// by a tiled kernel: a workgroup stages the two 64-point panels in LDS (transposed, [dim][point]) and
// every thread produces a 4 x 4 block of K, written as 32-byte row segments (16 threads -> 512
// contiguous bytes per row).  HBM-write bound: 8 bytes per element, x is read once per tile from L2.
#include "gpmi_internal.h"
#include "kmath.h"

namespace {

constexpr int KT = 64;  // output tile edge

// Covariance function of N elements at once (kmath.h: a thread's elements go through exp / log1p in lockstep, so a
// polynomial coefficient is fetched once per N FMAs; the library's exp() inlined per element re-materialised its
// constants at every use - as many v_mov_b32 as arithmetic - and its pow() is ~300 instructions per element).
// KERNEL is a template parameter of the builders: one covariance function per kernel keeps the code within the
// instruction cache two CUs share.   s = sum_k 0.5 * dx_k^2 / l_k^2  (>= 0)
template <int KERNEL, int N>
__device__ inline void kfun(const KParams& p, const double (&s)[N], double (&out)[N]) {
  double e[N];
  if (KERNEL == GPMI_KERNEL_SE) {  // exp(-s)  (covariance.py:254)
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = -s[i];
  } else {  // (1 + s / kappa)^-kappa = exp(-kappa log1p(s / kappa))  (covariance.py:348)
    const double ik = 1.0 / p.kappa;
    double z[N], l[N];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = s[i] * ik;
    kmath::log1p_pos(z, l);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = -p.kappa * l[i];
  }
  kmath::exp_neg(e, out);
}

// SQUARE: U == V, jitter + noise on the diagonal, identity in the padding (rows/cols >= n).
template <bool SQUARE, int KERNEL>
__device__ inline void kbuild_body(const KParams& p, const double* __restrict__ U, int64_t nu,
                                   const double* __restrict__ V, int64_t nv,
                                   const double* __restrict__ noise, double* __restrict__ out,
                                   int64_t ld, int lower_only);

template <bool SQUARE, int KERNEL>
__global__ __launch_bounds__(256) void kbuild_kernel(KParams p, const double* __restrict__ U,
                                                     int64_t nu, const double* __restrict__ V,
                                                     int64_t nv, const double* __restrict__ noise,
                                                     double* __restrict__ out, int64_t ld,
                                                     int lower_only) {
  kbuild_body<SQUARE, KERNEL>(p, U, nu, V, nv, noise, out, ld, lower_only);
}

// batched square build: problem z uses hyper-parameters pdev[z] and writes matrix out + z * stride
template <int KERNEL>
__global__ __launch_bounds__(256) void kbuild_batched_kernel(const KParams* __restrict__ pdev,
                                                             const double* __restrict__ x, int64_t n,
                                                             const double* __restrict__ noise,
                                                             double* __restrict__ out, int64_t ld,
                                                             int64_t stride, int64_t noise_stride) {
  // noise_stride > 0: every problem has noise variances of its own (HeteroscedasticNoise: they are hyper-parameters)
  kbuild_body<true, KERNEL>(pdev[blockIdx.z], x, n, x, n, noise + (int64_t)blockIdx.z * noise_stride,
                            out + (int64_t)blockIdx.z * stride, ld, 2);
}

template <bool SQUARE, int KERNEL>
__device__ inline void kbuild_body(const KParams& p, const double* __restrict__ U, int64_t nu,
                                   const double* __restrict__ V, int64_t nv,
                                   const double* __restrict__ noise, double* __restrict__ out,
                                   int64_t ld, int lower_only) {
  int ti = blockIdx.y, tj = blockIdx.x;
  const int tile_off = lower_only >> 8;  // mode 2: the triangle starts at tile (tile_off, tile_off)
  lower_only &= 0xff;
  if (SQUARE && lower_only == 2) {
    // one-dimensional grid over the lower tiles only (row by row): id = ti (ti + 1) / 2 + tj.  (Half of a square
    // grid's 61 504 workgroups at N = 16384 did nothing but start and exit: 0.48 -> 0.45 ms per build; the rest is the f64 exp, VALU-bound.)
    const int id = blockIdx.x;
    ti = (int)((sqrt(8.0 * id + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    while (ti * (ti + 1) / 2 > id) --ti;
    tj = id - ti * (ti + 1) / 2 + tile_off;
    ti += tile_off;
  } else if (SQUARE && lower_only && tj > ti) {
    return;
  }
  // the two point panels, [dim][point]: 2 x d x 64 doubles of dynamic LDS (a static 2 x GPMI_MAX_D x 64 = 64 KiB held
  // the kernel at two workgroups per CU whatever d was; at d = 8 it needs 8 KiB)
  extern __shared__ double kb_lds[];
  const int tid = threadIdx.x;
  const int d = p.d;
  double* su = kb_lds;
  double* sv = kb_lds + d * KT;
  const int64_t i0 = (int64_t)ti * KT, j0 = (int64_t)tj * KT;
  // stage panels transposed: s[k][pt]
  for (int idx = tid; idx < KT * d; idx += 256) {
    int pt = idx / d, k = idx - pt * d;
    int64_t gi = i0 + pt, gj = j0 + pt;
    su[k * KT + pt] = (gi < nu) ? U[gi * d + k] : 0.0;
    sv[k * KT + pt] = (gj < nv) ? V[gj * d + k] : 0.0;
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
    double ur[4], vc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ur[r] = su[k * KT + ty * 4 + r];
#pragma unroll
    for (int c = 0; c < 4; ++c) vc[c] = sv[k * KT + tx * 4 + c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double dx = ur[r] - vc[c];
        s[r][c] = fma(0.5 * dx * dx, il2, s[r][c]);  // distances_k / l_k^2 accumulated (covariance.py:254)
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // two halves of eight elements: 16 at once cost occupancy, 8 already amortise the constants
    double sv8[8], cf8[8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sv8[4 * r + c] = s[2 * h + r][c];
    kfun<KERNEL>(p, sv8, cf8);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) s[2 * h + r][c] = cf8[4 * r + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t gi = i0 + ty * 4 + r;
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t gj = j0 + tx * 4 + c;
      double val;
      if (gi < nu && gj < nv) {
        double cfun = s[r][c];
        if (SQUARE && gi == gj) {
          // a^2 (C + 1e-12) + WhiteNoise + sig   (covariance.py:254-255, 163-169; regression.py:239)
          val = p.a2 * (cfun + 1e-12);
          val += p.extra_diag;
          val += noise[gi];
        } else {
          val = p.a2 * cfun;
        }
      } else {
        val = (SQUARE && gi == gj) ? 1.0 : 0.0;  // padding: identity keeps the factorisation trivial
      }
      v[c] = val;
    }
    double* dst = out + gi * ld + j0 + tx * 4;
    *reinterpret_cast<d2_t*>(dst) = d2_t{v[0], v[1]};
    *reinterpret_cast<d2_t*>(dst + 2) = d2_t{v[2], v[3]};
  }
}

__global__ void add_full_kernel(double* __restrict__ A, int64_t ld, const double* __restrict__ Y,
                                int64_t n) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = blockIdx.y;
  if (j < n) A[i * ld + j] += Y[i * n + j];
}

}  // namespace

static size_t kb_lds_bytes(int d) { return sizeof(double) * 2 * (size_t)d * KT; }

// dispatch on the covariance function (a template parameter of the kernels)
template <bool SQUARE>
static void launch_kb(dim3 grid, hipStream_t s, const KParams& p, const double* U, int64_t nu, const double* V,
                      int64_t nv, const double* noise, double* out, int64_t ld, int lower_only) {
  if (p.kernel == GPMI_KERNEL_SE)
    hipLaunchKernelGGL((kbuild_kernel<SQUARE, GPMI_KERNEL_SE>), grid, dim3(256), kb_lds_bytes(p.d), s, p, U, nu, V, nv,
                       noise, out, ld, lower_only);
  else
    hipLaunchKernelGGL((kbuild_kernel<SQUARE, GPMI_KERNEL_RQ>), grid, dim3(256), kb_lds_bytes(p.d), s, p, U, nu, V, nv,
                       noise, out, ld, lower_only);
}

void launch_kbuild_square(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                          const double* noise, double* A, int64_t ld, bool lower_only) {
  const unsigned nt = (unsigned)(np / KT);
  dim3 grid = lower_only ? dim3(nt * (nt + 1) / 2) : dim3(nt, nt);
  launch_kb<true>(grid, s, p, x, n, x, n, noise, A, ld, lower_only ? 2 : 0);
}

// The lower tiles in two launches: part 1 = the first `split_cols` columns (all rows), part 2 = everything to the right
// of them.  The fit starts factoring the first outer panel behind part 1 while part 2 is still being built on the
// update stream (api.hip: enqueue_factor_and_forward).
void launch_kbuild_square_part(hipStream_t s, const KParams& p, const double* x, int64_t n, int64_t np,
                               const double* noise, double* A, int64_t ld, int part, int split_cols) {
  const unsigned nt = (unsigned)(np / KT), sp = (unsigned)(split_cols / KT);
  if (part == 1) {
    launch_kb<true>(dim3(sp, nt), s, p, x, n, x, n, noise, A, ld, 1);
  } else if (nt > sp) {
    const unsigned m = nt - sp;
    launch_kb<true>(dim3(m * (m + 1) / 2), s, p, x, n, x, n, noise, A, ld, 2 | (int)(sp << 8));
  }
}

void launch_kbuild_square_batched(hipStream_t s, int kernel, const KParams* pdev, int batch, const double* x,
                                  int64_t n, int64_t np, const double* noise, double* A, int64_t ld,
                                  int64_t stride, int d, int64_t noise_stride) {
  const unsigned nt = (unsigned)(np / KT);
  dim3 grid(nt * (nt + 1) / 2, 1, (unsigned)batch);  // lower tiles only, one-dimensional (kbuild_body, mode 2)
  if (kernel == GPMI_KERNEL_SE)
    hipLaunchKernelGGL(kbuild_batched_kernel<GPMI_KERNEL_SE>, grid, dim3(256), kb_lds_bytes(d), s, pdev, x, n, noise, A,
                       ld, stride, noise_stride);
  else
    hipLaunchKernelGGL(kbuild_batched_kernel<GPMI_KERNEL_RQ>, grid, dim3(256), kb_lds_bytes(d), s, pdev, x, n, noise, A,
                       ld, stride, noise_stride);
}

void launch_kbuild_cross(hipStream_t s, const KParams& p, const double* U, int64_t mu, int64_t mp,
                         const double* V, int64_t n, int64_t np, double* out, int64_t ld) {
  dim3 grid((unsigned)(np / KT), (unsigned)(mp / KT));
  launch_kb<false>(grid, s, p, U, mu, V, n, nullptr, out, ld, 0);
}

void launch_add_full(hipStream_t s, double* A, int64_t ld, const double* Y, int64_t n) {
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
  hipLaunchKernelGGL(add_full_kernel, grid, dim3(256), 0, s, A, ld, Y, n);
}
