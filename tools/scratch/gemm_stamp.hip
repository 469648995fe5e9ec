// Diagnostic build of the fp64 MFMA GEMM main loop with s_memtime stamps (never shipped):
// where does a k-step spend its cycles?  Prints per-wave averages of
//   t_read  : barrier release -> first MFMA can issue (LDS fragment latency)
//   t_mfma  : first MFMA issue -> last MFMA issued
//   t_stage : last MFMA issued -> staging stores done
//   t_bar   : waiting at the barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BK = 16, LDS_STRIDE = 18, TILE_DOUBLES = BM * LDS_STRIDE;

__device__ inline unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int STAMP>
__global__ __launch_bounds__(256, 2) void gemm_diag(const double* A, const double* B, double* C, long lda, long ldc,
                                                    int k, int ntc, unsigned long long* dbg, unsigned long long* wgt) {
  unsigned long long wg_t0 = 0;
  if (STAMP && threadIdx.x == 0) wg_t0 = __builtin_amdgcn_s_memrealtime();
  __shared__ double smem[2 * 2 * TILE_DOUBLES];
  const int ti = blockIdx.x / ntc, tj = blockIdx.x % ntc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const double* Ag = A + (long)ti * BM * lda;
  const double* Bg = B + (long)tj * BM * lda;
  const int lrow = tid >> 3, lkc = (tid & 7) * 2;
  const int lkc_sw = lkc ^ ((((lrow & 15) >= 4) && ((lrow & 15) < 12)) ? 4 : 0);
  d2_t ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const d2_t*>(Ag + (long)(lrow + 32 * i) * lda + k0 + lkc);
      rb[i] = *reinterpret_cast<const d2_t*>(Bg + (long)(lrow + 32 * i) * lda + k0 + lkc);
    }
  };
  auto sstore = [&](int buf) {
    double* sa = smem + buf * 2 * TILE_DOUBLES;
    double* sb = sa + TILE_DOUBLES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<d2_t*>(sa + (lrow + 32 * i) * LDS_STRIDE + lkc_sw) = ra[i];
      *reinterpret_cast<d2_t*>(sb + (lrow + 32 * i) * LDS_STRIDE + lkc_sw) = rb[i];
    }
  };
  d4_t acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0, 0, 0, 0};
  const int fr = lane & 15, fk = lane >> 4, sw = (fr >= 4 && fr < 12) ? 1 : 0;
  const int a_off = (wr * 64 + fr) * LDS_STRIDE + ((fk ^ sw) << 2);
  const int b_off = (wc * 64 + fr) * LDS_STRIDE + ((fk ^ sw) << 2);
  gload(0); sstore(0); __syncthreads();
  const int nk = k / BK;
  unsigned long long t0a = 0, s_gl = 0, s_read = 0, s_mfma = 0, s_stage = 0, s_bar = 0, t0 = 0, t1, t2, t3, t4, tb = STAMP ? stamp() : 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (STAMP) t0 = stamp();
    if (kt + 1 < nk) gload((kt + 1) * BK);
    if (STAMP) { __builtin_amdgcn_sched_barrier(0); t0a = stamp(); __builtin_amdgcn_sched_barrier(0); }
    const double* sa = smem + cur * 2 * TILE_DOUBLES;
    const double* sb = sa + TILE_DOUBLES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      d2_t a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const d2_t*>(sa + a_off + t * 16 * LDS_STRIDE + 2 * h);
        b[t] = *reinterpret_cast<const d2_t*>(sb + b_off + t * 16 * LDS_STRIDE + 2 * h);
      }
      if (STAMP && h == 0) { __builtin_amdgcn_sched_barrier(0); t1 = stamp(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    if (STAMP) { __builtin_amdgcn_sched_barrier(0); t2 = stamp(); __builtin_amdgcn_sched_barrier(0); }
    if (kt + 1 < nk) sstore(cur ^ 1);
    if (STAMP) { __builtin_amdgcn_sched_barrier(0); t3 = stamp(); __builtin_amdgcn_sched_barrier(0); }
    __syncthreads();
    if (STAMP) { t4 = stamp(); s_gl += t0a - t0; s_read += t1 - t0a; s_mfma += t2 - t1; s_stage += t3 - t2; s_bar += t4 - t3; }
  }
  unsigned long long tep = STAMP ? stamp() : 0;
  double* Cg = C + ((long)ti * BM + wr * 64) * ldc + (long)tj * BM + wc * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double* rowp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rowp[r] = Cg + (long)(i * 16 + fk + 4 * r) * ldc + fr;
    double cv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[j][r] = rowp[r][j * 16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) rowp[r][j * 16] = cv[j][r] - acc[i][j][r];
  }
  if (STAMP && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    wgt[blockIdx.x * 3] = wg_t0; wgt[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
    wgt[blockIdx.x * 3 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  if (STAMP && lane == 0) {
    unsigned long long te = stamp();
    unsigned long long* d = dbg + ((long)blockIdx.x * 4 + wave) * 6;
    d[0] = s_read; d[5] = s_gl; d[1] = s_mfma; d[2] = s_stage; d[3] = s_bar; d[4] = te - tb;
  }
}

int main() {
  const int n = 8192, k = 512, nt = n / 128; const long ld = n + 32, lda = k + 32;
  double *A, *C; unsigned long long* dbg;
  hipMalloc(&A, sizeof(double) * n * lda); hipMalloc(&C, sizeof(double) * n * ld);
  hipMalloc(&dbg, 8 * 6 * 4 * nt * nt);
  unsigned long long* wgt; hipMalloc(&wgt, 8 * 3 * nt * nt);
  hipMemset(A, 0, sizeof(double) * n * lda); hipMemset(C, 0, sizeof(double) * n * ld);
  // non-trivial data
  std::vector<double> h((size_t)n * lda); for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(gemm_diag<0>, dim3(nt * nt), dim3(256), 0, 0, A, A, C, lda, ld, k, nt, dbg, wgt);
      else hipLaunchKernelGGL(gemm_diag<1>, dim3(nt * nt), dim3(256), 0, 0, A, A, C, lda, ld, k, nt, dbg, wgt);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("variant %d (stamps %s): %.3f ms  %.2f TFLOP/s\n", variant, variant ? "on" : "off", ms, 2.0 * n * n * k / ms / 1e9);
    }
  }
  std::vector<unsigned long long> d((size_t)6 * 4 * nt * nt);
  hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost);
  double s[5] = {0, 0, 0, 0, 0}; size_t nw = (size_t)4 * nt * nt;
  for (size_t w = 0; w < nw; ++w) for (int j = 0; j < 5; ++j) s[j] += (double)d[w * 6 + j];
  double nk = k / 16;
  double ep = 0; for (size_t w = 0; w < nw; ++w) ep += (double)d[w * 6 + 5];
  printf("per k-step per wave (cycles): read %.0f  mfma %.0f  stage %.0f  barrier %.0f ; wave lifetime (loop+epilogue) %.0f, gload-issue per k-step %.0f\n",
         s[0] / nw / nk, s[1] / nw / nk, s[2] / nw / nk, s[3] / nw / nk, s[4] / nw, ep / nw / nk);
  {
    std::vector<unsigned long long> w((size_t)3 * nt * nt);
    hipMemcpy(w.data(), wgt, w.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0; double dsum = 0, dmin = 1e30, dmax = 0;
    size_t nwg = (size_t)nt * nt;
    for (size_t b = 0; b < nwg; ++b) { tmin = std::min(tmin, w[b*3]); tmax = std::max(tmax, w[b*3+1]); double d = (double)(w[b*3+1]-w[b*3]); dsum += d; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
    printf("kernel span %.1f us; WG duration avg %.1f us min %.1f max %.1f (10 ns ticks); sum of WG durations / (512 slots * span) = %.3f\n",
           (tmax - tmin) * 0.01, dsum / nwg * 0.01, dmin * 0.01, dmax * 0.01, dsum / (512.0 * (tmax - tmin)));
    // finish-time profile: how many WGs end in each 10% bucket of the kernel span; start-time of the first 512
    int hist[10] = {0}; for (size_t b = 0; b < nwg; ++b) { int q = (int)(10.0 * (w[b*3+1] - tmin) / (double)(tmax - tmin + 1)); hist[q]++; }
    printf("WG end-time histogram (10 buckets):"); for (int q = 0; q < 10; ++q) printf(" %d", hist[q]); printf("\n");
    double late = 0; int nl = 0; for (size_t b = 0; b < 512; ++b) { late += (double)(w[b*3] - tmin); ++nl; }
    printf("mean start delay of the first 512 WGs: %.1f us\n", late / nl * 0.01);
    // per XCD: last end time
    unsigned long long xend[8] = {0}; int xcnt[8] = {0};
    for (size_t b = 0; b < nwg; ++b) { int x = (int)((w[b*3+2] >> 32) & 7); xend[x] = std::max(xend[x], w[b*3+1]); xcnt[x]++; }
    printf("per-XCD tiles / last end (us):"); for (int x = 0; x < 8; ++x) printf(" %d/%.0f", xcnt[x], (xend[x] - tmin) * 0.01); printf("\n");
  }
  return 0;
}
