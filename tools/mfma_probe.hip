// Probe: layout and issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_probe(const double* A, const double* B, double* D) {
  // A: 16x4 row-major [i][k], B: 4x16 row-major [k][j]
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void mfma_rate(double* out, int iters, long long* cyc) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void fma_rate(double* out, int iters, long long* cyc) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 0.5 + k * 7;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + j * j * 0.25 - k;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_probe<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) + 4 * r, col = l & 15;
    if (std::fabs(D[l * 4 + r] - ref[row * 16 + col]) > 1e-9) ++bad;
  }
  printf("layout check row=(l>>4)+4r col=l&15 : %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
  // rate
  double* out; long long* cyc; CK(hipMalloc(&out, 8 * 1024 * 1024)); CK(hipMalloc(&cyc, 8));
  long long hc; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  for (int waves = 1; waves <= 2; ++waves) {
    // one block per CU, 4*waves waves per block
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      mfma_rate<8><<<p.multiProcessorCount, 256 * waves>>>(out, iters, cyc);
      hipEventRecord(e1); CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, e0, e1); CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      double flops = (double)p.multiProcessorCount * 4 * waves * iters * 8 * 2048.0;
      printf("mfma f64 16x16x4: %d waves/SIMD, %.3f ms, %.2f TFLOP/s, %.1f memtime-ticks per MFMA per wave\n", waves, ms, flops / ms / 1e9, (double)hc / (iters * 8.0));
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    mfma_rate<1><<<p.multiProcessorCount, 256>>>(out, iters, cyc);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1); CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    printf("mfma f64 dependent chain (1 acc): %.3f ms, %.1f ticks per MFMA\n", ms, (double)hc / iters);
  }
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipEventRecord(e0);
    fma_rate<<<p.multiProcessorCount, 256 * waves>>>(out, iters, cyc);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)p.multiProcessorCount * 256 * waves * iters * 16 * 2.0;
    printf("v_fma_f64: %d waves/SIMD %.3f ms, %.2f TFLOP/s\n", waves, ms, flops / ms / 1e9);
  }
  return 0;
}
