// Probe: what does the chip SUSTAIN in fp64 MFMA with nothing else going on?  Every wave keeps 16 independent
// v_mfma_f64_16x16x4_f64 accumulator chains in registers (no LDS, no memory in the loop) - the issue-bound ideal of the
// trailing-update kernel - on all CUs (or the first `cus` of the mask) for 2 ms .. 200 ms.  Prints wall-clock TFLOP/s
// against the 78.6 TFLOP/s of 256 CUs x 2.4 GHz and the shader clock the run got (s_memtime cycles per s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_sustained tools/probes/mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// RANDOM = false: one constant operand pair per lane (the multipliers see the same bits every time: a lower bound on the
// power an MFMA draws); true: eight pseudo-random operand values per lane in rotation, mantissas fully populated - what
// real matrix data looks like to the multiplier array
template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void burn(double* out, int iters, unsigned long long* clk) {
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a[8], b[8];
  unsigned long long h = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1) + 0xD1B54A32D192ED03ull * (blockIdx.x + 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    a[i] = RANDOM ? (double)(long long)h * (1.0 / 9.3e18) : 1e-3 * threadIdx.x;
    h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
    b[i] = RANDOM ? (double)(long long)h * (1.0 / 9.3e18) : 1.0 + 1e-4 * threadIdx.x;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  double* out = nullptr;
  unsigned long long* clk = nullptr;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int random = 0; random < 2; ++random)
  for (int wgs_per_cu : {1, 2}) {
    for (int iters : {20000, 200000, 1000000}) {
      const int grid = ncu * wgs_per_cu;
      hipLaunchKernelGGL(burn<false>, dim3(grid), dim3(256), 0, 0, out, 1000, clk);  // warm
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      if (random) hipLaunchKernelGGL(burn<true>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
      else hipLaunchKernelGGL(burn<false>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2];
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)grid * 4 /*waves*/ * iters * 16.0 * 2048.0;
      std::printf("%s operands, %d CUs x %d workgroup(s) of 4 waves, %8d iterations: %8.2f ms, %6.2f TFLOP/s (%.3f of 78.6), shader clock %.3f GHz\n",
                  random ? "random  " : "constant", ncu, wgs_per_cu, iters, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 78.6e12,
                  (double)h[0] / (double)h[1] * 0.1);
    }
  }
  return 0;
}
