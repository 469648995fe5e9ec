// Probe: what takes the shader clock away from the trailing-update kernel?  tools/probes/mfma_sustained.hip shows the chip
// sustaining 77.8 TFLOP/s of fp64 MFMA at 2.39 GHz when the MFMAs run from registers; the update kernel gets 2.10 GHz on 256
// CUs.  Here the same MFMA loop is fed the way the kernel feeds it, one ingredient at a time:
//   mode 0  registers only
//   mode 1  + LDS fragment reads at the kernel's rate (8 x ds_read_b128 per 16 MFMAs and wave: a 64 x 64 wave tile, k = 4)
//   mode 2  + L2-resident global loads at the kernel's rate (2 x 16 B per lane and 16 MFMAs: the operand slabs of a 128 x 128 tile)
//   mode 3  both
//   mode 4  both + the global loads streaming from a buffer far larger than L2 + Infinity Cache (HBM)
//   mode 8  one global load in four from that buffer (about the kernel's 1 TB/s of fabric traffic), the others from L2
//   mode 16 a workgroup barrier every 64 MFMAs (the ring's stage barrier)
//   mode 32 twice the global loads (what wave-private operand slabs - a ring without barriers - would fetch)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power tools/probes/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void burn(double* out, const double* __restrict__ src, size_t src_mask, int iters,
                                               unsigned long long* clk) {
  __shared__ double lds[4096];  // 32 KiB
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1e-3 * ((i * 2654435761u) >> 20);
  __syncthreads();
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = 1e-3 * (threadIdx.x + 3 * i);
    b[i] = 1.0 - 1e-4 * (threadIdx.x + 5 * i);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  d2 g0 = {0.0, 0.0}, g1 = {0.0, 0.0};
  // every workgroup walks its own 64 KiB stripe sequence through the buffer (16 B per lane, 1 KiB per wave and load)
  size_t goff = ((size_t)blockIdx.x * 65536 + (size_t)wave * 16384 + (size_t)lane * 16) / 8;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
      // 8 x 16 B per lane from LDS (conflict-free: consecutive lanes, consecutive 16 B)
      const double* p = lds + ((it & 3) * 1024 + lane * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const d2 va = *reinterpret_cast<const d2*>(p + i * 128);
        const d2 vb = *reinterpret_cast<const d2*>(p + 512 + i * 128);
        a[2 * i] = va[0]; a[2 * i + 1] = va[1];
        b[2 * i] = vb[0]; b[2 * i + 1] = vb[1];
      }
    }
    if (MODE & 2) {
      if (MODE & 8) {
        // three loads in four from the first 2 MiB of the buffer (L2), the fourth from a fresh stripe (HBM)
        const bool far = (it & 1) == 0;
        g0 = *reinterpret_cast<const d2*>(src + (far ? (goff & src_mask) : ((goff * 7) & ((size_t)(2 << 20) / 8 - 1))));
        g1 = *reinterpret_cast<const d2*>(src + (((goff * 3 + 1024) & ((size_t)(2 << 20) / 8 - 1))));
        goff += (size_t)gridDim.x * 65536 / 8;
      } else {
        g0 = *reinterpret_cast<const d2*>(src + (goff & src_mask));
        g1 = *reinterpret_cast<const d2*>(src + ((goff + 1024) & src_mask));
        goff += (MODE & 4) ? (size_t)gridDim.x * 65536 / 8 : 2048;  // (HBM: a fresh stripe every time; L2: 16 KiB steps in a small window)
      }
    }
    if (MODE & 32) {  // (the second copy of the slabs: two more 16 B loads per lane from the L2 window)
      const d2 g2 = *reinterpret_cast<const d2*>(src + ((goff * 5 + 512) & ((size_t)(2 << 20) / 8 - 1)));
      const d2 g3 = *reinterpret_cast<const d2*>(src + ((goff * 11 + 1536) & ((size_t)(2 << 20) / 8 - 1)));
      g0[1] += g2[0];
      g1[0] += g3[1];
    }
    if ((MODE & 16) && (it & 3) == 3) __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
    if (MODE & 2) {  // consume the loads (keeps them in the loop) without a dependency in front of the MFMAs
      acc[0][0] += g0[0] * 1e-30;
      acc[1][0] += g1[1] * 1e-30;
    }
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

template <int MODE>
void run(const char* what, int grid, double* out, const double* src, size_t mask, int iters, unsigned long long* clk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(burn<MODE>, dim3(grid), dim3(256), 0, 0, out, src, mask, 2000, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(burn<MODE>, dim3(grid), dim3(256), 0, 0, out, src, mask, iters, clk);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flop = (double)grid * 4 * iters * 16.0 * 2048.0;
  const double gbytes = (MODE & 2) ? (double)grid * 256 * iters * ((MODE & 32) ? 64.0 : 32.0) : 0.0;
  std::printf("%-46s %8.2f ms  %6.2f TFLOP/s (%.3f of 78.6)  clock %.3f GHz  global loads %.2f TB/s\n", what, ms,
              flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 78.6e12, (double)h[0] / (double)h[1] * 0.1,
              gbytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = argc > 1 ? std::atoi(argv[1]) : 150000;
  double *out = nullptr, *small = nullptr, *big = nullptr;
  unsigned long long* clk = nullptr;
  const size_t small_bytes = (size_t)2 << 20, big_bytes = (size_t)8 << 30;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  hipMalloc(&small, small_bytes);
  hipMalloc(&big, big_bytes);
  hipMemset(small, 0, small_bytes);
  hipMemset(big, 0, big_bytes);
  hipMalloc(&clk, 16);
  hipDeviceSynchronize();
  const int grid = 2 * ncu;
  run<0>("registers only", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<1>("+ LDS fragment reads", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<2>("+ global loads (2 MiB window: L2)", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<3>("+ LDS reads + global loads (L2)", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<7>("+ LDS reads + global loads (8 GiB: HBM)", grid, out, big, big_bytes / 8 - 1, iters / 4, clk);
  run<11>("+ LDS reads + global loads (1 in 4 from HBM)", grid, out, big, big_bytes / 8 - 1, iters / 2, clk);
  run<19>("+ LDS reads + global loads (L2) + stage barriers", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<27>("+ LDS + loads (1 in 4 HBM) + stage barriers", grid, out, big, big_bytes / 8 - 1, iters / 2, clk);
  run<3 + 32>("+ LDS + 2 x global loads (L2), no barrier", grid, out, small, small_bytes / 8 - 1, iters, clk);
  run<3 + 32>("  the same, ONE workgroup per CU", ncu, out, small, small_bytes / 8 - 1, iters, clk);
  run<3>("+ LDS + global loads (L2), ONE workgroup per CU", ncu, out, small, small_bytes / 8 - 1, iters, clk);
  run<19>("+ LDS + loads (L2) + barriers, ONE workgroup/CU", ncu, out, small, small_bytes / 8 - 1, iters, clk);
  run<0>("registers only (again)", grid, out, small, small_bytes / 8 - 1, iters, clk);
  return 0;
}
