// Probe: does hipExtAnyOrderLaunch let the second kernel of a stream start while the first still runs (gfx950)?
// build: hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void producer(unsigned long long* t, int* flag, int spin_us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    t[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
    t[1] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void consumer(unsigned long long* t, int* flag, int wait) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    t[2] = __builtin_amdgcn_s_memrealtime();
    if (wait) {
      long n = 0;
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 && n < (1 << 24)) ++n;
    }
    t[3] = __builtin_amdgcn_s_memrealtime();
  }
}

int main() {
  unsigned long long* t;
  int* flag;
  hipMalloc(&t, 64);
  hipMalloc(&flag, 4);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(t, 0, 64);
      hipMemset(flag, 0, 4);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, s, t, flag, 100);
      if (mode == 0)
        hipLaunchKernelGGL(consumer, dim3(64), dim3(256), 0, s, t, flag, 0);
      else
        hipExtLaunchKernelGGL(consumer, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t, flag, 1);
      hipStreamSynchronize(s);
      unsigned long long h[4];
      hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
      printf("%s: producer %.1f us; consumer start %+.1f us after producer end, consumer done %+.1f us after producer end\n",
             mode ? "anyorder" : "in-order", (h[1] - h[0]) * 0.01, ((double)h[2] - (double)h[1]) * 0.01,
             ((double)h[3] - (double)h[1]) * 0.01);
    }
  }
  return 0;
}
