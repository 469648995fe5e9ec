// Launch-gap probe (tools only): chains of dependent dummy kernels on one stream, true end->start gaps from
// in-kernel s_memrealtime stamps (100 MHz).  Varies grid size, LDS size, bytes written and run time to find
// what makes the 10 us gaps seen between the panel kernels of the factorisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

struct K { int grid; int lds; int wbytes; int spin_us; int inplace; };

__global__ void dummy(unsigned long long* stamp, double* buf, int wbytes, int spin_us, int inplace) {
  extern __shared__ double sm[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) atomicMin(stamp, t0);
  const int n = wbytes / 8;
  double* p = buf + (size_t)blockIdx.x * n;
  double acc = 0.0;
  if (inplace) for (int i = threadIdx.x; i < n; i += blockDim.x) acc += p[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(4);
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = sm[(i + 1) & 255] + 1.0;
  __builtin_amdgcn_s_waitcnt(0);
  if (threadIdx.x == 0) atomicMax(stamp + 1, __builtin_amdgcn_s_memrealtime());
}

int main(int argc, char** argv) {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double* buf; hipMalloc(&buf, (size_t)1 << 30);
  const int reps = 30;
  struct Case { const char* name; std::vector<K> seq; };
  std::vector<Case> cases = {
    {"potrf-like: diag(1wg,77K,60us) trsm(110wg,55K,8us,64KB inplace) upd(600wg,36K,8us,32KB inplace)",
     {{1, 77000, 131072, 60, 1}, {110, 55296, 65536, 8, 1}, {600, 36864, 32768, 8, 1}}},
    {"same, no global writes", {{1, 77000, 0, 60, 0}, {110, 55296, 0, 8, 0}, {600, 36864, 0, 8, 0}}},
    {"same, all LDS 36K", {{1, 36864, 131072, 60, 1}, {110, 36864, 65536, 8, 1}, {600, 36864, 32768, 8, 1}}},
    {"predict-like: g1(128wg,36K,20us,32KB) upd(1000wg,72K,100us,128KB inplace)",
     {{128, 36864, 32768, 20, 0}, {1000, 73728, 131072, 100, 1}}},
    {"two short: a(110wg,55K,8us,64KB) b(600wg,36K,8us,32KB)", {{110, 55296, 65536, 8, 1}, {600, 36864, 32768, 8, 1}}},
    {"two short, no writes", {{110, 55296, 0, 8, 0}, {600, 36864, 0, 8, 0}}},
    {"two short, same kernel config (110wg,36K)", {{110, 36864, 65536, 8, 1}, {110, 36864, 65536, 8, 1}}},
    {"tiny: 1wg 0 LDS 5us x2", {{1, 1024, 0, 5, 0}, {1, 1024, 0, 5, 0}}},
    {"big grid short: 2000wg 36K 8us x2", {{2000, 36864, 32768, 8, 1}, {2000, 36864, 32768, 8, 1}}},
  };
  for (auto& c : cases) {
    const int n = (int)c.seq.size() * reps;
    unsigned long long* st; hipMalloc(&st, sizeof(unsigned long long) * 2 * n);
    std::vector<unsigned long long> init(2 * n);
    for (int i = 0; i < n; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    hipMemcpy(st, init.data(), sizeof(unsigned long long) * 2 * n, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)dummy, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int i = 0; i < n; ++i) {
      const K& k = c.seq[i % c.seq.size()];
      hipLaunchKernelGGL(dummy, dim3(k.grid), dim3(256), k.lds, s, st + 2 * i, buf, k.wbytes, k.spin_us, k.inplace);
    }
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(2 * n);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
    printf("%s\n", c.name);
    const int m = (int)c.seq.size();
    for (int j = 0; j < m; ++j) {
      double gsum = 0, dsum = 0; int cnt = 0;
      for (int i = m * 5 + j; i < n; i += m) {  // skip warm-up
        if (i == 0) continue;
        gsum += (double)(h[2 * i] - h[2 * (i - 1) + 1]) / 100.0;
        dsum += (double)(h[2 * i + 1] - h[2 * i]) / 100.0;
        ++cnt;
      }
      printf("   kernel %d: gap before %.2f us, in-kernel duration %.2f us\n", j, gsum / cnt, dsum / cnt);
    }
    hipFree(st);
  }
  return 0;
}
