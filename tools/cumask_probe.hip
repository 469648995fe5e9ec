// Probe: does hipExtStreamCreateWithCUMask confine workgroups, and how do mask bits map to XCDs / CUs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID bits [3:0]
    unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hwid;
  }
  // burn a little time so that blocks spread
  long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}
int main() {
  const int nb = 2048;
  unsigned* d; hipMalloc(&d, nb * 8);
  std::vector<unsigned> h(nb * 2);
  for (int test = 0; test < 4; ++test) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (test == 0) for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
    if (test == 1) mask[0] = 0xffffffffu;                 // first 32 bits
    if (test == 2) mask[0] = 0x000000ffu;                 // first 8 bits
    if (test == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[0] = 0; }  // all but first 32
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("test %d: hipExtStreamCreateWithCUMask failed: %s\n", test, hipGetErrorString(e)); continue; }
    hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> xcc; std::map<unsigned long long, int> cu;
    for (int b = 0; b < nb; ++b) {
      xcc[h[b * 2] & 0xf]++;
      unsigned hw = h[b * 2 + 1];
      unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cu[((unsigned long long)(h[b * 2] & 0xf) << 16) | (se << 8) | (sh << 4) | cu_id]++;
    }
    printf("test %d: distinct XCC %zu, distinct (xcc,se,sh,cu) %zu ; per-XCC counts:", test, xcc.size(), cu.size());
    for (auto& kv : xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
