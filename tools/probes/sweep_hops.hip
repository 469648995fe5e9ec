// Where a step of the single-launch triangular sweeps spends its time: 10 ns wall-clock stamps of wave 0 at the phases of
// every step (compiled into the kernels by -DGPMI_SWEEP_STAMPS; the product build has none of it).
//   slot 0: the LAST poll (the producer's 128 values) has returned      slot 1: partial sums written / behind the barrier
//   slot 2: behind the first fold barrier   slot 3: u complete (barrier)   slot 4: own 128 values published
// hop k = publish_k - publish_{k-1} = (slot0_k - slot4_{k-1}: the values' way through memory to the consumer) + the phases.
// build + run:  bash tools/probes/sweep_hops.sh   (links the library's other objects; never part of libgpmi.so)
#define GPMI_SWEEP_STAMPS 1
#include "../../inference-tools_amd/csrc/solve.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void report(const char* name, int nt) {
  static unsigned long long h[8][1024];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sweep_stamp), sizeof(h));
  double sum[6] = {0, 0, 0, 0, 0, 0};
  int cnt = 0;
  for (int k = 2; k < nt; ++k) {  // steps with a producer on the critical path
    const double hop = 0.01 * (double)(h[4][k] - h[4][k - 1]);
    const double way = 0.01 * (double)((long long)h[0][k] - (long long)h[4][k - 1]);
    sum[0] += hop;
    sum[1] += way;
    for (int p = 1; p <= 4; ++p) sum[1 + p] += 0.01 * (double)(h[p][k] - h[p - 1][k]);
    ++cnt;
  }
  for (int lo = 2; lo < nt; lo += (nt >= 64 ? nt / 8 : nt)) {  // the hop along the sweep (early steps share HBM with every later workgroup's stream)
    const int hi = std::min(nt, lo + (nt >= 64 ? nt / 8 : nt));
    printf("   steps %3d..%3d: %.2f us/step\n", lo, hi - 1, 0.01 * (double)(h[4][hi - 1] - h[4][lo - 1]) / (hi - lo));
  }
  printf("%s nt=%d: hop %.2f us = way-to-consumer %.2f + phase1 %.2f + phase2 %.2f + phase3 %.2f + phase4 %.2f   (whole sweep %.1f us)\n",
         name, nt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt,
         0.01 * (double)(h[4][nt - 1] - h[0][1]));
}

int main(int argc, char** argv) {
  for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
    const int64_t n = argc > 1 ? atoll(argv[a]) : 16384;
    const int nt = (int)(n / NB);
    const int64_t ld = n;
    double *L, *invD, *r, *v;
    int* err;
    (void)hipMalloc(&L, sizeof(double) * n * ld);
    (void)hipMalloc(&invD, sizeof(double) * nt * NB * NB);
    (void)hipMalloc(&r, sizeof(double) * n);
    (void)hipMalloc(&v, sizeof(double) * n);
    (void)hipMalloc(&err, sizeof(int));
    (void)hipMemset(err, 0, sizeof(int));
    {  // small finite numbers: timing does not depend on the values
      std::vector<double> hl((size_t)n * 64);
      for (size_t i = 0; i < hl.size(); ++i) hl[i] = 1e-6 * (double)((i * 2654435761u) % 1000);
      for (int64_t off = 0; off < n * ld; off += (int64_t)hl.size())
        (void)hipMemcpy(L + off, hl.data(), sizeof(double) * hl.size(), hipMemcpyHostToDevice);
      for (int64_t off = 0; off < (int64_t)nt * NB * NB; off += (int64_t)hl.size())
        (void)hipMemcpy(invD + off, hl.data(), sizeof(double) * std::min<int64_t>((int64_t)hl.size(), (int64_t)nt * NB * NB - off),
                        hipMemcpyHostToDevice);
      (void)hipMemcpy(r, hl.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    }
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((n + 255) / 256), 1, 1), dim3(256), 0, s, v, n, (int64_t)0);
      hipLaunchKernelGGL(trsv_fwd_flow_kernel, dim3(nt), dim3(FLOW_THREADS), 0, s, L, ld, invD, r, v, err, (int64_t)0, (int64_t)0,
                         (int64_t)0, nt, 1);
      (void)hipStreamSynchronize(s);
    }
    report("forward ", nt);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((n + 255) / 256), 1, 1), dim3(256), 0, s, v, n, (int64_t)0);
      hipLaunchKernelGGL(trsv_bwd_flow_kernel, dim3(nt), dim3(FLOW_THREADS), 0, s, L, ld, invD, r, v, err, nt, (int64_t)0, (int64_t)0,
                         (int64_t)0, 1);
      (void)hipStreamSynchronize(s);
    }
    report("backward", nt);
    int herr = 0;
    (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
    if (herr) printf("err flag %d\n", herr);
    (void)hipFree(L); (void)hipFree(invD); (void)hipFree(r); (void)hipFree(v); (void)hipFree(err);
  }
  return 0;
}
