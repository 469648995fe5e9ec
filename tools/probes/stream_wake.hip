// Does the time from launch to completion of a tiny kernel depend on WHICH stream of a process it is launched on?
// (round 6: whole handles whose lockstep batch calls took 28 ms instead of 4 - the first kernels of a call completed
// 15 - 25 ms after their launch; the handle's streams were the 11th / 12th the process had created.)
// Creates `nstreams` non-blocking streams one after another; on each: `reps` times {pause `pause_us` on the host, launch
// `burst` tiny kernels, synchronise}, prints the median and the maximum launch -> completion time per stream.
// build: hipcc --offload-arch=gfx950 -O2 -o stream_wake stream_wake.hip     usage: ./stream_wake [nstreams] [reps] [pause_us] [destroy]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void tiny(double* p) { p[threadIdx.x] += 1.0; }

int main(int argc, char** argv) {
  const int nstreams = argc > 1 ? std::atoi(argv[1]) : 16;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 40;
  const int pause_us = argc > 3 ? std::atoi(argv[3]) : 300;
  const int destroy = argc > 4 ? std::atoi(argv[4]) : 0;  // 1: destroy each stream before the next is created
  double* buf = nullptr;
  if (hipMalloc(&buf, 4096) != hipSuccess) return 1;
  std::vector<hipStream_t> streams;
  for (int k = 0; k < nstreams; ++k) {
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
    streams.push_back(s);
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
      std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
      const auto t0 = std::chrono::steady_clock::now();
      for (int b = 0; b < 6; ++b) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, buf);
      (void)hipStreamSynchronize(s);
      t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    std::printf("stream %2d: median %9.1f us   max %9.1f us\n", k, t[t.size() / 2], t.back());
    if (destroy) {
      (void)hipStreamDestroy(s);
      streams.pop_back();
    }
  }
  for (auto s : streams) (void)hipStreamDestroy(s);
  return 0;
}
