#!/bin/bash
cd "$(dirname "$0")/../.."
for k in 256 512 1024 2048; do
  for hot in 0 1; do
    timeout 120 python tools/bench_gemm.py 15872 $k 1 5 $hot 2>&1 | tail -1
  done
done
timeout 120 python tools/bench_gemm.py 7936 512 1 10 0 2>&1 | tail -1
timeout 120 python tools/bench_gemm.py 7936 512 0 5 0 2>&1 | tail -1
