#!/bin/bash
# A/B helper: bench step time with and without the per-launch stamps, alternating
for i in 1 2 3; do
  for v in 1 ""; do
    echo -n "BENCH_NO_PROF='$v': "
    BENCH_NO_PROF=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms')"
  done
done
