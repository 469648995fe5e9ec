#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; s=d.get('sharded') or {}
c3=s.get('config3',{}); c5=s.get('config5',{})
print('ms %.3f  all_trailing %.2f | cfg3 %.3f s  cfg5 %.0f evals/s' % (d['ms_per_step'], r['all_trailing']['achieved'], c3.get('seconds',0), c5.get('lml_evals_per_s',0)))"; }
run() { echo "== headline $*"; env "$@" BENCH_CFG3_POINTS=16 BENCH_CFG5_LADDERS=16 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
c2() { echo "== cfg2 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-150; }
pt() { echo "== pt $*"; env "$@" timeout 300 python tools/config5_bench.py 20 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s' % d['lml_evals_per_s'])"; }
for rep in 1 2; do
  run X=1
  run GPU_MAX_HW_QUEUES=8
done
run GPU_MAX_HW_QUEUES=6
run GPU_MAX_HW_QUEUES=16
c2 X=1
c2 GPU_MAX_HW_QUEUES=8
pt X=1
pt GPU_MAX_HW_QUEUES=8
