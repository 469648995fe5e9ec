#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
t=[k for k in r['kernels'] if 'trsv' in k['kernel']]
print('ms %.3f  sweeps avg %.3f ms %.0f GB/s' % (d['ms_per_step'], t[0]['avg_ms'], t[0]['achieved']))"; }
run() { echo "== headline $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
c2() { echo "== cfg2 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-150; }
for rep in 1 2; do
  run GPMI_SWEEP_FOLD=0
  run GPMI_SWEEP_FOLD=1
  c2 GPMI_SWEEP_FOLD=0
  c2 GPMI_SWEEP_FOLD=1
done
