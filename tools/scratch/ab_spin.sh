#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms %.3f' % d['ms_per_step'])"; }
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 30 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
pt() { echo "== pt $*"; env "$@" timeout 300 python tools/config5_bench.py 20 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s' % d['lml_evals_per_s'])"; }
c2() { echo "== cfg2 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 cfg4 2>&1 | tail -2 | cut -c1-150; }
for rep in 1 2; do
  run X=1
  run GPMI_SYNC_SPIN=1
  run GPMI_SYNC_SPIN=2
  run GPMI_SYNC_SPIN=3
done
pt X=1
pt GPMI_SYNC_SPIN=1
c2 X=1
c2 GPMI_SYNC_SPIN=1
