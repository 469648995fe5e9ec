#!/bin/bash
# A/B of two lists per workgroup (short tasks / K = 512 tiles) against one list in virtual-time order
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms %.3f  all_trailing %.2f  flow_tail %.3f ms' % (d['ms_per_step'], r['all_trailing']['achieved'], r['flow_tail']['ms_per_step']))"; }
run() { echo "== headline $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
c2() { echo "== cfg2 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-150; }
for rep in 1 2; do
  c2 GPMI_FLOW_TWO_LISTS=0
  c2 GPMI_FLOW_TWO_LISTS=1
  run GPMI_FLOW_TWO_LISTS=0
  run GPMI_FLOW_TWO_LISTS=1
done
bash tools/scratch/flow_steps.sh gpurun_out/flows 2>&1 | grep -v "^step [3-5][0-9]"
