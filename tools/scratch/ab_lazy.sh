#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms %.3f  dominant %.2f TF (%d launches, avg %.3f ms, clock %.3f)  all_trailing %.2f  flow_tail %.3f ms' % (d['ms_per_step'], r['achieved'], r['launches'], r['avg_launch_ms'], r['clock_ghz'], r['all_trailing']['achieved'], r['flow_tail']['ms_per_step']))"; }
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
run X=1
run GPMI_LAZY2=1
run X=1
run GPMI_LAZY2=1
