#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; s=d.get('sharded') or {}
print('ms %.3f  all_trailing %.2f flow %.2f | cfg3 %.3f s' % (d['ms_per_step'], r['all_trailing']['achieved'], r['flow_tail']['ms_per_step'], s.get('config3',{}).get('seconds',0)))"; }
run() { echo "== $*"; env "$@" BENCH_CFG3_POINTS=8 BENCH_CFG5_LADDERS=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
run GPMI_STREAM_POOL=0
run GPMI_STREAM_POOL=4
run GPMI_STREAM_POOL=3
run GPMI_STREAM_POOL=2
run GPMI_STREAM_POOL=0
run GPMI_STREAM_POOL=4
