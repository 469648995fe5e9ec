#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
t=[k for k in r['kernels'] if 'trsm_rows' in k['kernel']]
print('ms %.3f  all_trailing %.2f  dominant %.2f  trsm %.2f TF/s %.3f ms' % (d['ms_per_step'], r['all_trailing']['achieved'], r['achieved'], t[0]['achieved'] if t else 0, t[0]['avg_ms'] if t else 0))"; }
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
export GPMI_LA_STREAM=0
run GPMI_TRSM_SPLIT=0
run GPMI_TRSM_SPLIT=1 GPMI_TRSM_AUX=1
run GPMI_TRSM_SPLIT=1 GPMI_TRSM_AUX=2
run GPMI_TRSM_SPLIT=1 GPMI_TRSM_AUX=3
run GPMI_TRSM_SPLIT=1 GPMI_TRSM_AUX=0 GPU_MAX_HW_QUEUES=8
run GPMI_TRSM_SPLIT=0 GPU_MAX_HW_QUEUES=8
