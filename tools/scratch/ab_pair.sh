#!/bin/bash
# A/B of the flag-ordered tail's own CU-masked pair (8 | 248) against the look-ahead pair (32 | 224)
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
  c2 GPMI_FLOW_PAIR=0
  c2 GPMI_FLOW_PAIR=1
  c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_CHAIN_CUS=16
  run GPMI_FLOW_PAIR=0
  run GPMI_FLOW_PAIR=1
done
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_NEAR_WGS=64
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_STATS=1
python - <<'PY'
import time, sys, os
sys.path[:0] = ["inference-tools_amd", "."]
import numpy as np
from inference_amd.gp import GpRegressor
rng = np.random.default_rng(1)
x = rng.uniform(0, 1, (8192, 4)); y = np.sin(x.sum(1))
t0 = time.perf_counter(); gp = GpRegressor(x, y, y_err=np.full(8192, 0.1), hyperpars=np.array([0.0, 0.0, -0.5, -0.5, -0.5, -0.5])); t1 = time.perf_counter()
gp.engine.close() if hasattr(gp.engine, "close") else None
del gp
import gc; gc.collect()
t2 = time.perf_counter()
print("construct %.2f s, destroy %.2f s" % (t1 - t0, t2 - t1))
PY
