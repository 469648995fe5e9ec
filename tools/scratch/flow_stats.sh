#!/bin/bash
# diagnostics of the flag-ordered tail: in-kernel statistics + timings.  usage: tools/flow_stats.sh outdir
out=${1:-gpurun_out/flows}; mkdir -p $out
cd "$(dirname "$0")/.."
GPMI_FLOW=0 timeout 300 python tools/fit_digest.py $out/ref_8192.npz 8192 > /dev/null 2>&1
timeout 300 python tools/fit_digest.py $out/new_8192.npz 8192 > $out/new.log 2>&1 || tail -2 $out/new.log
python - <<PY
import numpy as np
a, b = np.load("$out/ref_8192.npz"), np.load("$out/new_8192.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
print("n=8192:", "bit-identical" if not bad else f"DIFFERENT in {bad}")
PY
GPMI_FLOW_STATS=1 timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -4
for v in "GPMI_FLOW_WGS=1" "GPMI_FLOW_WGS=2" "GPMI_FLOW=0"; do
  echo "== $v"; env $v timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1
done
timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-200
