#!/bin/bash
# A/B of the flag-ordered tail (GPMI_FLOW=1, default) against the stream-ordered schedule (GPMI_FLOW=0):
# bit-identity of the fit at N = 8192 / 16384 / ragged sizes, then timings.  usage: tools/flow_probe.sh [outdir]
out=${1:-gpurun_out/flow}; mkdir -p $out
cd "$(dirname "$0")/.."
for n in 6500 8192 16384; do
  for f in 0 1; do
    GPMI_FLOW=$f timeout 300 python tools/fit_digest.py $out/d_${n}_$f.npz $n > $out/d_${n}_$f.log 2>&1 || echo "digest n=$n flow=$f FAILED rc=$?" 
  done
  python - <<PY
import numpy as np
try:
    a, b = np.load("$out/d_${n}_0.npz"), np.load("$out/d_${n}_1.npz")
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        err = float(np.abs(a[k]-b[k]).max()/max(np.abs(a[k]).max(),1e-300))
        print("n=$n", k, "bit-identical" if same else f"DIFFERENT rel {err:.3e}")
except Exception as e:
    print("n=$n compare failed:", e)
PY
done
for f in 0 1; do
  echo "== GPMI_FLOW=$f"
  GPMI_FLOW=$f timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -2
  GPMI_FLOW=$f timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'))"
done
