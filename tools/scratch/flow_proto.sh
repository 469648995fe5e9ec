#!/bin/bash
# protocol variants of the flag-ordered tail: correctness (bit-identity with the stream-ordered schedule) and time
out=${1:-gpurun_out/flowp}; mkdir -p $out
cd "$(dirname "$0")/.."
GPMI_FLOW=0 timeout 300 python tools/fit_digest.py $out/ref_8192.npz 8192 > /dev/null 2>&1
GPMI_FLOW=0 timeout 300 python tools/fit_digest.py $out/ref_16384.npz 16384 > /dev/null 2>&1
for p in ${PROTOS:-0 1 2 3}; do
  for n in 8192 16384; do
    GPMI_FLOW_PROTO=$p timeout 300 python tools/fit_digest.py $out/p${p}_$n.npz $n > $out/p${p}_$n.log 2>&1 || echo "proto $p n=$n: run FAILED ($(tail -1 $out/p${p}_$n.log))"
    python - <<PY
import numpy as np
try:
    a, b = np.load("$out/ref_$n.npz"), np.load("$out/p${p}_$n.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("proto $p n=$n:", "bit-identical" if not bad else f"DIFFERENT in {bad}")
except Exception as e:
    print("proto $p n=$n: compare failed:", e)
PY
  done
  GPMI_FLOW_PROTO=$p timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1
  GPMI_FLOW_PROTO=$p timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-160
done
