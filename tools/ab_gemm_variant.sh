#!/bin/bash
# A/B on ONE box of a library variant (tools/build_gemm_variant.sh <name> ...): the trailing-update kernel alone, sustained
# (40 launches), the fit digest (bits) and the headline.  usage: tools/ab_gemm_variant.sh <name>
cd "$(dirname "$0")/.."
D=$PWD/inference-tools_amd/inference_amd/lib
v=libgpmi_$1.so
for lib in libgpmi.so $v; do GPMI_LIB=$D/$lib python tools/fit_digest.py gpurun_out/dig_$lib.npz 8192 > /dev/null 2>&1; done
python - <<PY
import numpy as np
a=dict(np.load("gpurun_out/dig_libgpmi.so.npz")); b=dict(np.load("gpurun_out/dig_$v.npz"))
print("fit digest N=8192: bit-identical" if all(np.array_equal(a[k],b[k]) for k in a) else "fit digest DIFFERS: max rel " + str(max(float(np.abs(a[k]-b[k]).max()/np.abs(a[k]).max()) for k in a)))
PY
rm -f gpurun_out/dig_*.npz
for rep in 1 2 3; do
  for lib in libgpmi.so $v; do
    echo -n "$lib: "; GPMI_LIB=$D/$lib python tools/bench_gemm.py 15872 512 1 40 | tr '\n' ' '
    GPMI_LIB=$D/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('| headline', round(d['ms_per_step'],2), 'ms/step; update', round(d['roofline']['achieved'],2), 'TFLOP/s')"
  done
done
