#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab_lib.sh <lib name under inference_amd/lib> ...
# (extra environment for every run: AB_ENV="GPMI_GEMM_DMA=1"); the current libgpmi.so is always included
run() { env "$@" $AB_ENV python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step; update', round(d['roofline']['achieved'],2), 'TFLOP/s')"; }
D=$PWD/inference-tools_amd/inference_amd/lib
for rep in 1 2; do
  for lib in libgpmi.so "$@"; do
    echo -n "$lib: "; run GPMI_LIB=$D/$lib
  done
done
