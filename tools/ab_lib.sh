#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab_lib.sh <lib name under inference_amd/lib> ...  (each also run
# as the stand-alone trailing-update GEMM); the current libgpmi.so is always included
run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step; update', round(d['roofline']['achieved'],2), 'TFLOP/s')"; }
D=$PWD/inference-tools_amd/inference_amd/lib
for rep in 1 2; do
  for lib in libgpmi.so "$@"; do
    echo -n "$lib: "; run GPMI_LIB=$D/$lib GPMI_PANEL16_MIN=0
    echo -n "   gemm K=512: "; GPMI_LIB=$D/$lib python tools/bench_gemm.py 12288 512 1 4
  done
done
