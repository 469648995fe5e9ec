#!/bin/bash
# A/B timing on ONE box: alternates the round-1 library (inference_amd/lib/libgpmi_r1.so, built from git history) and
# the current one, optionally with environment settings:  tools/ab_bench.sh "GPMI_PANEL16_MIN=92" "GPMI_PANEL16_MIN=0"
run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['achieved'],2))"; }
R1=$PWD/inference-tools_amd/inference_amd/lib/libgpmi_r1.so
for rep in 1 2; do
  [ -f $R1 ] && { echo -n "r1 lib: "; run GPMI_LIB=$R1; }
  echo -n "current: "; run X=1
  for cfg in "$@"; do echo -n "current $cfg: "; run $cfg; done
done
