#!/bin/bash
# A/B helper: bench step time for several values of one environment variable.  usage: env_sweep.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  echo -n "$var=$v: "
  env $var=$v BENCH_NO_PROF=1 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms')"
done
