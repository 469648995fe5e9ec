#!/bin/bash
# A/B helper: bench step time for several look-ahead thresholds (tile rows)
for v in "$@"; do
  echo -n "GPMI_LOOKAHEAD_MIN=$v: "
  GPMI_LOOKAHEAD_MIN=$v BENCH_NO_PROF=1 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms')"
done
