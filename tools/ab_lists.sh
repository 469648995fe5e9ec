#!/bin/bash
# A/B of task lists for the flag-ordered tail on ONE box: the library's own against lists from tools/sim/flow_sched.py
# (GPMI_FLOW_LISTS=<prefix>): tools/ab_lists.sh <prefix> [<prefix> ...]
# (round 6: the files describe the one-list layout, which the library builds - and accepts from files - with GPMI_FLOW_SPLIT=0 only)
cd "$(dirname "$0")/.."
export GPMI_FLOW_SPLIT=0 GPMI_FLOW_QUARTER=99999
for p in "$@"; do
  echo "== parity with $p"; GPMI_FLOW_LISTS=$p python -m pytest tests/test_gpu_parity.py -x -q -k "flow_tail_is_bit_identical" 2>&1 | tail -1
done
for rep in 1 2; do
  for p in "" "$@"; do
    echo "== lists: ${p:-library} (rep $rep)"
    GPMI_FLOW_LISTS=$p python tools/probe_lml.py base 2>&1 | grep -v "^\[flow\]" | tail -1
    GPMI_FLOW_LISTS=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['ms_per_step'],2), 'ms/step; flow tail', round(d['roofline']['flow_tail']['ms_per_step'],2), 'ms')"
  done
done
