#!/bin/bash
# the look-ahead / flow switch with the 33 us chain step
cd $GRAFT_REPO_ROOT
for la in 48 52 56 60; do
  echo "== GPMI_LOOKAHEAD_MIN=$la"
  for i in 1 2; do
  GPMI_LOOKAHEAD_MIN=$la python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   ', round(d['ms_per_step'],3), round(d['roofline']['flow_tail']['ms_per_step'],3))"
  done
  GPMI_LOOKAHEAD_MIN=$la python3 tools/config_bench.py cfg2 | tail -1
  GPMI_LOOKAHEAD_MIN=$la python3 tools/config_bench.py cfg3 | tail -1
done
