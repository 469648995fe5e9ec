#!/bin/bash
# sweep of the flag-ordered tail's deal parameters and of the switch point.  usage: tools/flow_sweep.sh
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-60; }
run GPMI_FLOW=0
run GPMI_FLOW=1
run GPMI_LOOKAHEAD_MIN=64
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR_WGS=16
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR_WGS=64
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR_D=2
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR_D=5 GPMI_FLOW_NEAR_WGS=64
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR=0
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_NEAR=8
run GPMI_LOOKAHEAD_MIN=64 GPMI_FLOW_PROTO=2
for la in 60 68 76 84; do
  echo "== headline GPMI_LOOKAHEAD_MIN=$la"
  GPMI_LOOKAHEAD_MIN=$la timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
