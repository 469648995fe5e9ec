#!/bin/bash
# A/B of the flag-ordered tail's schedule variants (round 6): second lists for the K = 512 chunks (GPMI_FLOW_SPLIT) and the
# chunks as 64 x 64 quarter tasks from outer panel q0 on (GPMI_FLOW_QUARTER=q0); N = 8192 fit in steady state, two rounds.
# usage (GPU box): tools/ab_flow_split.sh [N]
N=${1:-8192}
for round in 1 2; do
  for v in "A=0" "GPMI_FLOW_SPLIT=1" "GPMI_FLOW_QUARTER=0" "GPMI_FLOW_SPLIT=1 GPMI_FLOW_QUARTER=0" "GPMI_FLOW_SPLIT=1 GPMI_FLOW_QUARTER=2" "GPMI_FLOW_SPLIT=1 GPMI_FLOW_QUARTER=4" "GPMI_FLOW_SPLIT=1 GPMI_FLOW_QUARTER=6"; do
    echo -n "$v : "
    env $v GPMI_FLOW_STATS=1 python tools/fit_timeline.py $N 3 2>&1 | grep "^\[flow\]" | tail -1 | cut -c1-150
    echo -n "      "
    env $v python tools/fit_timeline.py $N 60 | tail -1
  done
done
