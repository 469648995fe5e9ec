#!/bin/bash
# A/B of the urgent-chunk lists of the flag-ordered tail (round 6); N = 8192 fit in steady state + the tail's own statistics.
N=${1:-8192}
export GPMI_FLOW_SPLIT=1
for round in 1 2; do
  for v in "GPMI_FLOW_SPLIT=0" "GPMI_FLOW_QUARTER=0" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=1" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=2" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=3" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=4" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=6" "GPMI_FLOW_QUARTER=0 GPMI_FLOW_URGENT=99"; do
    echo -n "$v : "
    env $v GPMI_FLOW_STATS=1 python tools/fit_timeline.py $N 3 2>&1 | grep "^\[flow\]" | tail -1 | cut -c30-140
    echo -n "      "
    env $v python tools/fit_timeline.py $N 60 | tail -1
  done
done
