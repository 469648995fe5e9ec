#!/bin/bash
# Sweep of the flag-ordered tail's list knobs on top of split lists + quarter chunks (round 6): N = 8192 fit, steady state.
# usage (GPU box): tools/ab_flow_knobs.sh
export GPMI_FLOW_SPLIT=${GPMI_FLOW_SPLIT:-1} GPMI_FLOW_QUARTER=${GPMI_FLOW_QUARTER:-0}
run() { echo -n "$* : "; env "$@" python tools/fit_timeline.py 8192 60 | tail -1; }
run A=0
for near in 2 8 16 64; do run GPMI_FLOW_NEAR=$near; done
for d in 2 5 8; do run GPMI_FLOW_NEAR_D=$d; done
for w in 16 64 96; do run GPMI_FLOW_NEAR_WGS=$w; done
run GPMI_FLOW_NEAR=8 GPMI_FLOW_NEAR_D=5 GPMI_FLOW_NEAR_WGS=64
run GPMI_FLOW_NEAR=16 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=96
run GPMI_FLOW_NEAR=64 GPMI_FLOW_NEAR_D=3 GPMI_FLOW_NEAR_WGS=64
run A=0
