#!/bin/bash
# the slice model after round 4's faster panel chain: alternating runs, best and median of six
cd $GRAFT_REPO_ROOT
one() {
  env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3))"
}
declare -A res
cfgs=("GPMI_SLICE_PCT=100" "GPMI_SLICE_PCT=115" "GPMI_SLICE_PCT=130" "GPMI_SLICE_PCT=145")
for i in 1 2 3 4 5 6; do
  for c in "${cfgs[@]}"; do res[$c]="${res[$c]} $(one $c)"; done
done
for c in "${cfgs[@]}"; do echo "$c: ${res[$c]}" | python3 -c "
import sys
l=sys.stdin.read().split(':'); v=sorted(float(x) for x in l[1].split()); print(l[0], 'min', v[0], 'median', (v[2]+v[3])/2, v)"; done
