#!/bin/bash
# flag-ordered tail at 52 rows: its own knobs once more
cd $GRAFT_REPO_ROOT
one() {
  env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
for c in "GPMI_FLOW_WGS=2" "GPMI_FLOW_WGS=1" "GPMI_FLOW_NEAR=8" "GPMI_FLOW_NEAR=2" "GPMI_FLOW_NEAR_WGS=24" "GPMI_FLOW_NEAR_D=2" "GPMI_FLOW_NEAR_D=5 GPMI_FLOW_NEAR_WGS=48"; do
  echo "== $c: $(one $c) | $(one $c) | $(one $c)"
done
