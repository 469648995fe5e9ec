#!/bin/bash
# flow tail: window of the list scan against the band of dedicated near workgroups
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg2 2>&1 | tail -1
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   headline ms_per_step', round(d['ms_per_step'],3), 'flow tail', round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
run GPMI_FLOW_WINDOW=1
run GPMI_FLOW_WINDOW=8
run GPMI_FLOW_WINDOW=16
run GPMI_FLOW_WINDOW=32
run GPMI_FLOW_WINDOW=64
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=5 GPMI_FLOW_NEAR_WGS=48
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=64
run GPMI_FLOW_WINDOW=16 GPMI_FLOW_NEAR_D=1 GPMI_FLOW_NEAR_WGS=16
timeout 600 python3 -m pytest tests -m gpu -x -q -k "flow or potrf or fit" 2>&1 | tail -3
