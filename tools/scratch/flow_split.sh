#!/bin/bash
# flow tail: K = 512 chunks in lists of their own (GPMI_FLOW_SPLIT_Z) against the band of dedicated near workgroups
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg2 2>&1 | tail -1
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   headline ms_per_step', round(d['ms_per_step'],3), 'flow tail', round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
run GPMI_FLOW_SPLIT_Z=0
run GPMI_FLOW_SPLIT_Z=1
run GPMI_FLOW_SPLIT_Z=1 GPMI_FLOW_NEAR_D=5 GPMI_FLOW_NEAR_WGS=48
run GPMI_FLOW_SPLIT_Z=1 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=64
run GPMI_FLOW_SPLIT_Z=1 GPMI_FLOW_NEAR_D=8 GPMI_FLOW_NEAR_WGS=96
timeout 600 python3 -m pytest tests -m gpu -x -q -k "flow or potrf or fit" 2>&1 | tail -3
