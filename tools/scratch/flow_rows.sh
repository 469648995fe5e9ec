#!/bin/bash
cd $GRAFT_REPO_ROOT
GPMI_FLOW_NEAR_DEPTH=2 timeout 600 python3 -m pytest tests -m gpu -x -q -k "flow or schedule" 2>&1 | tail -1
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg2 2>&1 | tail -1
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   headline ms_per_step', round(d['ms_per_step'],3), 'flow tail', round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
run GPMI_FLOW_ROWTASKS=0 GPMI_FLOW_WINDOW=1
run GPMI_FLOW_NEAR_DEPTH=2
run GPMI_FLOW_NEAR_DEPTH=2 GPMI_FLOW_NEAR_WGS=96
run GPMI_FLOW_NEAR_DEPTH=2 GPMI_FLOW_R_REMAIN=28
run GPMI_FLOW_NEAR_DEPTH=2 GPMI_FLOW_R_REMAIN=200
run GPMI_FLOW_NEAR_DEPTH=3 GPMI_FLOW_NEAR_WGS=96
GPMI_FLOW_NEAR_DEPTH=2 N=8192 tools/flow_tr.sh gpurun_out/flow8k 2>&1 | head -3; python tools/flow_panels.py gpurun_out/flow8k/trace.bin | grep -v busy
