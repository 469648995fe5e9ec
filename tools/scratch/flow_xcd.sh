#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -m gpu -x -q -k "flow or potrf or fit or schedule" 2>&1 | tail -1
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg2 2>&1 | tail -1
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   headline ms_per_step', round(d['ms_per_step'],3), 'flow tail', round(d['roofline']['flow_tail']['ms_per_step'],3))"
}
run GPMI_FLOW_XCD=0
run GPMI_FLOW_XCD=1
run GPMI_FLOW_XCD=0
run GPMI_FLOW_XCD=1
N=8192 tools/flow_tr.sh gpurun_out/flow8k 2>&1 | head -2; python tools/flow_panels.py gpurun_out/flow8k/trace.bin | grep -v busy
