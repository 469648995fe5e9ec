#!/bin/bash
# lockstep batches: thresholds re-checked after the one-problem-per-XCD layout
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" python3 tools/config_bench.py cfg5 | tail -1 | cut -c1-110
  env "$@" python3 tools/config5_bench.py 30 | cut -c150-200
}
run GPMI_BATCH_OUTER=4
run GPMI_BATCH_OUTER=2
run GPMI_BATCH_OUTER=8
run GPMI_BATCH_SPLIT=0
run GPMI_BIG_MIN=256
run GPMI_BIG_MIN=512
run GPMI_M32_MAX=128
run GPMI_M32_MAX=256
