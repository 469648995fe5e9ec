#!/bin/bash
# round 4: the look-ahead's switch point and the slice model with the faster potrf_diag (one box, alternating)
cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % l['ms_per_step'])"; }
for rep in 1 2; do
for cfg in "A=0" "GPMI_SLICE_CHAIN_US=200" "GPMI_SLICE_CHAIN_US=140" "GPMI_LOOKAHEAD_MIN=52" "GPMI_LOOKAHEAD_MIN=68" "GPMI_LOOKAHEAD_MIN=76" "GPMI_SLICE_PCT=120" "GPMI_SLICE_PCT=80"; do
  echo -n "$cfg: "; run $cfg
done; done
