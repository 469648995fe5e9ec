#!/bin/bash
# other configurations with the re-fitted slice model and the switch at 52 rows; LOOKAHEAD_MIN re-checked with it
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q -k "schedule or fit or flow or headline" 2>&1 | tail -1
one() {
  env "$@" python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3))"
}
declare -A res
cfgs=("GPMI_LOOKAHEAD_MIN=52" "GPMI_LOOKAHEAD_MIN=48" "GPMI_LOOKAHEAD_MIN=56")
for i in 1 2 3 4 5; do
  for c in "${cfgs[@]}"; do res[$c]="${res[$c]} $(one $c)"; done
done
for c in "${cfgs[@]}"; do echo "$c: ${res[$c]}" | python3 -c "
import sys
l=sys.stdin.read().split(':'); v=sorted(float(x) for x in l[1].split()); print(l[0], 'min', v[0], 'median', v[len(v)//2], v)"; done
python3 tools/config_bench.py cfg2 | tail -1; python3 tools/config_bench.py cfg3 | tail -1; python3 tools/grad_times.py 16384 | tail -1
