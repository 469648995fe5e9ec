#!/bin/bash
cd "$(dirname "$0")/../.."
pt() { echo "== $*"; env "$@" timeout 300 python tools/config5_bench.py ${STEPS:-20} ${LAD:-8} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s  %.1f chain steps/s  evals %d swaps %d' % (d['lml_evals_per_s'], d['chain_steps_per_s'], d['lml_evaluations'], d['swaps_accepted']))"; }
for rep in 1 2; do
  pt GPMI_PT_ASYNC=0
  pt GPMI_PT_ASYNC=1
done
LAD=64 STEPS=10 pt GPMI_PT_ASYNC=0
LAD=64 STEPS=10 pt GPMI_PT_ASYNC=1
LAD=16 STEPS=10 pt GPMI_PT_ASYNC=0
LAD=16 STEPS=10 pt GPMI_PT_ASYNC=1
