#!/bin/bash
cd "$(dirname "$0")/../.."
pt() { echo "== $*"; env "$@" timeout 300 python tools/config5_bench.py ${STEPS:-20} ${LAD:-8} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s  %.1f chain steps/s  evals %d swaps %d' % (d['lml_evals_per_s'], d['chain_steps_per_s'], d['lml_evaluations'], d['swaps_accepted']))"; }
pt GPMI_PT_ASYNC=0
pt GPMI_PT_GROUPS=2
pt GPMI_PT_GROUPS=4
pt GPMI_PT_GROUPS=8
pt GPMI_PT_ASYNC=0
pt GPMI_PT_GROUPS=2
pt GPMI_PT_GROUPS=4
pt GPMI_PT_GROUPS=8
LAD=64 STEPS=10 pt GPMI_PT_ASYNC=0
LAD=64 STEPS=10 pt GPMI_PT_GROUPS=8
LAD=64 STEPS=10 pt GPMI_PT_GROUPS=16
