#!/bin/bash
# headline step against the CUs reserved for the panel chain (GPMI_PANEL_CUS, default 32), sustained (20 steps each)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in 32 24 16 8; do
  GPMI_PANEL_CUS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sharded --no-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('GPMI_PANEL_CUS=$v', round(d['ms_per_step'],2), 'ms/step; update', round(r['achieved'],2), 'TFLOP/s, clock', r.get('clock_ghz', r.get('shader_clock_ghz')), '; flow tail', round(r['flow_tail']['ms_per_step'],2), 'ms')"
done; done
