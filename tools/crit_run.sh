#!/bin/bash
# trace of the tail at N = 8192 and the walk back from stalled chain steps (tools/flow_crit.py); GPMI_FLOW_ZSPLIT=0 lists
cd "$(dirname "$0")/.."
out=gpurun_out/crit; mkdir -p $out
GPMI_FLOW_ZSPLIT=0 GPMI_FLOW_TRACE=$out/trace.bin timeout 300 python tools/fit_digest.py $out/t.npz 8192 2>&1 | tail -1
for k in ${STEPS:-7 11 15}; do python tools/flow_crit.py $out/trace.bin $k; done
python tools/flow_panels.py $out/trace.bin 2>/dev/null | head -40
rm -f $out/trace.bin $out/t.npz
