#!/bin/bash
out=${1:-gpurun_out/flowt}; mkdir -p $out
cd "$(dirname "$0")/../.."
GPMI_FLOW_TRACE=$out/trace.bin timeout 300 python tools/fit_digest.py $out/t.npz ${N:-8192} 2>&1 | tail -1
python tools/flow_trace.py $out/trace.bin 0 64 | grep "^step\|^m=\|tasks:" 
rm -f $out/trace.bin $out/t.npz
