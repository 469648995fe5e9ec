#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for p in 0 2 1; do
  echo "== GPMI_STREAM_POOL=$p"
  GPMI_STREAM_POOL=$p timeout 300 python tools/grad_times.py 16384 2>&1 | tail -1
  GPMI_STREAM_POOL=$p timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-140
done
done
