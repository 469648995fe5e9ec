#!/bin/bash
# lockstep chunk size of the batched likelihood (config 5's 512 evaluations at N = 2048): GPMI_BATCH_GIB / GPMI_BATCH_MAX A/B
cd "$(dirname "$0")/.."
for rep in 1 2; do
for cfg in "6 256 1" "6 256 0" "24 512 1" "24 512 0" "48 1024 1" "48 1024 0"; do
  set -- $cfg
  echo -n "GIB=$1 MAX=$2 SPLIT=$3: "; GPMI_BATCH_GIB=$1 GPMI_BATCH_MAX=$2 GPMI_BATCH_SPLIT=$3 python tools/lml_batch_time.py 2>&1 | tail -1
done; done
