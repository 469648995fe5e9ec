#!/bin/bash
# cfg2 (N = 8192) fit time against the end of the look-ahead regime and the flow kernel's workgroup count
cd $GRAFT_REPO_ROOT
for la in 60 48 40 32 24; do
  echo "== GPMI_LOOKAHEAD_MIN=$la"
  GPMI_LOOKAHEAD_MIN=$la python3 tools/config_bench.py cfg2 2>&1 | tail -2
done
echo "== GPMI_FLOW=0"
GPMI_FLOW=0 python3 tools/config_bench.py cfg2 2>&1 | tail -2
