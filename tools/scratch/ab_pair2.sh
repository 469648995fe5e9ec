#!/bin/bash
cd "$(dirname "$0")/../.."
c2() { echo "== cfg2 $*"; env "$@" timeout 300 python tools/config_bench.py cfg2 2>&1 | tail -1 | cut -c1-90; }
c2 GPMI_FLOW_PAIR=0
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_NWG=480
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_NWG=464
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_NWG=448
c2 GPMI_FLOW_PAIR=1 GPMI_FLOW_NWG=488
c2 GPMI_FLOW_PAIR=0
