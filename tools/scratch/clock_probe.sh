#!/bin/bash
# Effective GPU clock of the stand-alone GEMM launches: GRBM_GUI_ACTIVE cycles (summed over the 8 XCDs) over the
# kernel duration of the same launches from a kernel trace.  usage: tools/clock_probe.sh [n] [k]
n=${1:-15872}; k=${2:-512}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/clk_pmc gpurun_out/clk_trace
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_pmc -- python3 tools/bench_gemm.py $n $k 1 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/clk_trace -- python3 tools/bench_gemm.py $n $k 1 4 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
c = [float(r['Counter_Value']) for f in glob.glob('gpurun_out/clk_pmc/*/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'gemm_nt' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for f in glob.glob('gpurun_out/clk_trace/*/*kernel_trace.csv') for r in csv.DictReader(open(f)) if 'gemm_nt' in r['Kernel_Name']]
cyc = sum(c[-4:]) / 4 / 8; dur = sum(d[-4:]) / 4
print(f"cycles per XCD per launch {cyc:.0f}, duration {dur/1e3:.1f} us -> {cyc/dur:.3f} GHz")
PY
