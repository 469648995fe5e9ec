#!/bin/bash
# Wave-level stall counters of the stand-alone trailing-update GEMM on 128 CUs (not clock-limited there).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_gemm2; rm -rf $out; mkdir -p $out
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  d=$out/$(echo $c | tr ' ' '+' | cut -c1-40)
  GPMI_DEV_CUS=128 timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_gemm.py 12288 512 1 3 > $d.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_dma_kernel' in r['Kernel_Name'] or 'gemm_nt_kernel<1, 0, 0, 128, 128>' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
