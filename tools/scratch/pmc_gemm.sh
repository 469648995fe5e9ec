#!/bin/bash
# Collect PMC counters for the GEMM micro-benchmark (separate passes; see MI355X_MICROARCH.md).
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $c --output-format csv -d $out -- python3 tools/bench_gemm.py "$@" > /dev/null 2>&1
done
python3 - "$out" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print("%-32s n=%d mean=%.4g"%(k,len(v),sum(v)/len(v)))
PY
