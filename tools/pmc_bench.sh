#!/bin/bash
# HBM traffic and MFMA utilisation counters of the bench's dominant kernel (separate --pmc passes of the same
# command, as MI355X_MICROARCH.md prescribes).  usage: tools/pmc_bench.sh <outdir under gpurun_out>
out=$1; mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  d=$out/$(echo $c | tr ' ' '+')
  # (GPMI_FLOW=0: counter collection serialises kernels; the flag-ordered tail needs its two launches side by side)
  GPMI_FLOW=0 BENCH_NO_PROF=1 timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded > $d.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = 'update128' if ('gemm_dma_kernel<1, 0>' in n or 'gemm_nt_kernel<1, 0, 0, 128, 128>' in n) else 'kbuild' if 'kbuild_kernel<true' in n else \
              'trsv_fwd' if 'trsv_fwd_flow' in n else None
        if key:
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in agg.items()}
print(json.dumps(res, indent=1))
PY
