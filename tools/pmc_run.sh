#!/bin/bash
# Separate rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) of ONE command, and
# a per-kernel summary (mean of every counter over the sampled launches) as JSON on stdout.
# usage: tools/pmc_run.sh <outdir under gpurun_out> <program and arguments: the program itself directly after -->
#   e.g. tools/pmc_run.sh gpurun_out/prof_r04/pmc_bench python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded
# GPMI_FLOW=0 throughout: counter collection serialises kernels, the flag-ordered tail needs its two launches side by
# side.  The command's own output of every pass is kept (<group>.log): the FETCH_SIZE pass's bench line states the
# algorithmic bytes per launch for exactly the schedule the counters saw.
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  d=$out/$(echo $c | tr ' ' '+')
  GPMI_FLOW=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -- "$@" > $d.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
PAT = [("update128", ("gemm_dma_kernel<1, 0>",)), ("predict_trsm", ("gemm_dma_kernel<0, 0>",)), ("kbuild", ("kbuild_kernel<true",)),
       ("kbuild_batched", ("kbuild_batched_kernel",)), ("trsv_fwd", ("trsv_fwd_flow",)), ("trsv_bwd", ("trsv_bwd_flow",)),
       ("potrf_diag", ("potrf_diag_kernel",)), ("lml_grad", ("lml_grad_kernel",)), ("update64", ("gemm_dma64_kernel<1, 0>",)),
       ("panel_trsm", ("gemm_nt_kernel<0, 1, 0, 64, 128>", "gemm_nt_kernel<0, 1, 0, 32, 128>")), ("syrk_kskip", ("gemm_dma_kernel<1, 1>", "gemm_dma_kernel<0, 1>"))]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for key, pats in PAT:
            if any(p in n for p in pats):
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
                break
res = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in agg.items()}
print(json.dumps(res, indent=1))
PY
