#!/bin/bash
# LDS bank-conflict audit of every kernel the path launches: one rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE,
# SQ_INSTS_LDS, SQ_WAVE_CYCLES; counter collection serialises kernels, hence GPMI_FLOW=0 - the flag-ordered tail needs two
# kernels resident at once) over a fit + predict at N = 8192, LML + gradient at N = 4096, config 4 (EI with gradient) and a
# short config-5 run, then per kernel: conflict cycles / LDS-array cycles.  The program itself directly after `--`.
# usage: tools/lds_audit.sh <outdir under gpurun_out>     -> <outdir>/lds_audit.txt
out=${1:-gpurun_out/lds_audit}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPMI_FLOW=0
c="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"
timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/fit -- python3 tools/fit_digest.py /tmp/lds_o.npz 8192 > $out/fit.log 2>&1
timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/grad -- python3 tools/grad_times.py 4096 > $out/grad.log 2>&1
timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/cfg4 -- python3 tools/config_bench.py cfg4 > $out/cfg4.log 2>&1
timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/cfg5 -- python3 tools/config5_bench.py 3 > $out/cfg5.log 2>&1
python3 - "$out" <<'PY' | tee $out/lds_audit.txt
import csv, glob, collections, sys, re
for run in ("fit", "grad", "cfg4", "cfg5"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(f"{sys.argv[1]}/{run}/*/*counter_collection.csv") + glob.glob(f"{sys.argv[1]}/{run}/*/*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); n[k] += 1
    print(f"== {run}: kernel, launches, LDS instructions/launch, LDS-array cycles/launch, bank-conflict cycles/launch (share of the array cycles), wave cycles/launch")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0)):
        a, b = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
        print(f"  {k:60s} {n[k]:5d} {v.get('SQ_INSTS_LDS', 0)/n[k]:12.0f} {a/n[k]:12.0f} {b/n[k]:12.0f} ({(b/a if a else 0):5.1%}) {v.get('SQ_WAVE_CYCLES', 0)/n[k]:14.0f}")
PY
