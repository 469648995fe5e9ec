#!/bin/bash
# Where the trailing-update kernel's issue slots go: separate rocprofv3 --pmc passes of the stand-alone
# GEMM (tools/bench_gemm.py: lower-triangular K = 512 update, operands from memory and L1-hot) and of the bench command with
# GPMI_FLOW=0 (counter collection serialises kernels).  The program itself directly after `--`.
# usage: tools/pmc_stalls.sh <outdir under gpurun_out>
out=${1:-gpurun_out/pmc_stalls}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > $out/avail.txt 2>&1
groups=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU")
for tag in mem hot; do
  hot=0; [ $tag = hot ] && hot=1
  for c in "${groups[@]}"; do
    d=$out/gemm_$tag/$(echo $c | tr ' ' '+')
    mkdir -p $out/gemm_$tag
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_gemm.py 15872 512 1 3 $hot > $d.log 2>&1
  done
done
for c in "${groups[@]}"; do
  d=$out/bench/$(echo $c | tr ' ' '+')
  mkdir -p $out/bench
  GPMI_FLOW=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded --no-configs > $d.log 2>&1
done
python3 - "$out" <<'PY'
# per run and queue (bench: the update stream's full-round launches, the slices on the panel stream's 32 CUs and the first
# update on the full chip are different queues): every counter averaged per launch, launches matched across the passes by
# their order, and the ratios DESIGN.md section 4.2 quotes
import csv, glob, collections, sys, json
res = {}
for run in ("gemm_mem", "gemm_hot", "bench"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{sys.argv[1]}/{run}/*/*/*counter_collection.csv"):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "gemm_dma_kernel<1, 0>" in r["Kernel_Name"]:
                d = per[int(r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["_workgroups"] = int(r["Grid_Size"]) / int(r["Workgroup_Size"])
                d["_queue"] = int(r["Queue_Id"])
        for n, (_, v) in enumerate(sorted(per.items())):
            rows[n].update(v)
    byq = collections.defaultdict(list)
    for v in rows.values():
        byq[v["_queue"]].append(v)
    res[run] = {}
    for q, l in sorted(byq.items()):
        g = lambda k: sum(v.get(k, 0.0) for v in l) / len(l)
        act = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
        cu = g("SQ_BUSY_CU_CYCLES")
        wc = g("SQ_WAVE_CYCLES")
        e = {"launches": len(l), "workgroups_avg": g("_workgroups"),
             "counters_per_launch": {k: g(k) for k in sorted(l[0]) if not k.startswith("_")}}
        if act > 0 and cu > 0 and wc > 0:
            e["derived"] = {
                "cus_busy_on_average": cu / act,
                "mfma_busy_of_the_busy_cus_simd_cycles": g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * cu),
                "mfma_busy_of_all_simd_cycles_of_the_chip": g("SQ_VALU_MFMA_BUSY_CYCLES") / (act * 1024.0),
                "wave_cycles_waiting_for_any_instruction": g("SQ_WAIT_INST_ANY") / wc,
                "wave_cycles_waiting_for_lds": g("SQ_WAIT_INST_LDS") / wc,
                "wave_cycles_waiting_any": g("SQ_WAIT_ANY") / wc,
                "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
                "lds_busy_of_busy_cu_cycles": g("SQ_LDS_IDX_ACTIVE") / cu,
            }
        res[run][f"queue_{q}"] = e
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1)
print(json.dumps({r: {q: v.get("derived") for q, v in d.items()} for r, d in res.items()}, indent=1))
PY
find $out -name "*counter_collection.csv" -delete
find $out -name "*agent_info.csv" -delete
