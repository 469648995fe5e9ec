"""Copy the judged summaries of a profiling session (tools/profile_round.sh <tag>, run through gpurun) from
gpurun_out/prof_<tag>/ into profiles/ (tracked).   usage: python tools/make_profiles.py <tag>   e.g. r02"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
go = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
pr = os.path.join(ROOT, "profiles")


def stats_of(sub):
    f = sorted(glob.glob(os.path.join(go, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    return f[-1] if f else None  # the newest: gpurun merges a session into what earlier sessions left behind


for sub in ("bench", "cfg2", "cfg3", "cfg4", "cfg5", "lmlgrad", "pt"):
    st = stats_of(sub)
    if st:
        shutil.copy(st, os.path.join(pr, f"{tag}_{sub}_kernel_stats.csv"))
for name in ("cfg2.txt", "cfg3.txt", "cfg4.txt", "cfg5.txt", "lmlgrad.txt", "pt.json", "pt16.json", "pt_traced.json", "propose.json", "search.json", "bench_timeline.txt",
             "vendor.json", "flow_curve_n8192.txt"):
    src = os.path.join(go, name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_{name}"))
bench = json.loads(open(os.path.join(go, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(pr, f"{tag}_bench.json"), "w"), indent=1)

# rocprofv3's average duration of the dominant kernel over the traced run, beside the bench's own stamps
rows = list(csv.DictReader(open(stats_of("bench"))))
upd = next(r for r in rows if "gemm_dma_kernel<1, 0>" in r["Name"] or "gemm_nt_kernel<1, 0, 0, 128, 128>" in r["Name"])
upd_us = float(upd["AverageNs"]) / 1e3

for name, dst in (("pmc_stalls/summary.json", "pmc_stalls.json"), ("flow_trace_n8192.txt", "flow_trace_n8192.txt")):
    src = os.path.join(go, name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_{dst}"))
for name in ("bench_kernel_stats_by_queue.csv",):
    src = os.path.join(go, name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_{name}"))


def load(name):
    try:
        return json.load(open(os.path.join(go, name)))
    except (OSError, ValueError):
        return {}


pmc_sets = {"bench": load("pmc_bench.json"), "lml_gradient": load("pmc_grad.json"), "config5": load("pmc_cfg5.json")}
# the bench line printed by the FETCH_SIZE pass itself (GPMI_FLOW=0): the algorithmic bytes per launch for exactly the
# schedule whose traffic the counters saw
alg = {}
try:
    pl = json.loads([l for l in open(os.path.join(go, "pmc_bench_line.txt")).read().splitlines() if l.startswith("{")][-1])
    alg = pl["roofline"]["same_kernel_name_all_launches"]
except (OSError, IndexError, KeyError, ValueError):
    pass

KB = 1024.0
NAMES = {"update128": "gemm_dma_kernel<1, 0> (trailing update, 128x128 tiles)",
         "predict_trsm": "gemm_dma_kernel<0, 0> (updates of the many-right-hand-side solves: predict TRSM, L^-T; regression.py:213, 556)",
         "kbuild": "kbuild_kernel<true, SE> (covariance build, lower tiles)",
         "kbuild_batched": "kbuild_batched_kernel (lockstep covariance build)",
         "trsv_fwd": "trsv_fwd_flow_kernel (forward sweep)", "trsv_bwd": "trsv_bwd_flow_kernel (backward sweep)",
         "potrf_diag": "potrf_diag_kernel (128 x 128 diagonal block: factor + inverse; regression.py:241)",
         "lml_grad": "lml_grad_kernel (fused trace contraction of the likelihood gradient; regression.py:565-566)",
         "update64": "gemm_dma64_kernel<1, 0> (64x64-tile updates: remainders, lockstep batches)",
         "panel_trsm": "gemm_nt_kernel<0, 1, 0, {64,32}, 128> (panel TRSM as a product with the inverse block)",
         "syrk_kskip": "gemm_dma_kernel<., 1> (k-skipped SYRK K^-1 = L^-T L^-1; regression.py:556-557)"}
out = {
    "command": "tools/pmc_run.sh: rocprofv3 --pmc <one group per pass> --output-format csv -- <command>, GPMI_FLOW=0; passes: FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE; commands: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded | tools/grad_times.py 16384 | tools/config5_bench.py 4",
    "correction": "FETCH_SIZE x 2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md), WRITE_SIZE exact; both in KiB",
    "same_schedule": "traffic and algorithmic bytes of the dominant kernel come from the SAME run: the FETCH_SIZE pass prints its own bench line (GPMI_FLOW=0: stream-ordered tail), whose same_kernel_name_all_launches.algorithmic_bytes_per_launch_avg is quoted below",
    "kernels": {},
    "cross_check": {"rocprof_kernel_stats_avg_us_of_the_dominant_kernel": upd_us, "launches": int(upd["Calls"]),
                    "bench_stamp_avg_us": bench["roofline"]["avg_launch_ms"] * 1e3,
                    "bench_stamp_avg_us_all_launches_of_this_kernel_name": bench["roofline"].get("same_kernel_name_all_launches", {}).get("avg_launch_ms", float("nan")) * 1e3,
                    # (since round 6 the line's `achieved` is already the rocprofv3-duration figure of the PREVIOUS committed
                    # profile; the live stamp figure is `achieved_stamps`)
                    "bench_achieved_tflops": bench["roofline"].get("achieved_stamps", bench["roofline"]["achieved"]),
                    "achieved_tflops_with_rocprof_durations": bench["roofline"].get("achieved_stamps", bench["roofline"]["achieved"])
                    * bench["roofline"].get("same_kernel_name_all_launches", {}).get("avg_launch_ms", float("nan")) * 1e3 / upd_us},
}
for cmd, pmc in pmc_sets.items():
    for key, cnt in pmc.items():
        def mean(c):
            return cnt[c]["mean"] if c in cnt else float("nan")
        fetch = 2.0 * mean("FETCH_SIZE") * KB
        write = mean("WRITE_SIZE") * KB
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
        ent = {"name": NAMES.get(key, key), "command": cmd, "launches_sampled": cnt.get("FETCH_SIZE", {}).get("n", 0),
               "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
               "hbm_bytes_per_launch": fetch + write,
               "l2_hit_rate": mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum")),
               "gpu_cycles_per_launch": cyc}
        if mean("SQ_VALU_MFMA_BUSY_CYCLES") > 0:
            ent["mfma_busy_fraction_of_all_simd_cycles"] = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 256 * 4)
            ent["mfma_flop_per_launch"] = mean("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512.0
        if key == "update128" and cmd == "bench" and alg:
            ent["algorithmic_bytes_per_launch_same_run"] = alg.get("algorithmic_bytes_per_launch_avg")
            ent["traffic_over_algorithmic"] = (fetch + write) / alg["algorithmic_bytes_per_launch_avg"] if alg.get("algorithmic_bytes_per_launch_avg") else None
        out["kernels"][key if cmd == "bench" else f"{cmd}:{key}"] = ent
json.dump(out, open(os.path.join(pr, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(out["cross_check"], indent=1))
print(json.dumps(out["kernels"].get("update128", {}), indent=1))
