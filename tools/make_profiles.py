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
for name in ("cfg2.txt", "cfg3.txt", "cfg4.txt", "cfg5.txt", "lmlgrad.txt", "pt.json", "pt16.json", "pt_traced.json", "propose.json", "bench_timeline.txt"):
    src = os.path.join(go, name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(pr, f"{tag}_{name}"))
bench = json.loads(open(os.path.join(go, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(pr, f"{tag}_bench.json"), "w"), indent=1)

# rocprofv3's average duration of the dominant kernel over the traced run, beside the bench's own stamps
rows = list(csv.DictReader(open(stats_of("bench"))))
upd = next(r for r in rows if "gemm_dma_kernel<1, 0>" in r["Name"] or "gemm_nt_kernel<1, 0, 0, 128, 128>" in r["Name"])
upd_us = float(upd["AverageNs"]) / 1e3

pmc = json.load(open(os.path.join(go, "pmc.json")))


def mean(k, c):
    return pmc[k][c]["mean"]


KB = 1024.0
out = {
    "command": "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sharded, GPMI_FLOW=0  (tools/pmc_bench.sh; passes: FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE)",
    "correction": "FETCH_SIZE x 2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md), WRITE_SIZE exact; both in KiB",
    "kernels": {},
    "cross_check": {"rocprof_kernel_stats_avg_us_of_the_dominant_kernel": upd_us, "launches": int(upd["Calls"]),
                    "bench_stamp_avg_us": bench["roofline"]["avg_launch_ms"] * 1e3,
                    "bench_stamp_avg_us_all_launches_of_this_kernel_name": bench["roofline"].get("same_kernel_name_all_launches", {}).get("avg_launch_ms", float("nan")) * 1e3,
                    "bench_achieved_tflops": bench["roofline"]["achieved"],
                    # the kernel_stats row mixes update-stream launches and slices (same kernel name): durations are
                    # compared over that same set; the stamps' rate scaled by the ratio is what rocprofv3's clock gives
                    "achieved_tflops_with_rocprof_durations": bench["roofline"]["achieved"]
                    * bench["roofline"].get("same_kernel_name_all_launches", {}).get("avg_launch_ms", float("nan")) * 1e3 / upd_us},
}
for key, name in (("update128", "gemm_dma_kernel<1, 0> (trailing update, 128x128 tiles)"),
                  ("kbuild", "kbuild_kernel<true, SE> (covariance build, lower tiles)"),
                  ("trsv_fwd", "trsv_fwd_flow_kernel (forward sweep)")):
    if key not in pmc:
        continue
    fetch = 2.0 * mean(key, "FETCH_SIZE") * KB
    write = mean(key, "WRITE_SIZE") * KB
    ent = {"name": name, "launches_sampled": pmc[key]["FETCH_SIZE"]["n"],
           "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
           "hbm_bytes_per_launch": fetch + write,
           "l2_hit_rate": mean(key, "TCC_HIT_sum") / (mean(key, "TCC_HIT_sum") + mean(key, "TCC_MISS_sum"))}
    cyc = mean(key, "GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
    ent["gpu_cycles_per_launch"] = cyc
    if mean(key, "SQ_VALU_MFMA_BUSY_CYCLES") > 0:
        ent["mfma_busy_fraction_of_all_simd_cycles"] = mean(key, "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 256 * 4)
        ent["mfma_flop_per_launch"] = mean(key, "SQ_INSTS_VALU_MFMA_MOPS_F64") * 512.0
    out["kernels"][key] = ent
json.dump(out, open(os.path.join(pr, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(out["cross_check"], indent=1))
print(json.dumps(out["kernels"]["update128"], indent=1))
