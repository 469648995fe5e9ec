"""Copy the judged summaries of a gpurun profiling session from gpurun_out/ into profiles/ (tracked).
usage: python tools/make_profiles.py <tag>   e.g. r01   (expects gpurun_out/prof_<tag>e, pmc_<tag>e.json, bench_<tag>e.json)"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
go = os.path.join(ROOT, "gpurun_out")
pr = os.path.join(ROOT, "profiles")
stats = glob.glob(os.path.join(go, f"prof_{tag}e", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(pr, f"{tag}_bench_kernel_stats.csv"))
bench = json.loads(open(os.path.join(go, f"bench_{tag}e.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(pr, f"{tag}_bench.json"), "w"), indent=1)

# rocprof average of the stamped launches (the last `steps` fits of the traced run) for the cross-check
trace = glob.glob(os.path.join(go, f"prof_{tag}e", "*", "*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(trace)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
fits = [i for i, k in enumerate(ks) if "kbuild_kernel<true>" in k[2]]
last = ks[fits[-5]:]
def avg_us(sub):
    sel = [k for k in last if sub in k[2]]
    return len(sel), sum(k[1] - k[0] for k in sel) / max(len(sel), 1) / 1e3
n_upd, upd_us = avg_us("gemm_nt_kernel<1, 0, 0, 128, 128>")

pmc = json.load(open(os.path.join(go, f"pmc_{tag}e.json")))
def mean(k, c):
    return pmc[k][c]["mean"]
KB = 1024.0
out = {
    "command": "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline  (tools/pmc_bench.sh; passes: FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE)",
    "correction": "FETCH_SIZE x 2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md), WRITE_SIZE exact; both in KiB",
    "kernels": {},
    "cross_check": {"rocprof_kernel_trace_avg_us_of_stamped_launches": upd_us, "launches": n_upd,
                    "bench_stamp_avg_us": bench["roofline"]["avg_launch_ms"] * 1e3},
}
for key, name in (("update128", "gemm_nt_kernel<1, 0, 0, 128, 128> (trailing update)"),
                  ("kbuild", "kbuild_kernel<true> (covariance build, lower tiles)"),
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
print(json.dumps(out, indent=1))
