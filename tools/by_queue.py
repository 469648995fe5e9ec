"""Per-(kernel, queue) split of a rocprofv3 kernel trace: a kernel_stats row blends the launches of one kernel NAME
over all queues - for the trailing-update kernel the 224-CU launches of the update stream and the slices on the
panel stream's 32 CUs.  This table lets both figures of the bench line (`roofline.frac`: the update-stream launches,
`roofline.frac_all_launches`: the blend) be re-derived from a committed file.
usage: python tools/by_queue.py <rocprof dir> <out.csv> [K of the trailing update, default 512]
Columns: kernel, queue, launches, total_us, avg_us, workgroups (sum), and for the 128 x 128-tile update kernel
(gemm_dma_kernel<1, 0>, every workgroup = one tile with K = 512 in the bench command) the algorithmic TFLOP/s =
workgroups x 2 x 128^2 x K / total time."""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, r.get("Queue_Id", "?"))
    a = agg.setdefault(key, [0, 0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[2] += int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r.get("Grid_Size_Y", 1)), 1) * max(
        int(r.get("Grid_Size_Z", 1)), 1)
with open(out, "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["kernel", "queue", "launches", "total_us", "avg_us", "workgroups", "algorithmic_tflops_if_tile_update"])
    for (name, q), (n, ns, wgs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = ""
        if name.startswith("gemm_dma_kernel<1, 0>"):
            tf = "%.2f" % (wgs * 2.0 * 128 * 128 * K / (ns * 1e-9) / 1e12)
        w.writerow([name, q, n, "%.1f" % (ns / 1e3), "%.2f" % (ns / 1e3 / n), wgs, tf])
print("wrote", out, len(agg), "rows")
