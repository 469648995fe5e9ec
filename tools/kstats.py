"""Print a rocprofv3 *kernel_stats.csv (found under the directory given) as a short table: python tools/kstats.py <dir> [rows]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%-46s calls %5d avg %8.1f us total %8.1f ms %5.1f%%" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f ms" % (tot / 1e6))
