"""Print the kernel timeline of the last fit in a rocprofv3 kernel-trace CSV (debug helper)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][27:60], r['Grid_Size_X'], r['Stream_Id']) for r in rows]
ks.sort()
idx = [i for i, k in enumerate(ks) if 'kbuild_kernel<true' in k[2]]
s = idx[-1]
seg = ks[s:s + n]
t0 = seg[0][0]
for (b, e, nme, g, st) in seg:
    print("%9.1f -> %9.1f us dur %8.1f  %-34s grid %8s stream %s" % ((b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3, nme, g, st))
full = ks[s:]
print("step wall %.2f ms" % ((max(k[1] for k in full) - full[0][0]) / 1e6))
