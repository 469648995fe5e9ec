"""Dump the kernel timeline (start, end, duration, kernel, grid, stream) of the last bench step found in a
rocprofv3 kernel-trace CSV to a text file: usage  python tools/timeline_full.py <rocprof dir> <out.txt>"""
import csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Grid_Size_X'], r['Workgroup_Size_X'],
             r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows)
idx = [i for i, k in enumerate(ks) if 'kbuild_kernel<true' in k[2]]
s = idx[-1]
# (the build is split over two streams when the factorisation starts in its look-ahead regime: both launches belong to the step)
if len(idx) > 1 and ks[idx[-1]][0] - ks[idx[-2]][0] < 200000:
    s = idx[-2]
full = ks[s:]
t0 = full[0][0]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:44]
with open(out, 'w') as g:
    for (b, e, nme, gx, wx, st) in full:
        g.write("%9.1f %9.1f %8.1f  %-44s wg %6d  q %s\n" % ((b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3, short(nme), int(gx) // max(int(wx), 1), st))
    g.write("step wall %.3f ms\n" % ((max(k[1] for k in full) - t0) / 1e6))
print("wrote", out, len(full), "kernels")
