"""Timeline of the kernels behind the last idle gap >= 30 ms of a rocprofv3 kernel trace (tools/call_timeline.py sleeps 50 ms before
its last call).  usage: python tools/trace_last_call.py <rocprof dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('Queue_Id', '?'))
               for r in csv.DictReader(open(f))))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 30e6:
        cut = i
rows = rows[cut:]
t0 = rows[0][0]
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:46]
prev_end = {}
for b, e, nme, gx, wx, q in rows:
    print("%9.1f %9.1f %8.1f  %-46s wg %6d  q %s" % ((b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3, short(nme), int(gx) // max(int(wx), 1), q))
print("call wall %.3f ms, %d kernels" % ((max(r[1] for r in rows) - t0) / 1e6, len(rows)))
