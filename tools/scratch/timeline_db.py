"""Kernel timeline of the last bench step from a rocprofv3 rocpd database (debug helper).
usage: timeline_db.py <results.db> [fit|predict|stats]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "stats"
rows = db.execute("select start, end, name, grid_x, workgroup_x, stream_id from kernels order by start").fetchall()
ks = [(s, e, n, g // max(w, 1), st) for (s, e, n, g, w, st) in rows]
if mode == "stats":
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, g, st in ks:
        agg[n][0] += 1; agg[n][1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("name,calls,total_ms,avg_us,pct")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%.3f,%.2f,%.2f' % (n, c, t / 1e6, t / c / 1e3, 100.0 * t / tot))
    sys.exit()
sq = [i for i, k in enumerate(ks) if 'kbuild_kernel<true' in k[2]]
cr = [i for i, k in enumerate(ks) if 'kbuild_kernel<false' in k[2]]
s = sq[-1]; e = [i for i in cr if i > s][0]
seg = ks[s:e] if mode == "fit" else ks[e:]
t0 = seg[0][0]
print("%s: %d launches, %.2f ms" % (mode, len(seg), (max(k[1] for k in seg) - t0) / 1e6))
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 60
prev = None
show = seg if len(seg) <= 2 * lim else seg[:lim] + [None] + seg[-lim:]
for k in show:
    if k is None:
        print("   ..."); prev = None; continue
    (b, en, n, g, st) = k
    gap = (b - prev) / 1e3 if prev else 0.0
    print("%9.1f us gap %6.1f dur %7.1f  %-46s grid %6d st %s" % ((b - t0) / 1e3, gap, (en - b) / 1e3, n[:46], g, st))
    prev = en
