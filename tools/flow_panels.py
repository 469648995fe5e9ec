"""Per outer panel summary of a GPMI_FLOW_TRACE file: chain D starts, Z chunk windows, workgroup busy fraction per 25 us."""
import struct
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
m, nl, ntasks, cw = struct.unpack("4q", raw[:32]); p = 32
off = np.frombuffer(raw, np.int32, nl + 1, p); p += 4 * (nl + 1)
tasks = np.frombuffer(raw, np.dtype([("type", "u1"), ("s", "u1"), ("fadd", "u1"), ("pad", "u1"), ("i", "u2"), ("j", "u2"), ("k", "u2"), ("pad2", "u2")]), ntasks, p); p += 12 * ntasks
tr = np.frombuffer(raw, np.uint64, 4 * ntasks + m * cw, p).astype(np.int64)
tt = tr[:4 * ntasks].reshape(ntasks, 4); ct = tr[4 * ntasks:].reshape(m, cw)
t0 = ct[0, 0]
us = lambda x: (x - t0) * 0.01
Z = tasks['type'] == 2
for k in np.unique(tasks['k'][Z]):
    sel = Z & (tasks['k'] == k)
    print(f"Z k={k}: n={sel.sum()} ready {us(tt[sel,1].min()):.0f}..{us(tt[sel,1].max()):.0f} done {us(tt[sel,2].min()):.0f}..{us(tt[sel,2].max()):.0f} body {np.median((tt[sel,2]-tt[sel,1])*0.01):.0f} us median")
print("D starts", [round(us(ct[k, 0])) for k in range(m)])
T = int(us(tt[:, 3].max())) + 1
bins = np.zeros(T // 25 + 1)
for n in range(ntasks):
    a, b = us(tt[n, 1]), us(tt[n, 2])
    if b <= a: continue
    for i in range(int(a // 25), int(b // 25) + 1):
        lo = max(a, i * 25); hi = min(b, (i + 1) * 25)
        if hi > lo: bins[i] += hi - lo
nwg = 448
print("busy fraction per 25us:", " ".join(f"{x/25/nwg:.2f}" for x in bins))
for ty in range(4):
    sel = tasks['type'] == ty
    if not sel.any():
        continue
    d = (tt[sel, 2] - tt[sel, 1]) * 0.01
    print("TUZR"[ty], "body median", np.median(d), "mean", d.mean(), "sum/wg", d.sum() / nwg)
