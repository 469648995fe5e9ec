"""Progress of the flag-ordered tail from a GPMI_FLOW_TRACE file: when potrf_diag(k) started (every 4th column), against the
first poll of the task kernel, how busy the task workgroups were in between, and the Z chunks by panel.
usage: python tools/flow_curve.py trace.bin"""
import struct
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
m, nwg, ntasks, cw = struct.unpack("4q", raw[:32])  # (nwg: the number of LISTS)
# round 6: GPMI_FLOW_SPLIT gives every workgroup two lists, three with GPMI_FLOW_URGENT (the launch has at most 512 workgroups)
nwg_phys = nwg if nwg <= 512 else (nwg // 3 if nwg % 3 == 0 and nwg // 3 <= 512 and nwg // 2 > 512 else nwg // 2)
p = 32
off = np.frombuffer(raw, np.int32, nwg + 1, p); p += 4 * (nwg + 1)
tasks = np.frombuffer(raw, np.dtype([("type", "u1"), ("s", "u1"), ("fadd", "u1"), ("pad", "u1"), ("i", "u2"), ("j", "u2"), ("k", "u2"), ("pad2", "u2")]), ntasks, p); p += 12 * ntasks
tr = np.frombuffer(raw, np.uint64, 4 * ntasks + m * cw, p).astype(np.int64)
tt = tr[: 4 * ntasks].reshape(ntasks, 4)
ct = tr[4 * ntasks:].reshape(m, cw)
t0 = tt[:, 0][tt[:, 0] > 0].min()
us = lambda x: (x - t0) * 0.01
end = max(us(tt[:, 3].max()), us(ct[m - 1, 8]))
print(f"m={m} tasks={ntasks}: task kernel's first poll = 0, last publish {us(tt[:, 3].max()):.0f} us, chain ends {us(ct[m-1, 8]):.0f} us")
print("D(k) starts [us]:", " ".join(f"{k}:{us(ct[k, 0]):.0f}" for k in range(0, m, 4)))
# busy fraction of the task workgroups per 250 us window
body = np.stack([us(tt[:, 1]), us(tt[:, 2])], 1)
edges = np.arange(0, end + 250, 250)
row = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(body[:, 1], b) - np.maximum(body[:, 0], a), 0, None).sum()
    row.append(ov / ((b - a) * nwg_phys))
print("task workgroups busy, per 250 us:", " ".join(f"{x:.2f}" for x in row))
z = (tasks["type"] == 2) | (tasks["type"] == 3)  # K = 512 chunks, whole or as quarters (FT_ZS)
for q in np.unique(tasks["k"][z]):
    sel = z & (tasks["k"] == q)
    print(f"  Z k={q}: {sel.sum()} tiles, bodies from {us(tt[sel, 1].min()):.0f} to {us(tt[sel, 2].max()):.0f} us, median body {np.median((tt[sel, 2] - tt[sel, 1]) * 0.01):.0f} us")
