"""Reads a GPMI_FLOW_TRACE file (potrf_flow.hip) and prints, per chain step, when each launch ran and when the tasks the
chain waits for were seen ready / finished.  usage: python tools/flow_trace.py trace.bin [first_step] [steps]"""
import struct
import sys

import numpy as np

path = sys.argv[1]
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 4
raw = open(path, "rb").read()
m, nwg, ntasks, cw = struct.unpack("4q", raw[:32])
p = 32
off = np.frombuffer(raw, np.int32, nwg + 1, p); p += 4 * (nwg + 1)
tasks = np.frombuffer(raw, np.dtype([("type", "u1"), ("s", "u1"), ("fadd", "u1"), ("pad", "u1"), ("i", "u2"), ("j", "u2"), ("k", "u2"), ("pad2", "u2")]), ntasks, p); p += 12 * ntasks
tr = np.frombuffer(raw, np.uint64, 4 * ntasks + m * cw, p).astype(np.int64)
tt = tr[: 4 * ntasks].reshape(ntasks, 4)
ct = tr[4 * ntasks:].reshape(m, cw)
owner = np.zeros(ntasks, int)
for b in range(nwg):
    owner[off[b]:off[b + 1]] = b
t0 = ct[0, 0]
us = lambda x: (x - t0) * 0.01
names = "TUZQ"  # (Q: a quarter of a K = 512 chunk, FT_ZS, round 6)
print(f"m={m} lists={nwg} tasks={ntasks}; whole chain {us(ct[m-1, 8]):.0f} us")
dur = (tt[:, 2] - tt[:, 1]) * 0.01
for ty in range(4):
    sel = tasks["type"] == ty
    if not sel.any():
        continue
    print(f"  {names[ty]} tasks: {sel.sum()}  body {np.median(dur[sel]):.1f} us median, {dur[sel].mean():.1f} mean, {np.percentile(dur[sel], 95):.1f} p95;"
          f" publish {np.median((tt[sel, 3] - tt[sel, 2]) * 0.01):.2f} us")
# the chain step: start of potrf_diag(k) to start of potrf_diag(k + 1); in the last third of the tail the chain never waits for
# the task kernel, so the median there is the bare step (D + the two products behind it + launch boundaries)
steps = np.array([(ct[k + 1, 0] - ct[k, 0]) * 0.01 for k in range(m - 1)])
diag = np.array([(ct[k, 8] - ct[k, 0]) * 0.01 for k in range(m)])
tail = slice(max(0, 2 * (m - 1) // 3), m - 1)
print(f"  chain step (D(k) start -> D(k+1) start): median {np.median(steps):.1f} us over all {m - 1}, "
      f"{np.median(steps[tail]):.1f} us median / {steps[tail].min():.1f} min over the last third; potrf_diag itself "
      f"{np.median(diag):.1f} us median; behind potrf_diag's end to the next start {np.median(steps[tail] - diag[:-1][tail]):.1f} us")
for k in range(k0, min(k0 + nk, m - 1)):
    d0, d1 = us(ct[k, 0]), us(ct[k, 8])
    tc0, tc1 = us(ct[k, 24]), us(ct[k, 25])
    uc0, uc1 = us(ct[k, 26]), us(ct[k, 27])
    print(f"step {k}: D {d0:.1f}..{d1:.1f} ({d1-d0:.1f}) | Tc enters {tc0:.1f} waits until {tc1:.1f} | Uc enters {uc0:.1f} waits until {uc1:.1f} | next D at {us(ct[k+1,0]):.1f}")
    # the tasks Tc(k+1) waits for: T(k+2, k) slabs and U(k+2, k+1, k) sub-tiles
    for ty, i, j in ((0, k + 2, 0), (1, k + 2, k + 1), (1, k + 2, k + 2)):
        sel = np.where((tasks["type"] == ty) & (tasks["i"] == i) & (tasks["k"] == k) & ((ty == 0) | (tasks["j"] == j)))[0]
        for n in sel:
            print(f"     {names[ty]}({i},{j if ty else k},{k}) s={tasks['s'][n]} wg {owner[n]:3d}: polled from {us(tt[n,0]):.1f}, ready seen {us(tt[n,1]):.1f}, body done {us(tt[n,2]):.1f}, published {us(tt[n,3]):.1f}")
