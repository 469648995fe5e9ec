"""Walks back from a chain step of a GPMI_FLOW_TRACE file along the input that arrived last.  usage: flow_crit.py trace.bin step"""
import struct
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
K = int(sys.argv[2])
m, nl, ntasks, cw = struct.unpack("4q", raw[:32]); p = 32
off = np.frombuffer(raw, np.int32, nl + 1, p); p += 4 * (nl + 1)
tasks = np.frombuffer(raw, np.dtype([("type", "u1"), ("s", "u1"), ("fadd", "u1"), ("pad", "u1"), ("i", "u2"), ("j", "u2"), ("k", "u2"), ("pad2", "u2")]), ntasks, p); p += 12 * ntasks
tr = np.frombuffer(raw, np.uint64, 4 * ntasks + m * cw, p).astype(np.int64)
tt = tr[:4 * ntasks].reshape(ntasks, 4); ct = tr[4 * ntasks:].reshape(m, cw)
t0 = ct[0, 0]
us = lambda x: (x - t0) * 0.01
owner = np.zeros(ntasks, int)
for b in range(nl):
    owner[off[b]:off[b + 1]] = b
idx = {}
for n in range(ntasks):
    t = tasks[n]
    idx.setdefault((int(t["type"]), int(t["i"]), int(t["j"]) if t["type"] in (1, 2) else 0, int(t["k"])), []).append(n)
DEPTH = int(__import__("os").environ.get("GPMI_FLOW_NEAR_DEPTH", "1"))
lazyp = lambda i, j: max((j // 4 - DEPTH) if i < 4 * (j // 4) + 8 else j // 4, 0)


R_FROM = 12
rrow = lambda i, k: (3, i, 0, k // 4) in idx


def producers_of_tile(i, j, upto):
    """tasks that bring tile (i, j) up to column `upto` (exclusive)"""
    out = []
    for q in range(lazyp(i, j)):
        out += idx.get((2, i, j, q), [])
    for k in range(4 * lazyp(i, j), upto):
        out += idx.get((3, i, 0, k // 4), []) if rrow(i, k) else idx.get((1, i, j, k), [])
    return out


def t_tasks(i, k):
    return idx.get((3, i, 0, k // 4), []) if rrow(i, k) else idx.get((0, i, 0, k), [])


def inputs(n):
    t = tasks[n]
    ty, i, j, k = int(t["type"]), int(t["i"]), int(t["j"]), int(t["k"])
    if ty == 0:
        return producers_of_tile(i, k, k), ("D", k)
    if ty == 1:
        return t_tasks(i, k) + t_tasks(j, k) + producers_of_tile(i, j, k), None
    if ty == 3:
        return [x for c in range(4) for q in range(k) for x in idx.get((2, i, 4 * k + c, q), [])], ("D", 4 * k + 3)
    cols = range(4 * k, 4 * k + 4)
    return [x for c in cols for x in t_tasks(i, c) + t_tasks(j, c)] + idx.get((2, i, j, k - 1), []), None


def describe(n):
    t = tasks[n]
    return f"{'TUZR'[t['type']]}({t['i']},{t['j'] if t['type'] in (1, 2) else t['k']},{t['k']}) s={t['s']} list {owner[n]}: polled {us(tt[n,0]):.0f} ready {us(tt[n,1]):.0f} done {us(tt[n,2]):.0f} pub {us(tt[n,3]):.0f}"


print(f"step {K}: D {us(ct[K,0]):.0f}..{us(ct[K,8]):.0f}; Tc enters {us(ct[K,24]):.0f} waits until {us(ct[K,25]):.0f}; Uc enters {us(ct[K,26]):.0f} until {us(ct[K,27]):.0f}")
# Tc(K) waits for tile (K+1, K) up to column K
cur = max(producers_of_tile(K + 1, K, K), key=lambda n: tt[n, 3])
for depth in range(40):
    print("  " * 0 + describe(cur))
    ins, d = inputs(cur)
    ins = [n for n in ins if n != cur]
    late = max(ins, key=lambda n: tt[n, 3]) if ins else None
    # previous task on the same list / workgroup
    wait_dep = tt[cur, 1] - tt[cur, 0]
    if late is None:
        break
    tl = us(tt[late, 3])
    if d is not None:
        dk = d[1]
        print(f"      (D({dk}) ended {us(ct[dk,8]):.0f}, Tc({dk}) entered {us(ct[dk,24]):.0f})")
    if us(tt[cur, 1]) - tl > 15 and us(tt[cur, 1]) - us(tt[cur, 0]) < 3:
        print(f"      -> resource: its workgroup was busy until {us(tt[cur,0]):.0f} (last input at {tl:.0f})")
        # find the task on the same workgroup that ended just before
        wg = owner[cur] % 448
        same = [n for n in range(ntasks) if owner[n] % 448 == wg and tt[n, 3] <= tt[cur, 0] + 200 and n != cur]
        prev = max(same, key=lambda n: tt[n, 3])
        print("         busy with " + describe(prev))
    cur = late
    if us(tt[cur, 3]) < 200:
        break

if len(sys.argv) > 3:
    # walk back from the Z chunk of panel q that became ready last among rows <= rmax
    q, rmax = int(sys.argv[3]), int(sys.argv[4])
    cand = [n for n in range(ntasks) if tasks[n]["type"] == 2 and tasks[n]["k"] == q and tasks[n]["i"] <= rmax]
    cur = max(cand, key=lambda n: tt[n, 1])
    print(f"--- Z chunk walk, panel {q}, rows <= {rmax}")
    for depth in range(30):
        print(describe(cur))
        ins, d = inputs(cur)
        ins = [n for n in ins if n != cur]
        if not ins:
            break
        late = max(ins, key=lambda n: tt[n, 3])
        if d is not None:
            print(f"      (D({d[1]}) ended {us(ct[d[1],8]):.0f}, next launch entered {us(ct[d[1],24]):.0f})")
        if us(tt[cur, 1]) - us(tt[late, 3]) > 15 and us(tt[cur, 1]) - us(tt[cur, 0]) < 3:
            print(f"      -> resource: list busy until {us(tt[cur,0]):.0f} (last input at {us(tt[late,3]):.0f})")
        cur = late
