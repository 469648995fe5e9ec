"""Timing model of the flag-ordered tail AS SHIPPED (csrc/potrf_flow.hip: static in-order task lists per workgroup + the
chain's one launch per column) and a test bench for list builders.  The device kernel runs whatever lists flow_build()
hands it, so a schedule is a function  tasks -> (workgroup, position); this file prices such functions before they are
ported to C++.

  python tools/sim/flow_sched.py [m] [policy ...]      policies: shipped, heft
"""
import heapq
import sys
from collections import defaultdict

OBT = 4


def lazy_panels(i, j, near=4):
    P = j // OBT
    e = P - 1 if i < OBT * P + OBT + near else P
    return max(e, 0)


ZSPLIT = 1  # 4: every K = 512 chunk as four 64 x 64 quarter tasks (what-if); "last": only a tile's last chunk (ZS tasks)


def build_tasks(m, near=4):
    """The decomposition of potrf_flow.hip: flow_build (which tasks exist is fixed: bit-identity with the stream-ordered
    factorisation).  Returns list of dicts."""
    tasks = []
    for k in range(m):
        for i in range(k + 2, m):
            for s in range(4):
                tasks.append(dict(t="T", i=i, j=0, k=k, s=s, add=1))
            for j in range(k + 1, i + 1):
                if OBT * lazy_panels(i, j, near) > k:
                    continue
                for s in range(4):
                    if i == j and s == 1:
                        continue
                    tasks.append(dict(t="U", i=i, j=j, k=k, s=s, add=2 if (i == j and s == 0) else 1))
    for i in range(m):
        for j in range(i + 1):
            for q in range(lazy_panels(i, j, near)):
                if ZSPLIT == 1 or (ZSPLIT == "last" and q != j // OBT - 1):
                    tasks.append(dict(t="Z", i=i, j=j, k=q, s=0, add=4 * OBT))
                else:
                    for s in range(4):
                        if i == j and s == 1:
                            continue
                        tasks.append(dict(t="Z", i=i, j=j, k=q, s=s, add=(2 if (i == j and s == 0) else 1) * OBT, quarter=True))
    return tasks


class Model:
    """Durations in us (measured: profiles/r05 traces): single-column task bodies 7, hop (flag visible -> body starts) 1.8,
    K = 512 chunk 125 with a busy neighbour on its CU / 76 alone, chain: potrf_diag 21.5, DDONE visible 0.5 after it, the
    launch ends 6.5 after max(potrf_diag's end, the worker flags seen + 3), next launch 2.2 later."""
    cT = 7.2
    cU = 7.2
    cZ2 = 124.0
    cZ1 = 76.0
    cQ2 = 37.0  # a 64 x 64 quarter of a chunk beside a busy neighbour (measured: 35 - 39) / alone (23 - 26)
    cQ1 = 25.0
    hop = 1.8
    vis = 0.4
    cD = 21.5
    cTail = 6.5
    cLoad = 3.0
    cB = 2.2


def simulate(m, lists, mod=Model, near=4, verbose=False):
    """lists: list (per workgroup) of task dicts in execution order.  Workgroups b and b + len(lists)//2 share a CU."""
    nw = len(lists)
    half = nw // 2
    Lcnt = [0] * m
    F = [[0] * (m) for _ in range(m)]
    DD = [0]
    t = [0.0]
    ev = []
    seq = [0]

    def at(time, fn, *a):
        seq[0] += 1
        heapq.heappush(ev, (time, seq[0], fn, a))

    def ready(x):
        ty, i, j, k = x["t"], x["i"], x["j"], x["k"]
        if ty == "T":
            return DD[0] >= k + 1 and F[i][k] == 4 * k
        if ty == "U":
            return Lcnt[i] >= 4 * (k + 1) and Lcnt[j] >= 4 * (k + 1) and F[i][j] >= 4 * k
        return Lcnt[i] >= 4 * OBT * (k + 1) and Lcnt[j] >= 4 * OBT * (k + 1) and F[i][j] >= 4 * OBT * k

    pos = [0] * nw
    busy = [False] * nw
    zrun = [False] * nw
    busy_us = [0.0]
    stall = [0.0]
    starts = {}

    def try_wg(b):
        if busy[b] or pos[b] >= len(lists[b]):
            return
        x = lists[b][pos[b]]
        if not ready(x):
            return
        busy[b] = True
        if x["t"] == "Z":
            mate = (b + half) % nw
            if x.get("quarter"):
                dur = mod.cQ2 if zrun[mate] else mod.cQ1
            else:
                dur = mod.cZ2 if zrun[mate] else mod.cZ1
            # (a chunk that starts beside a running one slows that one too; the model prices only the newcomer, and the
            # mate's remaining time is stretched in proportion)
            zrun[b] = True
        else:
            dur = mod.cT if x["t"] == "T" else mod.cU
        starts[id(x)] = t[0] + mod.hop
        busy_us[0] += dur
        at(t[0] + mod.hop + dur, done, b, x)

    def done(b, x):
        busy[b] = False
        zrun[b] = False
        pos[b] += 1
        at(t[0] + mod.vis, publish, x)
        try_wg(b)

    def publish(x):
        if x["t"] == "T":
            Lcnt[x["i"]] += x["add"]
        else:
            F[x["i"]][x["j"]] += x["add"]
        wake()

    # the chain: one launch per column
    ch = dict(k=0, dend=None, waiting=False, t_launch=0.0, ends=[])

    def launch(k):
        ch["k"] = k
        ch["t_launch"] = t[0]
        if k > 0:
            Lcnt[k] = 4 * k
        ch["dend"] = t[0] + mod.cD
        at(ch["dend"] + 0.5, ddone, k)
        ch["waiting"] = True
        wake()

    def ddone(k):
        DD[0] = k + 1
        wake()

    def chain_check():
        if not ch["waiting"]:
            return
        k = ch["k"]
        if k + 1 >= m:
            ch["waiting"] = False
            at(max(ch["dend"], t[0]) + 1.0, finish_launch, k)
            return
        if F[k + 1][k] >= 4 * k and F[k + 1][k + 1] >= 4 * k:
            ch["waiting"] = False
            flags_seen = t[0] + 1.0
            end = max(ch["dend"], flags_seen + mod.cLoad) + mod.cTail
            stall[0] += max(0.0, flags_seen + mod.cLoad - ch["dend"])
            at(end, finish_launch, k)

    def finish_launch(k):
        ch["ends"].append(t[0])
        if k + 1 < m:
            F[k + 1][k + 1] = 4 * (k + 1)
            at(t[0] + mod.cB, launch, k + 1)

    def wake():
        chain_check()
        for b in range(nw):
            try_wg(b)

    at(0.0, launch, 0)
    while ev:
        t[0], _, fn, a = heapq.heappop(ev)
        fn(*a)
    left = sum(len(l) - p for l, p in zip(lists, pos))
    assert left == 0 and len(ch["ends"]) == m, (left, len(ch["ends"]))
    return dict(total=t[0], chain_end=ch["ends"][-1], stall=stall[0], util=busy_us[0] / (nw * t[0]), starts=starts,
                col_ends=ch["ends"])


# ---- list builders ----------------------------------------------------------------------------------------------------
def shipped_lists(m, nwg=448, near_d=3, near_wgs=32, near=4):
    """potrf_flow.hip: flow_build as of round 4."""
    tasks = build_tasks(m, near)
    nearl, rest, zt = [], [], []
    for x in tasks:
        if x["t"] == "T":
            key = (4 * x["k"], x["i"], 0, x["s"])
        elif x["t"] == "U":
            key = (4 * x["k"] + 2, x["i"], x["j"], x["s"])
        else:
            key = (4 * (OBT * x["k"] + OBT - 1) + 1, x["i"], x["j"], 0)
        x["key"] = key
        if x["t"] == "Z":
            zt.append(x)
        elif x["i"] - x["k"] <= near_d:
            nearl.append(x)
        else:
            rest.append(x)
    nearl.sort(key=lambda x: x["key"])
    rest.sort(key=lambda x: x["key"])
    zt.sort(key=lambda x: x["key"])
    nn = near_wgs if nwg > 2 * near_wgs else 0
    lists = [[] for _ in range(nwg)]
    if nn == 0:
        rest = sorted(rest + nearl, key=lambda x: x["key"])
        nearl = []
    for n, x in enumerate(nearl):
        lists[n % nn].append(x)
    for n, x in enumerate(rest):
        lists[nn + n % (nwg - nn)].append(x)
    for n, x in enumerate(zt):
        lists[nn + (n + len(rest)) % (nwg - nn)].append(x)
    for l in lists:
        l.sort(key=lambda x: x["key"])
    return lists


# ---- dependency graph + list scheduling ("simulate, then freeze") ------------------------------------------------------
def graph(m, tasks, near=4):
    """preds[n] = list of task indices / ('L', k) chain launch starts / ('D', k) potrf_diag ends that task n waits for;
    chain_preds[k] = tasks whose flags launch k's workers wait for."""
    idx = {}
    for n, x in enumerate(tasks):
        idx[(x["t"], x["i"], x["j"], x["k"], x["s"])] = n

    def zq(i, j, q):
        return [idx[("Z", i, j, q, s)] for s in range(4) if ("Z", i, j, q, s) in idx]

    def subs(i, j, k):  # the sub-updates of column k on tile (i, j)
        return [idx[("U", i, j, k, s)] for s in range(4) if not (i == j and s == 1)]

    def tile_state(i, j, upto):
        """tasks that bring tile (i, j) to 'all columns < upto applied' (only the last link of the per-tile sequence)"""
        lz = lazy_panels(i, j, near)
        if upto - 1 >= OBT * lz and upto >= 1 and ("U", i, j, upto - 1, 0) in idx:
            return subs(i, j, upto - 1)
        if lz > 0 and upto >= OBT * lz:
            return zq(i, j, lz - 1)
        return []

    def row_col(i, k):  # L(i, k) final
        if i >= k + 2:
            return [idx[("T", i, 0, k, s)] for s in range(4)]
        return [("L", k + 1)]  # i == k + 1: the chain's strip, published at the start of launch k + 1

    preds = []
    for x in tasks:
        t, i, j, k = x["t"], x["i"], x["j"], x["k"]
        if t == "T":
            p = [("D", k)] + tile_state(i, k, k)
        elif t == "U":
            p = row_col(i, k) + row_col(j, k) + tile_state(i, j, k)
        else:
            p = row_col(i, OBT * k + OBT - 1) + row_col(j, OBT * k + OBT - 1)
            if k > 0:
                p += zq(i, j, k - 1)
        preds.append(p)
    chain_preds = []
    for k in range(m):
        p = []
        if k + 1 < m and k >= 1:
            p = tile_state(k + 1, k, k) + tile_state(k + 1, k + 1, k)
        chain_preds.append(p)
    return preds, chain_preds


def heft_lists(m, nwg=448, short_wgs=64, mod=Model, near=4, zprio=1.0, verbose=False, urgent_wgs=0, urgent=None):
    """urgent: set of task keys (t, i, j, k, s) of chunks that may use the `urgent_wgs` workgroups behind the short-only ones"""
    tasks = build_tasks(m, near)
    preds, chain_preds = graph(m, tasks, near)
    n = len(tasks)
    dur = [mod.cT + mod.hop if x["t"] != "Z" else 0.5 * (mod.cZ1 + mod.cZ2) + mod.hop for x in tasks]
    step = mod.cD + mod.cTail + mod.cB
    # successors; chain nodes: ('L', k) = launch k start, ('D', k) = potrf_diag(k) end
    succ = defaultdict(list)
    for v, p in enumerate(preds):
        for u in p:
            succ[u].append(v)
    for k, p in enumerate(chain_preds):
        for u in p:
            succ[u].append(("W", k))  # workers of launch k
    # upward rank by reverse topological order: process chain backwards, tasks by descending "level".  A simple way:
    # memoised recursion (depth is bounded by ~m * 10; use an explicit stack)
    rank = {}
    sys.setrecursionlimit(1000000)

    def r(u):
        if u in rank:
            return rank[u]
        if isinstance(u, tuple):
            kind, k = u
            if kind == "L":  # launch k start: -> D(k) end -> ...; and its own successors (tasks waiting for the strip)
                v = max([mod.cD + r(("D", k))] + [r(s) for s in succ.get(u, [])])
            elif kind == "D":  # potrf_diag(k) end -> launch end (tail) -> next launch
                nxt = (mod.cTail + mod.cB + r(("L", k + 1))) if k + 1 < m else 0.0
                v = max([nxt] + [r(s) for s in succ.get(u, [])])
            else:  # 'W': workers of launch k have their flags: -> tail -> next launch
                v = (mod.cLoad + mod.cTail + mod.cB + r(("L", k + 1))) if k + 1 < m else 0.0
        else:
            v = dur[u] + max([0.0] + [r(s) for s in succ.get(u, [])])
        rank[u] = v
        return v

    for k in range(m - 1, -1, -1):
        r(("L", k))
    for u in range(n):
        r(u)
    prio = [rank[u] * (zprio if tasks[u]["t"] == "Z" else 1.0) for u in range(n)]
    # ---- list scheduling with the timing model
    half = nwg // 2
    remaining = [len(set(map(lambda q: q if not isinstance(q, tuple) else q, p))) for p in preds]
    # count distinct predecessor events (a tuple event counts once)
    pred_sets = [set(p) for p in preds]
    remaining = [len(s) for s in pred_sets]
    users = defaultdict(list)
    for v, s in enumerate(pred_sets):
        for u in s:
            users[u].append(v)
    cp_sets = [set(p) for p in chain_preds]
    cp_left = [len(s) for s in cp_sets]
    cusers = defaultdict(list)
    for k, s in enumerate(cp_sets):
        for u in s:
            cusers[u].append(k)
    t = [0.0]
    ev = []
    seq = [0]

    def at(time, fn, *a):
        seq[0] += 1
        heapq.heappush(ev, (time, seq[0], fn, a))

    ready_short, ready_z, ready_uz = [], [], []  # heaps of (-prio, n)
    free_short = list(range(short_wgs))           # workgroups that never take a chunk
    free_urg = list(range(short_wgs, short_wgs + urgent_wgs))  # short tasks and urgent chunks
    free_any = list(range(short_wgs + urgent_wgs, nwg))
    is_urgent = [urgent is not None and (x['t'], x['i'], x['j'], x['k'], x['s']) in urgent for x in tasks]
    zrun = [False] * nwg
    lists = [[] for _ in range(nwg)]
    order = []

    def release(u):
        for v in users.get(u, []):
            remaining[v] -= 1
            if remaining[v] == 0:
                push(v)
        for k in cusers.get(u, []):
            cp_left[k] -= 1
            if cp_left[k] == 0:
                chain_flags(k)
        dispatch()

    def push(v):
        if tasks[v]["t"] != "Z":
            heapq.heappush(ready_short, (-prio[v], v))
        elif is_urgent[v]:
            heapq.heappush(ready_uz, (-prio[v], v))
        else:
            heapq.heappush(ready_z, (-prio[v], v))

    def startz(b, v):
        mate = (b + half) % nwg
        zrun[b] = True
        start(b, v, mod.cZ2 if zrun[mate] else mod.cZ1)

    def dispatch():
        while ready_short and (free_short or free_urg or free_any):
            b = free_short.pop() if free_short else (free_urg.pop() if free_urg else free_any.pop())
            _, v = heapq.heappop(ready_short)
            start(b, v, mod.cT)
        while ready_uz and (free_urg or free_any):
            b = free_urg.pop() if free_urg else free_any.pop()
            _, v = heapq.heappop(ready_uz)
            startz(b, v)
        while ready_z and free_any:
            b = free_any.pop()
            _, v = heapq.heappop(ready_z)
            startz(b, v)

    def start(b, v, d):
        lists[b].append(tasks[v])
        tasks[v]["sim_start"] = t[0] + mod.hop
        order.append(v)
        at(t[0] + mod.hop + d, fin, b, v)

    def fin(b, v):
        zrun[b] = False
        (free_short if b < short_wgs else (free_urg if b < short_wgs + urgent_wgs else free_any)).append(b)
        at(t[0] + mod.vis, release, v)
        dispatch()

    ch = dict(dend=0.0, flags=None, k=0)

    def launch(k):
        ch["k"] = k
        ch["dend"] = t[0] + mod.cD
        ch["flags"] = None
        at(ch["dend"] + 0.5, release, ("D", k))
        release(("L", k))
        if k + 1 >= m:
            at(ch["dend"] + 1.0, launch_end, k)
        elif cp_left[k] == 0:
            chain_flags(k)

    def chain_flags(k):
        if ch["k"] != k or ch["flags"] is not None:
            return
        ch["flags"] = t[0] + 1.0
        at(max(ch["dend"], ch["flags"] + mod.cLoad) + mod.cTail, launch_end, k)

    ends = []

    def launch_end(k):
        ends.append(t[0])
        if k + 1 < m:
            at(t[0] + mod.cB, launch, k + 1)

    for v in range(n):
        if remaining[v] == 0:
            push(v)
    at(0.0, launch, 0)
    while ev:
        t[0], _, fn, a = heapq.heappop(ev)
        fn(*a)
    assert len(order) == n and len(ends) == m, (len(order), n, len(ends))
    if verbose:
        print(f"  list scheduling itself: {t[0]:.0f} us")
    heft_lists.last = dict(rank=rank, tasks=tasks, makespan=t[0])
    return lists


def write_lists(path, m, lists):
    """<prefix>_m<m>.bin for GPMI_FLOW_LISTS (potrf_flow.hip: flow_lists_override)"""
    import struct

    code = {"T": 0, "U": 1, "Z": 2}
    off = [0]
    body = b""
    for l in lists:
        for x in l:
            body += struct.pack("<4B4H", code[x["t"]], x["s"], x["add"], 0, x["i"], x["j"] if x["t"] != "T" else 0, x["k"], 0)
        off.append(off[-1] + len(l))
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", m, len(lists), off[-1]))
        f.write(struct.pack(f"<{len(off)}i", *off))
        f.write(body)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("m", type=int, nargs="?", default=48)
    ap.add_argument("--policy", default="both", choices=["shipped", "heft", "both"])
    ap.add_argument("--short", type=int, default=64, help="workgroups that never take a K = 512 chunk")
    ap.add_argument("--nwg", type=int, default=448)
    ap.add_argument("--write", help="file prefix: writes <prefix>_m<m>.bin (the heft lists) for GPMI_FLOW_LISTS")
    a = ap.parse_args()
    m = a.m
    if a.policy in ("shipped", "both"):
        lists = shipped_lists(m, a.nwg)
        r = simulate(m, lists)
        ce = r["col_ends"]
        print(f"shipped: {sum(len(l) for l in lists)} tasks, total {r['total']:.0f} us, chain stalls {r['stall']:.0f} us, workgroup "
              f"utilisation {r['util']:.2f}; panel periods {[round(ce[k + 4] - ce[k]) for k in range(3, m - 4, 4)]}")
    if a.policy in ("heft", "both"):
        lists = heft_lists(m, a.nwg, short_wgs=a.short, verbose=True)
        r = simulate(m, lists)
        ce = r["col_ends"]
        print(f"heft (short-only workgroups {a.short}): total {r['total']:.0f} us, chain stalls {r['stall']:.0f} us, utilisation "
              f"{r['util']:.2f}; panel periods {[round(ce[k + 4] - ce[k]) for k in range(3, m - 4, 4)]}")
        if a.write:
            write_lists(f"{a.write}_m{m}.bin", m, lists)
            print("wrote", f"{a.write}_m{m}.bin")


# ---- a DYNAMIC scheduler (what-if, round 5): no per-workgroup lists - every free workgroup claims the first READY task in
# priority order among the first `window` unclaimed ones of the classes it serves ---------------------------------------
def simulate_dynamic(m, short_wgs=96, window=64, nwg=448, mod=Model, near=4, hop=None, short_takes_q=True, any_takes_short=False,
                     zwindow=None):
    tasks = build_tasks(m, near)
    for x in tasks:
        if x["t"] == "T":
            x["key"] = (4 * x["k"], x["i"], 0, x["s"])
        elif x["t"] == "U":
            x["key"] = (4 * x["k"] + 2, x["i"], x["j"], x["s"])
        else:
            x["key"] = (4 * (OBT * x["k"] + OBT - 1) + 1, x["i"] - (1 << 20) if x.get("quarter") else x["i"], x["j"], x["s"])
    is_short = lambda x: x["t"] != "Z" or (short_takes_q and x.get("quarter"))
    qs = sorted([x for x in tasks if is_short(x)], key=lambda x: x["key"])
    qz = sorted([x for x in tasks if not is_short(x)], key=lambda x: x["key"])
    hop = mod.hop if hop is None else hop
    zwindow = window if zwindow is None else zwindow
    half = nwg // 2
    Lcnt = [0] * m
    F = [[0] * m for _ in range(m)]
    DD = [0]
    t = [0.0]
    ev = []
    seq = [0]

    def at(time, fn, *a):
        seq[0] += 1
        heapq.heappush(ev, (time, seq[0], fn, a))

    def ready(x):
        ty, i, j, k = x["t"], x["i"], x["j"], x["k"]
        if ty == "T":
            return DD[0] >= k + 1 and F[i][k] == 4 * k
        if ty == "U":
            return Lcnt[i] >= 4 * (k + 1) and Lcnt[j] >= 4 * (k + 1) and F[i][j] >= 4 * k
        return Lcnt[i] >= 4 * OBT * (k + 1) and Lcnt[j] >= 4 * OBT * (k + 1) and F[i][j] >= 4 * OBT * k

    free_s = list(range(short_wgs))
    free_z = list(range(short_wgs, nwg))
    zrun = [False] * nwg
    busy_us = [0.0]
    stall = [0.0]

    def take(queue, w):
        for n, x in enumerate(queue[:w]):
            if ready(x):
                del queue[n]
                return x
        return None

    def dispatch():
        while free_s:
            x = take(qs, window)
            if x is None:
                break
            start(free_s.pop(), x)
        while free_z:
            x = take(qs, window) if any_takes_short else None
            if x is None:
                x = take(qz, zwindow)
            if x is None:
                break
            start(free_z.pop(), x)

    def start(b, x):
        if x["t"] == "Z":
            mate = (b + half) % nwg
            if x.get("quarter"):
                dur = mod.cQ2 if zrun[mate] else mod.cQ1
            else:
                dur = mod.cZ2 if zrun[mate] else mod.cZ1
            zrun[b] = True
        else:
            dur = mod.cT
        busy_us[0] += dur
        at(t[0] + hop + dur, done, b, x)

    def done(b, x):
        zrun[b] = False
        (free_s if b < short_wgs else free_z).append(b)
        at(t[0] + mod.vis, publish, x)
        dispatch()

    def publish(x):
        if x["t"] == "T":
            Lcnt[x["i"]] += x["add"]
        else:
            F[x["i"]][x["j"]] += x["add"]
        wake()

    ch = dict(k=0, dend=None, waiting=False, ends=[])

    def launch(k):
        ch["k"] = k
        if k > 0:
            Lcnt[k] = 4 * k
        ch["dend"] = t[0] + mod.cD
        at(ch["dend"] + 0.5, ddone, k)
        ch["waiting"] = True
        wake()

    def ddone(k):
        DD[0] = k + 1
        wake()

    def chain_check():
        if not ch["waiting"]:
            return
        k = ch["k"]
        if k + 1 >= m:
            ch["waiting"] = False
            at(max(ch["dend"], t[0]) + 1.0, finish_launch, k)
            return
        if F[k + 1][k] >= 4 * k and F[k + 1][k + 1] >= 4 * k:
            ch["waiting"] = False
            flags_seen = t[0] + 1.0
            stall[0] += max(0.0, flags_seen + mod.cLoad - ch["dend"])
            at(max(ch["dend"], flags_seen + mod.cLoad) + mod.cTail, finish_launch, k)

    def finish_launch(k):
        ch["ends"].append(t[0])
        if k + 1 < m:
            F[k + 1][k + 1] = 4 * (k + 1)
            at(t[0] + mod.cB, launch, k + 1)

    def wake():
        chain_check()
        dispatch()

    at(0.0, launch, 0)
    while ev:
        t[0], _, fn, a = heapq.heappop(ev)
        fn(*a)
    assert not qs and not qz and len(ch["ends"]) == m
    return dict(total=t[0], stall=stall[0], util=busy_us[0] / (nwg * t[0]), col_ends=ch["ends"])
