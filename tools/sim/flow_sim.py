"""Discrete-event model of the flag-ordered ("dataflow") tile Cholesky of the factorisation's tail
(csrc/potrf_flow.hip).  Used to choose the task decomposition and the two queue orders before any device code
was written, and kept as the executable statement of the schedule: `--check` replays the same task list with NumPy
tile operations in a randomised admissible order and compares with numpy.linalg.cholesky.

Model: one chain stream (potrf_diag D(k), Tc(k) = panel TRSM of tile (k+1, k), Uc(k) = update of tile (k+1, k+1) by
column k: ordinary launches, `boundary` us apart) and S persistent workgroups that pull tasks from two in-order
queues, H (single-column tasks, priority) and Z (K = 512 chunks), a task being claimed only when its flags are set.
"""
import argparse
import heapq
import math


def build(m, near=4, alpha=0.25, zorder="tau", OB=4, d1=8, classes=None):
    """Task lists for a tail of m tile rows.  Returns (H, Z, lazy_end); every task carries its virtual time `tau`
    (every dependency of a task has a smaller tau, so in-order queues sorted by tau cannot deadlock) and `near`
    (row within d1 of the column being applied: the priority queue)."""
    H, Z = [], []

    def lazy_end(i, j):
        P = j // OB
        e = OB * (P - 1) if i < OB * P + OB + near else OB * P
        return max(e, 0)

    for k in range(m):
        for i in range(k + 2, m):
            key = k + alpha * (i - k - 1)
            nr = (i - k) <= d1
            cls = 0
            if classes:
                while cls < len(classes) and (i - k) > classes[cls]:
                    cls += 1
            for s in range(4):
                H.append(dict(t="T", i=i, k=k, s=s, key=(key, i, 0, 0), near=nr, cls=cls))
            # singles of column k on row i: tiles (i, j), k < j <= i, with lazy_end(i, j) <= k
            for j in range(k + 1, i + 1):
                if lazy_end(i, j) <= k:
                    for s in range(4):
                        H.append(dict(t="U", i=i, j=j, k=k, s=s, key=(key + 0.5, i, 1, j), near=nr, cls=cls))
    H.sort(key=lambda t: t["key"])
    for i in range(m):
        for j in range(i + 1):
            for q in range(lazy_end(i, j) // OB):
                Z.append(dict(t="Z", i=i, j=j, q=q))
    if zorder == "tau":
        Z.sort(key=lambda t: ((OB * t["q"] + OB - 1) * (1 - alpha) + alpha * (t["i"] - 1) + 0.25, t["i"], t["j"]))
    elif zorder == "qij":
        Z.sort(key=lambda t: (t["q"], t["i"], t["j"]))
    elif zorder == "qji":
        Z.sort(key=lambda t: (t["q"], t["j"], t["i"]))
    return H, Z, lazy_end


def simulate(m, S=448, near=4, alpha=0.25, zorder="tau", d1=8, reserve=64, reserve2=0, W=1, WZ=1, cD=29.5, cB=2.0, cTc=5.0, cUc=5.0, cT=9.0, cU=9.0,
             cZ=68.0, vis=1.5, verbose=False):
    Hall, Z, lazy_end = build(m, near, alpha, zorder, d1=d1)
    H = [x for x in Hall if x["near"]]
    H2 = [x for x in Hall if not x["near"]]
    OB = 4
    # flags
    Ddone = 0
    Lcnt = [0] * m          # T slabs done per row (4 per column); chain rows set directly
    F = [[0] * m for _ in range(m)]  # sub-updates applied (4 per column)
    t = 0.0
    ev = []  # (time, seq, fn)
    seq = [0]

    def at(time, fn):
        seq[0] += 1
        heapq.heappush(ev, (time, seq[0], fn))

    # ---- chain ----
    chain_wait = [0.0]
    st = dict(k=0, phase="D", busy=False, finished=False, tend=0.0)

    def chain_try():
        nonlocal Ddone
        if st["busy"] or st["finished"]:
            return
        k = st["k"]
        ph = st["phase"]
        if ph == "D":
            ok = True  # stream order: after Uc(k - 1)
            dur = cD
        elif ph == "Tc":
            ok = F[k + 1][k] == 4 * k
            dur = cTc
        else:
            ok = F[k + 1][k + 1] == 4 * k
            dur = cUc
        if not ok:
            st["stall_from"] = st.get("stall_from", t)
            return
        if "stall_from" in st:
            chain_wait[0] += t - st.pop("stall_from")
        st["busy"] = True

        def done():
            nonlocal Ddone
            st["busy"] = False
            k = st["k"]
            if ph == "D":
                # published by the next chain kernel's prologue (boundary later)
                def pub():
                    nonlocal Ddone
                    Ddone = k + 1
                    wake()
                at(t + cB + vis, pub)
                if k + 1 < m:
                    st["phase"] = "Tc"
                else:
                    st["finished"] = True
                    st["tend"] = t
            elif ph == "Tc":
                def pub():
                    Lcnt[k + 1] = 4 * (k + 1)
                    wake()
                at(t + cB + vis, pub)
                st["phase"] = "Uc"
            else:
                F[k + 1][k + 1] = 4 * (k + 1)
                st["phase"] = "D"
                st["k"] = k + 1
            at(t + cB, chain_try)
        at(t + dur, done)

    # ---- bulk ----
    hi = [0]
    h2 = [0]
    zi = [0]
    # slots: class 0 takes only near H tasks, class 1 near + far H tasks, class 2 anything
    free = [reserve, reserve2, S - reserve - reserve2]
    busy_time = [0.0]

    def ready(task):
        ty = task["t"]
        if ty == "T":
            return Ddone >= task["k"] + 1 and F[task["i"]][task["k"]] == 4 * task["k"]
        if ty == "U":
            i, j, k = task["i"], task["j"], task["k"]
            return Lcnt[i] >= 4 * (k + 1) and Lcnt[j] >= 4 * (k + 1) and 4 * k <= F[i][j] < 4 * k + 4
        i, j, q = task["i"], task["j"], task["q"]
        return Lcnt[i] >= 16 * (q + 1) and Lcnt[j] >= 16 * (q + 1) and F[i][j] == 16 * q

    def finish(task):
        def fn():
            ty = task["t"]
            if ty == "T":
                Lcnt[task["i"]] += 1
            elif ty == "U":
                F[task["i"]][task["j"]] += 1
            else:
                F[task["i"]][task["j"]] += 16
            free[task["slot"]] += 1
            wake()
        return fn

    claimedH = [False] * len(H)
    claimedH2 = [False] * len(H2)
    claimedZ = [False] * len(Z)

    def scan(q, claimed, head, w):
        """first ready task among the first w unclaimed ones (a wave examines w tasks at once)"""
        while head[0] < len(q) and claimed[head[0]]:
            head[0] += 1
        n = 0
        x = head[0]
        while x < len(q) and n < w:
            if not claimed[x]:
                n += 1
                if ready(q[x]):
                    return x
            x += 1
        return -1

    def bulk_try():
        while True:
            task = None
            if sum(free) > 0:
                x = scan(H, claimedH, hi, W)
                if x >= 0:
                    task = H[x]
                    claimedH[x] = True
                    dur = cT if task["t"] == "T" else cU
                    slot = 0 if free[0] > 0 else (1 if free[1] > 0 else 2)
            if task is None and free[1] + free[2] > 0:
                x = scan(H2, claimedH2, h2, W)
                if x >= 0:
                    task = H2[x]
                    claimedH2[x] = True
                    dur = cT if task["t"] == "T" else cU
                    slot = 1 if free[1] > 0 else 2
            if task is None and free[2] > 0:
                x = scan(Z, claimedZ, zi, WZ)
                if x >= 0:
                    task = Z[x]
                    claimedZ[x] = True
                    dur = cZ
                    slot = 2
            if task is None:
                return
            free[slot] -= 1
            task["slot"] = slot
            busy_time[0] += dur
            at(t + dur + vis, finish(task))

    def wake():
        chain_try()
        bulk_try()

    at(0.0, wake)
    while ev:
        t, _, fn = heapq.heappop(ev)
        fn()
    assert st["finished"] and all(claimedH) and all(claimedH2) and all(claimedZ), (st, hi[0], len(H), h2[0], len(H2), zi[0], len(Z))
    return dict(total_us=t, chain_end=st["tend"], chain_stall=chain_wait[0], nH=len(Hall), nZ=len(Z),
                util=busy_time[0] / (S * t))


def check(m, nb=8, seed=0, near=4):
    """Replay the decomposition with NumPy tile operations in a random admissible order."""
    import numpy as np

    rng = np.random.default_rng(seed)
    n = m * nb
    X = rng.normal(size=(n, n))
    A0 = X @ X.T + n * np.eye(n)
    A = A0.copy()
    H, Z, lazy_end = build(m, near)
    tile = lambda i, j: A[i * nb:(i + 1) * nb, j * nb:(j + 1) * nb]
    invD = {}
    Ddone = 0
    Lcnt = [0] * m
    F = [[0] * m for _ in range(m)]
    chain = []
    for k in range(m):
        chain.append(("D", k))
        if k + 1 < m:
            chain += [("Tc", k), ("Uc", k)]
    ci, pend = 0, H + Z
    done = [False] * len(pend)
    ndone = 0
    hb = nb // 2

    def sub(s):
        return slice((s >> 1) * hb, (s >> 1) * hb + hb), slice((s & 1) * hb, (s & 1) * hb + hb)

    while ci < len(chain) or ndone < len(pend):
        progressed = False
        # chain step (random chance to lag)
        if ci < len(chain) and rng.random() < 0.5:
            ty, k = chain[ci]
            ok = True
            if ty == "Tc":
                ok = F[k + 1][k] == 4 * k
            elif ty == "Uc":
                ok = F[k + 1][k + 1] == 4 * k
            if ok:
                if ty == "D":
                    assert F[k][k] == 4 * k
                    L = np.linalg.cholesky(tile(k, k))
                    tile(k, k)[:] = L
                    invD[k] = np.linalg.inv(L)
                    Ddone = k + 1
                elif ty == "Tc":
                    tile(k + 1, k)[:] = tile(k + 1, k) @ invD[k].T
                    Lcnt[k + 1] = 4 * (k + 1)
                else:
                    tile(k + 1, k + 1)[:] -= tile(k + 1, k) @ tile(k + 1, k).T
                    F[k + 1][k + 1] = 4 * (k + 1)
                ci += 1
                progressed = True
        cand = [x for x in rng.permutation(len(pend))[:64] if not done[x]]
        for x in cand:
            task = pend[x]
            ty = task["t"]
            if ty == "T":
                i, k, s = task["i"], task["k"], task["s"]
                if Ddone >= k + 1 and F[i][k] == 4 * k:
                    q = nb // 4
                    tile(i, k)[s * q:(s + 1) * q] = tile(i, k)[s * q:(s + 1) * q] @ invD[k].T
                    Lcnt[i] += 1
                else:
                    continue
            elif ty == "U":
                i, j, k, s = task["i"], task["j"], task["k"], task["s"]
                if Lcnt[i] >= 4 * (k + 1) and Lcnt[j] >= 4 * (k + 1) and 4 * k <= F[i][j] < 4 * k + 4:
                    r, c = sub(s)
                    tile(i, j)[r, c] -= tile(i, k)[r] @ tile(j, k)[c].T
                    F[i][j] += 1
                else:
                    continue
            else:
                i, j, q = task["i"], task["j"], task["q"]
                if Lcnt[i] >= 16 * (q + 1) and Lcnt[j] >= 16 * (q + 1) and F[i][j] == 16 * q:
                    for k in range(4 * q, 4 * q + 4):
                        tile(i, j)[:] -= tile(i, k) @ tile(j, k).T
                    F[i][j] += 16
                else:
                    continue
            done[x] = True
            ndone += 1
            progressed = True
        if not progressed and ci >= len(chain):
            # nothing sampled was admissible: scan all
            left = [x for x in range(len(pend)) if not done[x]]
            assert left, "stuck"
    Lref = np.linalg.cholesky(A0)
    err = np.abs(np.tril(A) - Lref).max() / np.abs(Lref).max()
    for i in range(m):
        for j in range(i + 1):
            assert F[i][j] == 4 * j, (i, j, F[i][j])
    return err


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    if a.check:
        for m in (1, 2, 3, 5, 9, 14):
            print(m, check(m, seed=m))
    else:
        for near in (0, 4):
            for res in ((0, 0), (32, 0)):
                for d1 in ((1000, 1, 1), (1000, 64, 1), (1000, 64, 64), (1000, 256, 64), (8, 64, 64)):
                    alpha = 0.0
                    r = simulate(a.m, near=near, alpha=alpha, d1=d1[0], reserve=res[0], reserve2=res[1], W=d1[1], WZ=d1[2])
                    print(f"m={a.m} near={near} res={res} d1={d1}: total {r['total_us']:.0f} us, chain stall "
                          f"{r['chain_stall']:.0f}, util {r['util']:.2f}, nH {r['nH']} nZ {r['nZ']}")
