"""CPU checks of round 3's new host-side logic (no GPU):

* the task lists of the flag-ordered factorisation that libgpmi actually ships (`gpmi_flow_task_lists`, the C++ builder
  of csrc/potrf_flow.hip) replayed with NumPy tile operations: every workgroup's list in order, a task only when the
  flags of its inputs are set, the chain launches in stream order - in a randomised interleaving - must reproduce
  numpy.linalg.cholesky (numpy.linalg.cholesky is what the path replaces: regression.py:241, 537, 555);
* every list is in virtual-time order and a task's inputs precede it (the progress argument of potrf_flow.hip);
* the Python model of the same decomposition (tools/sim/flow_sim.py) agrees with the shipped lists;
* csrc/kmath.h (exp / log1p of the covariance build) against long double on the host (tools/kmath_check.cpp).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shipped_lists(m, nwg):
    from inference_amd import _lib

    lib = _lib.load()
    n = C.c_int64(0)
    assert lib.gpmi_flow_task_lists(m, nwg, 0, None, C.byref(n)) == 0
    out = np.zeros((n.value, 8), dtype=np.int32)
    assert lib.gpmi_flow_task_lists(m, nwg, n.value, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)) == 0
    return out


def replay(m, nwg, nb=8, seed=0):
    """Replays the shipped lists; returns (relative error of the factor, number of tasks)."""
    tasks = shipped_lists(m, nwg)
    # (column 6: the LIST; with GPMI_FLOW_SPLIT a workgroup has two, list nwg + b being the second of workgroup b - replayed
    # as independent in-order lists, which admits every interleaving the device can produce and more)
    nwg = max(nwg, int(tasks[:, 6].max()) + 1 if len(tasks) else nwg)
    lists = [tasks[tasks[:, 6] == w] for w in range(nwg)]
    assert sum(len(l) for l in lists) == len(tasks)
    rng = np.random.default_rng(seed)
    n = m * nb
    X = rng.normal(size=(n, n))
    A0 = X @ X.T + n * np.eye(n)
    A = A0.copy()
    tile = lambda i, j: A[i * nb:(i + 1) * nb, j * nb:(j + 1) * nb]
    invD = {}
    Ddone, Lcnt, F = 0, [0] * m, np.zeros((m, m), dtype=int)
    pos = [0] * nwg
    chain = [("D", 0)] + [x for k in range(m - 1) for x in (("Tc", k), ("Uc", k), ("D", k + 1))]
    ci = 0
    hb, qb = nb // 2, nb // 4
    left = len(tasks)
    guard = 0
    while ci < len(chain) or left:
        guard += 1
        assert guard < 200 * (len(tasks) + len(chain)), "no progress: a task waits for something that never comes"
        if ci < len(chain) and rng.random() < 0.3:
            ty, k = chain[ci]
            if ty == "D":
                assert F[k, k] == 4 * k
                Lk = np.linalg.cholesky(tile(k, k))
                tile(k, k)[:] = Lk
                invD[k] = np.linalg.inv(Lk)
                ci += 1
                # (Ddone is published by the NEXT chain launch's prologue; the last column has no consumers)
            elif ty == "Tc" and F[k + 1, k] >= 4 * k:
                Ddone = k + 1
                tile(k + 1, k)[:] = tile(k + 1, k) @ invD[k].T
                ci += 1
            elif ty == "Uc" and F[k + 1, k + 1] >= 4 * k:
                Lcnt[k + 1] = 4 * (k + 1)
                tile(k + 1, k + 1)[:] -= tile(k + 1, k) @ tile(k + 1, k).T
                F[k + 1, k + 1] = 4 * (k + 1)  # not a flag anybody reads: D(k + 1) follows in stream order
                ci += 1
        for w in rng.permutation(nwg)[:8]:
            if pos[w] >= len(lists[w]):
                continue
            ty, i, j, k, s, fadd = (int(v) for v in lists[w][pos[w]][:6])
            if ty == 0:
                if not (Ddone >= k + 1 and F[i, k] == 4 * k):
                    continue
                tile(i, k)[s * qb:(s + 1) * qb] = tile(i, k)[s * qb:(s + 1) * qb] @ invD[k].T
                Lcnt[i] += fadd
            elif ty == 1:
                if not (Lcnt[i] >= 4 * (k + 1) and Lcnt[j] >= 4 * (k + 1) and F[i, j] >= 4 * k):
                    continue
                r, c = slice((s >> 1) * hb, (s >> 1) * hb + hb), slice((s & 1) * hb, (s & 1) * hb + hb)
                tile(i, j)[r, c] -= tile(i, k)[r] @ tile(j, k)[c].T
                F[i, j] += fadd
            elif ty == 3:  # a quarter of chunk k (round 6): sub-tile s takes the panel's four columns
                if not (Lcnt[i] >= 16 * (k + 1) and Lcnt[j] >= 16 * (k + 1) and F[i, j] >= 16 * k):
                    continue
                r, c = slice((s >> 1) * hb, (s >> 1) * hb + hb), slice((s & 1) * hb, (s & 1) * hb + hb)
                for col in range(4 * k, 4 * k + 4):
                    tile(i, j)[r, c] -= tile(i, col)[r] @ tile(j, col)[c].T
                F[i, j] += fadd
            else:
                if not (Lcnt[i] >= 16 * (k + 1) and Lcnt[j] >= 16 * (k + 1) and F[i, j] == 16 * k):
                    continue
                for col in range(4 * k, 4 * k + 4):
                    tile(i, j)[:] -= tile(i, col) @ tile(j, col).T
                F[i, j] += fadd
            pos[w] += 1
            left -= 1
    Lref = np.linalg.cholesky(A0)
    # a diagonal tile keeps only its lower triangle current (the upper-right sub-tile is skipped): compare lower parts
    return float(np.abs(np.tril(A) - Lref).max() / np.abs(Lref).max()), len(tasks)


@pytest.mark.parametrize("m,nwg", [(1, 8), (2, 8), (5, 16), (9, 448), (14, 72), (23, 448)])
def test_shipped_task_lists_factor_a_matrix(m, nwg):
    err, ntasks = replay(m, nwg, seed=m)
    assert err < 1e-13, err
    assert (ntasks == 0) == (m <= 2)


@pytest.mark.parametrize("env", [{"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "0", "GPMI_FLOW_URGENT": "0"},
                                 {"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "2", "GPMI_FLOW_URGENT": "0"},
                                 {"GPMI_FLOW_SPLIT": "0", "GPMI_FLOW_QUARTER": "0"},
                                 {"GPMI_FLOW_SPLIT": "0", "GPMI_FLOW_QUARTER": "99999"},  # the schedule of rounds 3-5
                                 {"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "99999"},
                                 {"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "0", "GPMI_FLOW_URGENT": "1"},
                                 {"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "0", "GPMI_FLOW_URGENT": "3"},
                                 {"GPMI_FLOW_SPLIT": "1", "GPMI_FLOW_QUARTER": "3", "GPMI_FLOW_URGENT": "2"}])
def test_split_lists_and_quarter_chunks_factor_a_matrix(env):
    """(round 6) The schedule variants of csrc/potrf_flow.hip - the K = 512 chunks in second lists (GPMI_FLOW_SPLIT) and / or as
    four 64 x 64 sub-tile tasks (GPMI_FLOW_QUARTER) - replayed the same way, in a child process (the library reads the
    switches once)."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import test_flow_cpu as t\n"
            "for m, nwg in ((9, 448), (14, 72), (23, 448)):\n"
            "    err, n = t.replay(m, nwg, seed=m)\n"
            "    assert err < 1e-13, (m, nwg, err)\n"
            "tl = t.shipped_lists(23, 448)\n"
            "print('ok', int((tl[:, 0] == 3).sum()), int((tl[:, 0] == 2).sum()), int(tl[:, 6].max()))\n"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "inference-tools_amd")))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr[-3000:]
    ok, nzs, nz, maxlist = res.stdout.split()
    if env.get("GPMI_FLOW_QUARTER") == "0":
        assert int(nzs) > 0 and int(nz) == 0
    if env.get("GPMI_FLOW_SPLIT") == "1":
        assert int(maxlist) >= 448  # second lists exist
    else:
        assert int(maxlist) < 448


def lazy_panels(i, j, near=4):
    """csrc/potrf_flow.hip: flow_lazy_panels - the outer panels a tile takes as K = 512 chunks"""
    P = j // 4
    return max(P - 1 if i < 4 * P + 4 + near else P, 0)


def test_lists_are_in_dependency_order_with_inputs_first():
    """The shipped default (round 6): workgroup b holds list b (panel TRSMs and one-column updates, virtual-time order),
    list nwg + b (the last three chunks of a tile, as quarter tasks, ordered by the panel that needs them: right in front
    of the tile's first single-column task) and list 2 nwg + b (the older chunks, by outer panel q).  Every list is sorted
    by a key under which each task follows all its inputs - the progress argument of potrf_flow.hip."""
    m, nwg = 30, 448
    t = shipped_lists(m, nwg)
    assert int(t[:, 6].max()) < 3 * nwg and not (t[:, 0] == 2).any()  # every chunk is a quarter task (type 3)
    ty, ti, tj, tk = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    lazy = np.array([lazy_panels(int(i), int(j)) for i, j in zip(ti, tj)])
    urgent = (ty == 3) & (lazy - tk <= 3)
    key = np.where(ty == 0, 4 * tk, np.where(ty == 1, 4 * tk + 2, np.where(urgent, 16 * lazy - 1, 4 * (4 * tk + 3) + 1)))
    # the key of a task exceeds the keys of its inputs: T(i, 4q + 3) -> chunk q -> the tile's next chunk -> its first
    # single-column task (U at 16 lazy + 2, or the panel TRSM T(i, 4 lazy) at 16 lazy)
    ch = ty == 3
    assert np.all(key[ch] > 4 * (4 * tk[ch] + 3)) and np.all(key[ch] < 16 * lazy[ch])
    for w in range(3 * nwg):
        sel = t[:, 6] == w
        assert np.all(np.diff(key[sel]) >= 0), w
        if w < nwg:
            assert not ch[sel].any()           # first lists: short tasks only
        elif w < 2 * nwg:
            assert urgent[sel].all()           # second lists: the chunks a tile's own panel waits for
        else:
            assert (ch[sel] & ~urgent[sel]).all()
        if sel.any() and w >= nwg:             # the chunks of one tile in ascending q within a list
            for (i, j) in {(int(a), int(b)) for a, b in zip(ti[sel], tj[sel])}:
                q = tk[sel & (ti == i) & (tj == j)]
                assert np.all(np.diff(q) >= 0)
    # every (tile, column) is applied exactly once, the lazy chunks before the single columns
    applied = {}
    for ty_, i, j, k, s, fadd, w, _ in t:
        if ty_ == 1:
            applied.setdefault((i, j), []).append((4 * k + 2, fadd))
        elif ty_ in (2, 3):
            applied.setdefault((i, j), []).append((4 * (4 * k + 3) + 1, fadd))
    for (i, j), lst in applied.items():
        # an off-diagonal tile takes its j columns from the lists; a diagonal tile's last column comes from the chain
        assert sum(f for _, f in lst) == (4 * j if i > j else 4 * (j - 1)), (i, j)
    assert set(applied) == {(i, j) for i in range(m) for j in range(1, i + 1) if (i, j) != (1, 1)}
    # rows next to the chain go to the first 32 workgroups only, which take no chunks
    near = t[(t[:, 0] < 2) & (t[:, 1] - t[:, 3] <= 3)]
    assert near[:, 6].max() < 32 and (t[ch][:, 6] % nwg).min() >= 32


def test_python_model_agrees_with_the_shipped_lists():
    sys.path.insert(0, os.path.join(ROOT, "tools", "sim"))
    import flow_sim

    m = 17
    H, Z, _ = flow_sim.build(m, near=4)
    model = {("T", x["i"], 0, x["k"], x["s"]) for x in H if x["t"] == "T"}
    model |= {("U", x["i"], x["j"], x["k"], x["s"]) for x in H if x["t"] == "U" and not (x["i"] == x["j"] and x["s"] == 1)}
    model |= {("Z", x["i"], x["j"], x["q"], 0) for x in Z}
    names = "TUZZ"  # (type 3: a quarter of a chunk - the model counts whole chunks)
    shipped = {(names[ty], i, j if ty else 0, k, s if ty < 2 else 0) for ty, i, j, k, s, *_ in shipped_lists(m, 448)}
    assert model == shipped
    assert flow_sim.check(6, seed=3) < 1e-13


def test_kmath_accuracy_on_the_host(tmp_path):
    exe = str(tmp_path / "kmath_check")
    src = os.path.join(ROOT, "tools", "kmath_check.cpp")
    inc = os.path.join(ROOT, "inference-tools_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-I", inc, src, "-o", exe], check=True)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr  # exp <= 1 ulp, log1p <= 1.5 ulp, exact end points
