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


def test_lists_are_in_virtual_time_order_with_inputs_first():
    m, nwg = 30, 448
    t = shipped_lists(m, nwg)
    vt = np.where(t[:, 0] == 0, 4 * t[:, 3], np.where(t[:, 0] == 1, 4 * t[:, 3] + 2, 4 * (4 * t[:, 3] + 3) + 1))
    for w in range(nwg):
        v = vt[t[:, 6] == w]
        assert np.all(np.diff(v) >= 0), w
    # every (tile, column) is applied exactly once, the lazy chunks before the single columns
    applied = {}
    for ty, i, j, k, s, fadd, w, _ in t:
        if ty == 1:
            applied.setdefault((i, j), []).append((4 * k + 2, fadd))
        elif ty == 2:
            applied.setdefault((i, j), []).append((4 * (4 * k + 3) + 1, fadd))
    for (i, j), lst in applied.items():
        # an off-diagonal tile takes its j columns from the lists; a diagonal tile's last column comes from the chain
        assert sum(f for _, f in lst) == (4 * j if i > j else 4 * (j - 1)), (i, j)
    assert set(applied) == {(i, j) for i in range(m) for j in range(1, i + 1) if (i, j) != (1, 1)}
    # rows next to the chain go to the first 32 workgroups only
    near = t[(t[:, 0] < 2) & (t[:, 1] - t[:, 3] <= 3)]
    assert near[:, 6].max() < 32 and t[t[:, 0] == 2][:, 6].min() >= 32


def test_python_model_agrees_with_the_shipped_lists():
    sys.path.insert(0, os.path.join(ROOT, "tools", "sim"))
    import flow_sim

    m = 17
    H, Z, _ = flow_sim.build(m, near=4)
    model = {("T", x["i"], 0, x["k"], x["s"]) for x in H if x["t"] == "T"}
    model |= {("U", x["i"], x["j"], x["k"], x["s"]) for x in H if x["t"] == "U" and not (x["i"] == x["j"] and x["s"] == 1)}
    model |= {("Z", x["i"], x["j"], x["q"], 0) for x in Z}
    names = "TUZ"
    shipped = {(names[ty], i, j if ty else 0, k, s) for ty, i, j, k, s, *_ in shipped_lists(m, 448)}
    assert model == shipped
    assert flow_sim.check(6, seed=3) < 1e-13


def test_kmath_accuracy_on_the_host(tmp_path):
    exe = str(tmp_path / "kmath_check")
    src = os.path.join(ROOT, "tools", "kmath_check.cpp")
    inc = os.path.join(ROOT, "inference-tools_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-I", inc, src, "-o", exe], check=True)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr  # exp <= 1 ulp, log1p <= 1.5 ulp, exact end points
