"""256 x 128 tile experiment (GPMI_GEMM_256=1): correctness against numpy at n = 3584, then timing at the trailing-update shapes."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inference-tools_amd"))
from inference_amd import _lib

h = _lib.Handle(0)
rng = np.random.default_rng(0)


def run(n, k, lower, reps, check):
    ld, ldp = n + 32, k + 32
    Cm = rng.standard_normal((n, ld))
    P = rng.standard_normal((n, ldp)) * 0.05
    dC, dP = C.c_void_p(), C.c_void_p()
    h.call("gpmi_dev_alloc", Cm.nbytes, C.byref(dC))
    h.call("gpmi_dev_alloc", P.nbytes, C.byref(dP))
    h.call("gpmi_dev_upload", dC, Cm.ctypes.data_as(C.c_void_p), Cm.nbytes)
    h.call("gpmi_dev_upload", dP, P.ctypes.data_as(C.c_void_p), P.nbytes)
    if check:
        h.call("gpmi_dev_gemm_nt", dC, ld, dP, ldp, dP, ldp, n, n, k, lower)
        out = np.empty_like(Cm)
        h.call("gpmi_dev_download", out.ctypes.data_as(C.c_void_p), dC, Cm.nbytes)
        ref = Cm[:, :n] - P[:, :k] @ P[:, :k].T
        got = out[:, :n]
        if lower:
            m = np.tril(np.ones((n, n), bool))
            err = np.abs(got - ref)[m].max()
            up = np.abs(got - Cm[:, :n])[~m & ~np.kron(np.eye(n // 128, dtype=bool), np.ones((128, 128), bool))].max()
            print(f"n={n} k={k} lower: max err {err:.3e}; strictly-upper tiles changed by {up:.3e}")
        else:
            print(f"n={n} k={k} rect: max err {np.abs(got - ref).max():.3e}")
    else:
        for _ in range(2):
            h.call("gpmi_dev_gemm_nt", dC, ld, dP, ldp, dP, ldp, n, n, k, lower)
        t0 = time.perf_counter()
        for _ in range(reps):
            h.call("gpmi_dev_gemm_nt", dC, ld, dP, ldp, dP, ldp, n, n, k, lower)
        dt = (time.perf_counter() - t0) / reps
        nt = n // 128
        tiles = nt * (nt + 1) / 2 if lower else nt * nt
        fl = tiles * 2.0 * 128 * 128 * k
        print(f"n={n} k={k} lower={lower}: {dt * 1e3:.3f} ms, {fl / dt / 1e12:.2f} TFLOP/s")
    h.call("gpmi_dev_free", dC)
    h.call("gpmi_dev_free", dP)


print("GPMI_GEMM_256 =", os.environ.get("GPMI_GEMM_256"))
run(3584, 256, 1, 1, True)
run(3584, 256, 0, 1, True)
for n in (15872, 8192):
    for k in (512, 1024):
        for lower in (1, 0):
            run(n, k, lower, 6, False)
