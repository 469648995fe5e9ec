"""Micro-benchmark of the fp64 MFMA GEMM (trailing-update shape) through the C-ABI device entry points.
usage: python tools/bench_gemm.py [n] [k] [lower] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inference-tools_amd"))
from inference_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
k = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lower = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
hot = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # 1: operand row pitch 0 (every tile reads the same rows: L1-hot operands)
h = _lib.Handle(0)
ld = n + 32
ldp = k + 32
ldo = 0 if hot else ldp
rng = np.random.default_rng(0)
Cm = rng.standard_normal((n, ld))
P = rng.standard_normal((n, ldp)) * 0.05
dC, dP = C.c_void_p(), C.c_void_p()
h.call("gpmi_dev_alloc", Cm.nbytes, C.byref(dC))
h.call("gpmi_dev_alloc", P.nbytes, C.byref(dP))
h.call("gpmi_dev_upload", dC, Cm.ctypes.data_as(C.c_void_p), Cm.nbytes)
h.call("gpmi_dev_upload", dP, P.ctypes.data_as(C.c_void_p), P.nbytes)
h.call("gpmi_profile_enable", 1)
for _ in range(2):
    h.call("gpmi_dev_gemm_nt", dC, ld, dP, ldo, dP, ldo, n, n, k, lower)
h.call("gpmi_profile_reset")
for _ in range(reps):
    h.call("gpmi_dev_gemm_nt", dC, ld, dP, ldo, dP, ldo, n, n, k, lower)
cnt = C.c_int64()
ms, fl, by = C.c_double(), C.c_double(), C.c_double()
h.call("gpmi_profile_read", _lib.PROF_SYRK, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by))
ghz = C.c_double()
h.call("gpmi_profile_clock", C.byref(ghz))
print(f"clock {ghz.value:.3f} GHz | n={n} k={k} lower={lower} hot={hot}: {ms.value / cnt.value:.3f} ms/launch, {fl.value / ms.value / 1e9:.2f} TFLOP/s "
      f"({100 * fl.value / ms.value / 1e9 / 78.6:.1f}% of 78.6)")
