"""True (in-kernel wall-clock) timeline of the factorisation's panel chain: potrf_diag -> panel TRSM -> inner update
per tile column, for a matrix the size of the tail (N = 5632: in order on the full chip).
usage: GPMI_CHAIN_TRACE=1 python tools/chain_trace.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("GPMI_CHAIN_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inference-tools_amd"))
from inference_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5632
h = _lib.Handle(0)
ld = n + 32
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 64))
Am = B @ B.T + n * np.eye(n)
buf = np.zeros((n, ld))
buf[:, :n] = Am
d = C.c_void_p()
h.call("gpmi_dev_alloc", buf.nbytes, C.byref(d))
info = C.c_int()
for rep in range(3):
    h.call("gpmi_dev_upload", d, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    if rep == 2:
        h.call("gpmi_profile_enable", 2 << _lib.PROF_SYRK)  # stamps only: no event records in the stream
        h.call("gpmi_profile_reset")
    h.call("gpmi_dev_potrf", d, n, ld, C.byref(info))
cnt = C.c_int64()
ms, fl, by = C.c_double(), C.c_double(), C.c_double()
sys.stderr.flush()
h.call("gpmi_profile_read", _lib.PROF_SYRK, C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by))  # prints the [chain] lines
