"""Run INSIDE a child interpreter by tests/test_sanitizers_cpu.py with the AddressSanitizer runtime preloaded and
GPMI_LIB pointing at csrc/build_asan/libgpmi_asan.so (host code of every translation unit compiled with
AddressSanitizer + UndefinedBehaviorSanitizer).  Exercises what the library does without a device; any sanitizer report aborts the
process (non-zero exit), which is what the test looks at.  Prints "sanitize ok <n calls>" at the end."""
import ctypes as C
import os
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
import numpy as np
from inference_amd import _lib

lib = _lib.load()
calls = 0
ERR_ARG = -1

# 1. every entry point that takes a handle, with a NULL handle and zeroed / NULL arguments: GPMI_ERR_ARG, nothing touched
for name, (restype, argtypes) in sorted(_lib.SIGNATURES.items()):
    if not argtypes or argtypes[0] is not _lib._vp or name in ("gpmi_destroy", "gpmi_last_error"):
        continue
    args = [None]
    for t in argtypes[1:]:
        args.append(t() if t in (C.c_int, C.c_int64, C.c_double, C.c_float) else None)
    rc = getattr(lib, name)(*args)
    assert rc == ERR_ARG, (name, rc)
    calls += 1
assert lib.gpmi_destroy(None) == 0
assert isinstance(lib.gpmi_last_error(None), bytes)

# 2. the entry points without a handle
assert lib.gpmi_version() == 100
n = C.c_int(-7)
rc = lib.gpmi_device_count(C.byref(n))
assert rc in (0, -4) and n.value >= 0, (rc, n.value)
assert lib.gpmi_device_count(None) == ERR_ARG
buf = C.create_string_buffer(64)
assert lib.gpmi_device_pci_bus_id(0, buf, 64) in (0, -4, -1, -2)
assert lib.gpmi_device_pci_bus_id(0, None, 0) == ERR_ARG
ctx = C.c_void_p()
rc = lib.gpmi_create(0, C.byref(ctx))
if n.value == 0:
    assert rc == -4 and not ctx.value and b"no HIP device" in lib.gpmi_last_error(None)
else:  # (a GPU box: not where this driver is meant to run, but it must not trip either)
    assert rc == 0 and lib.gpmi_destroy(ctx) == 0
assert lib.gpmi_create(0, None) == ERR_ARG
calls += 8

# 3. the host-only task-list builder of the flag-ordered factorisation, m = 8 .. 64 tile rows, several workgroup counts
ip32 = C.POINTER(C.c_int32)
for m in list(range(8, 65, 4)) + [1, 2, 3, 63]:
    for nwg in (448, 64, 33, 1):
        cnt = C.c_int64(0)
        assert lib.gpmi_flow_task_lists(m, nwg, 0, None, C.byref(cnt)) == 0
        out = np.full((cnt.value, 8), -1, dtype=np.int32)
        assert lib.gpmi_flow_task_lists(m, nwg, cnt.value, out.ctypes.data_as(ip32), C.byref(cnt)) == 0
        # (the shipped default: up to three lists per workgroup, chunks as quarter tasks - type 3; GPMI_FLOW_SPLIT=0: one list)
        nl = nwg if os.environ.get("GPMI_FLOW_SPLIT") == "0" else 3 * nwg
        assert (out[:, 0] >= 0).all() and (out[:, 0] <= 3).all() and (out[:, 6] >= 0).all() and (out[:, 6] < nl).all()
        assert (np.diff(out[:, 6]) >= 0).all()  # list after list
        if cnt.value > 1:
            assert lib.gpmi_flow_task_lists(m, nwg, cnt.value - 1, out.ctypes.data_as(ip32), C.byref(cnt)) == ERR_ARG
        calls += 3
assert lib.gpmi_flow_task_lists(0, 448, 0, None, C.byref(C.c_int64(0))) == ERR_ARG
assert lib.gpmi_flow_task_lists(8, 0, 0, None, C.byref(C.c_int64(0))) == ERR_ARG
assert lib.gpmi_flow_task_lists(8, 448, 0, None, None) == ERR_ARG

# 4. GPMI_FLOW_LISTS override files (the prefix was set by the test before this process started): a well-formed
#    permutation is taken, malformed files (offsets not monotone / beyond the array, truncated, wrong header) are ignored
prefix = os.environ.get("GPMI_FLOW_LISTS")
if prefix:
    def lists(m, nwg):
        cnt = C.c_int64(0)
        assert lib.gpmi_flow_task_lists(m, nwg, 0, None, C.byref(cnt)) == 0
        out = np.zeros((cnt.value, 8), dtype=np.int32)
        assert lib.gpmi_flow_task_lists(m, nwg, cnt.value, out.ctypes.data_as(ip32), C.byref(cnt)) == 0
        return out

    for m in (9, 10, 11, 12, 13):  # files written by the test: see test_sanitizers_cpu.py
        got = lists(m, 64)
        assert got.shape[0] > 0
        calls += 2
print("sanitize ok", calls)
