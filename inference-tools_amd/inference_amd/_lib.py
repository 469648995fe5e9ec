"""
ctypes binding of libgpmi.so (C-ABI declared in include/gpmi.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C
inference-tools_amd/csrc` into `inference_amd/lib/libgpmi.so`.  Nothing here
falls back to a CPU implementation: a missing library or device raises
`GpmiUnavailable`.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPMI_LIB: another build of the same library (A/B timing of kernel variants: tools/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("GPMI_LIB") or os.path.join(_HERE, "lib", "libgpmi.so")

KERNEL_SE = 0
KERNEL_RQ = 1
PROF_KBUILD, PROF_SYRK, PROF_PANEL, PROF_SOLVE, PROF_SYRK_REST, PROF_TRSM, PROF_SYRK_SLICE, PROF_FLOW = 0, 1, 2, 3, 4, 5, 6, 7
OPT_LOCKSTEP_ALWAYS, OPT_RESERVE_POINTS, OPT_NO_FLOW = 1, 2, 3
_TRACE_MS = float(os.environ["GPMI_TRACE_CALLS"]) if os.environ.get("GPMI_TRACE_CALLS") else None
ERR_INTERNAL = -5  # GPMI_ERR_INTERNAL


class GpmiUnavailable(RuntimeError):
    """libgpmi.so (the HIP extension) or an MI355X device is missing."""


class GpmiError(RuntimeError):
    """A gpmi_* call returned a non-zero status."""


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/gpmi.h one to one
SIGNATURES = {
    "gpmi_version": (C.c_int, []),
    "gpmi_device_count": (C.c_int, [_ip]),
    "gpmi_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "gpmi_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gpmi_destroy": (C.c_int, [_vp]),
    "gpmi_last_error": (C.c_char_p, [_vp]),
    "gpmi_sync": (C.c_int, [_vp]),
    "gpmi_set_data": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _i64, _i64]),
    "gpmi_fit": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_lml": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _ip]),
    "gpmi_lml_batch": (C.c_int, [_vp, C.c_int, _i64, _dp, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_lml_batch_submit": (C.c_int, [_vp, C.c_int, _i64, _dp, C.c_int, _dp, _dp, _dp, C.c_int]),
    "gpmi_lml_batch_wait": (C.c_int, [_vp, C.c_int, _dp, _ip]),
    "gpmi_set_streams": (C.c_int, [_vp, C.c_int]),
    "gpmi_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gpmi_lml_grad": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_flow_task_lists": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gpmi_lml_grad_batch": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_lml_grad_batch_noise": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_predict": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_posterior": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_spatial_derivatives": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_gradient": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_covariance": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, C.c_int, _dp]),
    "gpmi_cross_covariance": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _i64, _dp]),
    "gpmi_get_K": (C.c_int, [_vp, _dp]),
    "gpmi_get_L": (C.c_int, [_vp, _dp]),
    "gpmi_loo_diag": (C.c_int, [_vp, _dp]),
    "gpmi_loo_terms": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_loo_grad": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_loo_grad_batch": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_loo_grad_batch_noise": (C.c_int, [_vp, C.c_int, C.c_int64, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_fit_mix": (C.c_int, [_vp, C.c_int, _ip, _dp, _ip, _dp, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_lml_mix": (C.c_int, [_vp, C.c_int, _ip, _dp, _ip, _dp, C.c_double, _dp, _dp, _ip]),
    "gpmi_lml_grad_mix": (C.c_int, [_vp, C.c_int, _ip, _dp, _ip, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_lml_grad_batch_mix": (C.c_int, [_vp, C.c_int, _ip, _i64, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_loo_grad_batch_mix": (C.c_int, [_vp, C.c_int, _ip, _i64, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_loo_terms_mix": (C.c_int, [_vp, C.c_int, _ip, _dp, _ip, _dp, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_predict_mix": (C.c_int, [_vp, _dp, _i64, _dp, _dp, _dp]),
    "gpmi_posterior_mix": (C.c_int, [_vp, _dp, _i64, _dp, _dp, _dp]),
    "gpmi_set_noise": (C.c_int, [_vp, _dp]),
    "gpmi_lml_grad_qdiag": (C.c_int, [_vp, _dp]),
    "gpmi_linv_set": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_linv_lml": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _ip]),
    "gpmi_linv_lml_grad": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_linv_posterior": (C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_linv_lml_dense": (C.c_int, [_vp, _dp, _dp, _dp, _ip]),
    "gpmi_linv_lml_grad_dense": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_linv_posterior_dense": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_append_point": (C.c_int, [_vp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _ip]),
    "gpmi_capacity": (C.c_int, [_vp, C.POINTER(_i64)]),
    "gpmi_fit_dense": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_lml_dense": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_loo_dense": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpmi_predict_dense": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_solve_rows": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpmi_prepare_gradient": (C.c_int, [_vp, C.c_int]),
    "gpmi_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gpmi_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "gpmi_comm_allgather": (C.c_int, [_vp, _dp, _dp, _i64]),
    "gpmi_comm_broadcast": (C.c_int, [_vp, _dp, _i64, C.c_int]),
    "gpmi_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gpmi_comm_destroy": (C.c_int, [_vp]),
    "gpmi_timer_start": (C.c_int, [_vp]),
    "gpmi_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "gpmi_profile_enable": (C.c_int, [_vp, C.c_int]),
    "gpmi_profile_read": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), _dp, _dp, _dp]),
    "gpmi_profile_reset": (C.c_int, [_vp]),
    "gpmi_profile_clock": (C.c_int, [_vp, _dp]),
    "gpmi_dev_alloc": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "gpmi_dev_free": (C.c_int, [_vp, _vp]),
    "gpmi_dev_upload": (C.c_int, [_vp, _vp, _vp, _i64]),
    "gpmi_dev_download": (C.c_int, [_vp, _vp, _vp, _i64]),
    "gpmi_dev_potrf": (C.c_int, [_vp, _vp, _i64, _i64, _ip]),
    "gpmi_dev_gemm_nt": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, C.c_int]),
}

_lib = None


def load():
    """Load libgpmi.so and declare every prototype.  Raises GpmiUnavailable if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpmiUnavailable(
                f"{LIB_PATH} not found - build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C inference-tools_amd/csrc)"
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        atexit.register(_close_all_handles)
    return _lib


def device_count() -> int:
    """Number of HIP devices visible to this process (0 without a GPU)."""
    cnt = C.c_int(0)
    load().gpmi_device_count(C.byref(cnt))
    return cnt.value


def device_identity(device: int) -> str:
    """Physical identity of visible device `device`: its PCI bus id, which - unlike the index - does not depend on the
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES mask of the process ("" if the runtime cannot tell)."""
    buf = C.create_string_buffer(64)
    rc = load().gpmi_device_pci_bus_id(int(device), buf, 64)
    return buf.value.decode() if rc == 0 else ""


def dptr(a):
    """double* view of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_live_handles = weakref.WeakSet()


def _close_all_handles():
    """Destroy every live context while the HIP runtime is still up.  Registered with atexit after the library is
    loaded, so it runs BEFORE the runtime's own exit handlers: a context that is only collected during interpreter
    shutdown (e.g. one kept alive by the traceback of a failed test) would otherwise synchronise streams of a
    runtime that is already gone and hang the process."""
    for h in list(_live_handles):
        try:
            h.close()
        except Exception:
            pass


class Handle:
    """Owns one gpmi_ctx (one device, one set of streams / workspaces)."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get("GPMI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            cnt = C.c_int(0)
            lib.gpmi_device_count(C.byref(cnt))
            if cnt.value > 0:
                device %= cnt.value
        self.lib = lib
        self.device = device
        self.ctx = _vp()
        rc = lib.gpmi_create(device, C.byref(self.ctx))
        if rc != 0:
            msg = lib.gpmi_last_error(None).decode()
            raise GpmiUnavailable(f"gpmi_create(device={device}) failed with status {rc}: {msg}")
        _live_handles.add(self)

    # entry points that factorise a matrix from their own inputs: safe to repeat after a failed attempt
    _REPEATABLE = frozenset(("gpmi_fit", "gpmi_fit_mix", "gpmi_fit_dense", "gpmi_lml", "gpmi_lml_batch", "gpmi_lml_grad",
                             "gpmi_lml_grad_batch", "gpmi_lml_mix", "gpmi_lml_grad_mix", "gpmi_lml_dense", "gpmi_loo_dense",
                             "gpmi_linv_lml", "gpmi_linv_lml_grad", "gpmi_linv_posterior", "gpmi_linv_lml_dense",
                             "gpmi_linv_lml_grad_dense", "gpmi_linv_posterior_dense", "gpmi_loo_terms", "gpmi_loo_grad",
                             "gpmi_loo_terms_mix", "gpmi_dev_potrf", "gpmi_loo_grad_batch", "gpmi_loo_grad_batch_noise", "gpmi_lml_grad_batch_noise",
                             "gpmi_lml_grad_batch_mix", "gpmi_loo_grad_batch_mix"))

    def call(self, name, *args):
        if _TRACE_MS is not None:  # GPMI_TRACE_CALLS=<ms>: report every entry-point call that takes longer (debugging aid)
            import time

            t0 = time.perf_counter()
            rc = getattr(self.lib, name)(self.ctx, *args)
            dt = (time.perf_counter() - t0) * 1e3
            if dt > _TRACE_MS:
                print(f"[gpmi trace] {name}: {dt:.2f} ms", file=sys.stderr, flush=True)
        else:
            rc = getattr(self.lib, name)(self.ctx, *args)
        if (rc == ERR_INTERNAL and name in self._REPEATABLE and not getattr(self, "_no_flow", False)
                and b"[flow-tail]" in self.lib.gpmi_last_error(self.ctx)):
            # a flag-ordered launch did not get its kernels side by side within its time limit (gpmi.h: GPMI_OPT_NO_FLOW):
            # once, with the stream-ordered schedule - the same factor, bit for bit, only slower.  (Only that time-out:
            # a triangular sweep that timed out is not cured by the option and must not cost the handle its fast path.)
            import warnings

            warnings.warn(f"{name}: {self.lib.gpmi_last_error(self.ctx).decode()} - repeating the call with the "
                          "stream-ordered schedule (GPMI_OPT_NO_FLOW) for the rest of this handle's life", RuntimeWarning)
            self._no_flow = True
            self.lib.gpmi_set_option(self.ctx, OPT_NO_FLOW, 1)
            rc = getattr(self.lib, name)(self.ctx, *args)
        if rc != 0:
            raise GpmiError(f"{name} failed with status {rc}: {self.lib.gpmi_last_error(self.ctx).decode()}")

    def close(self):
        if self.ctx:
            self.lib.gpmi_destroy(self.ctx)
            self.ctx = _vp()

    def __del__(self):
        if sys.is_finalizing():
            return  # the atexit hook has already closed what was alive; never touch the runtime from here
        try:
            self.close()
        except Exception:
            pass
