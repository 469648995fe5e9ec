"""
NumPy-facing wrapper around one gpmi handle: uploads the data once and exposes
the device operations of the GP hot path with ndarray arguments / results.
Used by `inference_amd.gp.regression.GpRegressor` and the covariance classes.
"""
import ctypes as C
import os

import numpy as np

from inference_amd import _lib
from inference_amd._lib import as_f64, dptr


class _DeviceCommMixin:
    """The library's RCCL communicator on the handle `self.h` (gpmi_comm_*): the start-up broadcast of the data set
    and the result gather of the sharded drivers (inference_amd.sharding)."""

    comm_world = 0

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = _lib.load().gpmi_comm_unique_id(buf)
        if rc != 0:
            raise _lib.GpmiError(f"gpmi_comm_unique_id failed with status {rc} (librccl missing?)")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        # RCCL prints a version banner on stdout from rank 0; keep stdout clean for callers that
        # emit machine-readable output (bench.py prints one JSON line) by sending it to stderr
        import os
        import sys

        libc = C.CDLL(None)
        sys.stdout.flush()
        libc.fflush(None)
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            self.h.call("gpmi_comm_init", int(rank), int(world), unique_id)
        finally:
            # the banner sits in C stdio's buffer when stdout is not a terminal: flush it while fd 1 still points
            # at stderr, or it would come out after the caller's own output at process exit
            libc.fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        self.comm_world = world

    def comm_allgather(self, values):
        send = as_f64(values).ravel()
        recv = np.empty(send.size * self.comm_world)
        self.h.call("gpmi_comm_allgather", dptr(send), dptr(recv), send.size)
        return recv.reshape(self.comm_world, send.size)

    def comm_broadcast(self, values, root: int = 0):
        """`values` of rank `root` on every rank (one ncclBroadcast); the other ranks pass an array of the same shape."""
        buf = as_f64(values).copy()
        self.h.call("gpmi_comm_broadcast", dptr(buf.ravel()), buf.size, int(root))
        return buf

    def comm_count(self) -> int:
        """Ranks RCCL itself sees in the communicator (ncclCommCount)."""
        n = C.c_int(0)
        self.h.call("gpmi_comm_count", C.byref(n))
        return n.value

    def comm_destroy(self):
        self.h.call("gpmi_comm_destroy")
        self.comm_world = 0


class DeviceComm(_DeviceCommMixin):
    """A communicator on a handle of its own, for what happens before a rank has any data: rank 0 broadcasts x, y,
    y_err (sharding.broadcast_dataset; the reference pickles the whole regressor into its worker processes,
    regression.py:597-601, mcmc/parallel.py:127-136), every rank then builds its own regressor.  The same object can
    serve the result gathers (`engine=` of the sharded drivers)."""

    def __init__(self, device=None):
        self.h = _lib.Handle(device)

    def close(self):
        self.h.close()


class GpEngine(_DeviceCommMixin):
    def __init__(self, x, y, noise_var=None, y_cov=None, device=None, reserve=0):
        self.h = _lib.Handle(device)
        self.x = as_f64(x)
        self.y = as_f64(y)
        self.n, self.d = self.x.shape
        if reserve:
            self.h.call("gpmi_set_option", _lib.OPT_RESERVE_POINTS, int(reserve))
        nv = None if noise_var is None else as_f64(noise_var)
        yc = None if y_cov is None else as_f64(y_cov)
        self.h.call("gpmi_set_data", dptr(self.x), dptr(self.y), dptr(nv), dptr(yc), self.n, self.d)

    # -- fit / likelihood -----------------------------------------------------------
    def fit(self, kernel, theta_cov, extra_diag, mu):
        theta = as_f64(theta_cov)
        mu = as_f64(mu)
        alpha = np.empty(self.n)
        logdet = C.c_double(0.0)
        info = C.c_int(0)
        self.h.call("gpmi_fit", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    dptr(alpha), C.byref(logdet), C.byref(info))
        return alpha, logdet.value, info.value

    def lml(self, kernel, theta_cov, extra_diag, mu):
        theta = as_f64(theta_cov)
        mu = as_f64(mu)
        out = C.c_double(0.0)
        info = C.c_int(0)
        self.h.call("gpmi_lml", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    C.byref(out), C.byref(info))
        return out.value, info.value

    def lml_batch(self, kernel, thetas_cov, extra_diag=None, mus=None, mu_const=None):
        thetas = as_f64(thetas_cov)
        if thetas.size == 0:
            return np.empty(0), np.zeros(0, dtype=np.int32)
        T, nt = thetas.shape
        ex = None if extra_diag is None else as_f64(extra_diag)
        mus = None if mus is None else as_f64(mus)
        mc = None if mu_const is None else as_f64(mu_const)
        out = np.empty(T)
        info = np.zeros(T, dtype=np.int32)
        self.h.call("gpmi_lml_batch", kernel, T, dptr(thetas), nt, dptr(ex), dptr(mus), dptr(mc),
                    dptr(out), info.ctypes.data_as(C.POINTER(C.c_int)))
        return out, info

    # evaluations per slot of the asynchronous form (gpmi.h); the library reads the same three environment variables
    ASYNC_MAX = int(os.environ.get("GPMI_ASYNC_SLOT_MAX", "256"))
    _BATCH_GIB = int(os.environ.get("GPMI_BATCH_GIB", "24"))
    _BATCH_MAX = int(os.environ.get("GPMI_BATCH_MAX", "512"))

    def async_slot_capacity(self):
        """Evaluations one asynchronous slot can hold: half of the lockstep workspace (api.hip: ensure_batch_ws keeps it
        below 512 matrices and 24 GiB; GPMI_BATCH_MAX / GPMI_BATCH_GIB)."""
        cap = self.capacity()
        per = cap * (cap + 32) * 8
        return max(0, min(self.ASYNC_MAX, min(self._BATCH_MAX, (self._BATCH_GIB << 30) // per) // 2))

    def lml_batch_submit(self, slot, kernel, thetas_cov, extra_diag=None, mus=None, mu_const=None):
        """gpmi_lml_batch_submit: enqueue the evaluations of `slot` (0 / 1) and return; `lml_batch_wait(slot)` delivers."""
        thetas = as_f64(thetas_cov)
        T, nt = thetas.shape
        ex = None if extra_diag is None else as_f64(extra_diag)
        mus = None if mus is None else as_f64(mus)
        mc = None if mu_const is None else as_f64(mu_const)
        self.h.call("gpmi_lml_batch_submit", kernel, T, dptr(thetas), nt, dptr(ex), dptr(mus), dptr(mc), int(slot))
        self._pending = getattr(self, "_pending", {})
        self._pending[int(slot)] = T

    def lml_batch_wait(self, slot):
        T = self._pending.pop(int(slot))
        out = np.empty(T)
        info = np.zeros(T, dtype=np.int32)
        self.h.call("gpmi_lml_batch_wait", int(slot), dptr(out), info.ctypes.data_as(C.POINTER(C.c_int)))
        return out, info

    # -- mixture covariance (ChangePoint): sub-kernel ids / parameter vectors + per-point weights ------------
    @staticmethod
    def _mix_args(kernels, thetas, weights):
        ks = np.ascontiguousarray(kernels, dtype=np.int32)
        nts = np.ascontiguousarray([len(t) for t in thetas], dtype=np.int32)
        th = as_f64(np.concatenate([np.asarray(t, dtype=float) for t in thetas]))
        g = as_f64(weights)
        ip = C.POINTER(C.c_int)
        return len(ks), ks, ks.ctypes.data_as(ip), th, nts, nts.ctypes.data_as(ip), g

    def fit_mix(self, kernels, thetas, weights, extra_diag, mu):
        nk, ks, kp, th, nts, ntp, g = self._mix_args(kernels, thetas, weights)
        mu = as_f64(mu)
        alpha = np.empty(self.n)
        logdet, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_fit_mix", nk, kp, dptr(th), ntp, dptr(g), float(extra_diag), dptr(mu), dptr(alpha),
                    C.byref(logdet), C.byref(info))
        return alpha, logdet.value, info.value

    def lml_mix(self, kernels, thetas, weights, extra_diag, mu):
        nk, ks, kp, th, nts, ntp, g = self._mix_args(kernels, thetas, weights)
        mu = as_f64(mu)
        out, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_lml_mix", nk, kp, dptr(th), ntp, dptr(g), float(extra_diag), dptr(mu), C.byref(out),
                    C.byref(info))
        return out.value, info.value

    def lml_grad_mix(self, kernels, thetas, weights, extra_diag, mu, row_weights=None):
        """row_weights (nk, 2, n): the caller's own row-sum weights (include/gpmi.h: hw); hrows then is (nk, 2, n)."""
        nk, ks, kp, th, nts, ntp, g = self._mix_args(kernels, thetas, weights)
        mu = as_f64(mu)
        lml, info = C.c_double(0.0), C.c_int(0)
        grad = np.empty(th.size)
        hw = None if row_weights is None else as_f64(np.asarray(row_weights, dtype=float).reshape(nk, 2, self.n))
        hrows = np.empty((nk, self.n) if hw is None else (nk, 2, self.n))
        alpha = np.empty(self.n)
        self.h.call("gpmi_lml_grad_mix", nk, kp, dptr(th), ntp, dptr(g), dptr(hw), float(extra_diag), dptr(mu),
                    C.byref(lml), dptr(grad), dptr(hrows), dptr(alpha), C.byref(info))
        return lml.value, grad, hrows, alpha, info.value

    def lml_grad_batch_mix(self, kernels, thetas, weights, extra_diag=None, mus=None, mu_const=None, want_qdiag=False,
                           row_weights=None):
        """gpmi_lml_grad_batch_mix: T evaluations of the mixture likelihood and its gradient pieces in one call.
        kernels: the nk sub-kernel ids; thetas: T lists of nk parameter vectors; weights: (T, nk, n) window weights;
        row_weights: None or (T, nk, 2, n) (include/gpmi.h: hw).
        Returns (lml (T,), grad (T, sum n_thetas), hrows (T, nk, n) - (T, nk, 2, n) with row_weights -, alpha (T, n),
        qdiag (T, n) or None, info (T,))."""
        ks = np.ascontiguousarray(kernels, dtype=np.int32)
        nk = len(ks)
        T = len(thetas)
        nts = np.ascontiguousarray([len(t) for t in thetas[0]], dtype=np.int32)
        th = as_f64(np.array([np.concatenate([np.asarray(v, dtype=float) for v in row]) for row in thetas]))
        g = as_f64(np.asarray(weights, dtype=float).reshape(T, nk, self.n))
        ex = None if extra_diag is None else as_f64(extra_diag)
        mus = None if mus is None else as_f64(mus)
        mc = None if mu_const is None else as_f64(mu_const)
        lml = np.empty(T)
        grad = np.empty((T, int(nts.sum())))
        hw = None if row_weights is None else as_f64(np.asarray(row_weights, dtype=float).reshape(T, nk, 2, self.n))
        hrows = np.empty((T, nk, self.n) if hw is None else (T, nk, 2, self.n))
        alpha = np.empty((T, self.n))
        qdiag = np.empty((T, self.n)) if want_qdiag else None
        info = np.zeros(T, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.h.call("gpmi_lml_grad_batch_mix", nk, ks.ctypes.data_as(ip), T, dptr(th), nts.ctypes.data_as(ip), dptr(g),
                    dptr(hw), dptr(ex), dptr(mus), dptr(mc), dptr(lml), dptr(grad), dptr(hrows), dptr(alpha), dptr(qdiag),
                    info.ctypes.data_as(ip))
        return lml, grad, hrows, alpha, qdiag, info

    def loo_grad_batch_mix(self, kernels, thetas, weights, extra_diag=None, mus=None, mu_const=None, row_weights=None):
        """gpmi_loo_grad_batch_mix: T evaluations of the mixture's leave-one-out pieces in one call (lockstep sizes only).
        Arguments as `lml_grad_batch_mix`.  Returns (alpha, ikdiag, pvec, mdiag (T, n each), grad (T, sum n_thetas),
        hrows (T, nk, n) - (T, nk, 2, n) with row_weights -, info (T,))."""
        ks = np.ascontiguousarray(kernels, dtype=np.int32)
        nk = len(ks)
        T = len(thetas)
        nts = np.ascontiguousarray([len(t) for t in thetas[0]], dtype=np.int32)
        th = as_f64(np.array([np.concatenate([np.asarray(v, dtype=float) for v in row]) for row in thetas]))
        g = as_f64(np.asarray(weights, dtype=float).reshape(T, nk, self.n))
        ex = None if extra_diag is None else as_f64(extra_diag)
        mus = None if mus is None else as_f64(mus)
        mc = None if mu_const is None else as_f64(mu_const)
        alpha, ikdiag, pvec, mdiag = (np.empty((T, self.n)) for _ in range(4))
        grad = np.empty((T, int(nts.sum())))
        hw = None if row_weights is None else as_f64(np.asarray(row_weights, dtype=float).reshape(T, nk, 2, self.n))
        hrows = np.empty((T, nk, self.n) if hw is None else (T, nk, 2, self.n))
        info = np.zeros(T, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.h.call("gpmi_loo_grad_batch_mix", nk, ks.ctypes.data_as(ip), T, dptr(th), nts.ctypes.data_as(ip), dptr(g),
                    dptr(hw), dptr(ex), dptr(mus), dptr(mc), dptr(alpha), dptr(ikdiag), dptr(pvec), dptr(mdiag), dptr(grad),
                    dptr(hrows), info.ctypes.data_as(ip))
        return alpha, ikdiag, pvec, mdiag, grad, hrows, info

    def loo_terms_mix(self, kernels, thetas, weights, extra_diag, mu):
        nk, ks, kp, th, nts, ntp, g = self._mix_args(kernels, thetas, weights)
        mu = as_f64(mu)
        alpha, ikdiag = np.empty(self.n), np.empty(self.n)
        info = C.c_int(0)
        self.h.call("gpmi_loo_terms_mix", nk, kp, dptr(th), ntp, dptr(g), float(extra_diag), dptr(mu),
                    dptr(alpha), dptr(ikdiag), C.byref(info))
        return alpha, ikdiag, info.value

    def predict_mix(self, pts, query_weights):
        p = as_f64(pts)
        gq = as_f64(query_weights)
        m = p.shape[0]
        mu, nss = np.empty(m), np.empty(m)
        self.h.call("gpmi_predict_mix", dptr(p), m, dptr(gq), dptr(mu), dptr(nss))
        return mu, nss

    def posterior_mix(self, pts, query_weights, mean_only=False):
        p = as_f64(pts)
        gq = as_f64(query_weights)
        m = p.shape[0]
        mu = np.empty(m)
        cov = None if mean_only else np.empty((m, m))
        self.h.call("gpmi_posterior_mix", dptr(p), m, dptr(gq), dptr(mu), dptr(cov))
        return mu, cov

    def set_noise(self, noise_var):
        self.h.call("gpmi_set_noise", dptr(as_f64(noise_var)))

    def lml_grad_qdiag(self):
        q = np.empty(self.n)
        self.h.call("gpmi_lml_grad_qdiag", dptr(q))
        return q

    def prepare_gradient(self, n_theta: int):
        """gpmi_prepare_gradient: the gradient path's lazily allocated workspaces, now."""
        self.h.call("gpmi_prepare_gradient", int(n_theta))

    def capacity(self):
        cap = C.c_int64(0)
        self.h.call("gpmi_capacity", C.byref(cap))
        return cap.value

    def append_point(self, x_new, y_new, noise_var_new, mu):
        """Append one training point at the fitted hyper-parameters (gpmi_append_point): (alpha, logdet, info)."""
        xn, mu = as_f64(np.ravel(x_new)), as_f64(mu)
        alpha = np.empty(self.n + 1)
        logdet, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_append_point", dptr(xn), float(y_new), float(noise_var_new), dptr(mu), dptr(alpha),
                    C.byref(logdet), C.byref(info))
        if info.value == 0:
            self.x = np.vstack([self.x, xn[None, :]])
            self.y = np.append(self.y, float(y_new))
            self.n += 1
        return alpha, logdet.value, info.value

    def set_option(self, option, value):
        self.h.call("gpmi_set_option", int(option), int(value))

    def set_streams(self, n):
        self.h.call("gpmi_set_streams", int(n))

    def lml_grad(self, kernel, theta_cov, extra_diag, mu):
        theta = as_f64(theta_cov)
        mu = as_f64(mu)
        lml = C.c_double(0.0)
        trq = C.c_double(0.0)
        info = C.c_int(0)
        grad = np.empty(theta.size)
        alpha = np.empty(self.n)
        self.h.call("gpmi_lml_grad", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    C.byref(lml), dptr(grad), C.byref(trq), dptr(alpha), C.byref(info))
        return lml.value, grad, trq.value, alpha, info.value

    def lml_grad_batch(self, kernel, thetas_cov, extra_diag, mus=None, mu_const=None):
        """T evaluations of `lml_grad` in one call (gpmi_lml_grad_batch: lockstep for padded N <= 4096):
        (lml (T,), grad (T, n_theta), trace_q (T,), alpha (T, n), info (T,))."""
        th = as_f64(np.atleast_2d(thetas_cov))
        T, nth = th.shape
        ex = as_f64(np.broadcast_to(np.asarray(extra_diag, dtype=float), (T,)))
        lml, trq = np.empty(T), np.empty(T)
        grad, alpha = np.empty((T, nth)), np.empty((T, self.n))
        info = np.zeros(T, dtype=np.int32)
        mus_p = dptr(as_f64(mus)) if mus is not None else None
        muc_p = dptr(as_f64(np.broadcast_to(np.asarray(mu_const, dtype=float), (T,)))) if mus is None else None
        self.h.call("gpmi_lml_grad_batch", kernel, T, dptr(th), nth, dptr(ex), mus_p, muc_p, dptr(lml), dptr(grad),
                    dptr(trq), dptr(alpha), info.ctypes.data_as(C.POINTER(C.c_int)))
        return lml, grad, trq, alpha, info

    def lml_grad_batch_noise(self, kernel, thetas_cov, extra_diag, noise_vars, mus=None, mu_const=None):
        """`lml_grad_batch` with noise variances of their own for every evaluation (gpmi_lml_grad_batch_noise):
        (lml, grad, trace_q, alpha, qdiag (T, n), info)."""
        th = as_f64(np.atleast_2d(thetas_cov))
        T, nth = th.shape
        ex = as_f64(np.broadcast_to(np.asarray(extra_diag, dtype=float), (T,)))
        nv = as_f64(np.atleast_2d(noise_vars))
        lml, trq = np.empty(T), np.empty(T)
        grad, alpha, qdiag = np.empty((T, nth)), np.empty((T, self.n)), np.empty((T, self.n))
        info = np.zeros(T, dtype=np.int32)
        mus_p = dptr(as_f64(mus)) if mus is not None else None
        muc_p = dptr(as_f64(np.broadcast_to(np.asarray(mu_const, dtype=float), (T,)))) if mus is None else None
        self.h.call("gpmi_lml_grad_batch_noise", kernel, T, dptr(th), nth, dptr(ex), mus_p, muc_p, dptr(nv), dptr(lml),
                    dptr(grad), dptr(trq), dptr(alpha), dptr(qdiag), info.ctypes.data_as(C.POINTER(C.c_int)))
        return lml, grad, trq, alpha, qdiag, info

    # -- prediction -------------------------------------------------------------------
    def predict(self, pts, want_var=True):
        p = as_f64(pts)
        m = p.shape[0]
        mu = np.empty(m)
        var = np.empty(m) if want_var else None
        self.h.call("gpmi_predict", dptr(p), m, dptr(mu), dptr(var))
        return mu, var

    def posterior(self, pts, mean_only=False):
        p = as_f64(pts)
        m = p.shape[0]
        mu = np.empty(m)
        cov = None if mean_only else np.empty((m, m))
        self.h.call("gpmi_posterior", dptr(p), m, dptr(mu), dptr(cov))
        return mu, cov

    def spatial_derivatives(self, pts):
        p = as_f64(pts)
        m = p.shape[0]
        dmu = np.empty((m, self.d))
        dvar = np.empty((m, self.d))
        self.h.call("gpmi_spatial_derivatives", dptr(p), m, dptr(dmu), dptr(dvar))
        return dmu, dvar

    def gradient(self, pts):
        p = as_f64(pts)
        m = p.shape[0]
        gmu = np.empty((m, self.d))
        gcov = np.empty((m, self.d, self.d))
        self.h.call("gpmi_gradient", dptr(p), m, dptr(gmu), dptr(gcov))
        return gmu, gcov

    def covariance(self, kernel, theta_cov, extra_diag=0.0, with_noise=False):
        theta = as_f64(theta_cov)
        K = np.empty((self.n, self.n))
        self.h.call("gpmi_covariance", kernel, dptr(theta), theta.size, float(extra_diag),
                    int(bool(with_noise)), dptr(K))
        return K

    def cross_covariance(self, kernel, theta_cov, pts):
        theta = as_f64(theta_cov)
        p = as_f64(pts)
        out = np.empty((p.shape[0], self.n))
        self.h.call("gpmi_cross_covariance", kernel, dptr(theta), theta.size, dptr(p), p.shape[0], dptr(out))
        return out

    def get_K(self):
        K = np.empty((self.n, self.n))
        self.h.call("gpmi_get_K", dptr(K))
        return K

    def get_L(self):
        L = np.empty((self.n, self.n))
        self.h.call("gpmi_get_L", dptr(L))
        return L

    def loo_diag(self):
        out = np.empty(self.n)
        self.h.call("gpmi_loo_diag", dptr(out))
        return out

    def loo_terms(self, kernel, theta_cov, extra_diag, mu):
        theta = as_f64(theta_cov)
        mu = as_f64(mu)
        alpha = np.empty(self.n)
        ikdiag = np.empty(self.n)
        info = C.c_int(0)
        self.h.call("gpmi_loo_terms", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    dptr(alpha), dptr(ikdiag), C.byref(info))
        return alpha, ikdiag, info.value

    def loo_grad(self, kernel, theta_cov, extra_diag, mu):
        theta = as_f64(theta_cov)
        mu = as_f64(mu)
        alpha, ikdiag, pvec = np.empty(self.n), np.empty(self.n), np.empty(self.n)
        grad = np.empty(theta.size)
        trq = C.c_double(0.0)
        info = C.c_int(0)
        self.h.call("gpmi_loo_grad", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    dptr(alpha), dptr(ikdiag), dptr(pvec), dptr(grad), C.byref(trq), C.byref(info))
        return alpha, ikdiag, pvec, grad, trq.value, info.value

    def loo_grad_batch(self, kernel, thetas_cov, extra_diag, mus=None, mu_const=None, noise_var=None):
        """T evaluations of `loo_grad` in one call (gpmi_loo_grad_batch: lockstep for padded N <= 4096):
        (alpha (T, n), ikdiag (T, n), pvec (T, n), grad (T, n_theta), trace_q (T,), info (T,)).  With `noise_var`
        (T, n: HeteroscedasticNoise, every evaluation its own data variances; gpmi_loo_grad_batch_noise) the tuple gains
        mdiag (T, n) = diag(K^-1 diag(c2) K^-1) in front of info."""
        th = as_f64(np.atleast_2d(thetas_cov))
        T, nth = th.shape
        ex = as_f64(np.broadcast_to(np.asarray(extra_diag, dtype=float), (T,)))
        alpha, ikdiag, pvec = np.empty((T, self.n)), np.empty((T, self.n)), np.empty((T, self.n))
        grad, trq = np.empty((T, nth)), np.empty(T)
        info = np.zeros(T, dtype=np.int32)
        mus_p = dptr(as_f64(mus)) if mus is not None else None
        muc_p = dptr(as_f64(np.broadcast_to(np.asarray(mu_const, dtype=float), (T,)))) if mus is None else None
        if noise_var is not None:
            nv = as_f64(noise_var)
            assert nv.shape == (T, self.n)
            mdiag = np.empty((T, self.n))
            self.h.call("gpmi_loo_grad_batch_noise", kernel, T, dptr(th), nth, dptr(ex), mus_p, muc_p, dptr(nv), dptr(alpha),
                        dptr(ikdiag), dptr(pvec), dptr(mdiag), dptr(grad), dptr(trq), info.ctypes.data_as(C.POINTER(C.c_int)))
            return alpha, ikdiag, pvec, grad, trq, mdiag, info
        self.h.call("gpmi_loo_grad_batch", kernel, T, dptr(th), nth, dptr(ex), mus_p, muc_p, dptr(alpha), dptr(ikdiag),
                    dptr(pvec), dptr(grad), dptr(trq), info.ctypes.data_as(C.POINTER(C.c_int)))
        return alpha, ikdiag, pvec, grad, trq, info

    # -- dense entry points (covariance functions that only implement the plugin ABC) ------------------
    def fit_dense(self, K, mu):
        K, mu = as_f64(K), as_f64(mu)
        alpha = np.empty(self.n)
        logdet, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_fit_dense", dptr(K), dptr(mu), dptr(alpha), C.byref(logdet), C.byref(info))
        return alpha, logdet.value, info.value

    def lml_dense(self, K, mu, want_alpha=False, want_inverse=False):
        K, mu = as_f64(K), as_f64(mu)
        lml, info = C.c_double(0.0), C.c_int(0)
        alpha = np.empty(self.n) if want_alpha else None
        iK = np.empty((self.n, self.n)) if want_inverse else None
        self.h.call("gpmi_lml_dense", dptr(K), dptr(mu), C.byref(lml), dptr(alpha), dptr(iK), C.byref(info))
        return lml.value, alpha, iK, info.value

    def loo_dense(self, K, mu, want_gradient_pieces=False):
        K, mu = as_f64(K), as_f64(mu)
        alpha, ikdiag = np.empty(self.n), np.empty(self.n)
        pvec = np.empty(self.n) if want_gradient_pieces else None
        W = np.empty((self.n, self.n)) if want_gradient_pieces else None
        info = C.c_int(0)
        self.h.call("gpmi_loo_dense", dptr(K), dptr(mu), dptr(alpha), dptr(ikdiag), dptr(pvec), dptr(W), C.byref(info))
        return alpha, ikdiag, pvec, W, info.value

    def predict_dense(self, Kq, want_var=True):
        Kq = as_f64(Kq)
        m = Kq.shape[0]
        ka = np.empty(m)
        ss = np.empty(m) if want_var else None
        self.h.call("gpmi_predict_dense", dptr(Kq), m, dptr(ka), dptr(ss))
        return ka, ss

    def solve_rows(self, Q, want_rows=True, want_gram=False):
        Q = as_f64(Q)
        m = Q.shape[0]
        X = np.empty((m, self.n)) if want_rows else None
        G = np.empty((m, m)) if want_gram else None
        self.h.call("gpmi_solve_rows", dptr(Q), m, dptr(X), dptr(G))
        return X, G

    # -- instrumentation ----------------------------------------------------------------
    def timer_start(self):
        self.h.call("gpmi_timer_start")

    def timer_stop(self):
        ms = C.c_float(0.0)
        self.h.call("gpmi_timer_stop", C.byref(ms))
        return ms.value

    def profile_enable(self, on=True):
        """True / 1: every class (HIP events around the launches: perturbs the overlap); 0: off;
        `2 << klass` (or an OR of them): only those classes — the trailing-update class PROF_SYRK is timed by
        in-kernel stamps and costs nothing."""
        self.h.call("gpmi_profile_enable", 1 if on is True else int(on))

    def profile_reset(self):
        self.h.call("gpmi_profile_reset")

    def profile_read(self, klass):
        n = C.c_int64(0)
        ms, fl, by = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        self.h.call("gpmi_profile_read", klass, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
        return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}

    def profile_clock(self):
        """Shader clock (GHz) over the stamped trailing-update launches since the last reset."""
        ghz = C.c_double(0.0)
        self.h.call("gpmi_profile_clock", C.byref(ghz))
        return ghz.value

    def sync(self):
        self.h.call("gpmi_sync")

    def close(self):
        self.h.close()


class LinvEngine:
    """Device side of `GpLinearInverter` (gpmi_linv_*): parameter positions, model matrix and data are
    uploaded once; every call evaluates one hyper-parameter vector."""

    def __init__(self, positions, model_matrix, y, y_err, device=None):
        self.h = _lib.Handle(device)
        self.x = as_f64(positions)
        self.n, self.d = self.x.shape
        self.A = as_f64(model_matrix)
        self.m = self.A.shape[0]
        zeros = np.zeros(self.n)
        self.h.call("gpmi_set_data", dptr(self.x), dptr(zeros), None, None, self.n, self.d)
        self.h.call("gpmi_linv_set", dptr(self.A), self.m, dptr(as_f64(y)), dptr(as_f64(y_err)))

    def lml(self, kernel, theta_cov, extra_diag, mu):
        theta, mu = as_f64(theta_cov), as_f64(mu)
        out, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_linv_lml", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    C.byref(out), C.byref(info))
        return out.value, info.value

    def lml_grad(self, kernel, theta_cov, extra_diag, mu):
        theta, mu = as_f64(theta_cov), as_f64(mu)
        lml, trq, info = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        grad, w = np.empty(theta.size), np.empty(self.n)
        self.h.call("gpmi_linv_lml_grad", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    C.byref(lml), dptr(grad), C.byref(trq), dptr(w), C.byref(info))
        return lml.value, grad, trq.value, w, info.value

    def posterior(self, kernel, theta_cov, extra_diag, mu, with_cov=True):
        theta, mu = as_f64(theta_cov), as_f64(mu)
        info = C.c_int(0)
        mean = np.empty(self.n)
        cov = np.empty((self.n, self.n)) if with_cov else None
        self.h.call("gpmi_linv_posterior", kernel, dptr(theta), theta.size, float(extra_diag), dptr(mu),
                    dptr(mean), dptr(cov), C.byref(info))
        return mean, cov, info.value

    # -- prior covariance evaluated by the caller (any CovarianceFunction object) ------------
    def _dense(self, K):
        K = as_f64(K)
        if K.shape != (self.n, self.n):
            raise ValueError(f"prior covariance must be ({self.n}, {self.n}), got {K.shape}")
        return K

    def lml_dense(self, K, mu):
        K, mu = self._dense(K), as_f64(mu)
        out, info = C.c_double(0.0), C.c_int(0)
        self.h.call("gpmi_linv_lml_dense", dptr(K), dptr(mu), C.byref(out), C.byref(info))
        return out.value, info.value

    def lml_grad_dense(self, K, mu):
        """(LML, G = A^T J^-1 A, w = A^T alpha, info): grad_j = 1/2 sum (w w^T - G) o dK_j."""
        K, mu = self._dense(K), as_f64(mu)
        lml, info = C.c_double(0.0), C.c_int(0)
        G, w = np.empty((self.n, self.n)), np.empty(self.n)
        self.h.call("gpmi_linv_lml_grad_dense", dptr(K), dptr(mu), C.byref(lml), dptr(G), dptr(w), C.byref(info))
        return lml.value, G, w, info.value

    def posterior_dense(self, K, mu, with_cov=True):
        K, mu = self._dense(K), as_f64(mu)
        info = C.c_int(0)
        mean = np.empty(self.n)
        cov = np.empty((self.n, self.n)) if with_cov else None
        self.h.call("gpmi_linv_posterior_dense", dptr(K), dptr(mu), dptr(mean), dptr(cov), C.byref(info))
        return mean, cov, info.value
