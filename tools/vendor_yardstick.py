"""A same-box yardstick beside the roofline: the VENDOR's fp64 routines at the shapes of this repository's hot path, timed
the way tools/bench_gemm.py times ours (device-resident operands, HIP events around each call, warm-up launches first,
enough repetitions for the shader clock to reach its plateau).  The reference publishes no numbers (README.md:1-40) and its
CPU path is not the target, so this is the only external comparator there is.

    rocsolver_dpotrf   N = 8192 and N = 16384 (lower)                      <-> gpmi_fit's factorisation (regression.py:241)
    rocblas_dtrsm      L X = B, N = 16384, M = 1024 right-hand sides          <-> gpmi_predict's solve (regression.py:213)
    rocblas_dsyrk      C -= A A^T, n = 15872, k = 512 (lower)                 <-> the trailing update, the dominant kernel

TOOLS ONLY: nothing under inference-tools_amd/ may name these libraries (tests/test_abi_cpu.py asserts it); the product
path is hand-written HIP throughout.  The libraries are loaded with ctypes (dlopen) if the image has them; without them, or
without a GPU, the tool prints {"available": false, ...} and exits 0.
usage: python tools/vendor_yardstick.py [--quick] [--reps R]"""
import ctypes as C
import json
import sys
import time

import numpy as np

FILL_LOWER, FILL_UPPER = 122, 121
OP_NONE, OP_TRANS = 111, 112
SIDE_LEFT, SIDE_RIGHT = 141, 142
DIAG_NON_UNIT = 131
H2D, D2H, D2D = 1, 2, 3


def load(name):
    for cand in (name, "/opt/rocm/lib/" + name):
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    return None


def main():
    quick = "--quick" in sys.argv
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else (3 if quick else 8)
    out = {"available": False}
    hip = load("libamdhip64.so")
    blas = load("librocblas.so")
    solver = load("librocsolver.so")
    if hip is None or blas is None or solver is None:
        out["why"] = "libamdhip64 / librocblas / librocsolver not found"
        print(json.dumps(out))
        return
    ndev = C.c_int(0)
    if hip.hipGetDeviceCount(C.byref(ndev)) != 0 or ndev.value < 1:
        out["why"] = "no HIP device"
        print(json.dumps(out))
        return
    vp = C.c_void_p
    hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
    hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [vp]
    hip.hipEventCreate.argtypes = [C.POINTER(vp)]
    hip.hipEventRecord.argtypes = [vp, vp]
    hip.hipEventSynchronize.argtypes = [vp]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with status {rc}")

    def dmalloc(nbytes):
        p = vp()
        chk(hip.hipMalloc(C.byref(p), nbytes), "hipMalloc")
        return p

    handle = vp()
    chk(blas.rocblas_create_handle(C.byref(handle)), "rocblas_create_handle")
    e0, e1 = vp(), vp()
    chk(hip.hipEventCreate(C.byref(e0)), "hipEventCreate")
    chk(hip.hipEventCreate(C.byref(e1)), "hipEventCreate")

    def timed(call, before=None, warm=2):
        """milliseconds of `reps` calls (after `warm` untimed ones), each bracketed by events on the null stream"""
        ms = []
        for r in range(warm + reps):
            if before is not None:
                before()
            chk(hip.hipEventRecord(e0, None), "hipEventRecord")
            chk(call(), "vendor call")
            chk(hip.hipEventRecord(e1, None), "hipEventRecord")
            chk(hip.hipEventSynchronize(e1), "hipEventSynchronize")
            t = C.c_float(0)
            chk(hip.hipEventElapsedTime(C.byref(t), e0, e1), "hipEventElapsedTime")
            if r >= warm:
                ms.append(float(t.value))
        return ms

    rng = np.random.default_rng(5)
    one, minus_one = C.c_double(1.0), C.c_double(-1.0)
    sizes = (4096,) if quick else (8192, 16384)
    res = {}
    big = max(sizes)
    # a diagonally dominant symmetric matrix (only the lower triangle is read): uniform(0, 1) / n off the diagonal + 2 I
    host = rng.random((big, big))
    host /= big
    host[np.diag_indices(big)] += 2.0
    pristine = dmalloc(big * big * 8)
    work = dmalloc(big * big * 8)
    chk(hip.hipMemcpy(pristine, host.ctypes.data_as(vp), big * big * 8, H2D), "hipMemcpy")
    del host
    info = dmalloc(8)
    for n in sizes:
        # (leading dimension `big`: the n x n leading block of the same matrix)
        def restore():
            chk(hip.hipMemcpy(work, pristine, big * big * 8, D2D), "hipMemcpy")

        ms = timed(lambda: solver.rocsolver_dpotrf(handle, FILL_LOWER, C.c_int(n), work, C.c_int(big), info), before=restore)
        inf = C.c_int(-1)
        chk(hip.hipMemcpy(C.byref(inf), info, 4, D2H), "hipMemcpy")
        fl = n**3 / 3.0
        res[f"potrf_{n}"] = {"ms_median": float(np.median(ms)), "ms_min": min(ms), "tflops_at_median": fl / np.median(ms) / 1e9,
                            "info": inf.value, "reps": reps}
    # the factor of the last potrf (n = max size) stays in `work`: the triangular solve with 1024 right-hand sides
    n = big
    m = 256 if quick else 1024
    rhs_host = rng.standard_normal((m, n))  # column-major n x m
    rhs0, rhs = dmalloc(n * m * 8), dmalloc(n * m * 8)
    chk(hip.hipMemcpy(rhs0, rhs_host.ctypes.data_as(vp), n * m * 8, H2D), "hipMemcpy")

    def restore_rhs():
        chk(hip.hipMemcpy(rhs, rhs0, n * m * 8, D2D), "hipMemcpy")

    ms = timed(lambda: blas.rocblas_dtrsm(handle, SIDE_LEFT, FILL_LOWER, OP_NONE, DIAG_NON_UNIT, C.c_int(n), C.c_int(m),
                                          C.byref(one), work, C.c_int(big), rhs, C.c_int(n)), before=restore_rhs)
    res[f"trsm_{n}_x_{m}"] = {"ms_median": float(np.median(ms)), "ms_min": min(ms),
                              "tflops_at_median": float(m) * n * n / np.median(ms) / 1e9, "reps": reps}
    # the trailing update's shape: C (n x n, lower) -= A A^T with A n x 512; 5 x the repetitions, like tools/bench_gemm.py
    ns, k = (3968, 512) if quick else (15872, 512)
    a = dmalloc(ns * k * 8)
    a_host = rng.standard_normal((k, ns)) * 1e-3
    chk(hip.hipMemcpy(a, a_host.ctypes.data_as(vp), ns * k * 8, H2D), "hipMemcpy")
    keep = reps
    reps = 5 * keep
    ms = timed(lambda: blas.rocblas_dsyrk(handle, FILL_LOWER, OP_NONE, C.c_int(ns), C.c_int(k), C.byref(minus_one), a, C.c_int(ns),
                                          C.byref(one), pristine, C.c_int(big)), warm=5)
    reps = keep
    res[f"syrk_{ns}_k{k}"] = {"ms_median": float(np.median(ms)), "ms_min": min(ms),
                              "tflops_at_median": float(ns) * ns * k / np.median(ms) / 1e9,
                              "tflops_at_min": float(ns) * ns * k / min(ms) / 1e9, "reps": 5 * keep,
                              "flops": "n^2 k (the lower triangle's n (n + 1) k, as tools/bench_gemm.py counts ours)"}
    ver = (C.c_char * 256)()
    try:
        blas.rocblas_get_version_string(ver, 256)
        out["rocblas_version"] = ver.value.decode()
    except Exception:
        pass
    out.update(available=True, results=res,
               timing="HIP events on the null stream around each call, operands resident; medians over `reps` calls after warm-up")
    print(json.dumps(out))
    for p in (pristine, work, info, rhs0, rhs, a):
        hip.hipFree(p)
    blas.rocblas_destroy_handle(handle)


if __name__ == "__main__":
    t0 = time.time()
    try:
        main()
    except Exception as err:  # a yardstick must never take its caller down
        print(json.dumps({"available": False, "why": f"{type(err).__name__}: {err}"}))
    sys.stderr.write(f"vendor_yardstick: {time.time() - t0:.1f} s\n")
