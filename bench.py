"""
bench.py — headline benchmark of the GP hot path on MI355X.

Metric (BASELINE.json): GpRegressor fit+predict wall-time and GFLOP/s at N=16384, d=8.
One "step" = one fit at fixed hyper-parameters (covariance build -> blocked Cholesky -> alpha solves,
reference regression.py:218-244) followed by a batched predict of M=1024 points (regression.py:188-216)
on the synthetic data of workloads.py, x / y / y_err already resident in HBM.
value = (N^3/3 + M N^2) FLOP per step * steps * n_gpus / wall  in GFLOP/s (whole job).

N GPUs: one process per GPU, each rank runs the same step at its own hyper-parameter vector (the path
shards over independent hyper-parameter evaluations, weak scaling) and the per-rank results
(log-determinant, alpha norm, predictive checksum of every step) are all-gathered over RCCL / xGMI in ONE
collective at the end of the timed region through the library's own communicator (gpmi_comm_*); barriers and
the max-over-ranks of the timing are RCCL all-gathers too.  The ranks launched by torch.distributed.run read RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT from the environment and exchange the RCCL unique id through a per-job
directory in /tmp (inference_amd.sharding.FileRendezvous): torch itself is NOT imported, because
importing it loads torch's bundled HIP / HSA runtime beside the system ROCm 7.2 one, and RCCL then
binds to the uninitialised copy.

`python bench.py --gpus N` without a launcher starts its own N ranks (launch_ranks): the parent never touches
the GPU, it spawns N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE set and relays rank 0's line.

Extra objects on the JSON line:
  roofline      the potrf trailing SYRK/GEMM update (fp64 MFMA bound), the dominant kernel: algorithmic FLOP of
                every launch divided by its duration, measured live inside the timed region by in-kernel
                s_memrealtime stamps (first workgroups' start -> last workgroup's end behind its epilogue stores;
                HIP events on the look-ahead streams would perturb the overlap).  `all_trailing` adds the launches
                that run other tile shapes (64 x 64 remainders, launches below 384 tiles); `clock_ghz` is the shader
                clock held during those launches (s_memtime / s_memrealtime; it needs tens of milliseconds of
                sustained load to reach its plateau, which the timed loop provides), `frac_at_clock` is the fraction of
                the MFMA peak AT THAT CLOCK.  `traffic` is HBM
                bytes per launch from the PMC passes of the same command committed under profiles/ (separate
                rocprofv3 --pmc runs cannot be part of this run); `kernels` lists the other kernels of a step
                (K-build, the two triangular sweeps, the predict TRSM) from an extra, un-timed pass with events.
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference, kind "port") timed on the host cores (rank 0,
                N=1 only).  `value` (round 5) = ONE run at the metric's own N, d, M with K-build / potrf / solves / predict
                seconds apart, after the bounded N=4096 sample (`sample_n4096`: warm-up + median of 3) has warmed the BLAS
                pool; --no-cpu-metric-size leaves the sample as `value`.  `configs` = the same oracle at the sizes of
                BASELINE configs 2-5, one bounded run each (--no-cpu-configs skips them: about two minutes of host time)
  configs       device timings of the BASELINE configurations the headline and `sharded` do not cover, after the timed
                region: config 2 (fit / predict / LML / LML + gradient at N=8192), config 4 (EI, -ln EI + gradient at 1000
                candidates, one propose_evaluation) and the LML gradient at the metric's own size (--no-configs skips them)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)



def _rank_cpu_setup():
    """A rank of a multi-rank job keeps to its own share of the host: cores (sched_setaffinity) and BLAS / OpenMP
    threads = cores // ranks, set HERE - in the child, before NumPy loads its BLAS and before anything touches the GPU.
    Eight ranks with a 64-thread OpenBLAS pool each on 64 cores would spin against one another and against the HIP
    runtime's completion threads (one stray BLAS call doubled the step time: see step() below).  BENCH_NO_PIN=1 leaves
    the affinity alone; thread counts the caller has set are respected."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ or world <= 1:
        return None
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % local_world
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = list(range(os.cpu_count() or 1))
    per = max(1, len(cpus) // local_world)
    mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
    pinned = False
    if not os.environ.get("BENCH_NO_PIN"):
        try:
            os.sched_setaffinity(0, mine)
            pinned = True
        except (AttributeError, OSError):
            pass
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        if var == "OMP_NUM_THREADS" and os.environ.get(var) == "1" and "TORCHELASTIC_RUN_ID" in os.environ:
            os.environ[var] = str(per)  # torch.distributed.run's own default of 1, not a choice of the caller
        else:
            os.environ.setdefault(var, str(per))
    return {"cores_per_rank": per, "pinned": pinned, "first_core": mine[0],
            "blas_threads": os.environ.get("OPENBLAS_NUM_THREADS")}


RANK_CPU = _rank_cpu_setup()

import numpy as np  # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2048 FLOP / 64 clk x 2.4 GHz (tools/mfma_probe.hip)


RCCL_INFO = {}  # rccl_ranks_seen, dataset_broadcast (multi-rank runs with a live communicator)


T_START = time.perf_counter()
# wall-clock budget of the whole default run: optional host-side blocks behind the timed region are skipped (and say so)
# when starting them would take the run beyond it
WALL_BUDGET_S = float(os.environ.get("BENCH_WALL_BUDGET_S", "330"))


def wall_left():
    return WALL_BUDGET_S - (time.perf_counter() - T_START)


def cpu_baseline(n_cpu, d, m_cpu, runs=3):
    """Oracle (port of the reference path: same NumPy / LAPACK calls) on the host cores: fit + batched predict at a
    bounded size, one warm-up at a quarter of the size, then the median of `runs` runs."""
    from oracle import gp_oracle as orc  # checker / baseline only
    import workloads as wl

    def one(n, m):
        x, y, e = wl.synthetic_dataset(2, n, d)
        theta = wl.timing_theta(wl.SE, y, d)
        pts = wl.query_points(2, m, d)
        t0 = time.perf_counter()
        gp = orc.OracleGp(x, y, e, kernel=orc.SE, hyperpars=theta)
        gp(pts)
        return time.perf_counter() - t0

    threads, blas = os.cpu_count() or 1, "unknown BLAS"
    try:
        from threadpoolctl import threadpool_info

        info = [i for i in threadpool_info() if i.get("user_api") == "blas"] or threadpool_info()
        if info:
            threads = max(i.get("num_threads", 1) for i in info)
            blas = "; ".join(sorted({f"{i.get('internal_api', '?')} {i.get('version', '')}".strip() for i in info}))
    except Exception:
        pass
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except Exception:
        pass
    try:
        import scipy

        versions = f"NumPy {np.__version__}, SciPy {scipy.__version__}"
    except Exception:
        versions = f"NumPy {np.__version__}"
    one(max(n_cpu // 4, 256), max(m_cpu // 4, 16))
    times = sorted(one(n_cpu, m_cpu) for _ in range(runs))
    dt = times[len(times) // 2]
    flops = n_cpu**3 / 3.0 + m_cpu * float(n_cpu) ** 2

    # second flavour: the reference's own memory / loop structure (N x N x d tensors, one triangular solve per
    # query point), at the largest BASELINE size whose tensors fit the host's free memory and a time budget (after a
    # first run at N = 3072 has measured the host's rate: the tensor part scales with N^2 d, the factorisation with N^3)
    def faithful_run(n_f, m_f):
        xf, yf, ef = wl.synthetic_dataset(2, n_f, d)
        tf = wl.timing_theta(wl.SE, yf, d)
        t0 = time.perf_counter()
        orc.faithful_se_fit_predict(xf, yf, ef, tf, wl.query_points(2, m_f, d))
        return time.perf_counter() - t0

    n_f, m_f = min(n_cpu, 3072), 64
    dt_f = faithful_run(n_f, m_f)
    ram_gb = host_ram_gb()
    budget_s = float(os.environ.get("BENCH_FAITHFUL_BUDGET_S", "45"))
    chosen = None
    for n_big in (16384, 8192):
        need_gb = 4.2 * n_big * n_big * d * 8 / 1e9  # dx, distances, distances / L2 and the exp's input + K, eye, L
        est = dt_f * (n_big / n_f) ** 2 + (n_big**3 / 3.0) / 1.0e11  # tensors ~ N^2 d; LAPACK potrf ~100 GFLOP/s on this class of host
        if ram_gb is not None and need_gb < 0.8 * ram_gb and est < budget_s and est + 100 < wall_left() and n_big > n_f:
            chosen = (n_big, need_gb, est)
            break
    if chosen is not None:
        n_f, m_f = chosen[0], 16
        dt_f = faithful_run(n_f, m_f)
    faithful = {
        "value": (n_f**3 / 3.0 + m_f * float(n_f) ** 2) / dt_f / 1e9,
        "unit": "GFLOP/s",
        "n": n_f, "m": m_f, "seconds": dt_f,
        "sample": f"same path with the reference's N x N x d tensors (covariance.py:218-219, 254) and its per-point predict "
        f"loop (regression.py:205-216) on {m_f} points, N={n_f} d={d}, 1 run: {dt_f:.1f} s"
        + (f" (tensors ~{chosen[1]:.0f} GB of {ram_gb:.0f} GB available)" if chosen else
           f" (larger sizes skipped: MemAvailable {ram_gb and round(ram_gb)} GB, time budget {budget_s:.0f} s, {wall_left():.0f} s of the run left)"),
    }
    return {
        "value": flops / dt / 1e9,
        "unit": "GFLOP/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"oracle fit+predict SE N={n_cpu} d={d} M={m_cpu} (same generator and theta as the workload; row-chunked "
        f"K-build as in the reference, numpy.linalg.cholesky, scipy.linalg.solve_triangular), warm-up + median of {runs} "
        f"runs: {dt:.1f} s each; {cpu}, {os.cpu_count()} logical CPUs, {blas} with {threads} threads, {versions}",
        "faithful": faithful,
        "blas": blas, "blas_threads": int(threads), "cpu": cpu, "host_ram_gb": ram_gb,
    }


def host_ram_gb():
    """MemAvailable of the host in GB (None if /proc/meminfo cannot be read)."""
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return float(line.split()[1]) * 1024 / 1e9
    except Exception:
        pass
    return None


def cpu_at_metric_size(n, d, m, runs=3):
    """The memory-lean oracle at the metric's own configuration, per phase: one
    warm-up run + the median of `runs` runs (by total time), the same NumPy / LAPACK calls as the reference path
    (regression.py:218-244, 188-216), K-build row-chunked as in oracle/gp_oracle.py (the reference's N x N x d tensors
    need 52 GB at N = 16384, d = 8: the `faithful` flavour of cpu_baseline runs them when the host has the memory)."""
    from oracle import gp_oracle as orc  # checker / baseline only
    import workloads as wl
    from numpy.linalg import cholesky
    from scipy.linalg import solve_triangular

    x, y, e = wl.synthetic_dataset(2, n, d)
    theta = wl.timing_theta(wl.SE, y, d)
    pts = wl.query_points(2, m, d)

    def one():
        t = [time.perf_counter()]

        def lap():
            t.append(time.perf_counter())
            return t[-1] - t[-2]

        K = orc.se_build(x, theta[1:])  # covariance.py:247-255
        K[np.diag_indices(n)] += e**2   # regression.py:239 "+ self.sig"
        t_build = lap()
        L = cholesky(K)                 # regression.py:241
        del K
        t_potrf = lap()
        alpha = solve_triangular(L.T, solve_triangular(L, y - theta[0], lower=True))  # regression.py:242-244
        t_solve = lap()
        K_qx = orc.se_cross(pts, x, theta[1:])                 # regression.py:209-210, batched over the query points
        mu = K_qx @ alpha + theta[0]
        v = solve_triangular(L, K_qx.T, lower=True)            # regression.py:213
        sig = np.sqrt(np.abs(np.exp(theta[1]) ** 2 - (v**2).sum(axis=0)))
        t_pred = lap()
        return {"k_build": t_build, "potrf": t_potrf, "alpha_solves": t_solve, "predict": t_pred, "total": t[-1] - t[0],
                "mu0": float(mu[0]), "sig0": float(sig[0])}

    warm = one()
    timed = sorted((one() for _ in range(runs)), key=lambda r: r["total"])
    med = timed[len(timed) // 2]
    total = med["total"]
    flops = n**3 / 3.0 + m * float(n) ** 2
    return {
        "value": flops / total / 1e9,
        "unit": "GFLOP/s",
        "seconds": {k: med[k] for k in ("k_build", "potrf", "alpha_solves", "predict", "total")},
        "runs": runs, "totals_s": [r["total"] for r in timed], "warmup_total_s": warm["total"],
        "potrf_gflops": n**3 / 3.0 / med["potrf"] / 1e9,
        "sample": f"memory-lean oracle at the metric's own size, SE N={n} d={d} M={m}: one warm-up run ({warm['total']:.1f} s) + "
        f"median of {runs} runs ({total:.1f} s; all: " + ", ".join("%.1f" % r_["total"] for r_ in timed) + "); "
        f"checksum mu[0]={med['mu0']:.12g} sig[0]={med['sig0']:.6g}",
    }


def _timeit(fn, reps=3, warm=1, steady=0.1):
    """Mean wall time of a call in STEADY STATE: the shader clock needs tens of milliseconds of sustained load to reach
    its plateau (tools/clock_ramp.py: an N = 8192 prediction takes 1.82 ms in the first calls after an idle spell and 1.62
    from the twentieth on; tools/bench_gemm.py: the trailing update 59 TFLOP/s over 5 launches, 66 over 40) - the headline's
    timed loop runs there, so do these: calls for at least `steady` seconds before the clock is read, and at least `reps`
    calls and `steady` seconds inside it."""
    t0 = time.perf_counter()
    n = 0
    while n < warm or time.perf_counter() - t0 < steady:
        fn()
        n += 1
    t0 = time.perf_counter()
    n = 0
    while n < reps or (time.perf_counter() - t0 < steady and n < 500):
        out = fn()
        n += 1
    return (time.perf_counter() - t0) / n, out


def _rate(flops, seconds):
    t = flops / seconds / 1e12
    return {"ms": seconds * 1e3, "tflops": t, "frac_of_fp64_mfma_peak": t / PEAK_FP64_MFMA_TFLOPS}


def device_configs(wl, dev, head_gp, head_theta, N):
    """The BASELINE.json configurations the headline does not cover, timed on the device AFTER the timed region through
    the public classes (steady state - see _timeit: warm-up and timed loop of at least 0.1 s each -, host wall clock around synchronous calls):
    config 2 (SE N = 8192 d = 8: fit, predict of 1024 points, LML, LML + gradient - regression.py:218-244,188-216,528-567),
    config 4 (SE N = 4096 d = 4: EI and -ln EI with gradient at 1000 candidates, one GpOptimiser.propose_evaluation -
    acquisition.py:76-125, optimisation.py:202-249) and the LML gradient at the metric's own size.  FLOP counts: potrf
    N^3/3, predict M N^2, LML + gradient N^3."""
    from inference_amd.gp import ExpectedImprovement, GpOptimiser, GpRegressor

    out = {}
    n2, d2, m2 = 8192, 8, 1024
    x, y, e = wl.synthetic_dataset(2, n2, d2)
    th = wl.timing_theta(wl.SE, y, d2)
    pts = wl.query_points(2, m2, d2)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th, device=dev)
    gp.prepare_gradient()
    fit, _ = _timeit(lambda: gp.set_hyperparameters(th))
    pred, _ = _timeit(lambda: gp(pts))
    lml, _ = _timeit(lambda: gp.marginal_likelihood(th))
    grad, _ = _timeit(lambda: gp.marginal_likelihood_gradient(th))
    out["config2"] = {
        "workload": f"GpRegressor SquaredExponential N={n2} d={d2}: set_hyperparameters / __call__ on M={m2} points / "
                    "marginal_likelihood / marginal_likelihood_gradient, fixed theta",
        "fit": _rate(n2**3 / 3.0, fit), "predict": _rate(m2 * float(n2) ** 2, pred),
        "lml": _rate(n2**3 / 3.0, lml), "lml_gradient": _rate(float(n2) ** 3, grad),
    }
    gp.engine.close()

    n4, d4 = 4096, 4
    x, y, e = wl.synthetic_dataset(4, n4, d4)
    th = wl.timing_theta(wl.SE, y, d4)
    cand = wl.query_points(4004, 1000, d4)
    opt = GpOptimiser(x, y, bounds=[(0.0, 1.0)] * d4, y_err=e, hyperpars=th, acquisition=ExpectedImprovement, device=dev)
    ei = opt.acquisition
    ei_v, _ = _timeit(lambda: ei.call_batch(cand))
    ei_g, _ = _timeit(lambda: ei.opt_func_gradient_batch(cand))

    def propose():
        np.random.seed(1)
        return opt.propose_evaluation()

    prop, where = _timeit(propose, reps=1, steady=0.0)
    out["config4"] = {
        "workload": f"GpOptimiser / ExpectedImprovement on SquaredExponential N={n4} d={d4}, fixed theta: 1000 candidates per "
                    f"call; one propose_evaluation = {n4} x 20 probes for the starting positions + {n4} L-BFGS-B runs in lockstep",
        "ei_1000_candidates": {"ms": ei_v * 1e3, "candidates_per_s": 1000 / ei_v},
        "minus_ln_ei_and_gradient_1000_candidates": {"ms": ei_g * 1e3, "candidates_per_s": 1000 / ei_g},
        "propose_evaluation": {"seconds": prop, "proposal": [float(v) for v in np.ravel(where)]},
    }
    opt.gp.engine.close()

    head_gp.prepare_gradient()
    hg, _ = _timeit(lambda: head_gp.marginal_likelihood_gradient(head_theta))
    out["lml_gradient_at_metric_size"] = {
        "workload": f"GpRegressor SquaredExponential N={N}: marginal_likelihood_gradient (K-build, potrf, L^-T by TRSM, "
                    "k-skipped SYRK, fused trace pass)", **_rate(float(N) ** 3, hg)}
    return out


def cpu_configs(wl, d_head):
    """The CPU oracle (kind "port": the reference's NumPy / LAPACK calls, oracle/gp_oracle.py) on the host cores at every
    BASELINE.json configuration's own size - one bounded run each, after the N = 4096 sample of
    cpu_baseline has warmed the BLAS pool.  Config 3: ONE of the 64 grid points (x 64 stated, not measured); config 4: the
    reference's per-candidate loop on a 50-candidate sample beside the batched restatement at 1000; config 5: LML
    evaluations per second of one process.  The LML gradient at N = 16384 is not timed (N^3 = 4.4 TFLOP of dtrtri-like
    work + 9 N x N gradient matrices: about two minutes)."""
    from oracle import gp_oracle as orc  # checker / baseline only

    out = {}

    def clock(fn):
        t0 = time.perf_counter()
        r = fn()
        return time.perf_counter() - t0, r

    # config 2
    n, d, m = 8192, 8, 1024
    x, y, e = wl.synthetic_dataset(2, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    pts = wl.query_points(2, m, d)
    t_fit, gp = clock(lambda: orc.OracleGp(x, y, e, kernel=orc.SE, hyperpars=th))
    t_pred, _ = clock(lambda: gp(pts))
    t_lml = t_fit  # (marginal_likelihood is the same K-build + cholesky + solves as the fit: not run a second time, round 6)
    del gp.K_xx, gp.L
    t_grad, _ = clock(lambda: gp.marginal_likelihood_gradient_lean(th))
    out["config2"] = {"fit_s": t_fit, "predict_s": t_pred, "lml_s": t_lml, "lml_gradient_s": t_grad,
                      "fit_gflops": n**3 / 3.0 / t_fit / 1e9, "lml_gradient_gflops": float(n) ** 3 / t_grad / 1e9,
                      "sample": f"one run each, SE N={n} d={d} M={m}; lml_s = fit_s (the same calls); gradient by the one-matrix-at-a-time form"}
    del gp
    # config 3: one grid point
    n, d = 16384, 16
    x, y, e = wl.synthetic_dataset(3, n, d)
    g0 = wl.theta_grid_cfg3(y, d)[0]
    o3 = orc.OracleGp(x, y, e, kernel=orc.RQ)
    t_one, _ = clock(lambda: o3.marginal_likelihood(g0))
    out["config3"] = {"one_grid_point_s": t_one, "grid_of_64_s_extrapolated": 64 * t_one, "lml_evals_per_s": 1.0 / t_one,
                      "sample": f"ONE marginal_likelihood of the 64-point grid, RQ N={n} d={d}; x 64 is an extrapolation"}
    del o3
    # config 4
    n, d = 4096, 4
    x, y, e = wl.synthetic_dataset(4, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    cand = wl.query_points(4004, 1000, d)
    gp = orc.OracleGp(x, y, e, kernel=orc.SE, hyperpars=th)
    mu_max = float(y.max())

    def loop_value():
        for p in cand[:50]:
            mu, sig = gp(p)
            orc.ei_value(mu, sig, mu_max)

    def loop_grad():
        for p in cand[:50]:
            mu, sig = gp(p)
            sm, sv = gp.spatial_derivatives(p)
            orc.ei_opt_func_gradient(mu, sig, sm, sv, mu_max)

    def batched():
        mu, sig = gp(cand)
        return orc.ei_value(mu, sig, mu_max)

    t_v, _ = clock(loop_value)
    t_g, _ = clock(loop_grad)
    t_b, _ = clock(batched)
    out["config4"] = {"ei_per_candidate_loop_candidates_per_s": 50 / t_v, "minus_ln_ei_gradient_loop_candidates_per_s": 50 / t_g,
                      "ei_batched_restatement_candidates_per_s": 1000 / t_b,
                      "sample": f"SE N={n} d={d}: the reference's call structure (one solve_triangular per candidate, "
                                "regression.py:205-216, 387-419) on 50 candidates; one TRSM over 1000 candidates for the batched form"}
    del gp
    # config 5
    n, d = 2048, 4
    x, y, e = wl.synthetic_dataset(5, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    gp = orc.OracleGp(x, y, e, kernel=orc.SE)
    rng = np.random.default_rng(0)
    ths = th + 0.05 * rng.standard_normal((12, th.size))
    gp.marginal_likelihood(ths[0])
    t_e, _ = clock(lambda: [gp.marginal_likelihood(t) for t in ths[1:]])
    out["config5"] = {"lml_evals_per_s": 11 / t_e,
                      "sample": f"11 marginal_likelihood calls in one process, SE N={n} d={d} (the reference runs one process per chain: "
                                "mcmc/parallel.py:127-136)"}
    return out



def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel, file it comes from): the committed PMC passes of this same
    command (profiles/rNN_pmc.json, written by tools/pmc_bench.sh + tools/make_profiles.py: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    coalesced reads on gfx950); (None, None) if absent."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["kernels"]["update128"]["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def rocprof_duration_ratio():
    """(rocprofv3's average duration of the dominant kernel / the in-kernel stamps' average over the same launches,
    file): from the committed kernel-trace summary of this command (profiles/rNN_pmc.json `cross_check`).  rocprofv3
    times a launch from its dispatch to its completion signal - the end-of-kernel write-back included -, the stamps
    from the first workgroup's first instruction to the last workgroup's last acknowledged store."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                x = json.load(f)["cross_check"]
            return x["rocprof_kernel_stats_avg_us_of_the_dominant_kernel"] / x["bench_stamp_avg_us_all_launches_of_this_kernel_name"], "profiles/" + name
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


def reference_rates():
    """What the chip sustains with nothing but the ingredient in question: builder-run probes, read from the committed file
    (None when it is absent) - context for `peak`, not a measurement of this run."""
    try:
        with open(os.path.join(ROOT, "profiles", "r05_sustained_rates.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def vendor_yardstick():
    """tools/vendor_yardstick.py in a child process (rocSOLVER / rocBLAS at the hot path's shapes; a side channel of the
    bench, never of the product): its `results` dict, or {"available": False, ...}.  BENCH_NO_VENDOR=1 skips it."""
    if os.environ.get("BENCH_NO_VENDOR"):
        return {"available": False, "why": "BENCH_NO_VENDOR"}
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vendor_yardstick.py")], capture_output=True, text=True,
                           timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as err:
        return {"available": False, "why": f"{type(err).__name__}: {err}"}


def flatten_for_the_record(line):
    """The driver's record keeps the SCALAR fields of `config`, `roofline` and `cpu_baseline` and drops nested values
    (BENCH_r05.json: extra_keys = ["configs", "sharded"], values gone).  Every headline number of the nested blocks is
    therefore repeated as a scalar here (tests/test_bench_cpu.py asserts that no nested value is a number's only home)."""
    cfg, roof, cpu = line["config"], line["roofline"], line.get("cpu_baseline")
    c = line.get("configs") or {}
    if "config2" in c:
        c2 = c["config2"]
        cfg.update(cfg2_fit_ms=c2["fit"]["ms"], cfg2_predict_ms=c2["predict"]["ms"], cfg2_lml_ms=c2["lml"]["ms"],
                   cfg2_lml_grad_ms=c2["lml_gradient"]["ms"], cfg2_fit_frac=c2["fit"]["frac_of_fp64_mfma_peak"],
                   cfg2_lml_frac=c2["lml"]["frac_of_fp64_mfma_peak"], cfg2_predict_frac=c2["predict"]["frac_of_fp64_mfma_peak"],
                   cfg2_lml_grad_frac=c2["lml_gradient"]["frac_of_fp64_mfma_peak"])
    if "config4" in c:
        c4 = c["config4"]
        cfg.update(cfg4_ei_ms=c4["ei_1000_candidates"]["ms"], cfg4_ei_grad_ms=c4["minus_ln_ei_and_gradient_1000_candidates"]["ms"],
                   cfg4_propose_s=c4["propose_evaluation"]["seconds"])
    if "lml_gradient_at_metric_size" in c:
        cfg.update(lml_grad_16k_ms=c["lml_gradient_at_metric_size"]["ms"],
                   lml_grad_16k_frac=c["lml_gradient_at_metric_size"]["frac_of_fp64_mfma_peak"])
    if "error" in c:
        cfg["configs_error"] = c["error"]
    sh = line.get("sharded") or {}
    if "gather" in sh:
        cfg["gather"] = sh["gather"]
    if "config3" in sh:
        cfg.update(cfg3_grid64_s=sh["config3"]["seconds"], cfg3_frac=sh["config3"]["frac_of_aggregate_fp64_mfma_peak"],
                   cfg3_lml_evals_per_s=sh["config3"]["lml_evals_per_s"], cfg3_checksum=sh["config3"]["checksum"])
    if "config5" in sh:
        cfg.update(cfg5_lml_evals_per_s=sh["config5"]["lml_evals_per_s"], cfg5_frac=sh["config5"]["frac_of_aggregate_fp64_mfma_peak"],
                   cfg5_chain_steps_per_s=sh["config5"]["chain_steps_per_s"], cfg5_seconds=sh["config5"]["seconds"])
    if "error" in sh:
        cfg["sharded_error"] = sh["error"]
    if "dataset_broadcast" in cfg:
        cfg["dataset_broadcast_ok"] = "identical" in str(cfg["dataset_broadcast"])
    v = line.get("vendor") or {}
    if v.get("available"):
        r = v["results"]
        cfg.update(vendor_potrf_16k_ms=r.get("potrf_16384", {}).get("ms_median"), vendor_potrf_8k_ms=r.get("potrf_8192", {}).get("ms_median"),
                   vendor_trsm_ms=r.get("trsm_16384_x_1024", {}).get("ms_median"),
                   vendor_syrk_tflops=r.get("syrk_15872_k512", {}).get("tflops_at_median"))
    else:
        cfg["vendor_available"] = False
    for row in roof.get("kernels") or []:
        name = row.get("kernel", "")
        key = ("kbuild" if name.startswith("kbuild") else "sweeps" if name.startswith("trsv") else
               "predict" if name.startswith("trsm_rows") else None)
        if key and "frac" in row:
            roof[key + "_frac"] = row["frac"]
            roof[key + "_avg_ms"] = row["avg_ms"]
        if name.startswith("panel chain"):
            roof["panel_chain_ms_per_step"] = row.get("total_ms_per_step")
    for blk in ("all_trailing", "flow_tail", "slices"):
        if blk in roof:
            roof[blk + "_tflops"] = roof[blk].get("achieved")
            roof[blk + "_ms_per_step"] = roof[blk].get("ms_per_step")
    same = roof.get("same_kernel_name_all_launches") or {}
    if roof.get("traffic") and same.get("algorithmic_bytes_per_launch_avg"):
        roof["algorithmic_bytes_per_launch_all_launches"] = same["algorithmic_bytes_per_launch_avg"]
        roof["traffic_over_algorithmic"] = roof["traffic"] / same["algorithmic_bytes_per_launch_avg"]
    rr = roof.get("reference_rates") or {}
    for k in ("fp64_mfma_from_registers_sustained_tflops", "this_kernel_alone_on_256_cus_sustained_tflops"):
        if k in rr:
            roof["ref_" + k] = rr[k]
    if cpu:
        sec = cpu.get("seconds") or {}
        if sec:
            cpu.update(seconds_total=sec.get("total"), k_build_s=sec.get("k_build"), potrf_s=sec.get("potrf"),
                       predict_s=sec.get("predict"))
        f = cpu.get("faithful") or {}
        if f:
            cpu.update(faithful_n=f.get("n"), faithful_gflops=f.get("value"), faithful_seconds=f.get("seconds"))
        sn = cpu.get("sample_n4096") or {}
        if sn:
            cpu["sample_n4096_gflops"] = sn.get("value")
        cc = cpu.get("configs") or {}
        if "config2" in cc:
            cpu.update(cfg2_fit_s=cc["config2"]["fit_s"], cfg2_lml_grad_s=cc["config2"]["lml_gradient_s"])
        if "config3" in cc:
            cpu["cfg3_one_grid_point_s"] = cc["config3"]["one_grid_point_s"]
        if "config4" in cc:
            cpu["cfg4_ei_loop_candidates_per_s"] = cc["config4"]["ei_per_candidate_loop_candidates_per_s"]
        if "config5" in cc:
            cpu["cfg5_lml_evals_per_s"] = cc["config5"]["lml_evals_per_s"]
    return line


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable with a plain copy)


def kernel_rows(eng, step, _lib, N, M):
    """Per-kernel roofline rows of the rest of a step from an extra pass with HIP events around the launches (it
    perturbs the overlap of the factorisation, so it runs AFTER the timed region and is not part of `value`)."""
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(2):
        step()
    eng.sync()
    kb, sv, ts, pn = (eng.profile_read(k) for k in (_lib.PROF_KBUILD, _lib.PROF_SOLVE, _lib.PROF_TRSM, _lib.PROF_PANEL))
    eng.profile_enable(0)
    rows = []

    def hbm(name, p, what):
        if p["ms"] > 0:
            a = p["bytes"] / (p["ms"] * 1e-3) / 1e9
            rows.append({"kernel": name, "bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": a / PEAK_HBM_GBS, "launches": p["launches"], "avg_ms": p["ms"] / p["launches"], "bytes": what})

    hbm("kbuild_kernel<true, SE> (K-build, lower tiles) + kbuild_kernel<false, SE> (cross-covariance of the query points)", kb,
        "4 N^2 B per fit (lower tiles written once) + 8 M N B per predict")
    hbm("trsv_fwd_flow_kernel + trsv_bwd_flow_kernel (alpha = L^-T L^-1 (y - mu))", sv, "4 N^2 B per sweep (L read once)")
    if ts["ms"] > 0:
        a = ts["flops"] / (ts["ms"] * 1e-3) / 1e12
        rows.append({"kernel": "trsm_rows_forward (predict: M right-hand sides, gemm_nt_kernel<0, *>)", "bound": "mfma",
                     "achieved": a, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_FP64_MFMA_TFLOPS,
                     "launches": ts["launches"], "avg_ms": ts["ms"] / ts["launches"], "flops": "M N^2 per predict"})
    if pn["ms"] > 0:
        rows.append({"kernel": "panel chain (potrf_diag_kernel + panel TRSM + inner K=128 updates; 32 CUs, hidden behind the "
                     "trailing update during the look-ahead regime, >= 52 trailing tile rows)", "bound": "latency", "launches": pn["launches"],
                     "total_ms_per_step": pn["ms"] / 2})
    return rows


def sharded_configs(args, wl, sharding, rank, world, local_rank, rdv, gather, eng_comm, _lib):
    """BASELINE configs 3 and 5 through the sharded drivers, AFTER the timed headline region (not part of `value`):
    config 3 = the 64-point RationalQuadratic hyper-parameter grid at N = 16384, d = 16 through
    `marginal_likelihood_sweep` (8 per GPU on 8 GPUs; reference: a Python loop over regression.py:528-542);
    config 5 = 64 ParallelTempering ladders x 8 temperatures over the GP log-marginal likelihood (N = 2048, d = 4)
    through `tempering_run`, whole ladders per rank (mcmc/parallel.py:127-136,190-231).  Both gather over the same
    path as the headline's result gather (`gather`).  Wall time = max over ranks (barrier before and after)."""
    from inference_amd.gp import GpRegressor, RationalQuadratic

    sharding.use_rendezvous(rdv)
    dev = local_rank % max(_lib.device_count(), 1)

    def timed(fn):
        if rdv is not None:
            rdv.barrier()
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        if rdv is not None:
            dt = max(float(v) for v in rdv.allgather_obj(dt, tag="sharded seconds"))
        return dt, out

    res = {"gather": gather, "note": "measured after the timed headline region; one process per GPU, contiguous blocks of "
           "units per rank, ONE all-gather of the results at the end"}
    grid_n = int(os.environ.get("BENCH_CFG3_POINTS", "64"))
    if grid_n > 0:
        n3, d3 = int(os.environ.get("BENCH_CFG3_N", "16384")), 16  # (the size is an aid of the tests only)
        x, y, e = wl.synthetic_dataset(3, n3, d3)
        grid = wl.theta_grid_cfg3(y, d3)[:grid_n]
        gp3 = GpRegressor(x, y, y_err=e, hyperpars=grid[0], kernel=RationalQuadratic, device=dev)
        gp3.engine.set_streams(2)
        sharding.marginal_likelihood_sweep(gp3, grid[: 2 * world], engine=eng_comm)  # warm-up: lanes, workspaces
        dt, vals = timed(lambda: sharding.marginal_likelihood_sweep(gp3, grid, engine=eng_comm))
        fl = len(grid) * n3**3 / 3.0
        res["config3"] = {
            "workload": f"RationalQuadratic N={n3} d={d3}: {len(grid)}-point hyper-parameter grid, marginal_likelihood_sweep",
            "seconds": dt, "lml_evals_per_s": len(grid) / dt, "tflops_aggregate": fl / dt / 1e12,
            "frac_of_aggregate_fp64_mfma_peak": fl / dt / 1e12 / (world * PEAK_FP64_MFMA_TFLOPS),
            "checksum": float(np.sum(vals)),
        }
        gp3.engine.close()
    n_lad = int(os.environ.get("BENCH_CFG5_LADDERS", "64"))
    if n_lad > 0:
        n5, d5, steps5 = 2048, 4, int(os.environ.get("BENCH_CFG5_STEPS", "10"))
        x, y, e = wl.synthetic_dataset(5, n5, d5)
        gp5 = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d5), device=dev)
        gp5.batch_independent_values(True)
        sharding.tempering_run(lambda k: wl.cfg5_ladder(gp5, k), world, 2, swap_interval=2, engine=eng_comm)  # warm-up
        dt, (state, evals) = timed(lambda: sharding.tempering_run(lambda k: wl.cfg5_ladder(gp5, k), n_lad, steps5,
                                                                  swap_interval=10, engine=eng_comm))
        chains = state.shape[0] * state.shape[1]
        res["config5"] = {
            "workload": f"{n_lad} ParallelTempering ladders x {state.shape[1]} temperatures, GibbsChain over the GP log-marginal "
                        f"likelihood (SE N={n5} d={d5}), {steps5} steps, swap interval 10, tempering_run (ladders built inside "
                        f"the timed call; the likelihood of their common start is evaluated once)",
            "seconds": dt, "chain_steps_per_s": chains * steps5 / dt, "lml_evals_per_s": evals / dt,
            "tflops_aggregate": evals / dt * (n5**3 / 3.0) / 1e12,
            "frac_of_aggregate_fp64_mfma_peak": evals / dt * (n5**3 / 3.0) / 1e12 / (world * PEAK_FP64_MFMA_TFLOPS),
            "checksum": float(np.sum(state[:, :, -1])),
        }
        gp5.engine.close()
    return res


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same script and relay rank
    0's JSON line.  The parent never touches the GPU (nothing GPU-related is imported before this point), the
    children are fresh interpreters (no fork of an initialised runtime, no exec from a GPU process); LOCAL_RANK
    selects the device modulo the number of visible devices, so on a box with fewer GPUs than ranks several
    ranks share a device (RCCL then refuses the duplicate device and the gather falls back to the rendezvous
    files — `config.parallelism` says which path ran)."""
    import secrets
    import subprocess

    env = dict(os.environ)
    # dmabuf IPC: on this image's host driver the legacy IPC mode fails RCCL's cross-process buffer registration with
    # `hipIpcGetMemHandle: invalid argument`; the image exports 0 already, setdefault keeps a caller's own choice
    # 
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["LOCAL_WORLD_SIZE"] = str(n_ranks)
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):  # (each child also pins itself: _rank_cpu_setup)
        env.setdefault(var, str(max(1, cores // n_ranks)))
    env.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=env.get("MASTER_PORT", "29511"),
               GPMI_RDV_KEY=f"bench_{os.getpid()}_{secrets.token_hex(6)}")
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)  # 2 s of GPU time: a sustained-clock figure, not a burst
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-metric-size", action="store_true",
                    help="skip the one CPU run at the metric's own size (about a minute of host time)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the config 3 / config 5 runs behind the timed region")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the device timings of BASELINE configs 2 and 4 and of the LML gradient behind the timed region")
    ap.add_argument("--no-cpu-configs", action="store_true",
                    help="skip the CPU oracle at the sizes of BASELINE configs 2-5 (about two minutes of host time)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)  # parent: spawns the ranks before anything touches the GPU, relays, exits
        return

    import workloads as wl
    from inference_amd import _lib

    _lib.load()  # system ROCm runtime first (see module docstring)
    from inference_amd import sharding
    from inference_amd.gp import GpRegressor, SquaredExponential

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the rendezvous must outlive the RCCL bootstrap watchdog below: a rank whose bootstrap returned at once waits in
    # allgather_obj for the ranks still inside th.join
    rccl_timeout = float(os.environ.get("GPMI_BENCH_RCCL_TIMEOUT", "120"))
    # (two watchdog waits at most: the bootstrap and the start-up broadcast)
    rdv = sharding.FileRendezvous(rank, world, timeout=2 * rccl_timeout + 90.0) if world > 1 else None
    try:
        run(args, wl, sharding, GpRegressor, SquaredExponential, _lib, rank, world, local_rank, rdv)
    finally:
        if rdv is not None:
            rdv.close()
        if STUCK:  # a thread is still inside RCCL: tearing the library down under it could block for ever
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0 if sys.exc_info()[0] is None else 1)
        # destroy the device contexts here, while the interpreter is fully alive: under rocprofv3 a context that is
        # only torn down during interpreter shutdown ended in a SIGSEGV inside the runtime's static destructors
        # (after the profile had been written)
        _lib._close_all_handles()


STUCK = []  # non-empty: a thread of this rank is still inside the RCCL bootstrap


def run(args, wl, sharding, GpRegressor, SquaredExponential, _lib, rank, world, local_rank, rdv):
    N, d, M = args.n, args.d, args.m
    x, y, e = wl.synthetic_dataset(2, N, d)
    thetas = wl.theta_set(wl.SE, y, d, max(world, 1), seed=11)
    theta = thetas[rank % len(thetas)] if world > 1 else thetas[0]
    pts = wl.query_points(2, M, d)

    dev_index = local_rank % max(_lib.device_count(), 1)
    gp = GpRegressor(x, y, y_err=e, hyperpars=theta, kernel=SquaredExponential, device=dev_index)
    eng = gp.engine
    gather = "none"
    if world > 1:
        # RCCL bootstrap + a first all-gather under a watchdog: a refusal (ranks sharing a device) or a bootstrap
        # that never returns must not cost the scaling run - the gather is 4 doubles per rank and falls back to the
        # rendezvous files; a rank left with a thread inside RCCL skips the library teardown at exit (STUCK)
        import threading

        box = {}
        rccl_limit = float(os.environ.get("GPMI_BENCH_RCCL_TIMEOUT", "120"))

        def bootstrap():
            try:
                sharding.init_device_comm_files(eng, rdv)
                eng.comm_allgather(np.zeros(1))
                box["ok"] = True
            except Exception as err:
                box["ok"], box["why"] = False, f"{type(err).__name__}: {err}"

        # ranks that share a device (bench.py --gpus 2 on a 1-GPU box) never try: RCCL refuses duplicate devices, and
        # while one rank is refused at once another can sit in the bootstrap waiting for it until the watchdog fires
        import socket

        # (a physical identity: with one HIP_VISIBLE_DEVICES mask per rank every rank's device has index 0)
        ident = _lib.device_identity(dev_index) or (
            f"index {dev_index} under HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')} "
            f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')}")
        where = rdv.allgather_obj(f"{socket.gethostname()}:{ident}", tag="device identity")
        if len(set(where)) < world:
            ok, why = False, "ranks share a device (RCCL refuses duplicate devices)"
        else:
            th = threading.Thread(target=bootstrap, daemon=True)
            th.start()
            th.join(rccl_limit)  # < the rendezvous' own time limit (main)
            if th.is_alive():
                STUCK.append(True)
                ok, why = False, "RCCL bootstrap did not return within the time limit"
            else:
                ok, why = box.get("ok", False), box.get("why", "")
        # every rank must take the same path
        oks = rdv.allgather_obj(ok, tag="rccl bootstrap")
        gather = "rccl" if all(oks) else f"file-fallback ({'; '.join(str(o) for o in oks)}{'' if ok else ' ' + why})"
        if gather == "rccl":
            # the start-up distribution of SURVEY section 8(e) on the live communicator: rank 0's data set to every rank by
            # ncclBroadcast, compared with the copy each rank generated itself; and the rank count RCCL itself reports.
            # Under the same watchdog as the bootstrap: a rank that raises before it enters the collective would leave the
            # others inside hipStreamSynchronize for ever
            def startup():
                try:
                    RCCL_INFO["rccl_ranks_seen"] = eng.comm_count()
                    bx, by, be = sharding.broadcast_dataset(x, y, e, comm=eng) if rank == 0 else sharding.broadcast_dataset(comm=eng)
                    same = bool(np.array_equal(bx, x) and np.array_equal(by, y) and np.array_equal(be, e))
                    RCCL_INFO["dataset_broadcast"] = "ncclBroadcast from rank 0: " + (
                        "identical to the locally generated copy" if same else "DIFFERS from the locally generated copy")
                except Exception as err:
                    RCCL_INFO["dataset_broadcast"] = f"failed: {type(err).__name__}: {err}"

            th = threading.Thread(target=startup, daemon=True)
            th.start()
            th.join(rccl_limit)
            alive = th.is_alive()
            if alive:
                STUCK.append(True)
                RCCL_INFO["dataset_broadcast"] = "did not return within the time limit"
            # a rank whose collective never returned takes every rank off the communicator
            if any(rdv.allgather_obj(alive, tag="rccl start-up")):
                gather = "file-fallback (the start-up broadcast over RCCL did not return on every rank)"
            RCCL_INFO["rccl_ranks_match"] = RCCL_INFO.get("rccl_ranks_seen") == world

    def allgather(vec):
        if gather == "rccl":
            return eng.comm_allgather(vec)  # RCCL over xGMI: the only collective of the path
        return np.array(rdv.allgather_obj(np.asarray(vec, dtype=float), tag=f"values {np.size(vec)}"))

    def step():
        gp.set_hyperparameters(theta)  # K-build + potrf + alpha
        mu, sig = gp(pts)  # cross-covariance + TRSM + reductions
        # plain NumPy reductions only: a BLAS call here (np.linalg.norm -> OpenBLAS nrm2) starts OpenBLAS's
        # spinning worker pool, which starves the HIP runtime's completion handling and doubled the
        # step time from the next step on (round-2 probe, profiles/HISTORY.md)
        return np.array([gp._logdet, float(np.sqrt(np.sum(gp.alpha**2))), float(mu.sum()), float(sig.sum())])

    for _ in range(args.warmup):
        step()

    def fence():
        eng.sync()  # every stream of the library on this device (nothing else runs on the GPU)
        if world > 1:
            allgather(np.zeros(1))  # barrier over all ranks

    if not os.environ.get("BENCH_NO_PROF"):
        eng.profile_enable(2 << _lib.PROF_SYRK)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    marks = []
    results = []
    for _ in range(args.steps):
        results.append(step())
        marks.append(time.perf_counter())
    # the ranks' results meet ONCE, still inside the timed region (the units are independent: a gather per step would
    # only make every rank wait for the slowest one at every step - and cost a file round trip per step whenever the
    # communicator could not be created)
    res = allgather(np.concatenate(results)) if world > 1 else results[-1]
    fence()
    dt = time.perf_counter() - t0
    if os.environ.get("BENCH_STEP_TIMES") and rank == 0:  # debugging aid: host-side completion time of each step
        print("step ms:", [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)], file=sys.stderr)
    prof = eng.profile_read(_lib.PROF_SYRK)
    prof_rest = eng.profile_read(_lib.PROF_SYRK_REST)
    prof_slice = eng.profile_read(_lib.PROF_SYRK_SLICE)
    prof_flow = eng.profile_read(_lib.PROF_FLOW)
    clock = eng.profile_clock()
    eng.profile_enable(0)

    if world > 1:
        dt = float(np.max(allgather(np.array([dt]))))  # MAX over ranks

    flops_step = N**3 / 3.0 + M * float(N) ** 2
    value = flops_step * args.steps * world / dt / 1e9

    sharded = None
    if not args.no_sharded:
        try:
            sharded = sharded_configs(args, wl, sharding, rank, world, local_rank, rdv, gather,
                                      eng if gather == "rccl" else None, _lib)
        except Exception as err:  # the headline line must not be lost to a failure behind the timed region
            sharded = {"error": f"{type(err).__name__}: {err}"}
            import traceback

            print(f"bench.py rank {rank}: the sharded configurations failed: {type(err).__name__}: {err}\n"
                  + "".join(traceback.format_exc(limit=8)), file=sys.stderr, flush=True)
            # this rank skips the exchanges the others are in: tell them, or they wait for the time limit and then read this
            # rank's closing barrier as the payload of the exchange it skipped
            if rdv is not None and not getattr(rdv, "aborted", False):
                rdv.abort(f"sharded configurations: {type(err).__name__}: {err}")

    if rank == 0:
        ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
        # the slices run on the panel stream's 32 CUs WHILE the launches of the other two classes run on the update
        # stream: their FLOPs count, their durations overlap the others' and do not add to the update stream's time
        all_ms = prof["ms"] + prof_rest["ms"]
        all_fl = prof["flops"] + prof_rest["flops"] + prof_slice["flops"]
        ach_slice = prof_slice["flops"] / (prof_slice["ms"] * 1e-3) / 1e12 if prof_slice["ms"] > 0 else 0.0
        ach_all = all_fl / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
        peak_at_clock = PEAK_FP64_MFMA_TFLOPS * clock / 2.4 if clock > 0 else None
        traffic, traffic_src = pmc_traffic()
        rp_ratio, rp_src = rocprof_duration_ratio()
        line = {
            "schema": "r06",  # r06: roofline.frac = rocprofv3-duration figure (stamps in frac_stamps); cpu_baseline.value = median of 3
                              # runs at the metric's size; scalar copies of every nested headline number (flatten_for_the_record)
            "metric": f"GpRegressor fit+predict wall-time and GFLOP/s at N={N}, d={d}; % fp64 MFMA peak",
            "value": value,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"GpRegressor SquaredExponential fit (fixed theta) + predict, N={N} d={d} M={M}",
                "flop_per_step": flops_step,
                "pct_fp64_mfma_peak_whole_step": 100.0 * value / world / 1e3 / PEAK_FP64_MFMA_TFLOPS,
                "parallelism": f"{world} independent hyper-parameter evaluations (one per GPU), result all-gather: {gather}",
                "gathered_values": int(np.size(res)),  # steps x 4 per rank, all ranks' in ONE collective inside the timed region
                # (schema note: until round 3 the ranks met in one all-gather PER STEP; multi-rank values of r03 and earlier
                # paid a wait for the slowest rank - and a file round trip on the fallback - every step and are not comparable)
                "gather_schedule": "once_at_end",
                **({"rank_cpu": RANK_CPU} if RANK_CPU else {}),
                **RCCL_INFO,
            },
            "roofline": {
                "kernel": "gemm_dma_kernel<1, 0> = <TILES_LOWER, OP_SUB>: 128x128 tiles, operands through an LDS-DMA ring (potrf trailing SYRK update, K=512; the full rounds of every launch with >= 384 tiles)",
                "timing": "in-kernel s_memrealtime stamps per launch: first workgroups' start -> last workgroup's end behind its epilogue stores",
                "cu_mask": "launches of the look-ahead regime run on 224 of 256 CUs (the other 32 factor the next panel)",
                "bound": "mfma",
                # `frac` / `achieved` with the kernel's duration as rocprofv3 --kernel-trace sees it (dispatch -> completion
                # signal; the committed profile's ratio to the stamps over the same launches, profiles/rNN_pmc.json cross_check):
                # the figure profiles/ reproduces.  `*_stamps`: the live in-kernel measurement of THIS run.
                "achieved": ach / rp_ratio if rp_ratio else ach,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (ach / rp_ratio if rp_ratio else ach) / PEAK_FP64_MFMA_TFLOPS,
                "frac_source": (f"in-kernel stamps of this run / {rp_ratio:.4f} (rocprofv3 duration ratio, {rp_src})" if rp_ratio
                                else "in-kernel stamps of this run (no committed rocprofv3 cross-check found)"),
                "frac_rocprofv3": (ach / rp_ratio / PEAK_FP64_MFMA_TFLOPS) if rp_ratio else None,
                "achieved_stamps": ach,
                "frac_stamps": ach / PEAK_FP64_MFMA_TFLOPS,
                "duration_ratio_rocprofv3_over_stamps": rp_ratio,
                # ... and over EVERY launch of this kernel name (the slices on the panel stream's 32 CUs included: what a
                # per-name kernel_stats row blends; profiles/rNN_bench_kernel_stats_by_queue.csv splits the row by queue)
                "achieved_all_launches": ((prof["flops"] + prof_slice["flops"]) / ((prof["ms"] + prof_slice["ms"]) * 1e-3) / 1e12)
                if prof["ms"] + prof_slice["ms"] > 0 else 0.0,
                "frac_all_launches": ((prof["flops"] + prof_slice["flops"]) / ((prof["ms"] + prof_slice["ms"]) * 1e-3) / 1e12
                                      / PEAK_FP64_MFMA_TFLOPS) if prof["ms"] + prof_slice["ms"] > 0 else 0.0,
                "traffic": traffic,
                "traffic_source": f"{traffic_src}: separate rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run; per launch, averaged over every launch of this kernel name (slices included: compare with same_kernel_name_all_launches.algorithmic_bytes_per_launch_avg)" if traffic_src else None,
                "launches": prof["launches"],
                "avg_launch_ms": prof["ms"] / max(prof["launches"], 1),
                "flop_per_launch_avg": prof["flops"] / max(prof["launches"], 1),
                # rocprofv3's kernel_stats row for this kernel NAME also holds the slice launches (same kernel on the panel
                # stream's 32 CUs): the average to hold against that row, and against the per-launch PMC traffic
                "same_kernel_name_all_launches": {
                    "launches": prof["launches"] + prof_slice["launches"],
                    "avg_launch_ms": (prof["ms"] + prof_slice["ms"]) / max(prof["launches"] + prof_slice["launches"], 1),
                    "flop_per_launch_avg": (prof["flops"] + prof_slice["flops"]) / max(prof["launches"] + prof_slice["launches"], 1),
                    "algorithmic_bytes_per_launch_avg": (prof["bytes"] + prof_slice["bytes"]) / max(prof["launches"] + prof_slice["launches"], 1),
                },
                "clock_ghz": clock,
                "peak_at_clock": peak_at_clock,
                "frac_at_clock": (ach / peak_at_clock) if peak_at_clock else None,
                # context for `peak` (builder-run probes read from profiles/, NOT measured in this run; None when absent)
                "reference_rates": reference_rates(),
                "all_trailing": {
                    "what": "every trailing-update launch of the factorisation: the 128x128-tile kernel above plus the 64x64-tile remainders and the launches below 384 tiles (gemm_nt_kernel<1, 0, 0, 64, 64>) on the update stream, plus the slices below, which run concurrently on the panel stream (FLOPs counted, time overlapped)",
                    "achieved": ach_all,
                    "frac": ach_all / PEAK_FP64_MFMA_TFLOPS,
                    "launches": prof["launches"] + prof_rest["launches"] + prof_slice["launches"],
                    "ms_per_step": all_ms / max(args.steps, 1),
                    "flop_per_step": all_fl / max(args.steps, 1),
                },
                "flow_tail_scope": "one flag-ordered tail per device at a time, process-wide: of several evaluation lanes factoring side by side (the config-3 sweep) only one gets it, the others take the stream-ordered schedule (same bits)",
                "flow_tail": {
                    "what": "the chain-bound last tile rows of the factorisation (below 52 trailing tile rows) as ONE persistent tile-task launch on the update stream's 224 CUs beside the bare panel chain (csrc/potrf_flow.hip): FLOPs of its K = 128 / K = 512 update tasks over the launch's whole duration - the launch waits for the chain most of the time, so this is the overlap achieved, not a kernel rate; not part of all_trailing",
                    "achieved": (prof_flow["flops"] / (prof_flow["ms"] * 1e-3) / 1e12) if prof_flow["ms"] > 0 else 0.0,
                    "launches": prof_flow["launches"],
                    "ms_per_step": prof_flow["ms"] / max(args.steps, 1),
                    "flop_per_step": prof_flow["flops"] / max(args.steps, 1),
                },
                "slices": {
                    "what": "the last tiles of a trailing update, run by the same 128x128-tile kernel on the 32 CUs reserved for the panel chain once the chain is through (peak share 32 / 256)",
                    "achieved": ach_slice,
                    "frac_of_share": ach_slice / (PEAK_FP64_MFMA_TFLOPS * 32 / 256),
                    "launches": prof_slice["launches"],
                    "ms_per_step": prof_slice["ms"] / max(args.steps, 1),
                    "flop_per_step": prof_slice["flops"] / max(args.steps, 1),
                },
            },
        }
        if sharded is not None:
            line["sharded"] = sharded
        if world == 1:
            line["roofline"]["kernels"] = kernel_rows(eng, step, _lib, N, M)
        if world == 1 and not args.no_configs:
            try:
                line["configs"] = device_configs(wl, dev_index, gp, theta, N)
            except Exception as err:  # the headline line must not be lost to a failure behind the timed region
                line["configs"] = {"error": f"{type(err).__name__}: {err}"}
        if world == 1 and not args.no_cpu_baseline:
            # the bounded N = 4096 sample first: it also warms the BLAS pool for the run at the metric's own size, whose
            # figure is the one `value` carries (round 5; until round 4 `value` was the sample's)
            sample = cpu_baseline(4096, d, 256)  # ~10-15 s of host work in all
            if args.no_cpu_metric_size:
                line["cpu_baseline"] = sample
            else:
                full = cpu_at_metric_size(N, d, M)
                line["cpu_baseline"] = {
                    "value": full["value"], "unit": "GFLOP/s", "cores": sample["cores"], "kind": "port",
                    "sample": full["sample"] + "; " + sample["sample"].split("; ", 1)[-1],
                    "seconds": full["seconds"], "potrf_gflops": full["potrf_gflops"],
                    "runs": full["runs"], "warmup_runs": 1, "totals_s": full["totals_s"],
                    "gpu_over_cpu": value / full["value"],
                    "blas_threads": sample["blas_threads"], "host_ram_gb": sample["host_ram_gb"],
                    "logical_cpus": os.cpu_count(),
                    "sample_n4096": {k: sample[k] for k in ("value", "unit", "sample")},
                    "faithful": sample["faithful"],
                }
            if not args.no_cpu_configs and wall_left() < 110:
                line["cpu_baseline"]["configs"] = {"skipped": f"{wall_left():.0f} s of the run's wall-clock budget left (needs ~100 s)"}
            elif not args.no_cpu_configs:
                try:
                    line["cpu_baseline"]["configs"] = cpu_configs(wl, d)
                except Exception as err:
                    line["cpu_baseline"]["configs"] = {"error": f"{type(err).__name__}: {err}"}
        if world == 1 and not args.no_configs:
            line["vendor"] = vendor_yardstick()
        print(json.dumps(flatten_for_the_record(line)), flush=True)


if __name__ == "__main__":
    main()
