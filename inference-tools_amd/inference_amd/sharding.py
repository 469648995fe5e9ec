"""
Multi-GPU sharding of independent hyper-parameter evaluations (BASELINE configs 3
and 5; reference counterpart: the `multiprocessing.Pool.map` task farming of
regression.py:597-601 and the per-chain processes of mcmc/parallel.py:127-136).

One process per GPU.  `torch.distributed` supplies the process group (rank /
world size / barrier; backend "gloo", CPU only — `import inference_amd` must come
BEFORE `import torch` so that the process runs on the system ROCm runtime, see
DESIGN.md section 6); the path's one collective — an all-gather of the per-rank
results, a few doubles per evaluation, latency-bound — goes over RCCL / xGMI
through the library's own communicator (`gpmi_comm_*`, bootstrapped by
broadcasting the RCCL unique id over the process group).  Without a device
communicator (CPU tests) the gather falls back to the process group itself.
There is no data-path collective: x, y are tiny and every rank builds them itself.
"""
import numpy as np


def world():
    """(rank, world_size) of the default process group, (0, 1) if not initialised."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (blocks differ by at most one)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_device_comm(engine):
    """Create the RCCL communicator of `engine` (a GpEngine): rank 0 makes the unique id, the
    process group broadcasts it, every rank joins."""
    rank, size = world()
    uid = [engine.comm_unique_id() if rank == 0 else None]
    if size > 1:
        import torch.distributed as dist

        dist.broadcast_object_list(uid, src=0)
    engine.comm_init(rank, size, uid[0])


def sharded_map(batch_fn, items, width: int = 1, engine=None):
    """Evaluate `batch_fn(items[lo:hi]) -> (hi - lo, width)` on every rank's block and all-gather
    (over RCCL when `engine` carries a device communicator, else over the process group).

    Returns the (n_items, width) array in the original order on every rank."""
    items = np.asarray(items)
    rank, size = world()
    lo, hi = shard_bounds(len(items), size, rank)
    local = np.asarray(batch_fn(items[lo:hi]), dtype=np.float64).reshape(hi - lo, width)
    if size == 1:
        return local
    if engine is not None and getattr(engine, "comm_world", 0) == size:
        per = -(-len(items) // size)
        send = np.zeros((per, width))
        send[: hi - lo] = local
        got = engine.comm_allgather(send).reshape(size, per, width)
        return np.concatenate([got[r, : b - a] for r, (a, b) in
                               enumerate(shard_bounds(len(items), size, r) for r in range(size))], axis=0)
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    per = -(-len(items) // size)  # equal-sized slots for all_gather
    buf = torch.zeros(per, width, dtype=torch.float64, device=dev)
    if hi > lo:
        buf[: hi - lo] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(size)]
    dist.all_gather(out, buf)
    parts = []
    for r, t in enumerate(out):
        a, b = shard_bounds(len(items), size, r)
        parts.append(t[: b - a].cpu().numpy())
    return np.concatenate(parts, axis=0)


def marginal_likelihood_sweep(gp, thetas):
    """Config 3: log-marginal likelihood of `gp` at every row of `thetas`, rows sharded over the
    ranks (each rank drives its own GPU), results all-gathered."""
    eng = gp.engine if getattr(gp.engine, "comm_world", 0) > 1 else None
    return sharded_map(lambda th: gp.marginal_likelihood_batch(th), np.asarray(thetas, dtype=float), engine=eng)[:, 0]
