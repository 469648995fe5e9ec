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
import os
import pickle
import time

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


class FileRendezvous:
    """Torch-free bootstrap for the ranks of ONE node (what `torch.distributed.run --nnodes=1`
    launches): exchanges small byte strings through a per-job directory in /tmp.  Used to hand the
    RCCL unique id to every rank — and as a last-resort gather if RCCL cannot be initialised — so
    that a multi-GPU job never has to import torch (importing it loads torch's bundled HIP / HSA
    runtime beside the system one, which RCCL then picks up uninitialised).
    All ranks are children of the same launcher process, whose pid keys the directory."""

    def __init__(self, rank=None, world=None, timeout=90.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.timeout = timeout
        key = os.environ.get("GPMI_RDV_KEY") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
        self.dir = os.path.join(os.environ.get("GPMI_RDV_DIR", "/tmp"), f"gpmi_rdv_{key}")
        os.makedirs(self.dir, exist_ok=True)
        self.round = 0

    def allgather_obj(self, obj):
        """Every rank contributes a picklable object; returns the list in rank order."""
        self.round += 1
        mine = os.path.join(self.dir, f"r{self.round}.{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f)
        os.replace(mine + ".tmp", mine)
        out, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, f"r{self.round}.{r}")
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rank {r} did not reach rendezvous round {self.round}")
                time.sleep(0.002)
            with open(path, "rb") as f:
                out.append(pickle.load(f))
        return out

    def broadcast_obj(self, obj, src=0):
        return self.allgather_obj(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather_obj(0)

    def close(self):
        """Leave the rendezvous: every rank removes its own files, rank 0 the directory."""
        self.barrier()
        time.sleep(0.05)  # let the slowest reader of the last round finish
        for name in os.listdir(self.dir):
            if name.endswith(f".{self.rank}"):
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass
        if self.rank == 0:
            t0 = time.time()
            while time.time() - t0 < 2.0:
                try:
                    os.rmdir(self.dir)
                    break
                except OSError:
                    time.sleep(0.02)


def init_device_comm_files(engine, rdv: "FileRendezvous"):
    """RCCL communicator bootstrap over a FileRendezvous (no torch in the process)."""
    uid = rdv.broadcast_obj(engine.comm_unique_id() if rdv.rank == 0 else None)
    engine.comm_init(rdv.rank, rdv.world, uid)
