"""
Multi-GPU sharding of independent hyper-parameter evaluations (BASELINE configs 3
and 5; reference counterpart: the `multiprocessing.Pool.map` task farming of
regression.py:597-601 and the per-chain processes of mcmc/parallel.py:127-136).

One process per GPU.  The path's one collective — an all-gather of the per-rank
results, a few doubles per evaluation, latency-bound — goes over RCCL / xGMI
through the library's own communicator (`gpmi_comm_*`).  The bootstrap (rank /
world size / handing round the RCCL unique id) is either a `torch.distributed`
process group (backend "gloo", CPU only — `import inference_amd` must come BEFORE
`import torch` so that the process runs on the system ROCm runtime, see DESIGN.md
section 6) or the torch-free `FileRendezvous` below (what bench.py uses).  Without a
device communicator (CPU tests) the gather falls back to the bootstrap channel.
There is no data-path collective.  The data set itself (x, y, y_err: a few hundred kilobytes) is either built by every
rank from the same generator or handed out once at start-up by `broadcast_dataset` (one ncclBroadcast).

Sharded units:
  * `marginal_likelihood_sweep`  config 3: a theta grid, contiguous blocks per rank;
  * `multistart_sweep`           L-BFGS starts of the hyper-parameter search
                                 (regression.py:597-601), blocks per rank, one gather of (theta*, f*);
  * `tempering_run`              config 5: whole ParallelTempering ladders per rank (every swap stays
                                 GPU-local, parallel.py:190-231), one gather of (theta, log-prob).
"""
import base64
import json
import os
import stat
import time

import numpy as np


# ---------------------------------------------------------------------------------------------
# process-group abstraction: torch.distributed if initialised, else a FileRendezvous, else serial
# ---------------------------------------------------------------------------------------------
_default_rdv = None


def use_rendezvous(rdv):
    """Make `rdv` (a FileRendezvous, or None) the bootstrap channel of this process."""
    global _default_rdv
    _default_rdv = rdv


def world():
    """(rank, world_size): the torch.distributed default group, else the active FileRendezvous, else (0, 1)."""
    try:
        import sys

        if "torch" in sys.modules:  # never import torch on behalf of a torch-free job
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    if _default_rdv is not None:
        return _default_rdv.rank, _default_rdv.world
    return 0, 1


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (blocks differ by at most one)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_device_comm(engine):
    """Create the RCCL communicator of `engine` (a GpEngine): rank 0 makes the unique id, the
    bootstrap channel broadcasts it, every rank joins."""
    rank, size = world()
    if _using_torch():
        import torch.distributed as dist

        uid = [engine.comm_unique_id() if rank == 0 else None]
        if size > 1:
            dist.broadcast_object_list(uid, src=0)
        engine.comm_init(rank, size, uid[0])
    elif _default_rdv is not None:
        init_device_comm_files(engine, _default_rdv)
    else:
        engine.comm_init(0, 1, engine.comm_unique_id())


def broadcast_dataset(x=None, y=None, y_err=None, comm=None, src: int = 0):
    """(x, y, y_err) of rank `src` on every rank: the start-up distribution of SURVEY section 8(e).  Over RCCL (one
    ncclBroadcast of n (d + 2) doubles, after a three-number header) when `comm` - a `DeviceComm` or an engine whose
    communicator spans the job - is given, else over the bootstrap channel.  The other ranks pass nothing.  y_err may be
    None on the source (it then is None everywhere)."""
    rank, size = world()
    if not 0 <= src < size:
        raise ValueError(f"broadcast_dataset: source rank {src} is not a rank of this job (world size {size})")
    err = None
    if rank == src:
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        err = None if y_err is None else np.ascontiguousarray(y_err, dtype=np.float64).ravel()
        if y.size != x.shape[0] or (err is not None and err.size != y.size):
            raise ValueError("broadcast_dataset: x, y, y_err disagree in length")
    if size == 1:
        return x, y, err
    if comm is not None and getattr(comm, "comm_world", 0) != size:
        # ranks that disagree about the channel would dead-lock, one inside ncclBroadcast and one in the bootstrap
        # broadcast: a communicator that does not span the job is an error, never a silent fall-back
        raise ValueError(f"broadcast_dataset: the communicator spans {getattr(comm, 'comm_world', 0)} rank(s), the job "
                         f"{size}; pass comm=None on EVERY rank to use the bootstrap channel")
    if comm is not None:
        head = np.array([x.shape[0], x.shape[1], 0.0 if err is None else 1.0]) if rank == src else np.zeros(3)
        n, d, has_err = (int(v) for v in comm.comm_broadcast(head, src))
        flat = np.zeros(n * (d + 1 + has_err))
        if rank == src:
            flat[: n * d] = x.ravel()
            flat[n * d : n * (d + 1)] = y
            if has_err:
                flat[n * (d + 1) :] = err
        flat = comm.comm_broadcast(flat, src)
        return (flat[: n * d].reshape(n, d).copy(), flat[n * d : n * (d + 1)].copy(),
                flat[n * (d + 1) :].copy() if has_err else None)
    payload = (x, y, err) if rank == src else None
    if _using_torch():
        import torch.distributed as dist

        box = [payload]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    got = _default_rdv.broadcast_obj(payload, src)
    xs, ys, es = got
    return (np.asarray(xs, dtype=np.float64).reshape(len(ys), -1), np.asarray(ys, dtype=np.float64),
            None if es is None else np.asarray(es, dtype=np.float64))


def _using_torch():
    import sys

    if "torch" not in sys.modules:
        return False
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _gather_rows(local, per, width, engine):
    """All-gather of one (per, width) float64 block per rank -> (world, per, width) on every rank."""
    rank, size = world()
    send = np.zeros((per, width))
    send[: local.shape[0]] = local
    if engine is not None and getattr(engine, "comm_world", 0) == size:
        return engine.comm_allgather(send).reshape(size, per, width)  # RCCL over xGMI
    if _using_torch():
        import torch
        import torch.distributed as dist

        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        buf = torch.from_numpy(send).to(dev)
        out = [torch.empty_like(buf) for _ in range(size)]
        dist.all_gather(out, buf)
        return np.stack([t.cpu().numpy() for t in out])
    return np.stack([np.asarray(a, dtype=float).reshape(per, width)
                     for a in _default_rdv.allgather_obj(send, tag=f"rows {per}x{width}")])


def sharded_map(batch_fn, items, width: int = 1, engine=None):
    """Evaluate `batch_fn(items[lo:hi]) -> (hi - lo, width)` on every rank's block and all-gather
    (over RCCL when `engine` carries a device communicator, else over the bootstrap channel).
    A rank whose block is empty (more ranks than items) skips `batch_fn` and contributes nothing.

    Returns the (n_items, width) array in the original order on every rank."""
    items = np.asarray(items)
    rank, size = world()
    n = len(items)
    lo, hi = shard_bounds(n, size, rank)
    if hi > lo:
        local = np.asarray(batch_fn(items[lo:hi]), dtype=np.float64).reshape(hi - lo, width)
    else:
        local = np.empty((0, width))
    if size == 1:
        return local
    per = max(-(-n // size), 1)  # equal-sized slots for the all-gather
    got = _gather_rows(local, per, width, engine)
    blocks = [shard_bounds(n, size, r) for r in range(size)]
    return np.concatenate([got[r, : b - a] for r, (a, b) in enumerate(blocks)], axis=0)


def _comm_engine(gp):
    eng = getattr(gp, "engine", None)
    return eng if eng is not None and getattr(eng, "comm_world", 0) > 1 else None


def marginal_likelihood_sweep(gp, thetas, engine=None):
    """Config 3: log-marginal likelihood of `gp` at every row of `thetas`, rows sharded over the
    ranks (each rank drives its own GPU), results all-gathered - over the device communicator of
    `engine` (default: the regressor's own engine) when it has one, else over the bootstrap channel."""
    return sharded_map(lambda th: gp.marginal_likelihood_batch(th), np.atleast_2d(np.asarray(thetas, dtype=float)),
                       engine=engine if engine is not None else _comm_engine(gp))[:, 0]


def multistart_sweep(gp, starting_positions):
    """The multi-start L-BFGS-B hyper-parameter search with its starts block-sharded over the ranks
    (the reference farms them over a multiprocessing.Pool, regression.py:597-601): every rank runs
    `gp.launch_bfgs` on its block, ONE all-gather returns (theta*, f*) of every start in order."""
    starts = np.atleast_2d(np.asarray(starting_positions, dtype=float))
    P = starts.shape[1]

    def run(block):
        out = np.empty((len(block), P + 1))
        for k, x0 in enumerate(block):
            res = gp.launch_bfgs(x0)
            out[k, :P] = res[0]
            out[k, P] = float(np.ravel(res[1])[0])
        return out

    got = sharded_map(run, starts, width=P + 1, engine=_comm_engine(gp))
    return got[:, :P], got[:, P]


def tempering_run(make_ladder, n_ladders: int, n_steps: int, swap_interval: int = 10, batch_posterior=None,
                  engine=None):
    """Config 5: `n_ladders` independent ParallelTempering ladders, whole ladders block-partitioned
    over the ranks (the reference runs one process per chain and pipes every position to the parent
    at each swap, parallel.py:127-136,190-231; here every swap is local to the rank that owns the
    ladder).  `make_ladder(k)` builds ladder k — it must seed the ladder's generators from k so
    that the run does not depend on the world size.  Every rank advances its ladders in lockstep
    (`advance_ladders`: one batched device evaluation per proposal round) and ONE all-gather at the
    end returns the final (theta, log-prob) of every chain.

    Returns (state, evaluations): `state` (n_ladders, n_chains, P + 1) on every rank, rows
    [theta | tempered log-prob]; `evaluations` the total number of posterior evaluations."""
    from inference_amd.mcmc.parallel import advance_ladders

    rank, size = world()
    lo, hi = shard_bounds(n_ladders, size, rank)
    ladders = [make_ladder(k) for k in range(lo, hi)]
    evals = advance_ladders(ladders, n_steps, swap_interval=swap_interval, batch_posterior=batch_posterior) if ladders else 0
    if ladders:
        local = np.array([[np.append(c.get_last(), c.probs[-1]) for c in lad.chains] for lad in ladders])
    else:
        local = None
    # shapes are needed on ranks without ladders too: they ride along in the gather
    shape = np.array([0.0, 0.0] if local is None else [local.shape[1], local.shape[2]])
    if size > 1:
        shapes = _gather_rows(shape[None, :], 1, 2, engine)[:, 0, :]
        n_chains, w = (int(v) for v in shapes.max(axis=0))
    else:
        n_chains, w = int(shape[0]), int(shape[1])
    flat = np.empty((0, n_chains * w + 1)) if local is None else np.concatenate(
        [local.reshape(hi - lo, n_chains * w), np.full((hi - lo, 1), evals / max(hi - lo, 1))], axis=1)
    if size == 1:
        got = flat
    else:
        per = max(-(-n_ladders // size), 1)
        g = _gather_rows(flat, per, n_chains * w + 1, engine)
        got = np.concatenate([g[r, : b - a] for r, (a, b) in
                              enumerate(shard_bounds(n_ladders, size, r) for r in range(size))], axis=0)
    return got[:, :-1].reshape(n_ladders, n_chains, w), int(round(got[:, -1].sum()))


# ---------------------------------------------------------------------------------------------
# torch-free bootstrap
# ---------------------------------------------------------------------------------------------
def _encode(obj):
    """JSON-safe form of the small payloads the ranks exchange (RCCL unique id, a few floats, flags):
    nothing is ever un-pickled, so a file dropped into the directory cannot run code."""
    if isinstance(obj, (bytes, bytearray)):
        return {"__b64__": base64.b64encode(bytes(obj)).decode("ascii")}
    if isinstance(obj, np.ndarray):
        return {"__nd__": np.asarray(obj, dtype=float).ravel().tolist(), "shape": list(obj.shape)}
    if isinstance(obj, (np.floating, np.integer, np.bool_)):
        return obj.item()
    if isinstance(obj, dict):
        return {str(k): _encode(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    raise TypeError(f"FileRendezvous cannot carry a {type(obj).__name__}")


def _decode(obj):
    if isinstance(obj, dict):
        if "__b64__" in obj:
            return base64.b64decode(obj["__b64__"])
        if "__nd__" in obj:
            return np.array(obj["__nd__"], dtype=float).reshape(obj["shape"])
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


class RendezvousAborted(RuntimeError):
    """Another rank left the job early (`FileRendezvous.abort`): every pending and later exchange raises this at once
    instead of waiting for files that will never appear."""


class RendezvousDesync(RuntimeError):
    """The ranks are not in the same exchange: a payload arrived under another tag than the reader's own."""


class FileRendezvous:
    """Torch-free bootstrap for the ranks of ONE node (what `torch.distributed.run --nnodes=1` or
    `bench.py --gpus N` launches): exchanges small JSON payloads through a per-job directory.  Used
    to hand the RCCL unique id to every rank — and as a last-resort gather if RCCL cannot be
    initialised — so that a multi-GPU job never has to import torch (importing it loads torch's
    bundled HIP / HSA runtime beside the system one, which RCCL then picks up uninitialised).

    The directory lives under $GPMI_RDV_DIR, else $XDG_RUNTIME_DIR, else /tmp; it is created with
    mode 0700 and refused unless it is a real directory owned by this user with no group / other
    access.  Its name carries the job key: $GPMI_RDV_KEY, else the launcher's run id
    ($TORCHELASTIC_RUN_ID) together with the launcher's pid and $MASTER_PORT — all ranks are
    children of the same launcher process.  Rank 0 clears left-overs of a crashed job with the same
    key before the first round; files are removed only after a final acknowledgement round."""

    def __init__(self, rank=None, world=None, timeout=90.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.timeout = timeout
        key = os.environ.get("GPMI_RDV_KEY")
        if not key:
            run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
            key = f"{run_id}_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
        key = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in key)
        base = os.environ.get("GPMI_RDV_DIR") or os.environ.get("XDG_RUNTIME_DIR") or "/tmp"
        self.dir = os.path.join(base, f"gpmi_rdv_{os.getuid()}_{key}")
        self._open_dir()
        self.round = 0
        self.closed = False
        self.aborted = False
        # Generation handshake, robust against the files of a crashed job with the same key: every other rank
        # keeps (re)writing a `hello` file with a fresh random token until it finds a `gen` file that quotes
        # its token; rank 0 first removes whatever is in the directory, then waits for every rank's hello
        # (re-written after the purge if it was caught by it) and publishes the generation token together
        # with the hello tokens it saw.  A stale `gen` quotes other tokens and is ignored.
        if self.rank == 0:
            self._purge()
            t0 = time.time()
            hellos = [None] + [self._read(f"hello.{r}", t0) for r in range(1, self.world)]
            self.gen = base64.b16encode(os.urandom(6)).decode("ascii").lower()
            self._write("gen", {"gen": self.gen, "hellos": hellos})
        else:
            mine = base64.b16encode(os.urandom(6)).decode("ascii").lower()
            t0 = time.time()
            while True:
                if not os.path.exists(os.path.join(self.dir, f"hello.{self.rank}")):
                    try:
                        self._write(f"hello.{self.rank}", mine)
                    except OSError:  # rank 0's purge took the half-written file from under the rename: write it again
                        continue
                try:
                    with open(os.path.join(self.dir, "gen")) as f:
                        g = json.load(f)
                    if g["hellos"][self.rank] == mine:
                        self.gen = g["gen"]
                        break
                except (OSError, ValueError, KeyError, IndexError, TypeError):
                    pass
                if time.time() - t0 > self.timeout:
                    raise TimeoutError("rank 0 did not open the rendezvous")
                time.sleep(0.005)

    def _purge(self, everything=False):
        for name in os.listdir(self.dir):
            if name.endswith(".tmp") and not everything:  # being written right now (truncated on re-use, never read)
                continue
            try:
                os.remove(os.path.join(self.dir, name))
            except OSError:
                pass

    def _open_dir(self):
        try:
            os.mkdir(self.dir, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(self.dir)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError(f"rendezvous directory {self.dir} is not a private directory of uid {os.getuid()}")

    def _write(self, name, obj):
        path = os.path.join(self.dir, name)
        fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_TRUNC | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            json.dump(_encode(obj), f)
        os.replace(path + ".tmp", path)

    def _read(self, name, t0, watch_abort=True):
        path = os.path.join(self.dir, name)
        polls = 0
        while True:
            try:
                with open(path) as f:
                    return _decode(json.load(f))
            except (FileNotFoundError, ValueError):  # not there yet, half-written or not JSON at all
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rendezvous file {name} did not appear within {self.timeout:.0f} s")
                polls += 1
                if watch_abort and polls % 25 == 0:
                    self._check_abort()
                time.sleep(0.002)

    def _check_abort(self):
        gen = getattr(self, "gen", None)
        if gen is None:
            return
        for name in os.listdir(self.dir):
            if name.startswith(f"{gen}.abort."):
                try:
                    with open(os.path.join(self.dir, name)) as f:
                        why = json.load(f)
                except (OSError, ValueError):
                    why = "?"
                self.aborted = True
                raise RendezvousAborted(f"rank {name.rsplit('.', 1)[-1]} left the job: {why}")

    def abort(self, reason=""):
        """Tell every rank that this one will not take part in further exchanges (it failed): their pending and later
        reads raise RendezvousAborted at once.  Without it a rank that skips an exchange leaves the others waiting for the
        time limit - and its NEXT exchange (the closing barrier) is read by them as the payload of the one it skipped."""
        self.aborted = True
        try:
            self._write(f"{self.gen}.abort.{self.rank}", str(reason)[:500])
        except (OSError, AttributeError):
            pass

    def allgather_obj(self, obj, tag=""):
        """Every rank contributes a JSON-representable object (numbers, strings, bytes, lists, dicts,
        float arrays); returns the list in rank order.  `tag` names the exchange: a payload written under another tag
        means the ranks are out of step (one of them skipped or added an exchange) and raises RendezvousDesync instead
        of being mistaken for this exchange's data."""
        if getattr(self, "aborted", False):
            raise RendezvousAborted("the job was aborted")
        self.round += 1
        self._write(f"{self.gen}.r{self.round}.{self.rank}", {"tag": str(tag), "v": obj})
        t0 = time.time()
        out = []
        for r in range(self.world):
            got = self._read(f"{self.gen}.r{self.round}.{r}", t0)
            if not isinstance(got, dict) or got.get("tag") != str(tag):
                raise RendezvousDesync(f"exchange {self.round} ({tag!r}): rank {r} is in "
                                       f"{got.get('tag') if isinstance(got, dict) else got!r}")
            out.append(got["v"])
        return out

    def broadcast_obj(self, obj, src=0):
        return self.allgather_obj(obj if self.rank == src else None, tag="broadcast")[src]

    def barrier(self):
        self.allgather_obj(0, tag="barrier")

    def close(self):
        """Leave the rendezvous.  Two rounds: after the first every rank has read everything it will
        ever read except the second round's files, which carry no payload; rank 0 then waits for
        every rank's `done` marker before it removes the directory's contents."""
        if self.closed:
            return
        self.closed = True
        try:
            if not getattr(self, "aborted", False):
                try:
                    self.barrier()
                except (RendezvousAborted, RendezvousDesync):
                    pass
            self._write(f"{self.gen}.done.{self.rank}", 1)
            if self.rank == 0:
                # (after an abort: a short wait, so that ranks still polling see the abort file before it is removed)
                if getattr(self, "aborted", False):
                    self.timeout = min(self.timeout, 20.0)
                t0 = time.time()
                for r in range(self.world):
                    self._read(f"{self.gen}.done.{r}", t0, watch_abort=False)
        except (TimeoutError, OSError, AttributeError):
            pass  # a peer died: clean up what we can
        if self.rank == 0:
            self._purge(everything=True)
            try:
                os.rmdir(self.dir)
            except OSError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def init_device_comm_files(engine, rdv: "FileRendezvous"):
    """RCCL communicator bootstrap over a FileRendezvous (no torch in the process)."""
    uid = rdv.broadcast_obj(engine.comm_unique_id() if rdv.rank == 0 else None)
    engine.comm_init(rdv.rank, rdv.world, uid)
