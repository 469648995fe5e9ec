"""
Multi-process coverage of the N>1 path on CPU: world_size-2 `gloo` process groups run the same
sharding / all-gather code that the GPU ranks run over RCCL.  The per-rank evaluator here is the
CPU oracle (test infrastructure); on the GPU box it is GpRegressor.marginal_likelihood_batch.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_items, out_dir):
    for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from inference_amd import sharding
    from oracle import gp_oracle as orc
    import workloads as wl

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, e = wl.synthetic_dataset(3, 96, 3)
    gp = orc.OracleGp(x, y, e, kernel=orc.RQ)
    thetas = wl.theta_set(wl.RQ, y, 3, n_items, seed=2)
    calls = []

    def batch(th):
        calls.append(len(th))
        return np.array([gp.marginal_likelihood(t) for t in th])

    res = sharding.sharded_map(batch, thetas)[:, 0]
    lo, hi = sharding.shard_bounds(n_items, world, rank)
    assert calls == [hi - lo]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [6, 7, 1])
def test_sharded_sweep_two_ranks_gloo(tmp_path, n_items):
    from oracle import gp_oracle as orc
    import workloads as wl

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    x, y, e = wl.synthetic_dataset(3, 96, 3)
    gp = orc.OracleGp(x, y, e, kernel=orc.RQ)
    thetas = wl.theta_set(wl.RQ, y, 3, n_items, seed=2)
    serial = np.array([gp.marginal_likelihood(t) for t in thetas])
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert np.array_equal(got, serial)  # every rank holds the full, ordered result


def test_shard_bounds_cover_everything():
    from inference_amd import sharding

    for n in (0, 1, 5, 8, 64, 65):
        for w in (1, 2, 3, 8):
            blocks = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
