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
    assert calls == ([hi - lo] if hi > lo else [])  # an empty block never reaches the evaluator
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


def _bcast_worker(rank, world, port, with_err, out_dir):
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
    # only the source rank holds the data (the reference pickles its regressor into the workers, regression.py:597-601)
    if rank == 1:
        x, y, e = wl.synthetic_dataset(3, 40, 3)
        got = sharding.broadcast_dataset(x, y, e if with_err else None, src=1)
    else:
        got = sharding.broadcast_dataset(src=1)
    x, y, e = got
    # ... and every rank can build its own model from what it received
    gp = orc.OracleGp(x, y, e if e is not None else np.full(y.size, 0.1), kernel=orc.SE)
    lml = gp.marginal_likelihood(wl.timing_theta(wl.SE, y, 3))
    np.savez(os.path.join(out_dir, f"b{rank}.npz"), x=x, y=y, e=np.zeros(0) if e is None else e, lml=lml)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("with_err", [True, False])
def test_broadcast_dataset_two_ranks_gloo(tmp_path, with_err):
    """Start-up distribution of SURVEY section 8(e) over the bootstrap channel (on the GPU ranks: one ncclBroadcast,
    gpmi_comm_broadcast): x, y, y_err arrive bit for bit, y_err = None stays None."""
    import workloads as wl

    port = _free_port()
    mp.spawn(_bcast_worker, args=(2, port, with_err, str(tmp_path)), nprocs=2, join=True)
    x, y, e = wl.synthetic_dataset(3, 40, 3)
    got = [np.load(tmp_path / f"b{r}.npz") for r in range(2)]
    for g in got:
        assert np.array_equal(g["x"], x) and np.array_equal(g["y"], y)
        assert np.array_equal(g["e"], e) if with_err else g["e"].size == 0
    assert float(got[0]["lml"]) == float(got[1]["lml"])


def test_broadcast_dataset_single_process_and_validation():
    from inference_amd import sharding
    import workloads as wl

    x, y, e = wl.synthetic_dataset(3, 12, 1)
    gx, gy, ge = sharding.broadcast_dataset(x[:, 0], y, e)  # 1-D x becomes a column, as GpRegressor does
    assert gx.shape == (12, 1) and np.array_equal(gy, y) and np.array_equal(ge, e)
    with pytest.raises(ValueError):
        sharding.broadcast_dataset(x, y[:-1], e)


def test_shard_bounds_cover_everything():
    from inference_amd import sharding

    for n in (0, 1, 5, 8, 64, 65):
        for w in (1, 2, 3, 8):
            blocks = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


# ---- config 5: whole ParallelTempering ladders per rank, one gather of (theta, log-prob) ---------------------
def _pt_problem():
    rng = np.random.default_rng(31)
    x = np.sort(rng.uniform(0, 6, 40))
    y = np.sin(x) + 0.3 * np.cos(2.5 * x) + 0.2 * rng.normal(size=40)
    return x, y, np.full(40, 0.2)


def _make_ladder_factory(gp):
    import random

    from numpy.random import default_rng

    from inference_amd.mcmc import GibbsChain, ParallelTempering

    start = np.array([gp.y.mean(), np.log(gp.y.std()), 0.0])

    def make(k):
        chains = []
        for t_i, temp in enumerate([1.0, 3.0, 9.0]):
            ch = GibbsChain(posterior=gp.marginal_likelihood, start=start, widths=[0.1, 0.2, 0.2], temperature=temp,
                            display_progress=False)
            for i, b in enumerate(gp.hp_bounds):
                ch.set_boundaries(i, b)
            ch.rng = default_rng(10_000 * k + 10 * t_i)
            for i, par in enumerate(ch.params):
                par.rng = default_rng(10_000 * k + 10 * t_i + 1 + i)
            chains.append(ch)
        pt = ParallelTempering(chains, batch_posterior=None)
        pt.rng = default_rng(77 + k)
        pt.pair_choice = random.Random(900 + k).choice  # own pairing stream: the run must not depend on the world size
        return pt

    return make


def _batch_of(gp):
    return lambda th: np.array([gp.marginal_likelihood(t) for t in th])


def _pt_worker(rank, world, port, n_ladders, out_dir):
    for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import random

    import torch.distributed as dist

    from inference_amd import sharding
    from oracle import gp_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, e = _pt_problem()
    gp = orc.OracleGp(x, y, e, kernel=orc.SE)
    state, evals = sharding.tempering_run(_make_ladder_factory(gp), n_ladders, 6, swap_interval=3,
                                          batch_posterior=_batch_of(gp))
    np.save(os.path.join(out_dir, f"pt{rank}.npy"), state)
    np.save(os.path.join(out_dir, f"ev{rank}.npy"), np.array([evals]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_ladders,world", [(3, 2), (1, 2), (4, 3)])
def test_tempering_run_two_ranks_gloo(tmp_path, n_ladders, world):
    """Ladders block-partitioned over the ranks give the state a single process gets (n_ladders = 1: rank 1 idle;
    4 ladders over 3 ranks: blocks of 2, 1, 1)."""
    import random

    from inference_amd import sharding
    from oracle import gp_oracle as orc

    port = _free_port()
    mp.spawn(_pt_worker, args=(world, port, n_ladders, str(tmp_path)), nprocs=world, join=True)
    x, y, e = _pt_problem()
    gp = orc.OracleGp(x, y, e, kernel=orc.SE)
    serial, evals = sharding.tempering_run(_make_ladder_factory(gp), n_ladders, 6, swap_interval=3,
                                           batch_posterior=_batch_of(gp))
    assert serial.shape == (n_ladders, 3, 4)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"pt{r}.npy"), serial)
        assert int(np.load(tmp_path / f"ev{r}.npy")[0]) == evals
    assert evals >= n_ladders * 3 * 6 * 3


def _ms_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from inference_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th, f = sharding.multistart_sweep(_FakeSearch(), _FakeSearch.starts)
    np.save(os.path.join(out_dir, f"ms{rank}.npy"), np.column_stack([th, f]))
    dist.barrier()
    dist.destroy_process_group()


class _FakeSearch:
    """Stands in for a GpRegressor: `launch_bfgs(x0)` with SciPy's return convention."""
    starts = np.arange(15.0).reshape(5, 3)

    def launch_bfgs(self, x0):
        return x0 * 0.5 + 1.0, np.array(float(np.sum(x0**2))), {"warnflag": 0}


def test_multistart_sweep_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_ms_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = np.column_stack([_FakeSearch.starts * 0.5 + 1.0, (_FakeSearch.starts**2).sum(axis=1)])
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"ms{r}.npy"), want)
