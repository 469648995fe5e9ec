"""CPU tests of the torch-free FileRendezvous used by the multi-GPU bench to bootstrap RCCL (real processes, as
`torch.distributed.run --nproc-per-node 2` or `bench.py --gpus 2` start them) and of bench.py's own launcher."""
import json
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "inference-tools_amd")) if p not in sys.path]


def _worker(rank, world, tmp, q, delay=0.0):
    import time

    sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29999", GPMI_RDV_DIR=tmp, GPMI_RDV_KEY="t1")
    from inference_amd.sharding import FileRendezvous

    time.sleep(delay)
    with FileRendezvous() as rdv:
        uid = rdv.broadcast_obj(b"x" * 128 if rank == 0 else None)
        got = rdv.allgather_obj({"rank": rank, "v": [rank * 1.5, 2.0], "a": np.array([[rank, 1.0 / 3.0]])})
        rdv.barrier()
        q.put((rank, uid, [(g["rank"], g["v"], g["a"].tolist()) for g in got]))


@pytest.mark.parametrize("late_rank", [None, 0, 1])
def test_file_rendezvous_two_processes(tmp_path, late_rank):
    """Exchange works whichever rank arrives first, also with the files of a crashed job with the same key present."""
    stale = tmp_path / f"gpmi_rdv_{os.getuid()}_t1"
    stale.mkdir(mode=0o700)
    (stale / "gen").write_text(json.dumps({"gen": "dead", "hellos": [None, "beef"]}))
    (stale / "hello.1").write_text(json.dumps("beef"))
    (stale / "dead.r1.0").write_text(json.dumps({"__b64__": "AAAA"}))
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path), q, 0.5 if r == late_rank else 0.0)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=60) for _ in range(2))
    [p.join(30) for p in procs]
    assert [p.exitcode for p in procs] == [0, 0]
    for rank, uid, got in res:
        assert uid == b"x" * 128
        assert [g[0] for g in got] == [0, 1] and got[1][1][0] == 1.5
        assert got[1][2] == [[1.0, 1.0 / 3.0]]  # floats survive the text round trip bit for bit
    assert not any(n.startswith("gpmi_rdv_") for n in os.listdir(tmp_path))  # cleaned up


def _worker_fail(rank, world, tmp, q, mode):
    import time

    sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29998", GPMI_RDV_DIR=tmp, GPMI_RDV_KEY="t2")
    from inference_amd.sharding import FileRendezvous, RendezvousAborted, RendezvousDesync

    rdv = FileRendezvous(timeout=60.0)
    t0 = time.time()
    try:
        rdv.allgather_obj(rank, tag="first")
        if mode == "abort" and rank == 1:
            rdv.abort("a failure of rank 1")       # skips the exchange the others enter next
        elif mode == "desync" and rank == 1:
            rdv.allgather_obj(0, tag="barrier")    # ... or enters ANOTHER exchange in its place
        else:
            rdv.allgather_obj([1.0, 2.0], tag="values")
        q.put((rank, "ok", time.time() - t0))
    except RendezvousAborted as err:
        q.put((rank, "aborted: " + str(err), time.time() - t0))
    except RendezvousDesync as err:
        q.put((rank, "desync: " + str(err), time.time() - t0))
    finally:
        rdv.close()


@pytest.mark.parametrize("mode", ["abort", "desync"])
def test_rendezvous_abort_and_desync_are_detected_at_once(tmp_path, mode):
    """Round 5 (what the first cold 8-rank bench run did): a rank that fails and skips an exchange no longer leaves the
    others waiting for the time limit, and its next exchange is never read as the payload of the one it skipped -
    `abort` ends the others' waits with RendezvousAborted, a payload under another tag raises RendezvousDesync."""
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fail, args=(r, 3, str(tmp_path), q, mode)) for r in range(3)]
    [p.start() for p in procs]
    res = dict((r, (what, dt)) for r, what, dt in (q.get(timeout=50) for _ in range(3)))
    [p.join(40) for p in procs]
    assert [p.exitcode for p in procs] == [0, 0, 0]
    assert all(dt < 10.0 for _, dt in res.values()), res  # nobody sat out the 60 s time limit
    if mode == "abort":
        assert res[1][0] == "ok" and res[0][0].startswith("aborted") and "a failure of rank 1" in res[2][0], res
    else:
        assert all(v[0].startswith("desync") for v in res.values()), res
    assert not any(n.startswith("gpmi_rdv_") for n in os.listdir(tmp_path))  # cleaned up


def test_rendezvous_refuses_foreign_or_open_directory(tmp_path, monkeypatch):
    from inference_amd.sharding import FileRendezvous

    monkeypatch.setenv("GPMI_RDV_DIR", str(tmp_path))
    monkeypatch.setenv("GPMI_RDV_KEY", "perm")
    d = tmp_path / f"gpmi_rdv_{os.getuid()}_perm"
    d.mkdir(mode=0o777)
    os.chmod(d, 0o777)
    with pytest.raises(PermissionError):
        FileRendezvous(0, 1)
    os.chmod(d, 0o700)
    os.rmdir(d)
    os.symlink(tmp_path, d)  # a symlink someone planted instead of the directory
    with pytest.raises(PermissionError):
        FileRendezvous(0, 1)


def test_rendezvous_payloads_are_never_unpickled(tmp_path, monkeypatch):
    """A pickle dropped where a rank's file is expected is not executed: it is not valid JSON, the read retries
    and times out."""
    import pickle

    from inference_amd.sharding import FileRendezvous

    monkeypatch.setenv("GPMI_RDV_DIR", str(tmp_path))
    monkeypatch.setenv("GPMI_RDV_KEY", "nopickle")
    rdv = FileRendezvous(0, 1, timeout=0.3)

    class Boom:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))

    rdv.world = 2
    with open(os.path.join(rdv.dir, f"{rdv.gen}.r1.1"), "wb") as f:
        pickle.dump(Boom(), f)
    with pytest.raises(TimeoutError):
        rdv.allgather_obj(1)
    assert not (tmp_path / "pwned").exists()
    with pytest.raises(TypeError):
        rdv.allgather_obj(object())  # only plain data may be sent


def test_bench_launcher_spawns_ranks_and_relays_rank0(tmp_path):
    """`bench.py --gpus N` without RANK in the environment: the parent starts N children with RANK / LOCAL_RANK /
    WORLD_SIZE / a per-job rendezvous key, relays rank 0's stdout and fails when a child fails.  The children here
    are a stand-in script (no GPU in this container); the launcher code is bench.py's own."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, json\n"
        "r = int(os.environ['RANK'])\n"
        "print(json.dumps({'rank': r, 'world': os.environ['WORLD_SIZE'], 'local': os.environ['LOCAL_RANK'],\n"
        "                  'key': os.environ['GPMI_RDV_KEY'], 'argv': sys.argv[1:]}))\n"
        "sys.exit(3 if os.environ.get('FAIL_RANK') == str(r) else 0)\n")
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '3', '--steps', '1']\n"
            f"import importlib.util as u; s = u.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r})\n"
            "b = u.module_from_spec(s); s.loader.exec_module(b)\n"
            f"b.__file__ = {str(child)!r}\n"
            "b.launch_ranks(3)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # exactly rank 0's line
    got = json.loads(lines[0])
    assert got["rank"] == 0 and got["world"] == "3" and got["local"] == "0" and got["key"].startswith("bench_")
    assert got["argv"] == ["--gpus", "3", "--steps", "1"]
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, FAIL_RANK="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "rank(s) failed" in bad.stderr
