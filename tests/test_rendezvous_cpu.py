"""CPU test of the torch-free FileRendezvous used by the multi-GPU bench to bootstrap RCCL
(two real processes, as `torch.distributed.run --nproc-per-node 2` would start them)."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, tmp, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "inference-tools_amd")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29999", GPMI_RDV_DIR=tmp)
    from inference_amd.sharding import FileRendezvous

    rdv = FileRendezvous()
    uid = rdv.broadcast_obj(b"x" * 128 if rank == 0 else None)
    got = rdv.allgather_obj({"rank": rank, "v": [rank * 1.5, 2.0]})
    rdv.barrier()
    q.put((rank, uid, got))
    rdv.close()


def test_file_rendezvous_two_processes(tmp_path):
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path), q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=60) for _ in range(2))
    [p.join(30) for p in procs]
    for rank, uid, got in res:
        assert uid == b"x" * 128
        assert [g["rank"] for g in got] == [0, 1] and got[1]["v"][0] == 1.5
    assert not any(n.startswith("gpmi_rdv_") for n in os.listdir(tmp_path))  # cleaned up
