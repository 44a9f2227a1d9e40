"""CPU, world_size 2, gloo: the host logic of the N > 1 path (LPT ownership, packed gather layout, re-assembly in genome
order) over a gloo transport defined HERE -- the product's transport is RCCL (consenrich_amd.sharding.RcclComm, GPU only);
torch / gloo appear in this test file only."""
import os
import socket
import sys

import numpy as np
import pytest


def _worker(rank, world, port, lengths, width, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    import torch

    from consenrich_amd.sharding import gather_tracks, lpt_assign

    class GlooTransport:
        def __init__(self):
            self.rank, self.world = dist.get_rank(), dist.get_world_size()

        def all_gather(self, send):
            t = torch.from_numpy(np.ascontiguousarray(send, np.float32))
            recv = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(recv, t)
            return np.stack([r.numpy() for r in recv])

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = lpt_assign(lengths, world)[rank]
        local = {i: (np.arange(lengths[i] * width, dtype=np.float32).reshape(lengths[i], width) + 1000.0 * i)
                 for i in mine}
        out = gather_tracks(local, lengths, width, GlooTransport())
        if rank == 0:
            ok = all(np.array_equal(out[i], np.arange(lengths[i] * width, dtype=np.float32).reshape(lengths[i], width)
                                    + 1000.0 * i) for i in range(len(lengths)))
            q.put(bool(ok))
        else:
            q.put(out is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gather_reassembles_genome_order():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    lengths = [37, 5, 120, 64, 9, 200, 1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def _rdzv_worker(rank, path, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from consenrich_amd.launch import JobFiles
    from consenrich_amd.sharding import exchange_unique_id

    files = JobFiles(rank, 3, directory=path, token="job-under-test")
    q.put((rank, exchange_unique_id(rank, lambda: bytes(range(128)) if rank == 0 else b"", files, timeout_s=60.0)))


@pytest.mark.timeout(120)
def test_unique_id_rendezvous_through_the_job_directory(tmp_path):
    """The torch-free rendezvous of the RCCL communicator: rank 0 publishes its 128-byte id atomically in the job's directory,
    the others poll (here 3 processes; the late starter is rank 0).  The directory already holds a STALE id of an earlier job
    (another token): the early ranks must not take it."""
    import multiprocessing as mp
    import time

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "job")
    os.makedirs(path, mode=0o700)
    with open(os.path.join(path, "rccl_id"), "wb") as fh:           # leftover of a job that died before it cleaned up
        fh.write(b"some-earlier-job\n" + bytes(128))
    procs = [ctx.Process(target=_rdzv_worker, args=(r, path, q)) for r in (1, 2)]
    for p in procs:
        p.start()
    time.sleep(0.5)
    assert q.empty()                                                # nobody accepted the stale file
    p0 = ctx.Process(target=_rdzv_worker, args=(0, path, q))
    p0.start()
    got = dict(q.get(timeout=90) for _ in range(3))
    for p in procs + [p0]:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1] == got[2] == bytes(range(128))
