"""CPU, world_size 2, gloo: the N > 1 path (LPT ownership + the single final track gather)."""
import os
import socket
import sys

import numpy as np
import pytest


def _worker(rank, world, port, lengths, width, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from consenrich_amd.sharding import gather_tracks, lpt_assign

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = lpt_assign(lengths, world)[rank]
        local = {i: (np.arange(lengths[i] * width, dtype=np.float32).reshape(lengths[i], width) + 1000.0 * i)
                 for i in mine}
        out = gather_tracks(local, lengths, width)
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
