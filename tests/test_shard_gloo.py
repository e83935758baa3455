"""N>1 path on CPU: world_size-2 gloo processes shard a batch with rabe_amd.shard, run the host-side
(string / Fr) part of the work on their block and gather the records in item order."""
import hashlib
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from rabe_amd import shard


def test_shard_range_partitions():
    for n in (0, 1, 7, 4096, 16385):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _work(i):
    from rabe_amd import hostprep as hp
    return hp.fr_le(hp.h_fr("item%d" % i)) + hashlib.sha3_256(b"%d" % i).digest()


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n, rank, world)
    local = [_work(i) for i in range(lo, hi)]
    allrec = shard.gather_records(local, dst=0)
    t = shard.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((allrec, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    allrec, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert allrec == [_work(i) for i in range(n)]
    assert t == 2.0
