"""The N > 1 path on CPU: world_size 2 over gloo (rendezvous on 127.0.0.1).  Covers the static
round-robin partition, pull scheduling over the store counter, and the whole-job aggregation that
bench.py prints (units summed, time = max over ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imageprocessor_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, lr, w = shard.init_from_env()
    assert (r, w) == (rank, world)
    # static partition of a uniform batch
    mine = shard.round_robin(10, rank, world)
    # pull scheduling of a mixed batch, largest first; rank 1 is "slow" and should claim less
    sizes = [(854, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)] * 5
    order = shard.lpt_order([shard.frame_cost(*s) for s in sizes])
    q = shard.WorkQueue(len(order), chunk=2, store=shard.default_store())
    dist.barrier()
    claimed = []
    import time
    while True:
        c = q.claim()
        if c is None:
            break
        claimed += [order[i] for i in c]
        time.sleep(0.02 if rank == 1 else 0.001)
    units, secs = shard.aggregate(len(claimed), 1.0 + rank)
    torch.save({"mine": mine, "claimed": claimed, "units": units, "secs": secs}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in range(world)]
    # round robin covers the batch exactly once
    assert sorted(res[0]["mine"] + res[1]["mine"]) == list(range(10))
    assert res[0]["mine"] == [0, 2, 4, 6, 8]
    # pull scheduling: every item exactly once, and the fast rank took more
    allc = res[0]["claimed"] + res[1]["claimed"]
    assert sorted(allc) == list(range(30))
    assert len(res[0]["claimed"]) > len(res[1]["claimed"])
    # both ranks see the same whole-job aggregate: units summed, time = max
    for r in res:
        assert r["units"] == 30.0 and r["secs"] == 2.0


def test_single_process_queue_and_order():
    q = shard.WorkQueue(5, chunk=2)
    got = []
    while True:
        c = q.claim()
        if c is None:
            break
        got += list(c)
    assert got == [0, 1, 2, 3, 4]
    costs = [shard.frame_cost(w, h) for w, h in [(854, 480), (7680, 4320), (1920, 1080)]]
    assert shard.lpt_order(costs) == [1, 2, 0]
    assert shard.aggregate(7, 0.5) == (7.0, 0.5)
