"""CPU test of the N>1 path: world_size-2 gloo processes shard independent problems and gather results to rank 0.
The per-problem 'solve' here is a deterministic stand-in (the HIP solve needs a GPU); what is tested is the partition,
the ragged gather and the result order that bench.py / a multi-GPU host rely on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_solve(ids):
    # a "result vector" whose length depends on the problem (ragged) and whose content identifies it
    return [np.arange(3 + (i % 4), dtype=np.float64) * 0.5 + 100.0 * i for i in ids]


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    from defslam_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = shard.solve_sharded(list(range(n_items)), _fake_solve, dist)
        if rank == 0:
            q.put([o.tolist() for o in out])
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_two_rank_sharded_solve_gathers_in_id_order(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = [v.tolist() for v in _fake_solve(list(range(n_items)))]
    assert got == expect


def test_shard_range_partitions_everything_once():
    from defslam_amd import shard
    for n in [0, 1, 7, 8, 9, 256, 1000]:
        for world in [1, 2, 3, 8]:
            seen = []
            for r in range(world):
                rr = shard.shard_range(n, r, world)
                seen += list(rr)
                assert len(rr) in (n // world, n // world + 1)
            assert seen == list(range(n))
