"""N>1 path on CPU: two processes, gloo.  Covers the batch split and the one collective (result gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nmrf_amd.parallel import gather_disparity, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(total * 3 * 5, dtype=torch.float32).view(total, 3, 5)     # the "disparities" of all pairs
        b, e = shard_range(total, rank, world)
        got = gather_disparity(full[b:e].clone(), total=total)
        q.put((rank, bool(torch.equal(got, full)), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_batch_shard_and_gather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok for _, ok, _ in res), res


def test_gather_is_identity_without_process_group():
    x = torch.rand(2, 4, 4)
    assert gather_disparity(x) is x
