"""N>1 path on the GPU: RCCL (`nccl` backend) ranks, one per visible MI355X -- min(2, device_count) of them.  On a 2+ GPU box the
gathered disparities of a batch sharded over two ranks must equal the single-process result bit for bit (same kernels, per-image
independent); on the 1-GPU box of the round-end tier the same code runs on a 1-rank RCCL group (init, one
all_gather_into_tensor on the side stream, teardown) -- the 2-rank leg then reports itself as skipped in the note."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, total, hw, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from nmrf_amd.parallel import OverlappedGather, gather_disparity, pin_to_gpu_numa, shard_range
        from nmrf_amd.utils.hashinit import synthetic_pair
        from tests.util import build_product
        numa = pin_to_gpu_numa(rank)
        model = build_product(128, dev)
        lo, hi = shard_range(total, rank, world)
        prs = [synthetic_pair(hw[0], hw[1], seed=900 + i)[:2] for i in range(lo, hi)]
        with torch.no_grad():
            disp = model({"img1": torch.stack([p[0] for p in prs]).to(dev), "img2": torch.stack([p[1] for p in prs]).to(dev)})["disp"]
            og = OverlappedGather(single_rank_too=True)
            a = og.submit(disp)                       # side-stream all_gather_into_tensor (the bench's placement)
            b2 = og.submit(disp)                      # second slot
            og.finish()
            torch.cuda.synchronize()
            c = gather_disparity(disp, total=total)   # the synchronous form (identity on a 1-rank group)
            ok = bool(torch.equal(a, b2)) and bool(torch.equal(a, c) if world > 1 else torch.equal(c, disp))
            want = None
            if rank == 0:                             # the single-process result of the WHOLE job on this rank's GPU
                allp = [synthetic_pair(hw[0], hw[1], seed=900 + i)[:2] for i in range(total)]
                want = model({"img1": torch.stack([p[0] for p in allp]).to(dev), "img2": torch.stack([p[1] for p in allp]).to(dev)})["disp"]
                ok = ok and bool(torch.equal(a, want))
        dist.barrier()
        q.put((rank, ok, tuple(a.shape), numa.get("pinned"), numa.get("why")))
    finally:
        dist.destroy_process_group()


def test_rccl_batch_shard_gather_equals_single_process():
    import torch.multiprocessing as mp
    from tests.conftest import record_note
    world = min(2, torch.cuda.device_count())
    total, hw = 4, (64, 104)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, total, hw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] == (total, hw[0], hw[1]) for r in res), res
    record_note("RCCL gather test ran on %d rank(s) (%d GPU(s) visible)%s; NUMA pinning: %s" % (
        world, torch.cuda.device_count(), "" if world > 1 else " -- the 2-rank leg needs a second GPU: 1-rank RCCL group only",
        ["pinned" if r[3] else "not pinned (%s)" % r[4] for r in sorted(res)]))
