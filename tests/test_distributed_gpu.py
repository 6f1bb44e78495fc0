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


def _train_rank(rank, world, port, q):
    """Gradients of the slice after backward + the overlapped reducer, on this rank's share of a batch of two; rank 0 also computes the
    single-process gradients of the whole batch."""
    import warnings
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from nmrf_amd.models.criterion import build_criterion
        from nmrf_amd.train import OverlappedGradientReducer, slice_parameters
        from nmrf_amd.utils.hashinit import synthetic_pair
        from tests.util import build_product, make_cfg
        cfg = make_cfg(128)
        crit = build_criterion(cfg)
        h, w = 64, 128

        def samples(idx):
            prs = [synthetic_pair(h, w, seed=700 + i) for i in idx]
            return {"img1": torch.stack([p[0] for p in prs]).to(dev), "img2": torch.stack([p[1] for p in prs]).to(dev),
                    "disp": torch.stack([p[2].clamp(1.0, 100.0) for p in prs]).to(dev),          # every pixel valid: equal counts per rank
                    "valid": torch.ones(len(idx), h, w, dtype=torch.bool, device=dev)}

        def grads(model, sample, reducer=None):
            for p in model.parameters():
                p.grad = None
            if reducer is not None:
                reducer.prepare()
            out = model(sample)
            ld = crit(out, {"disp": sample["disp"].clone(), "valid": sample["valid"]})
            sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
            if reducer is not None:
                reducer.finish()
            return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in slice_parameters(model)}

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = build_product(128, dev).train().enable_grad_slice()
            red = OverlappedGradientReducer([p for _, p in slice_parameters(model)], bucket_bytes=256 << 10)
            mine = list(range(2))[rank::world] if world > 1 else [0, 1]
            g1 = grads(model, samples(mine), red)            # step 1: the flat path (defines the live set)
            g2 = grads(model, samples(mine), red)            # step 2: buckets under the backward pass
            nb = 0 if red.buckets is None else len(red.buckets)
            red.close()
            same_steps = all((a is None) == (b is None) and (a is None or torch.equal(a, b)) for a, b in zip(g1.values(), g2.values()))
            worst, name = 0.0, ""
            if rank == 0:
                whole = grads(model, samples([0, 1]))         # one process, the whole batch
                for k, a in g2.items():
                    b = whole[k]
                    assert (a is None) == (b is None), k
                    if a is not None:
                        rel = float((a - b).norm() / b.norm().clamp(min=1e-12))
                        if rel > worst:
                            worst, name = rel, k
        dist.barrier()
        q.put((rank, same_steps, worst, name, nb))
    finally:
        dist.destroy_process_group()


def test_rccl_training_step_ranks_average_to_the_single_process_gradients():
    """a step over RCCL ranks on shares of a batch leaves the gradients of the one-process step on the whole
    batch (DDP's average of per-rank mean losses = the batch mean when every rank holds as many valid pixels: here all of them),
    through nmrf_amd.train.OverlappedGradientReducer -- flat on its first step, bucket by bucket under backward on the second, the
    same bits both times.  2 ranks when two GPUs are visible; on one GPU the same code runs on a 1-rank group (reducer inert)."""
    import torch.multiprocessing as mp
    from tests.conftest import record_note
    world = min(2, torch.cuda.device_count())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res                         # flat step == bucketed step, bit for bit, on every rank
    worst, name, nb = res[0][2], res[0][3], res[0][4]
    assert worst <= 1e-5, (worst, name)                         # (one rank: the same computation twice -> 0)
    record_note("RCCL training-step test on %d rank(s): gradients of the sharded step vs the one-process step on the whole batch, worst "
                "relative difference %.1e (%s); %d buckets%s" % (world, worst, name, nb, "" if world > 1 else
                                                                " -- one GPU visible: 1-rank group, the reducer is inert"))
