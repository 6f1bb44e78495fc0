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


def _bench(cmd, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, cwd=root, env=env, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


@pytest.mark.skipif(torch.cuda.is_available(), reason="plumbing mode is what bench.py does on a host WITHOUT a GPU")
def test_bench_launches_its_own_workers():
    """`python bench.py --gpus 2` (the driver's N=1 command shape at N=2, no torchrun): bench.py starts its two ranks itself
    (mp.start_processes, as the reference's main.py:129-142), shards the batch, gathers the results over gloo, rank 0 prints ONE
    JSON line.  Without a GPU the forward is a stand-in and the line says so ("valid": false, no value)."""
    r, recs = _bench(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(recs) == 1
    rec = recs[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["valid"] is False and rec["value"] is None
    assert rec["config"]["global_batch"] == 6 and rec["config"]["result_gather"] and rec["config"]["gather_correct"]
    assert rec["config4"]["global_batch"] == 16 and rec["config4"]["gather_correct"]        # config 4's leg: 8 pairs per rank


@pytest.mark.skipif(torch.cuda.is_available(), reason="plumbing mode is what bench.py does on a host WITHOUT a GPU")
def test_bench_under_torchrun_and_loud_failure_without_gpu():
    port = _free_port()
    r, recs = _bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and recs[0]["config"]["gather_correct"]
    r, recs = _bench(["bench.py", "--steps", "1", "--warmup", "1"])          # N=1 has nothing to fall back to: no line, an error
    assert r.returncode != 0 and not recs and "needs an MI355X" in r.stderr


def test_numa_pinning_from_a_sysfs_tree(tmp_path):
    """nmrf_amd.parallel.pin_to_gpu_numa: GPU PCI address -> numa_node -> that node's cpulist, intersected with the CPUs the
    process may use; -1 / missing entries leave the affinity alone and say why.  Runs in a child process (it changes affinity)."""
    import subprocess
    import sys
    allowed = sorted(os.sched_getaffinity(0))
    root = tmp_path / "sys"
    dev = root / "bus/pci/devices/0000:c5:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = root / "devices/system/node/node1"
    node.mkdir(parents=True)
    first = allowed[0]
    (node / "cpulist").write_text("%d-%d,%d\n" % (first, first, 10 ** 6))          # one CPU we own + one that does not exist
    single = root / "bus/pci/devices/0000:05:00.0"
    single.mkdir(parents=True)
    (single / "numa_node").write_text("-1\n")
    code = ("import os, json, sys; sys.path.insert(0, %r); from nmrf_amd.parallel import pin_to_gpu_numa, _parse_cpulist;"
            "assert _parse_cpulist('0-2,8,10-11') == [0, 1, 2, 8, 10, 11];"
            "a = pin_to_gpu_numa(0, sysfs=%r, pci='0000:05:00.0'); b0 = sorted(os.sched_getaffinity(0));"
            "m = pin_to_gpu_numa(0, sysfs=%r, pci='0000:aa:00.0');"
            "b = pin_to_gpu_numa(0, sysfs=%r, pci='0000:c5:00.0');"
            "print(json.dumps([a, b0, m, b, sorted(os.sched_getaffinity(0))]))") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(root), str(root), str(root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    import json
    a, b0, m, b, after = json.loads(r.stdout.strip().splitlines()[-1])
    assert a["pinned"] is False and a["numa_node"] == -1 and b0 == allowed
    assert m["pinned"] is False and "why" in m
    assert b["pinned"] is True and b["numa_node"] == 1 and b["cpus"] == 1 and after == [first]


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nmrf_amd.train import allreduce_gradients
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2)),
              torch.nn.Parameter(torch.zeros(4))]
        ps[2].requires_grad_(False)                                   # frozen: not part of the bucket
        ps[0].grad = torch.full((3, 5), float(rank + 1))
        if rank == 0:
            ps[1].grad = torch.arange(7.0)                             # rank 1 has no gradient for this one: zeros
        # ps[3]: trainable, but NO rank has a gradient (a loss term without weight): must stay None -- AdamW skips it as with one rank
        n = allreduce_gradients(ps)
        ok = (n == 22 and torch.allclose(ps[0].grad, torch.full((3, 5), 1.5)) and torch.allclose(ps[1].grad, torch.arange(7.0) / 2)
              and ps[2].grad is None and ps[3].grad is None)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nmrf_amd.train import OverlappedGradientReducer, allreduce_gradients

        def make():
            torch.manual_seed(0)
            return torch.nn.ModuleDict(dict(l1=torch.nn.Linear(4, 8), l2=torch.nn.Linear(8, 3), side=torch.nn.Linear(4, 3),
                                            unused=torch.nn.Linear(3, 2)))         # `unused`: no loss term reaches it on any rank

        def loss_of(m, x, with_side):
            y = m["l2"](torch.relu(m["l1"](x))).sum()
            return y + m["side"](x).pow(2).sum() if with_side else y
        a, b = make(), make()
        red = OverlappedGradientReducer(list(a.parameters()), bucket_bytes=64)      # tiny buckets: several all-reduces per step
        ok, launches = True, []
        for step in range(4):
            g = torch.Generator().manual_seed(100 * step + rank)
            x = torch.randn(5, 4, generator=g)
            with_side = not (rank == 1 and step == 2)                 # one rank, one step: a live parameter without a gradient -> zeros
            for m in (a, b):
                for p in m.parameters():
                    p.grad = None
            red.prepare()
            loss_of(a, x, with_side).backward()
            early = len(red._inflight)                                # buckets already in flight when backward returns
            n = red.finish()
            loss_of(b, x, with_side).backward()
            allreduce_gradients(list(b.parameters()))
            launches.append((early, n))
            for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
                if k.startswith("unused"):
                    ok = ok and pa.grad is None and pb.grad is None
                else:
                    ok = ok and pa.grad is not None and torch.equal(pa.grad, pb.grad)
        nb = len(red.buckets)
        red.close()
        q.put((rank, bool(ok), nb, launches))
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_reducer_world2():
    """nmrf_amd.train.OverlappedGradientReducer (the DDP step): buckets in reverse parameter order all-reduced
    while backward still runs, the same collectives in the same order on both ranks -- gradients equal, bit for bit, to the flat
    all-reduce behind the backward pass on every step, including the one where a rank has no gradient for a live parameter; a
    parameter no rank has a gradient for keeps None."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    for _, _, nb, launches in res:
        assert nb >= 3                                                # 64-byte buckets: l2, l1 and side end up in different buckets
        assert launches[0][0] == 0                                    # the first step is the flat path
        assert any(early > 0 for early, _ in launches[1:]), launches  # later steps: buckets left during the backward pass
    # one rank: the reducer is inert
    from nmrf_amd.train import OverlappedGradientReducer
    r1 = OverlappedGradientReducer([torch.nn.Parameter(torch.zeros(2))])
    assert not r1.active and r1.finish() == 0


def test_gradient_bucket_allreduce_world2():
    """nmrf_amd.train.allreduce_gradients: the DDP gradient average of main.py:334-339 as one flat bucket (gloo here, RCCL on GPUs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
    from nmrf_amd.train import allreduce_gradients
    assert allreduce_gradients([torch.nn.Parameter(torch.zeros(2))]) == 0           # no process group: identity


class _TinyStereo(torch.nn.Module):
    """A stand-in with the surface nmrf_amd.train.fit / train_step touch (grad_slice, freeze_bn, model(sample) -> {'disp'}): the loop, the
    gradient average and the rank-0 checkpoints are host logic -- the real model's forward needs the MI355X."""

    def __init__(self):
        super().__init__()
        self.grad_slice = True
        self.lin = torch.nn.Linear(3, 1)
        self.frozen_bn_calls = 0

    def freeze_bn(self):
        self.frozen_bn_calls += 1

    def forward(self, sample):
        return {"disp": self.lin(sample["img1"]).squeeze(-1)}


class _TinyCriterion:
    weight_dict = {"loss_disp": 2.0}

    def __call__(self, out, sample):
        return {"loss_disp": (out["disp"] - sample["disp"]).abs().mean(), "epe_train": (out["disp"] - sample["disp"]).abs().mean().detach()}


def _fit_worker(rank, world, port, q, ckpt_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nmrf_amd.config import get_cfg
        from nmrf_amd.train import fit
        cfg = get_cfg()
        cfg.merge_from_list(["SOLVER.MAX_ITER", 5, "SOLVER.CHECKPOINT_PERIOD", 2, "SOLVER.LATEST_CHECKPOINT_PERIOD", 4])
        cfg.freeze()
        torch.manual_seed(0)
        model = _TinyStereo()                                          # same initial weights on both ranks (as DDP broadcasts them)
        opt = torch.optim.AdamW(model.parameters(), lr=cfg.SOLVER.BASE_LR)
        g = torch.Generator().manual_seed(100 + rank)                  # each rank its own shard of the data
        batches = [{"img1": torch.randn(8, 3, generator=g), "img2": None, "disp": torch.randn(8, generator=g), "valid": torch.ones(8, dtype=torch.bool)}
                   for _ in range(2)]
        epochs, lrs = [], []
        step, epoch = fit(model, _TinyCriterion(), opt, batches, cfg, checkpoint_dir=ckpt_dir, set_epoch=epochs.append,
                          on_step=lambda s, lr, total, ld: lrs.append(lr))
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        q.put((rank, step, epoch, epochs, model.frozen_bn_calls, w.tolist(), len(lrs)))
    finally:
        dist.destroy_process_group()


def test_fit_loop_world2_rank0_checkpoints_and_gradient_average(tmp_path):
    """nmrf_amd.train.fit on two gloo ranks with a stand-in model: five steps over two-batch epochs (set_epoch 0, 1, 2; freeze_bn per epoch),
    the ranks see different data and end with IDENTICAL weights (the gradient average of main.py:334-339), rank 0 alone writes
    step_000002 / step_000004 / step_000005 (MAX_ITER) and checkpoint_latest (step 4)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, r1) = res
    assert r0[1:5] == (5, 2, [0, 1, 2], 3) and r1[1:5] == r0[1:5] and r0[6] == 5
    assert r0[5] == r1[5], (r0[5], r1[5])                             # bit-identical parameters on both ranks
    assert sorted(os.listdir(str(tmp_path))) == ["checkpoint_latest.pth", "step_000002.pth", "step_000004.pth", "step_000005.pth"]
    latest = torch.load(str(tmp_path / "checkpoint_latest.pth"))
    assert latest["step"] == 4 and latest["epoch"] == 1 and set(latest) == {"model", "optimizer", "step", "epoch"}
