"""StereoStream throughput at KITTI size, batch 1 and 8 (tools only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrf_amd.config import get_cfg                                                   # noqa: E402
from nmrf_amd.driver import StereoStream                                              # noqa: E402
from nmrf_amd.models import build_model                                               # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair                # noqa: E402

cfg = get_cfg()
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
l, r, _ = synthetic_pair(375, 1242, seed=1000)


def run(tag, drv, pairs, bs):
    list(drv.run(iter(pairs[:2 * bs])))
    torch.cuda.synchronize()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        n = sum(1 for _ in drv.run(iter(pairs)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, n / dt)
    print("batch %d %-34s %.1f pairs/s (%.2f ms per batch)" % (bs, tag, best, bs / best * 1e3), flush=True)


for bs in (1, 2, 8):
    pairs = [(i, l.to(torch.uint8), r.to(torch.uint8)) for i in range(8 * bs if bs > 1 else 32)]
    run("default", StereoStream(model, "cuda", batch=bs), pairs, bs)
    run("no clone", StereoStream(model, "cuda", batch=bs, copy_out=False), pairs, bs)
    run("eager", StereoStream(model, "cuda", batch=bs, graph=False), pairs, bs)
    pf = [(i, l, r) for i in range(8 * bs if bs > 1 else 32)]
    run("float32 host images", StereoStream(model, "cuda", batch=bs), pf, bs)
