#!/usr/bin/env python
"""Time every conv3x3_split call of one KITTI forward on its REAL operands, and again with the input replaced by noise of the
same scale (data-dependence probe: the conv heads ran 2x slower in the model than in tools/kernel_bench.py).
    python tools/conv_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrf_amd import kernels as K  # noqa: E402
from nmrf_amd.config import get_cfg  # noqa: E402
from nmrf_amd.models import build_model  # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair  # noqa: E402

os.environ["NMRF_OVERLAP"] = "0"
cfg = get_cfg()
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
l, r, _ = synthetic_pair(375, 1242, seed=1000)
sample = {"img1": l[None].cuda(), "img2": r[None].cuda()}
calls = []
orig = K.conv3x3_split


def spy(x, packed, co, stats=None, eps=1e-5):
    calls.append((x.clone(), packed, co, None if stats is None else stats.clone()))
    return orig(x, packed, co, stats, eps)


with torch.no_grad():
    model(sample)
    K.conv3x3_split = spy
    model(sample)
    K.conv3x3_split = orig
torch.cuda.synchronize()


def t(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for x, packed, co, stats in calls:
    g = torch.Generator(device="cuda").manual_seed(1)
    noise = (torch.rand(x.shape, device="cuda", generator=g) * 2 - 1) * float(x.abs().mean()) * 2
    sw = packed[0]
    wnoise = torch.randint(-2 ** 31, 2 ** 31 - 1, sw.shape, device="cuda", dtype=torch.int64).to(torch.int32)
    real = t(lambda: orig(x, packed, co, stats))
    rnd = t(lambda: orig(noise, packed, co, stats))
    finite = bool(torch.isfinite(x).all())
    print("conv3x3_split %s -> %d strips %d groups %d stats %d | real %.1f us  noise-input %.1f us | x: mean|.| %.3g max %.3g "
          "zeros %.1f%% finite %s | w stream: %.1f%% zero words" % (
              tuple(x.shape), co, packed[1], packed[2], stats is not None, real, rnd, float(x.abs().mean()), float(x.abs().max()),
              100 * float((x == 0).float().mean()), finite, 100 * float((sw == 0).float().mean())), flush=True)
