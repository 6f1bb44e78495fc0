#!/usr/bin/env python
"""Run-to-run determinism probe (GPU): every hand-written kernel, then whole forwards, must be bit-identical on
identical inputs.    python tools/determinism_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from nmrf_amd import kernels as K  # noqa: E402
from nmrf_amd.utils.hashinit import unit_noise, synthetic_pair  # noqa: E402

dev = "cuda"


def mk(key, *shape):
    import numpy as np
    return torch.from_numpy(unit_noise(key, int(np.prod(shape))).reshape(shape)).to(dev)


def rep(name, fn, n=10):
    ref = fn().clone()
    bad = 0
    for _ in range(n):
        o = fn()
        bad += int(not torch.equal(o, ref))
    print("%-40s %s" % (name, "deterministic" if bad == 0 else "DIFFERS in %d/%d runs, max %.3g" % (bad, n, float((o - ref).abs().max()))),
          flush=True)


for (b, hp, wp, n, win, sib) in ((2, 12, 18, 4, 6, True), (1, 48, 156, 4, 6, True), (2, 16, 28, 1, 4, False), (1, 96, 312, 1, 4, False)):
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", (2 * win - 1) ** 2, 384)
    for shift in (0, win // 2):
        rep("window %dx%dx%d B%d %dx%d shift %d" % (win, win, n, b, hp, wp, shift),
            lambda: K.window_attn(qkv, table, b, hp, wp, n, 4, win, shift, sib))
for (b, h, w, n) in ((2, 8, 13, 4), (1, 47, 156, 4)):
    qkv = mk("q2", b * h * w * n, 384)
    lv, lh = mk("lv", 64, 1, 3, 3), mk("lh", 64, 1, 3, 3)
    rep("stripe B%d %dx%d" % (b, h, w), lambda: K.stripe_attn(qkv, lv, lh, b, h, w, n))

from util import build_product  # noqa: E402
model = build_product(128, dev)
l, r, _ = synthetic_pair(64, 104, seed=50)
l2, r2, _ = synthetic_pair(64, 104, seed=51)
s = {"img1": torch.stack([l, l2]), "img2": torch.stack([r, r2])}
with torch.no_grad():
    for key in ("disp", "prob", "initial_proposal"):
        rep("model forward 64x104 B2 -> " + key, lambda: model(s)[key].float(), 6)
    for ov in ("0", "1"):
        os.environ["NMRF_OVERLAP"] = ov
        rep("model forward, NMRF_OVERLAP=" + ov, lambda: model(s)["disp"], 6)
    # stage-wise: backbone features, then the hot path from fixed features
    from nmrf_amd.frame_utils import InputPadder
    f = model.backbone(torch.cat([s["img1"], s["img2"]]).to(dev)) if hasattr(model, "backbone") else None
    if f is not None:
        rep("backbone (MIOpen) features", lambda: model.backbone(torch.cat([s["img1"], s["img2"]]).to(dev))[0], 6)
