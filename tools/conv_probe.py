#!/usr/bin/env python
"""MIOpen timings of the stock 3x3 convolutions of the backbone / conv heads at KITTI sizes, NCHW vs channels_last."""
import torch
import torch.nn.functional as F
dev = "cuda"


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


shapes = [(2, 64, 64, 188, 624, 1), (2, 64, 96, 188, 624, 2), (2, 96, 96, 94, 312, 1), (2, 96, 128, 94, 312, 1),
          (2, 128, 128, 94, 312, 1), (2, 128, 256, 94, 312, 1), (2, 256, 256, 47, 156, 1), (1, 256, 128, 47, 156, 1)]
for (b, ci, co, h, w, s) in shapes:
    x = torch.randn(b, ci, h, w, device=dev); wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    xl, wl = x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last)
    fl = 2.0 * b * ci * co * 9 * (h // s) * (w // s)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        a = t(lambda: F.conv2d(x, wt, None, s, 1))
        c = t(lambda: F.conv2d(xl, wl, None, s, 1))
        print("conv3x3 s%d %3d->%3d @%dx%dx%d  find=%d : NCHW %7.1f us (%5.1f TF/s)   NHWC %7.1f us (%5.1f TF/s)"
              % (s, ci, co, b, h, w, bench, a, fl / a / 1e6, c, fl / c / 1e6), flush=True)
