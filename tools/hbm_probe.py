#!/usr/bin/env python
"""Achievable HBM/MALL write and read rates for activation-sized tensors (context for the fused-linear kernels)."""
import torch
dev = "cuda"


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for mb in (15, 46, 184, 736, 2944):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.randn(n, device=dev)
    # rotate over several buffers so that successive iterations do not hit lines still resident in L2 / MALL
    k = max(1, min(8, 4096 // mb))
    xs = [torch.empty(n, device=dev) for _ in range(k)]
    ys = [torch.randn(n, device=dev) for _ in range(k)]
    i = [0]

    def fill():
        i[0] = (i[0] + 1) % k; xs[i[0]].fill_(1.0)

    def rd():
        i[0] = (i[0] + 1) % k; return ys[i[0]].sum()

    def cp():
        i[0] = (i[0] + 1) % k; xs[i[0]].copy_(ys[i[0]])
    us_f, us_r, us_c = t(fill), t(rd), t(cp)
    print("%5d MB x%d buffers: fill %7.1f us = %5.2f TB/s | sum-read %7.1f us = %5.2f TB/s | copy %7.1f us = %5.2f TB/s (r+w)"
          % (mb, k, us_f, mb * 1.048576 / us_f, us_r, mb * 1.048576 / us_r, us_c, 2 * mb * 1.048576 / us_c), flush=True)
