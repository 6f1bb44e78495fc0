#!/usr/bin/env python
"""hipBLASLt timings of the token linears at KITTI sizes (what a fused LN->GEMM->epilogue kernel has to beat)."""
import torch
import torch.nn.functional as F

dev = "cuda"
T = 29952


def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (k, n) in ((160, 384), (128, 384), (192, 384), (128, 128), (128, 512), (512, 128), (160, 128), (36, 128), (128, 64)):
    x = torch.randn(T, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    us = t(lambda: F.linear(x, w, b))
    print("linear  T=%d K=%3d N=%3d : %7.1f us  %6.1f TFLOP/s" % (T, k, n, us, 2.0 * T * k * n / us / 1e6), flush=True)
x = torch.randn(T, 512, device=dev)
print("gelu [T,512]            : %7.1f us" % t(lambda: F.gelu(x)))
x = torch.randn(T, 128, device=dev); y = torch.randn(T, 128, device=dev)
print("add  [T,128]            : %7.1f us" % t(lambda: x + y))
print("layer_norm [T,128]      : %7.1f us" % t(lambda: F.layer_norm(x, (128,))))
