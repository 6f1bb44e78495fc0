"""Where the window-attention backward kernel spends its time: the product build against the A/B builds of tools/build_ab_winbwd.sh that
return after phase k, one process per build (the library is loaded once per process).
    python tools/winbwd_phase_bench.py            # drives the sub-processes
Geometry: the inference stage of a 512x256 training crop, batch 2 (1/8 grid 32x64 padded to 36x66, 4 labels, window 6, shifted)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import torch
    import nmrf_amd._lib as L
    if sys.argv[1] != "product":
        L.LIB_PATH = os.path.join(ROOT, "nmrf_amd/lib", sys.argv[1], "libnmrf_hip.so")
    from nmrf_amd import kernels as K
    for (b, hp, wp, n, win, shift) in ((2, 36, 66, 4, 6, 3), (2, 64, 128, 1, 4, 2)):
        t = b * hp * wp * n
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = torch.randn(t, 384, device="cuda", generator=g)
        table = 0.3 * torch.randn((2 * win - 1) ** 2, 384, device="cuda", generator=g)
        dout = torch.randn(t, 128, device="cuda", generator=g)
        f = lambda: K.window_attn_backward(qkv, table, dout, b, hp, wp, n, 4, win, shift, n > 1)
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("%-8s win %d N %d (%d windows x 4 heads x %d): %8.1f us per call (kernel + the partial sums of the table)" % (
            sys.argv[1], win, n, (hp // win) * (wp // win), b, e0.elapsed_time(e1) * 200), flush=True)
else:
    for v in ["product"] + ["ab_wb%d" % k for k in range(5)]:
        if v == "product" or os.path.exists(os.path.join(ROOT, "nmrf_amd/lib", v, "libnmrf_hip.so")):
            subprocess.call([sys.executable, os.path.abspath(__file__), v])
