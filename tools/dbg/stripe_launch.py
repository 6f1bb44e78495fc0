import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if len(sys.argv) > 1 and sys.argv[1] == "preload":
    from nmrf_amd import build
    print("stamps", build.library_stamp(build.LIB), build.library_stamp(build.LIB.replace("libnmrf_hip.so", "libnmrf_hip_debug.so")))
from nmrf_amd import kernels as K
dev = "cuda"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
for (b, h, w, n) in ((2, 7, 13, 4), (1, 5, 40, 1), (1, 47, 156, 4)):
    qkv = rnd(b * h * w * n, 384, seed=1, scale=1.5).to(dev)
    lv, lh = rnd(64, 1, 3, 3, seed=1).to(dev), rnd(64, 1, 3, 3, seed=2).to(dev)
    for kv in ((False, True) if n == 4 else (False,)):
        try:
            x = K.to_kv16(qkv) if kv else qkv
            o = K.stripe_attn(x, lv, lh, b, h, w, n, kv16=kv)
            torch.cuda.synchronize()
            print((b, h, w, n), "kv16" if kv else "fp32", "ok", float(o.abs().max()))
        except Exception as e:
            print((b, h, w, n), "kv16" if kv else "fp32", "FAILED", repr(e)[:200])
            st = ctypes.CDLL("libamdhip64.so")
            st.hipGetErrorString.restype = ctypes.c_char_p
            print("   last error:", st.hipGetErrorString(st.hipGetLastError()))
