#!/usr/bin/env python
"""Find the first op whose output differs between two identical forwards (GPU).  Wraps every nmrf_amd.kernels entry
point plus F.linear / F.conv2d and compares the recorded outputs call by call."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nmrf_amd import kernels as K  # noqa: E402
from nmrf_amd.utils.hashinit import synthetic_pair  # noqa: E402
from util import build_product  # noqa: E402

os.environ["NMRF_OVERLAP"] = os.environ.get("NMRF_OVERLAP", "0")
log = []


def wrap(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        o = fn(*a, **k)
        outs = o if isinstance(o, tuple) else (o,)
        log.append((name, [t.detach().clone() for t in outs if torch.is_tensor(t)],
                    [tuple(t.shape) for t in a if torch.is_tensor(t)]))
        return o
    setattr(mod, name, w)


for nm in ("cost_volume", "dpn_filter_softmax", "nms_topk", "seed_features", "fourier_embed", "ln_concat", "add_ln_concat",
           "stripe_attn", "warp_corr_concat", "self_attn", "window_attn", "linear_smalln", "wta_median", "refine_epilogue",
           "instance_norm"):
    wrap(K, nm)
wrap(F, "linear")
wrap(F, "conv2d")
wrap(torch, "addmm")
wrap(torch, "mm")
wrap(torch, "matmul")

model = build_product(128, "cuda")
l, r, _ = synthetic_pair(64, 104, seed=50)
l2, r2, _ = synthetic_pair(64, 104, seed=51)
s = {"img1": torch.stack([l, l2]), "img2": torch.stack([r, r2])}
runs = []
with torch.no_grad():
    for i in range(3):
        log.clear()
        model(s)
        torch.cuda.synchronize()
        runs.append(list(log))
for a in (1, 2):
    print("run 0 vs run %d: %d / %d calls" % (a, len(runs[0]), len(runs[a])))
    shown = 0
    for i, ((n0, o0, sh), (n1, o1, _)) in enumerate(zip(runs[0], runs[a])):
        d = [float((x - y).abs().max()) for x, y in zip(o0, o1)]
        if any(v != 0 for v in d) and shown < 6:
            print("   call %3d %-18s inputs %s  max|diff| %s" % (i, n0, sh, d))
            shown += 1
    if not shown:
        print("   identical")
