"""How large are the tap bounding boxes of the MSDA calls in the Swin-T configuration (hash weights)?  Per 8 x 8 query tile: the box
over all heads and points (what msda_fwd_d8_tiled_kernel stages), per head, and per (head, point).   python tools/dbg/msda_box_stats.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrf_amd.config import get_cfg
from nmrf_amd.models import build_model
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair
from nmrf_amd import kernels as K

cfg = get_cfg()
cfg.merge_from_list(["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32, "BACKBONE.COMPAT", False,
                     "DPN.MAX_DISP", 256])
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
l, r, _ = synthetic_pair(1000, 1500, seed=1000)
calls = []
orig = K.msda_forward
def tap(value, shapes, start, loc, w):
    calls.append((shapes.clone(), loc.detach().clone()))
    return orig(value, shapes, start, loc, w)
K.msda_forward = tap
import nmrf_amd.ops.functions as F_
with torch.no_grad():
    model({"img1": l[None].cuda(), "img2": r[None].cuda()})
for shapes, loc in calls[:4]:
    h, w = int(shapes[0, 0]), int(shapes[0, 1])
    b, lq, m, _, p, _ = loc.shape
    kq = int(round((lq / (h * w)) ** 0.5))
    qh, qw = h * kq, w * kq
    x = (loc[..., 0] * w - 0.5).floor().reshape(b, qh, qw, m, p)      # [b, qh, qw, m, p]
    y = (loc[..., 1] * h - 0.5).floor().reshape(b, qh, qw, m, p)
    ref_x = ((torch.arange(qw, device=x.device) + 0.5) / kq - 0.5).view(1, 1, qw, 1, 1)
    ref_y = ((torch.arange(qh, device=x.device) + 0.5) / kq - 0.5).view(1, qh, 1, 1, 1)
    dx, dy = x - ref_x, y - ref_y
    print("level %dx%d (kq %d): offsets from the reference point, px: |dx| mean %.1f p99 %.1f max %.0f; |dy| mean %.1f p99 %.1f max %.0f" % (
        h, w, kq, float(dx.abs().mean()), float(dx.abs().flatten().kthvalue(int(0.99 * dx.numel()))[0]), float(dx.abs().max()),
        float(dy.abs().mean()), float(dy.abs().flatten().kthvalue(int(0.99 * dy.numel()))[0]), float(dy.abs().max())))
    th, tw = qh // 8, qw // 8
    xt = x[:, :th * 8, :tw * 8].reshape(b, th, 8, tw, 8, m, p)
    yt = y[:, :th * 8, :tw * 8].reshape(b, th, 8, tw, 8, m, p)
    def area(dims):
        wx = xt.amax(dims) - xt.amin(dims) + 2
        wy = yt.amax(dims) - yt.amin(dims) + 2
        return (wx * wy).float()
    a_all, a_head, a_hp = area((2, 4, 5, 6)), area((2, 4, 6)), area((2, 4))
    for name, a, cap in (("all heads and points", a_all, 504), ("per head", a_head, 504 // 8 * 8), ("per (head, point)", a_hp, 504)):
        print("   box of an 8x8 tile, %-22s pixels: median %6.0f  p90 %6.0f  max %7.0f;  <= 504: %5.1f %%;  <= 126 (per head, 8 boxes of 32 B rows in the same LDS): %5.1f %%" % (
            name, float(a.median()), float(a.flatten().kthvalue(int(0.9 * a.numel()))[0]), float(a.max()), 100 * float((a <= 504).float().mean()), 100 * float((a <= 126).float().mean())))
    # spread WITHIN a head across the tile vs the head's mean offset
    mean_off = torch.stack((dx.mean((1, 2)), dy.mean((1, 2))), -1)      # [b, m, p, 2]
    print("   mean offset per (head, point), image 0, px:", [[round(float(v), 1) for v in mean_off[0, mm, 0]] for mm in range(m)])
