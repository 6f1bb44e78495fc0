#!/usr/bin/env python
"""Which lines of nmrf_amd launch torch glue kernels (copies, cats, fills, elementwise) in one steady-state forward.
    python tools/glue_trace.py [--height 375 --width 1242]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrf_amd.config import get_cfg  # noqa: E402
from nmrf_amd.models import build_model  # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=375)
ap.add_argument("--width", type=int, default=1242)
args = ap.parse_args()
os.environ["NMRF_OVERLAP"] = "0"
cfg = get_cfg()
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
l, r, _ = synthetic_pair(args.height, args.width, seed=1000)
sample = {"img1": l[None].cuda(), "img2": r[None].cuda()}
with torch.no_grad():
    for _ in range(3):
        model(sample)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(sample)
        torch.cuda.synchronize()
avg = prof.key_averages(group_by_stack_n=12)
rows = []
for ev in avg:
    t = getattr(ev, "self_device_time_total", 0) or 0
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    frame = next((f for f in (ev.stack or []) if "nmrf_amd" in f and "kernels.py" not in f), (ev.stack or ["?"])[0])
    rows.append((t, ev.count, ev.key, frame.strip()[-120:]))
tot = 0.0
for t, n, name, frame in sorted(rows, reverse=True)[:60]:
    print("%8.1f us  x%-3d %-30s %s" % (t, n, name, frame))
    tot += t
print("listed total %.1f us" % tot)
