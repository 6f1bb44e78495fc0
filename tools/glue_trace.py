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
rows = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.stack:
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue                                        # count the innermost aten op only
    frame = next((f for f in ev.stack if "nmrf_amd" in f and "kernels.py" not in f), ev.stack[0])
    key = (ev.name, frame.strip()[-110:])
    rows[key][0] += 1
    rows[key][1] += ev.device_time_total
tot = 0.0
for (name, frame), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f us  x%-3d %-28s %s" % (t, n, name, frame))
    tot += t
print("listed total %.1f us" % tot)
