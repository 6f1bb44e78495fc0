"""Host cost of one hipGraph replay of the KITTI batch-1 forward against its GPU time: is the replay loop GPU- or host-bound?
python tools/graph_launch_probe.py [--batch B]   (run once per runtime setting, e.g. DEBUG_CLR_GRAPH_PACKET_CAPTURE=0/1)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nmrf_amd.config import get_cfg                                   # noqa: E402
from nmrf_amd.models import build_model                               # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().to(dev)
model.range_check = False
l, r, _ = synthetic_pair(375, 1242, seed=1000)
s = {"img1": l[None].repeat(args.batch, 1, 1, 1).to(dev), "img2": r[None].repeat(args.batch, 1, 1, 1).to(dev)}
with torch.no_grad():
    for _ in range(3):
        model(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(s)["disp"]
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
# (a) one replay at a time: host time of the call, then the GPU time of that replay alone
host, gpu = [], []
for _ in range(args.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    g.replay()
    host.append(time.perf_counter() - t0)
    e1.record()
    torch.cuda.synchronize()
    gpu.append(e0.elapsed_time(e1) * 1e-3)
# (b) back to back: enqueue loop alone, then the wall time until everything has run
t0 = time.perf_counter()
for _ in range(args.iters):
    g.replay()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
med = lambda v: sorted(v)[len(v) // 2]
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s batch %d: replay() host call median %.3f ms (min %.3f); one replay alone on the GPU %.3f ms; "
      "back to back: enqueue loop %.3f ms / replay, wall %.3f ms / replay -> %s-bound, %.1f pairs/s" % (
          os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset"), args.batch, med(host) * 1e3, min(host) * 1e3, med(gpu) * 1e3,
          t_enq / args.iters * 1e3, t_all / args.iters * 1e3, "host" if t_enq > 0.9 * t_all else "GPU", args.batch * args.iters / t_all))
