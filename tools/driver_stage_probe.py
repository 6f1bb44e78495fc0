"""Where a batch-1 pair spends its host time in nmrf_amd.driver.StereoStream: per-batch wall time of the producer's pieces
(staging copies, H2D + graph replay + D2H enqueue) and of the consumer's (event wait, evicting read-out).
python tools/driver_stage_probe.py [--batch B] [--pairs N]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nmrf_amd import kernels as K                                      # noqa: E402
from nmrf_amd.config import get_cfg                                    # noqa: E402
from nmrf_amd.driver import StereoStream                               # noqa: E402
from nmrf_amd.models import build_model                                # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--pairs", type=int, default=96)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().to(dev)
base = [tuple(t.to(torch.uint8) for t in synthetic_pair(375, 1242, seed=1000 + i)[:2]) for i in range(8)]
pairs = [(i,) + base[i % 8] for i in range(args.pairs)]
T = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w


ORIG = (K.host_copy_nt, K.host_read_evict)
for threaded in (True, False):
    T.clear()
    K.host_copy_nt, K.host_read_evict = ORIG
    drv = StereoStream(model, dev, batch=args.batch, threaded=threaded)
    list(drv.run(iter(pairs[:4 * args.batch])))
    torch.cuda.synchronize()
    K.host_copy_nt = timed("host_copy_nt (one view)", K.host_copy_nt)
    K.host_read_evict = timed("host_read_evict (one result)", K.host_read_evict)
    drv._enqueue = timed("_enqueue (staging + launches of a batch)", drv._enqueue)
    t0 = time.perf_counter()
    n = sum(1 for _ in drv.run(iter(pairs)))
    dt = time.perf_counter() - t0
    print("threaded=%s batch %d: %.1f pairs/s (%.2f ms per batch)" % (threaded, args.batch, n / dt, dt / n * args.batch * 1e3))
    for k, v in T.items():
        v = sorted(v)
        print("   %-44s median %.3f ms  p90 %.3f  max %.3f  calls %d" % (k, v[len(v) // 2] * 1e3, v[int(len(v) * 0.9)] * 1e3, v[-1] * 1e3, len(v)))
    K.host_copy_nt, K.host_read_evict = ORIG

# GPU side: the same graph replayed back to back on resident inputs, then with the driver's per-batch copies around it
plan = next(iter(drv.plans.values()))
g = plan.graph
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
ev[0].record()
for _ in range(50):
    g.replay()
ev[1].record()
for _ in range(50):
    plan.static_in.copy_(plan.dev_in[0], non_blocking=True)
    g.replay()
    plan.out_dev[0][:plan.static_out.shape[0]].copy_(plan.static_out, non_blocking=True)
ev[2].record()
torch.cuda.synchronize()
print("GPU: graph replay alone %.3f ms; with the device-side in / out copies of the driver %.3f ms" % (
    ev[0].elapsed_time(ev[1]) / 50, ev[1].elapsed_time(ev[2]) / 50))
h2d = torch.cuda.Stream(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(h2d):
    for _ in range(50):
        plan.dev_in[0].copy_(plan.pin_in[0], non_blocking=True)
h2d.synchronize()
t1 = time.perf_counter()
with torch.cuda.stream(h2d):
    for _ in range(50):
        plan.pin_out[0].copy_(plan.out_dev[0], non_blocking=True)
h2d.synchronize()
t2 = time.perf_counter()
print("copies alone: H2D of a batch %.3f ms, D2H of a result %.3f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t1) / 50 * 1e3))
# the same copies while the graph replays on the main stream
torch.cuda.synchronize()
ev[0].record()
for _ in range(50):
    g.replay()
    with torch.cuda.stream(h2d):
        plan.dev_in[1].copy_(plan.pin_in[1], non_blocking=True)
        plan.pin_out[1].copy_(plan.out_dev[1], non_blocking=True)
ev[1].record()
torch.cuda.synchronize()
print("graph replay with an H2D + a D2H in flight on another stream: %.3f ms per replay" % (ev[0].elapsed_time(ev[1]) / 50))
