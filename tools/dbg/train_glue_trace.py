"""Which Python lines launch the ATen glue kernels (copies, adds, fills, cats ...) of one training step?   python tools/dbg/train_glue_trace.py"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrf_amd.config import get_cfg
from nmrf_amd.models import build_model
from nmrf_amd.train import build_slice_optimizer, train_step
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair
warnings.simplefilter("ignore")
cfg = get_cfg(); cfg.freeze()
model, crit = build_model(cfg)
model = apply_hash_weights(model).to("cuda").train().enable_grad_slice(full=True)
opt = build_slice_optimizer(model, cfg)
prs = [synthetic_pair(256, 512, seed=100 + i) for i in range(2)]
gt = torch.stack([p[2] for p in prs]).float()
sample = {"img1": torch.stack([p[0] for p in prs]), "img2": torch.stack([p[1] for p in prs]), "disp": gt, "valid": (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)}
for _ in range(2):
    train_step(model, crit, opt, sample)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, crit, opt, sample)
    torch.cuda.synchronize()
avg = prof.key_averages(group_by_stack_n=16)
rows = []
for ev in avg:
    t = getattr(ev, "self_device_time_total", 0) or 0
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    frame = next((f for f in (ev.stack or []) if "nmrf_amd" in f), (ev.stack or ["?"])[0])
    rows.append((t, ev.count, ev.key, frame.strip()[-110:]))
tot = 0.0
for t, n, name, frame in sorted(rows, reverse=True)[:70]:
    print("%8.1f us  x%-4d %-34s %s" % (t, n, name, frame))
    tot += t
print("listed total %.1f us of aten device time %.1f us" % (tot, sum(r[0] for r in rows)))
