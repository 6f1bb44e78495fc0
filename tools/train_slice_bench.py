"""Time of one optimisation step on the part of the model this build can differentiate (nmrf_amd.train.train_step: training-mode forward
with the autograd tape, the reference's Criterion, backward through csrc/backward.hip, gradient clipping, AdamW) at a training-crop size,
and where it goes.  python tools/train_slice_bench.py [--height 256 --width 512 --batch 2 --steps 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nmrf_amd.config import get_cfg  # noqa: E402
from nmrf_amd.models import build_model  # noqa: E402
from nmrf_amd.train import build_slice_optimizer, slice_parameters, train_step  # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=256)
ap.add_argument("--width", type=int, default=512)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--full", action="store_true", help="enable_grad_slice(full=True): every parameter (convolutions on stock autograd)")
a = ap.parse_args()
cfg = get_cfg()
cfg.freeze()
model, crit = build_model(cfg)
model = apply_hash_weights(model).to("cuda").train().enable_grad_slice(full=a.full)
opt = build_slice_optimizer(model, cfg)
prs = [synthetic_pair(a.height, a.width, seed=100 + i) for i in range(a.batch)]
gt = torch.stack([p[2] for p in prs]).float()
sample = {"img1": torch.stack([p[0] for p in prs]), "img2": torch.stack([p[1] for p in prs]), "disp": gt, "valid": (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)}
import warnings
warnings.simplefilter("ignore")
train_step(model, crit, opt, sample)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = [train_step(model, crit, opt, sample)[0] for _ in range(a.steps)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
with torch.no_grad():
    model.eval()
    model(sample)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        model(sample)
    torch.cuda.synchronize()
    de = (time.perf_counter() - t1) / a.steps
n = sum(p.numel() for _, p in slice_parameters(model))
print("train_step (%s) at %dx%d, batch %d: %.1f ms per step (eval-mode forward of the same batch: %.1f ms); %d tensors / %.2f M parameters "
      "trained; loss %.2f -> %.2f over %d steps" % ("whole model" if a.full else "slice", a.width, a.height, a.batch, dt * 1e3, de * 1e3, len(slice_parameters(model)), n / 1e6,
                                                     losses[0], losses[-1], a.steps))
