"""Training on the MI355X end to end, against the reference's own run: the model built under torch.manual_seed(0) carries the REFERENCE's
initial weights bit for bit (same module tree, same init calls), tools/synthetic_crops.py deals the same batches in the same order as
tools/gen_trained_golden.py dealt the reference on CPU, and nmrf_amd.train.fit runs main.py's loop (reference optimizer groups, OneCycle
schedule, clip) with enable_grad_slice(full=True) -- every parameter trained.  The reference's loss curve of that run is stored in
tests/golden/e2e_t.npz (`loss_curve`), its result on the unseen 136x328 pair in `ref_epe_vs_gt`.  The two trajectories start at the same
loss and cannot stay bit-identical (MIOpen convolutions, split-fp16 products, discrete seed / winner-take-all decisions); what is compared
is the curve, window by window, and the final EPE.
    python tools/train_synthetic.py [--steps 4000] [--batch 2]"""
import argparse
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from synthetic_crops import Crops  # noqa: E402
from nmrf_amd.config import get_cfg  # noqa: E402
from nmrf_amd.models import build_model  # noqa: E402
from nmrf_amd.train import build_slice_optimizer, fit  # noqa: E402
from nmrf_amd.utils.hashinit import synthetic_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=4000)
ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
warnings.simplefilter("ignore")
ref = np.load(os.path.join(ROOT, "tests", "golden", "e2e_t.npz"))
ref_curve = ref["loss_curve"] if "loss_curve" in ref.files else None
torch.manual_seed(0)
cfg = get_cfg()
cfg.merge_from_list(["SOLVER.MAX_ITER", a.steps])
cfg.freeze()
model, crit = build_model(cfg)                                  # (CPU: the reference's initialisation order and generator)
model = model.to("cuda").train().enable_grad_slice(full=True)
opt = build_slice_optimizer(model, cfg)


class Stream:                                                   # one "epoch" = the whole run: `steps` batches of the shared crop stream
    def __iter__(self):
        data = Crops()
        for _ in range(a.steps):
            l, r, gt = data.batch(a.batch)
            yield {"img1": l, "img2": r, "disp": gt, "valid": (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)}


curve, t0 = [], time.time()


def on_step(step, lr, total, loss_dict):
    curve.append(total)
    if step == 1 or step % 250 == 0 or step == a.steps:
        w = curve[-50:]
        line = "step %4d  loss %9.4f  mean of the last %2d: %9.4f" % (step, total, len(w), sum(w) / len(w))
        if ref_curve is not None and step <= len(ref_curve):
            rw = ref_curve[max(0, step - 50):step]
            line += "   | reference (CPU): %9.4f  mean %9.4f" % (float(ref_curve[step - 1]), float(rw.mean()))
        print(line + "   %6.1f s" % (time.time() - t0), flush=True)


fit(model, crit, opt, Stream(), cfg, on_step=on_step)
torch.cuda.synchronize()
dt = time.time() - t0
print("%d steps of batch %d (96x192 crops) in %.1f s = %.1f ms per step, every parameter trained (%d tensors)" % (
    a.steps, a.batch, dt, dt / a.steps * 1e3, sum(1 for _ in model.parameters())))
model.eval()
h, w, seed = [int(v) for v in ref["pair_hws"][0]]
l, r, gt = synthetic_pair(h, w, seed=seed)
with torch.no_grad():
    out = model({"img1": l[None].clone(), "img2": r[None].clone()})
epe = float((out["disp"][0].cpu() - gt).abs().mean())
print("eval on the unseen %dx%d pair: EPE %.3f px against the analytic ground truth (the reference after its CPU run: %.3f px)" % (
    h, w, epe, float(ref["ref_epe_vs_gt"]) if "ref_epe_vs_gt" in ref.files else float("nan")))
