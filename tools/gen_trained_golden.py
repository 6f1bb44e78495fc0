"""Parity fixtures on weights WITH STRUCTURE: train the REAL reference (/root/reference, imported read-only
through tools/refshim.py) on CPU in the build container for a few thousand AdamW steps, and keep

  tests/golden/trained_sd.npz   the trained state dict (every floating entry, fp32, compressed) -- data, not source
  tests/golden/e2e_t.npz        the reference's eval-mode outputs on a 136x328 pair with those weights (prob, seeds, proposal,
                                cost volume, infer_tgt / infer_delta / infer_score, refine_tgt, disp, disp_pred, disp_curr) +
                                the reference's loss curve of the run

The run follows the reference's training loop (main.py:403-430): model.train() + freeze_bn(), the reference's own Criterion
(NMRF.py:276-429) with its weight_dict, main.py's own build_optimizer (AdamW groups) and OneCycleLR (main.py:383-390), gradient
clipping at SOLVER.GRAD_CLIP.  Data: random 96x192 crops of closed-form synthetic pairs (nmrf_amd.utils.hashinit.synthetic_pair on a
160x416 canvas, so the crops see disparities 8..54 px at different phases), the analytic disparity as ground truth.  Initial
weights are the reference's own initialisation under torch.manual_seed(0), NOT the hash fill: nothing of this fixture depends on
nmrf_amd/utils/hashinit.py's scale rules.

The committed fixture: --steps 4000 --batch 2 (66 minutes on 5 threads; loss 693 -> 4.7, training EPE 38 -> 0.2 px, 0.83 px on the
unseen 136x328 pair; curve: profiles/r05_trained_fixture_curve.txt).

Run:  python tools/gen_trained_golden.py [--steps 4000] [--batch 2] [--out DIR]
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from nmrf_amd.utils.hashinit import synthetic_pair  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def reference_main_module():
    """main.py of the checkout, for its build_optimizer (main.py:186-245); its heavy imports are the refshim stand-ins."""
    spec = importlib.util.spec_from_file_location("_ref_main", os.path.join(refshim.REF, "main.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
        return mod
    except Exception as e:                                    # tensorboard etc. missing: restate the grouping below
        print("main.py not importable here (%s): using the restated optimizer groups" % type(e).__name__)
        return None


def optimizer_for(model, cfg):
    m = reference_main_module()
    if m is not None:
        return m.build_optimizer(model, cfg)
    # main.py:186-245 restated: plain group at BASE_LR, backbone at BASE_LR * BACKBONE_LR_DECAY, norms / tables with their own decay
    base, dec = cfg.SOLVER.BASE_LR, cfg.SOLVER.BACKBONE_LR_DECAY
    norm_t = (torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d, torch.nn.LayerNorm)
    plain, norm, bb, table = [], [], [], []
    for mn, mod in model.named_modules():
        for pn, p in mod.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = "%s.%s" % (mn, pn)
            if full.startswith("image_encoder.backbone"):
                bb.append(p)
            elif "relative_position_enc_table" in pn:
                table.append(p)
            elif isinstance(mod, norm_t):
                norm.append(p)
            else:
                plain.append(p)
    groups = [{"params": plain, "lr": base}, {"params": norm, "lr": base, "weight_decay": cfg.SOLVER.WEIGHT_DECAY_NORM},
              {"params": bb, "lr": base * dec, "weight_decay": cfg.SOLVER.BACKBONE_WEIGHT_DECAY},
              {"params": table, "lr": base, "weight_decay": 0.0}]
    return torch.optim.AdamW([g for g in groups if g["params"]], lr=base, weight_decay=cfg.SOLVER.WEIGHT_DECAY)


from synthetic_crops import Crops  # noqa: E402  (shared with tools/train_synthetic.py: the same data stream on the MI355X)


def train(steps, batch, log_every=10):
    torch.manual_seed(0)
    refshim.install()
    from nmrf.config import get_cfg
    from nmrf.models import build_model
    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.MAX_ITER", steps])
    cfg.freeze()
    model, crit = build_model(cfg)
    opt = optimizer_for(model, cfg)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, cfg.SOLVER.BASE_LR, steps + 100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="cos")
    data = Crops()
    curve = []
    t0 = time.time()
    for it in range(steps):
        model.train()
        model.freeze_bn()
        l, r, gt = data.batch(batch)
        valid = (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)
        sample = {"img1": l, "img2": r, "disp": gt, "valid": valid}
        out = model(sample)
        loss_dict = crit(out, sample, log=False)
        wd = crit.weight_dict
        loss = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
        for p in model.parameters():
            p.grad = None
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.GRAD_CLIP)
        opt.step()
        sched.step()
        curve.append(float(loss))
        if it % log_every == 0 or it == steps - 1:
            epe = float((out["disp"].detach() - gt).abs()[valid].mean())          # (disp = 4 * disp_pred, NMRF.py:245)
            print("step %4d  loss %.4f  train-EPE %.3f px  %.1f s" % (it, float(loss), epe, time.time() - t0), flush=True)
    return model, cfg, np.asarray(curve, np.float32)


def capture(model, cfg, h=136, w=328, seed=3100):
    model.eval()
    caps = {}

    def hook(key, with_input=False):
        def f(m, i, o):
            caps[key] = o
            if with_input:
                caps[key + "_in"] = i
        return f

    model.inference.register_forward_hook(hook("infer_tgt", True))
    model.refinement.register_forward_hook(hook("refine_tgt", True))
    model.dpn.propagation.register_forward_hook(hook("prop", True))
    model.infer_head.register_forward_hook(hook("infer_delta"))
    model.infer_score_head.register_forward_hook(hook("infer_score"))
    l, r, gt = synthetic_pair(h, w, seed=seed)
    with torch.no_grad():
        out = model({"img1": l[None].clone(), "img2": r[None].clone()})
    d = {"pair_hws": np.asarray([(h, w, seed)], np.int64), "max_disp": np.int64(cfg.DPN.MAX_DISP),
         "img1": _np(l[None]).astype(np.uint8), "img2": _np(r[None]).astype(np.uint8), "gt": _np(gt),
         "prob": _np(out["prob"]), "seeds": _np(out["initial_proposal"]).astype(np.int16), "proposal": _np(out["proposal"]),
         "disp": _np(out["disp"]), "disp_pred": _np(out["disp_pred"]), "disp_curr": _np(caps["refine_tgt_in"][0]),
         "cost_volume": _np(caps["prop_in"][0]), "infer_tgt": _np(caps["infer_tgt"].reshape(-1, 128)),
         "infer_delta": _np(caps["infer_delta"].reshape(-1, 64)), "infer_score": _np(caps["infer_score"].reshape(-1, 64)),
         "refine_tgt": _np(caps["refine_tgt"].reshape(-1, 128))}
    epe = float((out["disp"][0] - gt).abs().mean())
    print("eval EPE of the trained reference on the %dx%d pair: %.3f px" % (h, w, epe))
    d["ref_epe_vs_gt"] = np.float32(epe)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=OUT, help="directory for trained_sd.npz / e2e_t.npz (default tests/golden)")
    a = ap.parse_args()
    OUT = a.out
    torch.set_num_threads(a.threads)
    os.makedirs(OUT, exist_ok=True)
    model, cfg, curve = train(a.steps, a.batch)
    sd = {k: _np(v).astype(np.float32) for k, v in model.state_dict().items() if torch.is_floating_point(v)}
    path = os.path.join(OUT, "trained_sd.npz")
    np.savez_compressed(path, **sd)
    print("trained_sd", len(sd), "tensors", sum(v.size for v in sd.values()), "values", os.path.getsize(path) // 1024, "KiB")
    d = capture(model, cfg)
    d["loss_curve"] = curve
    d["train_steps"], d["train_batch"] = np.int64(a.steps), np.int64(a.batch)
    path = os.path.join(OUT, "e2e_t.npz")
    np.savez_compressed(path, **d)
    print("e2e_t", {k: v.shape for k, v in d.items() if hasattr(v, "shape")}, os.path.getsize(path) // 1024, "KiB")
