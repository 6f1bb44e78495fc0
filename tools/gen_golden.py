"""Generate the golden parity fixtures under tests/golden/ by running the REAL
reference (/root/reference, imported read-only through tools/refshim.py) on
CPU in the build container.  Run:  python tools/gen_golden.py

Nothing of the reference travels: the fixtures hold only inputs (uint8 images,
crafted logits, MSDA operands) and the reference's outputs.  Weights are never
stored: they are the closed-form hash fill of nmrf_amd/utils/hashinit.py,
regenerated from state-dict keys wherever the fixtures are consumed.

Fixtures
  e2e_a.npz   52x100, MAX_DISP 128 (D=16), B=1: full per-stage captures
              (layer captures row-subsampled x2) - both pad branches odd
  e2e_b.npz   96x328, default MAX_DISP 320 (D=40), B=1: outputs + stage inputs / the tensors either side of the
              winner-take-all (cost volume, infer_tgt, infer_delta, infer_score, refine_tgt)
  e2e_c.npz   40x72,  MAX_DISP 128, B=2 (two different pairs): small outputs
  e2e_d.npz   136x1032, MAX_DISP 320 (D=40), B=2: small outputs; 1/8 grid 17x129 = 17 key tiles per horizontal stripe
              (the long-loop stripe kernel of the KITTI bench), images regenerated from (h, w, seed)
  e2e_k384.npz  48x392, MAX_DISP 384 (D=48: configs/kitti_mix_train.yaml:7, kitti_mix_2015_train.yaml:7), B=1: outputs + stage tensors
  e2e_z312.npz  40x320, MAX_DISP 312 (D=39, odd: configs/zero_shot_evaluation.yaml:11), B=1: outputs + stage tensors
  nms_cases.npz   crafted logits rows (ties, plateaus, NaN, ...) pushed through
              the reference's DPN.forward NMS+topk (DPN.py:119-125), D in {16,24,32,39,40,48}
  e2e_swin.npz / state_dict_keys.json   Swin-T + DeformNeck config: encoder features + outputs; key/shape listings
  e2e_train.npz   the reference in TRAINING mode (forward only): aux_outputs of every inference / refinement layer + its Criterion's
              losses on them (see run_train)
  msda.npz    ops/test.py known-answer case (seed 3) + model-shaped cases through
              ms_deform_attn_core_pytorch, fp32 outputs; gradients computed in fp64 autograd, stored fp32
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair, unit_noise  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def run_e2e(name, shapes_seeds, opts, full, store_images=True):
    model, cfg = refshim.build_reference_model(opts)
    apply_hash_weights(model)
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
    model.inference.ffn.register_forward_hook(hook("infer_ffn"))
    model.refinement.ffn.register_forward_hook(hook("refine_ffn"))
    model.dpn.proj.register_forward_hook(hook("context"))
    model.dpn.propagation.proj.register_forward_hook(hook("seed_embed"))
    for i, l in enumerate(model.dpn.propagation.layers):
        l.register_forward_hook(hook(f"prop_layer{i}"))
    for i, l in enumerate(model.inference.layers):
        l.register_forward_hook(hook(f"infer_layer{i}"))
        l.self_nmp.register_forward_hook(hook(f"infer_self{i}"))
    for i, l in enumerate(model.refinement.layers):
        l.register_forward_hook(hook(f"refine_layer{i}"))

    model.infer_head.register_forward_hook(hook("infer_delta"))
    model.infer_score_head.register_forward_hook(hook("infer_score"))

    lefts, rights = [], []
    for (h, w, seed) in shapes_seeds:
        l, r, _ = synthetic_pair(h, w, seed=seed)
        lefts.append(l)
        rights.append(r)
    img1, img2 = torch.stack(lefts), torch.stack(rights)
    with torch.no_grad():
        out = model({"img1": img1.clone(), "img2": img2.clone()})
    d = {"pair_hws": np.asarray(shapes_seeds, np.int64)}          # (h, w, seed) of nmrf_amd.utils.hashinit.synthetic_pair
    if store_images:
        d.update({"img1": _np(img1).astype(np.uint8), "img2": _np(img2).astype(np.uint8)})
    d.update({
        "max_disp": np.int64(cfg.DPN.MAX_DISP),
        "prob": _np(out["prob"]),
        "seeds": _np(out["initial_proposal"]).astype(np.int16),
        "proposal": _np(out["proposal"]),
        "disp": _np(out["disp"]),
        "disp_pred": _np(out["disp_pred"]),
        "disp_curr": _np(caps["refine_tgt_in"][0]),
    })
    if full:                                   # the tensors either side of the winner-take-all (NMRF.py:218-232) + stage inputs
        d["cost_volume"] = _np(caps["prop_in"][0])
        d["infer_tgt"] = _np(caps["infer_tgt"].reshape(-1, 128))
        d["infer_delta"] = _np(caps["infer_delta"].reshape(-1, 64))
        d["infer_score"] = _np(caps["infer_score"].reshape(-1, 64))          # Linear output, before the 0.25 of NMRF.py:221
        d["refine_tgt"] = _np(caps["refine_tgt"].reshape(-1, 128))
    if full is True:
        sub = lambda t: _np(t.reshape(-1, t.shape[-1]))[::2]
        d["context"] = _np(caps["context"])
        d["seed_embed_sub2"] = sub(caps["seed_embed"])
        d["prop_memory"] = _np(caps["prop"][0].reshape(-1, 128))
        d["infer_ffn_sub2"] = sub(caps["infer_ffn"])
        d["refine_ffn_sub2"] = sub(caps["refine_ffn"])
        for k in ("prop_layer0", "prop_layer1", "infer_self0", "infer_layer0", "infer_layer1",
                  "refine_layer0", "refine_layer1"):
            d[k + "_sub2"] = sub(caps[k])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(name, {k: v.shape for k, v in d.items() if hasattr(v, "shape")}, os.path.getsize(path) // 1024, "KiB")


def run_shipped_disp():
    """The disparity ranges of the reference's other shipped configs: MAX_DISP 384 -> D = 48 (the KITTI training
    configs) and MAX_DISP 312 -> D = 39, an odd number of hypotheses (zero-shot evaluation)."""
    run_e2e("e2e_k384", [(48, 392, 1020)], ["DPN.MAX_DISP", 384], full="stages")
    run_e2e("e2e_z312", [(40, 320, 1021)], ["DPN.MAX_DISP", 312], full="stages")


def run_train():
    """e2e_train.npz: the reference in TRAINING mode (model.train(), SOLVER.AUX_LOSS and NMP.RETURN_INTERMEDIATE at their defaults
    = True), forward only, under no_grad: no input padding (56x104 is a multiple of 8), `aux_outputs` of NMRF.py:259-273 -- one
    {disp_pred, logits_pred} per inference layer, one {disp_pred} per refinement layer but the last.  1/8 grid 7x13 (window
    padding 2+3 / 2+3), 1/4 grid 14x26 (1+1 / 1+1); B=1, MAX_DISP 128.  Also the reference Criterion's losses on that output
    against a seeded synthetic ground truth."""
    model, cfg = refshim.build_reference_model(["DPN.MAX_DISP", 128])
    apply_hash_weights(model)
    model.train()
    from nmrf.models import build_model
    crit = build_model(cfg)[1]
    shapes_seeds = [(56, 104, 1010)]
    ls, rs, gts = zip(*[synthetic_pair(h, w, seed=sd) for (h, w, sd) in shapes_seeds])
    img1, img2 = torch.stack(ls), torch.stack(rs)
    with torch.no_grad():
        out = model({"img1": img1.clone().float(), "img2": img2.clone().float()})
    assert "aux_outputs" in out and len(out["aux_outputs"]) == 5 + 4
    gt = torch.stack([torch.as_tensor(g) for g in gts]).float()
    valid = (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)
    with torch.no_grad():
        losses = crit(out, {"disp": gt, "valid": valid})
    d = {"pair_hws": np.asarray(shapes_seeds, np.int64), "max_disp": np.int64(cfg.DPN.MAX_DISP),
         "img1": _np(img1).astype(np.uint8), "img2": _np(img2).astype(np.uint8), "gt": _np(gt), "valid": _np(valid),
         "prob": _np(out["prob"]), "seeds": _np(out["initial_proposal"]).astype(np.int16), "proposal": _np(out["proposal"]),
         "disp": _np(out["disp"]), "disp_pred": _np(out["disp_pred"])}
    for i, a in enumerate(out["aux_outputs"]):
        for k, v in a.items():
            d["aux%d_%s" % (i, k)] = _np(v)
    for k, v in losses.items():
        d["loss/" + k] = _np(torch.as_tensor(v))
    # the reference's own parameter gradients of one training step's loss (main.py:413-420: sum_k weight_dict[k] * loss_dict[k], backward):
    # the three prediction heads and the two stage-final LayerNorms -- every parameter between the last attention kernel of a stage
    # and the loss (labels_curr and disp_curr are detached by the reference, NMRF.py:215,231: these gradients do not cross the stages)
    model.zero_grad(set_to_none=True)
    out_g = model({"img1": img1.clone().float(), "img2": img2.clone().float()})
    loss_g = crit(out_g, {"disp": gt, "valid": valid})
    total = sum(loss_g[k] * crit.weight_dict[k] for k in loss_g if k in crit.weight_dict)
    total.backward()
    d["loss_total"] = _np(total)

    def in_slice(name, heads, last):
        return name.startswith(heads) or (name.startswith(last) and name.split(".nmp.")[1].split(".")[0] in ("proj", "norm2", "mlp"))
    for name, p in model.named_parameters():                          # + the LAST block of either NMP stage: proj, norm2, mlp
        # (round 5, later: EVERY parameter of the refinement stage -- its window attention has a backward kernel now)
        # ... and of the inference stage: sibling + window attention backward; stored for its ffn, final norm and layers 0 and 4 (the
        # fixture would grow by 4 MB for the other three; the test asserts that they do get gradients)
        if name.startswith(("refinement.", "infer_head.", "infer_score_head.", "refine_head.", "inference.norm.", "inference.ffn.",
                            "inference.layers.0.", "inference.layers.4.", "dpn.mlp.")):         # dpn.mlp: the seed filter (the `init` loss)
            d["grad/" + name] = _np(p.grad)
    # (round 5, last: the convolutional side -- encoder, matching heads; `dpn.proj` is reached by the proposal loss only).  Stored in
    # full for a sample of tensors, and for EVERY one its norm and its projection on a fixed noise vector (utils.hashinit.unit_noise)
    conv_full = ("concatconv.3.weight", "gw.3.weight", "backbone.conv1.weight", "backbone.conv2.weight", "backbone.conv2.bias",
                 "backbone.layer1.0.conv1.weight", "backbone.layer2.0.downsample.0.weight", "backbone.layer2.0.downsample.0.bias",
                 "backbone.layer3.1.conv2.weight", "dpn.proj.3.weight")

    def conv_side(tag):
        from nmrf_amd.utils.hashinit import unit_noise
        for name, p in model.named_parameters():
            if not name.startswith(("backbone.", "concatconv.", "gw.", "dpn.proj.")):
                continue
            if p.grad is None:
                d[tag + "_none/" + name] = np.int8(1)
                continue
            g = _np(p.grad).astype(np.float64)
            d[tag + "_stat/" + name] = np.asarray([np.sqrt((g * g).sum()), (g.reshape(-1) * unit_noise("gproj/" + name, g.size)).sum()])
            if name in conv_full:
                d[tag + "/" + name] = _np(p.grad)
    conv_side("grad")
    # The proposal loss: Criterion.forward returns it as 'loss_prop' while the weight_dict of NMRF.py:432-447 names it 'proposal_disp',
    # so main.py:416's `if k in weight_dict` leaves it OUT of the trained loss and the propagation stage gets no gradient in the
    # reference's own step (checked: .grad is None).  Its gradient is still well defined: differentiated on its own here, for the
    # proposal head, the propagation's final norm and its last block.
    # (round 5, later: the whole propagation stage has a backward -- stored for the seed embedding, layers 0 and 4, the norm and the head)
    prop_names = [n for n, p in model.named_parameters()
                  if n.startswith(("dpn.prop_head.", "dpn.propagation.norm.", "dpn.propagation.cost_encoder.", "dpn.propagation.proj.",
                                   "dpn.propagation.layers.0.", "dpn.propagation.layers.4."))]
    assert all(dict(model.named_parameters())[n].grad is None for n in prop_names)
    model.zero_grad(set_to_none=True)
    out_p = model({"img1": img1.clone().float(), "img2": img2.clone().float()})
    loss_p = crit(out_p, {"disp": gt, "valid": valid})["loss_prop"]
    loss_p.backward()
    for n in prop_names:
        d["grad_prop/" + n] = _np(dict(model.named_parameters())[n].grad)
    conv_side("grad_prop")
    for k in ("disp", "disp_pred"):
        assert np.array_equal(_np(out_g[k]), d[k]), k                    # the same forward with and without no_grad
    path = os.path.join(OUT, "e2e_train.npz")
    np.savez_compressed(path, **d)
    print("e2e_train", {k: v.shape for k, v in d.items() if hasattr(v, "shape")}, os.path.getsize(path) // 1024, "KiB")


def run_train_b2():
    """e2e_train_b2.npz: the training step of run_train on a BATCH of two different pairs (64x128: 1/8 grid 8x16 -> window padding 4 / 2,
    1/4 grid 16x32): the reference's loss and, for EVERY parameter, the norm of its gradient and its projection on a fixed noise vector
    (nmrf_amd.utils.hashinit.unit_noise("gproj/<name>")) -- the batch dimension of every backward kernel against the reference's autograd."""
    model, cfg = refshim.build_reference_model(["DPN.MAX_DISP", 128, "SOLVER.LOSS_TYPE", "SMOOTH_L1"])
    apply_hash_weights(model)
    model.train()
    from nmrf.models import build_model
    crit = build_model(cfg)[1]                                  # (SMOOTH_L1, the Criterion's other loss type: its derivative is continuous, so
    #                                                             the comparison is free of the L1 sign flips that blur run_train's)
    shapes_seeds = [(64, 128, 1020), (64, 128, 1021)]
    ls, rs, gts = zip(*[synthetic_pair(h, w, seed=sd) for (h, w, sd) in shapes_seeds])
    img1, img2 = torch.stack(ls), torch.stack(rs)
    gt = torch.stack([torch.as_tensor(g) for g in gts]).float()
    valid = (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)
    d = {"pair_hws": np.asarray(shapes_seeds, np.int64), "max_disp": np.int64(cfg.DPN.MAX_DISP),
         "img1": _np(img1).astype(np.uint8), "img2": _np(img2).astype(np.uint8), "gt": _np(gt), "valid": _np(valid)}

    def stats(tag):
        for name, p in model.named_parameters():
            if p.grad is None:
                d[tag + "_none/" + name] = np.int8(1)
                continue
            g = _np(p.grad).astype(np.float64)
            d[tag + "_stat/" + name] = np.asarray([np.sqrt((g * g).sum()), (g.reshape(-1) * unit_noise("gproj/" + name, g.size)).sum(),
                                                   np.abs(g).max()])
    model.zero_grad(set_to_none=True)
    gw_outs, cc_outs = [], []
    if os.environ.get("NMRF_B2_FULL"):
        def keep(lst):
            def f(m, i, o):
                o.retain_grad()
                lst.append(o)
            return f
        model.gw.register_forward_hook(keep(gw_outs))
        model.concatconv.register_forward_hook(keep(cc_outs))
    out = model({"img1": img1.clone().float(), "img2": img2.clone().float()})
    losses = crit(out, {"disp": gt, "valid": valid})
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    for i, o in enumerate(gw_outs[:4]):
        d["dgw%d" % i] = _np(o.grad)
    for i, o in enumerate(cc_outs[:4]):
        d["dcc%d" % i] = _np(o.grad)
    d.update(loss_total=_np(total), seeds=_np(out["initial_proposal"]).astype(np.int16), disp_pred=_np(out["disp_pred"]))
    stats("grad")
    if os.environ.get("NMRF_B2_FULL"):                          # (diagnostic fixture: whole tensors of the matching heads)
        for name in ("gw.0.weight", "gw.3.weight", "concatconv.0.weight", "concatconv.3.weight"):
            d["grad/" + name] = _np(dict(model.named_parameters())[name].grad)
    model.zero_grad(set_to_none=True)
    out = model({"img1": img1.clone().float(), "img2": img2.clone().float()})
    crit(out, {"disp": gt, "valid": valid})["loss_prop"].backward()
    stats("grad_prop")
    path = os.path.join(OUT, "e2e_train_b2.npz")
    np.savez_compressed(path, **d)
    print("e2e_train_b2", len(d), "entries", os.path.getsize(path) // 1024, "KiB")


def run_train_trained():
    """e2e_train_t.npz: the training step at the TRAINED weights (tests/golden/trained_sd.npz) on the first batch of the training stream
    (tools/synthetic_crops.py: two 96x192 crops, default MAX_DISP 320 -> 40 disparity bins, L1 loss) -- the regime a real training run is
    in: structured weights, small losses, the sizes of tools/train_synthetic.py.  Per parameter: norm of the reference's gradient, its
    projection on a fixed noise vector, its largest entry."""
    from synthetic_crops import Crops
    model, cfg = refshim.build_reference_model([])
    with np.load(os.path.join(OUT, "trained_sd.npz")) as z:
        sd = {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in z.files}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("relative_position_index" in k or "indicator" in k for k in missing.missing_keys), missing
    model.train()
    from nmrf.models import build_model
    crit = build_model(cfg)[1]
    l, r, gt = Crops().batch(2)
    valid = (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)
    d = {"max_disp": np.int64(cfg.DPN.MAX_DISP), "img1": _np(l).astype(np.uint8), "img2": _np(r).astype(np.uint8), "gt": _np(gt),
         "valid": _np(valid)}
    assert np.array_equal(d["img1"].astype(np.float32), _np(l))
    model.zero_grad(set_to_none=True)
    out = model({"img1": l.clone(), "img2": r.clone()})
    losses = crit(out, {"disp": gt, "valid": valid})
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    d.update(loss_total=_np(total), seeds=_np(out["initial_proposal"]).astype(np.int16), disp_pred=_np(out["disp_pred"]))
    for name, p in model.named_parameters():
        if p.grad is None:
            d["grad_none/" + name] = np.int8(1)
            continue
        g = _np(p.grad).astype(np.float64)
        d["grad_stat/" + name] = np.asarray([np.sqrt((g * g).sum()), (g.reshape(-1) * unit_noise("gproj/" + name, g.size)).sum(), np.abs(g).max()])
    path = os.path.join(OUT, "e2e_train_t.npz")
    np.savez_compressed(path, **d)
    print("e2e_train_t", len(d), "entries", os.path.getsize(path) // 1024, "KiB; loss", float(total))


def run_train_swin():
    """e2e_train_swin.npz: the training step of the Swin-T + deformable-neck configuration (configs/sceneflow_swint.yaml keys) with
    BACKBONE.DROP_PATH 0 -- stochastic depth is random, so only the rate-0 step can be compared -- on one 64x128 pair: loss, seeds and per
    parameter the norm / noise projection / largest entry of the reference's gradient (trunk, neck incl. the MSDA offsets, heads, stages)."""
    opts = ["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32, "BACKBONE.COMPAT", False,
            "DPN.MAX_DISP", 128, "BACKBONE.DROP_PATH", 0.0]
    model, cfg = refshim.build_reference_model(opts)
    apply_hash_weights(model)
    model.train()
    from nmrf.models import build_model
    crit = build_model(cfg)[1]
    l, r, gt = synthetic_pair(64, 128, seed=2010)
    img1, img2, gt = l[None].float(), r[None].float(), torch.as_tensor(gt)[None].float()
    valid = (gt > 0) & (gt < cfg.SOLVER.MAX_DISP)
    d = {"max_disp": np.int64(cfg.DPN.MAX_DISP), "img1": _np(img1).astype(np.uint8), "img2": _np(img2).astype(np.uint8), "gt": _np(gt),
         "valid": _np(valid)}
    model.zero_grad(set_to_none=True)
    out = model({"img1": img1.clone(), "img2": img2.clone()})
    losses = crit(out, {"disp": gt, "valid": valid})
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()
    d.update(loss_total=_np(total), seeds=_np(out["initial_proposal"]).astype(np.int16), disp_pred=_np(out["disp_pred"]))
    for name, p in model.named_parameters():
        if p.grad is None:
            d["grad_none/" + name] = np.int8(1)
            continue
        g = _np(p.grad).astype(np.float64)
        d["grad_stat/" + name] = np.asarray([np.sqrt((g * g).sum()), (g.reshape(-1) * unit_noise("gproj/" + name, g.size)).sum(), np.abs(g).max()])
    path = os.path.join(OUT, "e2e_train_swin.npz")
    np.savez_compressed(path, **d)
    print("e2e_train_swin", len(d), "entries", os.path.getsize(path) // 1024, "KiB; loss", float(total))


def run_swin():
    """Swin-T + deformable neck config (configs/sceneflow_swint.yaml + MAX_DISP 256): encoder features, outputs,
    and the state-dict key/shape listing of both configs (for the strict-load contract tests)."""
    import json
    opts = ["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32,
            "BACKBONE.COMPAT", False, "DPN.MAX_DISP", 256]
    model, cfg = refshim.build_reference_model(opts)
    apply_hash_weights(model)
    caps = {}
    model.image_encoder.register_forward_hook(lambda m, i, o: caps.__setitem__("enc", o))
    model.refinement.register_forward_hook(lambda m, i, o: caps.__setitem__("refine_in", i))
    l, r, _ = synthetic_pair(60, 90, seed=2000)
    with torch.no_grad():
        out = model({"img1": l[None].clone(), "img2": r[None].clone()})
    d = {"img1": _np(l[None]).astype(np.uint8), "img2": _np(r[None]).astype(np.uint8),
         "feat4": _np(caps["enc"][0]), "prob": _np(out["prob"]), "seeds": _np(out["initial_proposal"]).astype(np.int16),
         "proposal": _np(out["proposal"]), "disp": _np(out["disp"]), "disp_curr": _np(caps["refine_in"][0])}
    path = os.path.join(OUT, "e2e_swin.npz")
    np.savez_compressed(path, **d)
    print("e2e_swin", {k: v.shape for k, v in d.items()}, os.path.getsize(path) // 1024, "KiB")
    keys = {"swin": {k: list(v.shape) for k, v in model.state_dict().items()}}
    model, _ = refshim.build_reference_model([])
    keys["default"] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)


def crafted_logits(d, rows_per_kind=24):
    """Rows of logits whose softmax exercises every NMS/top-k branch."""
    rows = []
    r = lambda key, n, salt: unit_noise(key, n, salt)
    for i in range(rows_per_kind):
        base = r("nms", d, i)
        rows.append(np.zeros(d, np.float32))                               # all tied
        rows.append(base.copy())                                           # flat-ish random: many maxima
        rows.append(base * 12)                                             # peaky: few bins > eps
        x = np.full(d, -20.0, np.float32); x[(7 * i) % d] = 5.0            # single peak, rest << eps
        rows.append(x)
        x = np.full(d, -20.0, np.float32); x[(3 * i) % d] = 4; x[(3 * i + 2) % d] = 4   # two equal peaks
        rows.append(x)
        x = base * 6; j = (5 * i) % (d - 3); x[j:j + 3] = x[j:j + 3].max() + 1          # plateau of 3
        rows.append(x)
        x = np.round(base * 3) * 2.0                                       # heavy exact ties between peaks
        rows.append(x.astype(np.float32))
        x = np.linspace(0, 6 + i * 0.3, d).astype(np.float32)              # monotone up (peak at D-1)
        rows.append(x)
        rows.append(x[::-1].copy())                                        # monotone down (peak at 0)
        x = base * 8; x[::2] = x[::2].max()                                # comb: equal maxima every other bin
        rows.append(x)
        x = base * 10; x[: d // 2] = -30                                   # left half dead (x<d zone look-alike)
        rows.append(x)
    nan_row = r("nmsn", d, 99) * 4
    nan_row[3] = np.nan
    rows.append(nan_row)
    return np.stack(rows).astype(np.float32)


def run_nms():
    out = {}
    for dmax in (128, 192, 256, 312, 320, 384):                       # D = 16, 24, 32, 39 (zero_shot_evaluation.yaml), 40, 48
        d = dmax // 8
        model, cfg = refshim.build_reference_model(["DPN.MAX_DISP", dmax])
        apply_hash_weights(model)
        logits = torch.from_numpy(crafted_logits(d))
        rows = logits.shape[0]

        class Fixed(torch.nn.Module):
            def forward(self, x):
                return logits[:, None, :]

        model.dpn.mlp = Fixed()
        cv = torch.zeros(1, 4, d, 1, rows)
        fmap = torch.from_numpy(unit_noise("f", 256 * rows, 1).reshape(1, 256, 1, rows))
        with torch.no_grad():
            _, prob, seeds, _ = model.dpn(cv, [fmap])
        out[f"logits_{d}"] = logits.numpy()
        out[f"prob_{d}"] = _np(prob)
        out[f"seeds_{d}"] = _np(seeds).astype(np.int16)
    path = os.path.join(OUT, "nms_cases.npz")
    np.savez_compressed(path, **out)
    print("nms_cases", {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")


def run_msda():
    refshim.install()
    from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as core
    out = {}

    def case(tag, n, m, dch, lq, shapes, p, gen):
        shapes_t = torch.as_tensor(shapes, dtype=torch.long)
        s = int(sum(h * w for h, w in shapes))
        l = len(shapes)
        value, loc, wgt = gen(n, s, m, dch, lq, l, p)
        res = core(value, shapes, loc, wgt)
        v64, l64, w64 = (t.double().requires_grad_(True) for t in (value, loc, wgt))
        r64 = core(v64, shapes, l64, w64)
        gout = torch.from_numpy(unit_noise("msda_g" + tag, r64.numel()).reshape(r64.shape)).double()
        gv, gl, gw = torch.autograd.grad(r64, (v64, l64, w64), gout)
        out.update({f"{tag}_value": _np(value), f"{tag}_shapes": shapes_t.numpy(), f"{tag}_loc": _np(loc),
                    f"{tag}_w": _np(wgt), f"{tag}_out": _np(res),
                    f"{tag}_gout": _np(gout).astype(np.float32),
                    f"{tag}_gvalue": _np(gv).astype(np.float32), f"{tag}_gloc": _np(gl).astype(np.float32),
                    f"{tag}_gw": _np(gw).astype(np.float32)})

    def gen_testpy(n, s, m, dch, lq, l, p):          # ops/test.py:14-35 operand recipe
        torch.manual_seed(3)
        value = torch.rand(n, s, m, dch) * 0.01
        loc = torch.rand(n, lq, m, l, p, 2)
        wgt = torch.rand(n, lq, m, l, p) + 1e-5
        wgt = wgt / wgt.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, wgt

    def gen_hash(tag):
        def g(n, s, m, dch, lq, l, p):
            value = torch.from_numpy(unit_noise(tag + "v", n * s * m * dch).reshape(n, s, m, dch))
            loc = torch.from_numpy(unit_noise(tag + "l", n * lq * m * l * p * 2).reshape(n, lq, m, l, p, 2)) * 0.6 + 0.5
            wgt = torch.from_numpy(unit_noise(tag + "w", n * lq * m * l * p).reshape(n, lq, m, l, p)).abs() + 1e-3
            wgt = wgt / wgt.sum((-1, -2), keepdim=True)
            return value, loc, wgt
        return g

    case("kat", 1, 2, 2, 2, [(6, 4), (3, 2)], 2, gen_testpy)
    case("neck", 2, 8, 8, 8 * 12, [(6, 10)], 4, gen_hash("neck"))          # DeformNeck shape family (1 level, D=8)
    case("ml", 2, 4, 16, 37, [(9, 7), (5, 4), (3, 2), (2, 1)], 3, gen_hash("ml"))
    case("odd", 1, 3, 5, 11, [(4, 5), (2, 3)], 2, gen_hash("odd"))
    path = os.path.join(OUT, "msda.npz")
    np.savez_compressed(path, **out)
    print("msda", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--train-b2-only" in sys.argv:
        run_train_b2()
        sys.exit(0)
    if "--nms-only" in sys.argv:
        run_nms()
        sys.exit(0)
    if "--shipped-disp-only" in sys.argv:
        run_shipped_disp()
        sys.exit(0)
    if "--post-only" in sys.argv:
        run_e2e("e2e_post", [(52, 100, 1006)], ["DPN.MAX_DISP", 128, "NMP.NORMALIZE_BEFORE", False], full=True)
        sys.exit(0)
    if "--train-swin-only" in sys.argv:
        run_train_swin()
        sys.exit(0)
    if "--train-trained-only" in sys.argv:
        run_train_trained()
        sys.exit(0)
    if "--train-only" in sys.argv:
        run_train()
        sys.exit(0)
    run_e2e("e2e_a", [(52, 100, 1000)], ["DPN.MAX_DISP", 128], full=True)
    run_e2e("e2e_b", [(96, 328, 1001)], [], full="stages")
    run_e2e("e2e_c", [(40, 72, 1002), (40, 72, 1003)], ["DPN.MAX_DISP", 128], full=False)
    # mid-size: 17 key tiles per horizontal stripe (129 pixels x 4 labels), both window paddings live; the images are the
    # closed-form synthetic pairs of these seeds and are regenerated where the fixture is consumed
    run_e2e("e2e_d", [(136, 1032, 1004), (136, 1032, 1005)], [], full=False, store_images=False)
    # NMP.NORMALIZE_BEFORE False (no shipped config sets it): the forward_post form of every message-passing block
    run_e2e("e2e_post", [(52, 100, 1006)], ["DPN.MAX_DISP", 128, "NMP.NORMALIZE_BEFORE", False], full=True)
    run_shipped_disp()
    run_nms()
    run_msda()
    run_swin()
    run_train()
    run_train_b2()
    run_train_swin()
    run_train_trained()            # (needs tests/golden/trained_sd.npz: tools/gen_trained_golden.py)
