"""Swin-T + deformable-neck configuration (BASELINE config 5 geometry; configs/sceneflow_swint.yaml keys).
CPU: strict state-dict contract for both configs against key/shape listings dumped from the real reference,
and the stock encoder code (everything except the MSDA operator, replaced here by the ORACLE's msda_core as the
checker) against the reference's encoder features.  GPU: the same encoder on the HIP MSDA operator, and the
hot path from the reference's encoder features."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import nmrf_oracle as O
from tests.util import GOLDEN, build_product, golden, report, t

SWIN_OPTS = ["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32, "BACKBONE.COMPAT", False]


def _keys(which):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return json.load(f)[which]


@pytest.mark.parametrize("which,opts", [("default", []), ("swin", SWIN_OPTS)])
def test_state_dict_matches_reference_listing(which, opts):
    sd = build_product(256 if which == "swin" else 320, opts=opts).state_dict()
    want = _keys(which)
    assert set(sd) == set(want), (sorted(set(want) - set(sd))[:5], sorted(set(sd) - set(want))[:5])
    bad = [k for k in want if list(sd[k].shape) != want[k]]
    assert not bad, bad[:5]


def test_swin_encoder_stock_code_vs_reference_features(monkeypatch):
    g = golden("e2e_swin")
    model = build_product(256, opts=SWIN_OPTS)
    from nmrf_amd.ops import functions as Fn

    def oracle_apply(value, shapes, start, loc, w, step):          # the oracle stands in for the GPU-only operator
        return O.msda_core(value, [tuple(int(v) for v in r) for r in shapes.tolist()], loc, w)

    monkeypatch.setattr(Fn.MSDeformAttnFunction, "apply", staticmethod(oracle_apply))
    img = torch.cat((t(g["img1"]).float(), t(g["img2"]).float()))
    img = F.pad(img, (0, (-img.shape[-1]) % 32, 0, (-img.shape[-2]) % 32), mode="replicate")
    with torch.no_grad():
        f4, f8 = model.image_encoder(img)
    report("swin feat 1/4", f4, t(g["feat4"]), 2e-4, 1e-4)
    assert f8.shape == (2, 128, f4.shape[2] // 2, f4.shape[3] // 2)


@pytest.mark.gpu
def test_swin_model_on_gpu():
    g = golden("e2e_swin")
    model = build_product(256, "cuda", opts=SWIN_OPTS)
    img1, img2 = t(g["img1"]).float(), t(g["img2"]).float()
    with torch.no_grad():
        out = model({"img1": img1, "img2": img2})
        img = torch.cat((img1, img2)).cuda()
        img = F.pad(img, (0, (-img.shape[-1]) % 32, 0, (-img.shape[-2]) % 32), mode="replicate")
        f4, _ = model.image_encoder(img)
        report("swin feat 1/4 (HIP MSDA)", f4.cpu(), t(g["feat4"]), 5e-4, 2e-4)
        # hot path from the reference's own encoder features: seeds must be bit-exact
        ref4 = t(g["feat4"]).cuda()
        ref8 = F.avg_pool2d(ref4, 2, 2)
        from tests.test_model_gpu import _gpu_chain_side, _oracle_chain_side
        from tests.util import check_chain, oracle_cfg, oracle_weights
        out_hw = g["disp"].shape[-2:]
        hp, cand = _gpu_chain_side(model, [ref8[:1].contiguous(), ref4[:1].contiguous()],
                                   [ref8[1:].contiguous(), ref4[1:].contiguous()], out_hw)
        w, cfg = oracle_weights(256, tuple(SWIN_OPTS)), oracle_cfg(256, divis_by=32)
        base = _oracle_chain_side(O.hot_path(w, cfg, ref8.cpu(), ref4.cpu(), None, out_hw, stages={}))
        # (one consistent chain: scores, decisions and disparities of the same oracle run on this host -- tests/test_model_gpu.py)
        l4, r4 = ref4[:1].cpu(), ref4[1:].cpu()
        check_chain("swin hot path from reference features", cand, base, lambda dq: O.refine_from(w, cfg, dq, l4, r4, out_hw)[0])
    assert torch.equal(hp["initial_proposal"].cpu().long(), t(g["seeds"]).long())
    report("prob", hp["prob"].cpu(), t(g["prob"]), 5e-6)
    assert out["disp"].shape == (1, 60, 90) and torch.isfinite(out["disp"]).all()
    mism = (out["initial_proposal"].cpu().long() != t(g["seeds"]).long()).any(-1).float().mean()
    assert mism < 0.05, f"{float(mism)} of the pixels changed seeds through the GPU encoder"


def test_shipped_swint_yaml_keys_build_verbatim():
    """configs/sceneflow_swint.yaml and kitti_mix_train_swint.yaml of the reference set BACKBONE.DROP_PATH 0.4; stochastic depth is
    the identity in eval mode and per-sample branch dropping in training mode."""
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    cfg = get_cfg()
    cfg.merge_from_list(["DATASETS.DIVIS_BY", 32, "BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128,
                         "BACKBONE.DROP_PATH", 0.4, "BACKBONE.COMPAT", False])
    cfg.freeze()
    model, criterion = build_model(cfg)
    assert criterion.weight_dict["loss_disp"] == cfg.SOLVER.LOSS_WEIGHTS[-1] and hasattr(model, "image_encoder") and model.divis_by == 32
    rates = [blk.drop_path for layer in model.image_encoder.backbone.layers for blk in layer.blocks]
    assert len(rates) == 12 and rates[0] == 0.0 and abs(rates[-1] - 0.4) < 1e-6        # the decay rule of swin.py:592-609
    # stochastic depth: per-sample masks in training mode (timm DropPath), the identity in eval mode
    blk = model.image_encoder.backbone.layers[3].blocks[1]
    x = torch.randn(6, 2, 2, 768, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(3)
    with torch.no_grad():
        y_eval = blk.eval()(x, None)
        y_train = blk.train()(x, None)
        same = [bool(torch.equal(a, b)) for a, b in zip(y_eval, y_train)]
    assert not all(same)                                         # (rate 0.4, two branches, six samples)
    with pytest.raises(RuntimeError, match="no CPU"):            # the training-mode forward runs (round 5) -- on the MI355X only
        model.train()({"img1": torch.zeros(1, 3, 32, 32), "img2": torch.zeros(1, 3, 32, 32)})
