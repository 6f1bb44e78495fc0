"""The training criterion (nmrf_amd/models/criterion.py) against golden losses produced by the reference's own
Criterion (tools/gen_criterion_golden.py; nmrf/models/NMRF.py:276-429).  CPU, plain PyTorch; tolerance 1e-5 relative (fp32
reductions in a different order)."""
import os

import numpy as np
import pytest
import torch

from nmrf_amd.config import get_cfg
from nmrf_amd.models.criterion import Criterion, build_criterion

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "criterion.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def _case(c):
    t = {k[len(c) + 1:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith(c + "/") and k != c + "/loss_type"}
    out = {k: t[k] for k in ("proposal", "prob", "disp", "disp_pred")}
    aux = []
    while f"aux{len(aux)}_disp_pred" in t:
        i = len(aux)
        aux.append({k[len(f"aux{i}_"):]: v for k, v in t.items() if k.startswith(f"aux{i}_")})
    if aux:
        out["aux_outputs"] = aux
    want = {k[5:]: v for k, v in t.items() if k.startswith("loss/")}
    return out, {"disp": t["gt"].clone(), "valid": t["valid"]}, want, str(GOLD[c + "/loss_type"])


@pytest.mark.parametrize("c", CASES)
def test_losses_match_the_reference_criterion(c):
    out, tgt, want, loss_type = _case(c)
    cfg = get_cfg()
    cfg.SOLVER.LOSS_TYPE = loss_type
    got = Criterion({}, cfg)(out, tgt)
    assert set(got) == set(want)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-5, atol=1e-6, msg=lambda m: f"{c} {k}: {m}")


def test_losses_are_differentiable_and_the_empty_target_is_a_zero_loss():
    out, tgt, _, _ = _case(CASES[1])
    for k in ("proposal", "prob", "disp_pred"):
        out[k].requires_grad_(True)
    for a in out["aux_outputs"]:
        for v in a.values():
            v.requires_grad_(True)
    crit = build_criterion(get_cfg())
    losses = crit(out, tgt)
    total = sum(v for k, v in losses.items() if k != "epe_train")
    total.backward()
    assert out["proposal"].grad.abs().sum() > 0 and out["prob"].grad.abs().sum() > 0 and out["disp_pred"].grad.abs().sum() > 0
    tgt = {"disp": tgt["disp"], "valid": torch.zeros_like(tgt["valid"])}
    losses = crit(out, tgt, log=False)
    assert float(losses["loss_disp"]) == 0 and float(losses["loss_coarse_disp_0"]) == 0 and float(losses["loss_prop"]) == 0
    assert "epe_train" not in losses


def test_weight_dict_follows_the_solver_keys():
    cfg = get_cfg()
    crit = build_criterion(cfg)
    n = cfg.NMP.NUM_INFER_LAYERS + cfg.NMP.NUM_REFINE_LAYERS
    wd = crit.weight_dict
    assert wd["loss_disp"] == cfg.SOLVER.LOSS_WEIGHTS[-1] and wd["init"] == 1 and wd["proposal_disp"] == 1
    assert [k for k in wd if k.startswith("loss_coarse_disp_")] == [f"loss_coarse_disp_{i}" for i in range(cfg.NMP.NUM_INFER_LAYERS)]
    assert len(wd) == 2 + n
    cfg.SOLVER.AUX_LOSS = False
    assert set(build_criterion(cfg).weight_dict) == {"proposal_disp", "init", "loss_disp"}
    cfg.SOLVER.LOSS_TYPE = "L2"
    with pytest.raises(AssertionError):
        build_criterion(cfg)
