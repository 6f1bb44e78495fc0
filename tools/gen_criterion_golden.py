"""Golden vectors for the training criterion: runs the REFERENCE's Criterion (imported from /root/reference through
tools/refshim.py -- build container only) on seeded synthetic output dictionaries and stores inputs + the losses it returns in
tests/golden/criterion.npz.  The fixture is data only; tests/test_criterion.py replays it through nmrf_amd.models.criterion.

Cases: (0) eval-style dictionary, L1; (1) with aux_outputs (coarse + refine heads), L1; (2) SMOOTH_L1 with aux_outputs;
(3) targets with out-of-range / invalid regions, disparities beyond the last histogram bin and beyond the image border.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim                                                # noqa: E402

refshim.install()
from nmrf.config import get_cfg                               # noqa: E402
from nmrf.models.NMRF import Criterion                        # noqa: E402


def case(seed, loss_type, aux, hard):
    g = torch.Generator().manual_seed(seed)
    b, h, w, n, d = 2, 4, 5, 4, 40
    H, W = 8 * h, 8 * w
    gt = torch.rand(b, H, W, generator=g) * (60 if not hard else 400)
    valid = torch.rand(b, H, W, generator=g) > (0.2 if not hard else 0.5)
    if hard:
        gt[0, :8, :8] = 0                                     # a cell without any valid pixel
        valid[1, 8:16] = False
        gt[1, 16:24, 16:24] = 318.5                           # past the last bin (D = 40 -> 39)
    out = {"proposal": torch.rand(b, h * w, n, generator=g) * 8,
           "prob": torch.softmax(torch.randn(b * h * w, d, generator=g) * 2, -1),
           "disp": torch.rand(b, H, W, generator=g) * 60,
           "disp_pred": torch.rand(b, H, W, generator=g) * 15}
    if aux:
        out["aux_outputs"] = [{"disp_pred": torch.rand(b, H, W, n, generator=g) * 8, "logits_pred": torch.randn(b, H, W, n, generator=g)},
                              {"disp_pred": torch.rand(b, H, W, n, generator=g) * 8, "logits_pred": torch.randn(b, H, W, n, generator=g)},
                              {"disp_pred": torch.rand(b, H, W, generator=g) * 15}]
    cfg = get_cfg()
    cfg.SOLVER.LOSS_TYPE = loss_type
    crit = Criterion({}, cfg)
    store = {"gt": gt.clone(), "valid": valid, **{k: v for k, v in out.items() if k != "aux_outputs"}}
    for i, a in enumerate(out.get("aux_outputs", ())):
        for k, v in a.items():
            store[f"aux{i}_{k}"] = v
    losses = crit(out, {"disp": gt, "valid": valid})
    for k, v in losses.items():
        store[f"loss/{k}"] = v
    return {k: v.numpy() for k, v in store.items()}, loss_type


def main():
    blob = {}
    for i, (lt, aux, hard) in enumerate([("L1", False, False), ("L1", True, False), ("SMOOTH_L1", True, False), ("L1", True, True)]):
        st, loss_type = case(100 + i, lt, aux, hard)
        blob[f"c{i}/loss_type"] = np.array(loss_type)
        for k, v in st.items():
            blob[f"c{i}/{k}"] = v
        print(i, lt, {k[5:]: float(v) for k, v in st.items() if k.startswith("loss/")})
    # one all-invalid target: the zero-loss branches
    out = os.path.join(HERE, "..", "tests", "golden", "criterion.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
