"""Shared helpers for the parity tests: golden loading, hash weights, oracle configs."""
import functools
import os

import numpy as np
import torch

from nmrf_amd.config import get_cfg
from nmrf_amd.models import build_model
from nmrf_amd.utils.hashinit import hash_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def make_cfg(max_disp=320, opts=()):
    cfg = get_cfg()
    cfg.merge_from_list(["DPN.MAX_DISP", max_disp] + list(opts))
    cfg.freeze()
    return cfg


def build_product(max_disp=320, device="cpu", opts=()):
    """The nmrf_amd model with the closed-form hash weights (same fill the goldens were made with)."""
    model = build_model(make_cfg(max_disp, opts))[0].eval()
    sd = model.state_dict()
    new = hash_state_dict(sd)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(v)
    return model.to(device)


@functools.lru_cache(maxsize=None)
def oracle_weights(max_disp=320):
    """Flat weight dict for the oracle: keys/shapes come from the product model's state dict."""
    model = build_model(make_cfg(max_disp))[0]
    return hash_state_dict(model.state_dict())


def oracle_cfg(max_disp=320, **kw):
    from oracle.nmrf_oracle import OracleCfg
    return OracleCfg(max_disp=int(max_disp), **kw)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def report(name, got, want, atol, rtol=0.0):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} beyond tol; max|d|={float(err.max()):.3e} "
                             f"(ref max {float(want.abs().max()):.3e}); first at {idx}: got "
                             f"{float(got[tuple(idx)]):.6g} want {float(want[tuple(idx)]):.6g}")
    return float(err.max())
