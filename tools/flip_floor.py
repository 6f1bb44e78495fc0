"""The fp32 flip floor of the NMRF path: how far is the REFERENCE's own fp32 output from the same algorithm in exact (fp64)
arithmetic, and where do the GPU paths sit relative to that?   ("settle the EPE contract with evidence".)

    python tools/flip_floor.py [--kitti] [--out profiles/r03_flip_floor.md]

For every golden fixture of tests/golden (outputs of the reference itself, tools/gen_golden.py) and optionally one synthetic
KITTI-size pair it evaluates
    fp64      the oracle in double precision from the same images and weights        (the "exact" answer)
    ref32     the reference's fp32 output stored in the fixture
    oracle32  the oracle in fp32 (what the GPU tests compare against)
    gpu-split / gpu-fp32   this build's hot path on the MI355X from the oracle's fp32 encoder features, default split-fp16
              arithmetic and NMRF_LINEAR=fp32 (only when a GPU is present)
and prints, per pair (X vs Y): pixels whose winner-take-all decision differs (NMRF.py:228), the largest reference-side score
margin among them, pixels off by > 0.5 px, raw EPE, median, max, and the EPE of X against the oracle refinement of X's OWN
disp_curr ("same decisions").  Test infrastructure: imports oracle/, never imported by the product."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import nmrf_oracle as O                                                     # noqa: E402
from tests.util import (build_product, disp_stats, golden, golden_images, oracle_cfg,  # noqa: E402
                        oracle_weights, t, unshuffle_heads)


def _pack(out):
    st = out["stages"]
    return dict(score=st["score"], coarse=st["coarse"], disp_curr=st["disp_curr"], disp=out["disp"], proposal=out["proposal"],
                seeds=out["initial_proposal"])


def _gpu(mode, md, o32, out_hw):
    os.environ["NMRF_LINEAR"] = mode
    st32 = o32["stages"]
    model = build_product(md, "cuda")
    fl = [st32["fmap8_l"].cuda(), st32["fmap4_l"].cuda()]
    fr = [st32["fmap8_r"].cuda(), st32["fmap4_r"].cuda()]
    stages = {}
    with torch.no_grad():
        out = model.hot_path(fl, fr, out_hw, stages=stages)
    torch.cuda.synchronize()
    b, _, h8, w8 = fl[0].shape
    n = out["proposal"].shape[-1]
    coarse, score = unshuffle_heads(stages["infer_delta"].cpu(), stages["infer_score"].cpu(), out["proposal"].cpu().reshape(-1, n),
                                    (b, h8, w8, n))
    os.environ.pop("NMRF_LINEAR", None)
    return dict(score=score, coarse=coarse, disp_curr=stages["disp_curr"].cpu(), disp=out["disp"].cpu(),
                proposal=out["proposal"].cpu(), seeds=out["initial_proposal"].cpu())


def compare(x, y, refine_from=None):
    """x vs y (y = the side whose margins are quoted)."""
    sx, sy = x["score"].double(), y["score"].double()
    ix, iy = sx.max(-1).indices, sy.max(-1).indices
    flip = ix != iy
    margin = (sy.gather(-1, iy[..., None]) - sy.gather(-1, ix[..., None]))[..., 0]
    s = disp_stats(x["disp"], y["disp"])
    r = {"seeds_equal": bool((x["seeds"].double() == y["seeds"].double()).all()),
         "proposal_maxdiff": float((x["proposal"].double() - y["proposal"].double()).abs().max()),
         "score_maxdiff": float((sx - sy).abs().max()), "wta_flips": int(flip.sum()), "wta_rate": float(flip.double().mean()),
         "margin_max": float(margin[flip].max()) if flip.any() else 0.0, "px_gt_0p5": int(round(s["frac_gt_0p5"] * x["disp"].numel())),
         "rate_gt_0p5": s["frac_gt_0p5"], "epe": s["epe"], "median": s["median"], "max": s["max"]}
    if refine_from is not None:
        c = disp_stats(x["disp"], refine_from(x["disp_curr"].float()))
        r.update(cond_epe=c["epe"], cond_max=c["max"])
    return r


def run_case(name, img1, img2, md, ref=None, gpu=False):
    w, cfg = oracle_weights(md), oracle_cfg(md)
    w64 = {k: v.double() if v.is_floating_point() else v for k, v in w.items()}
    with torch.no_grad():
        t0 = time.time()
        o32 = O.forward(w, cfg, img1, img2, return_stages=True)
        t1 = time.time()
        o64 = O.forward(w64, cfg, img1.double(), img2.double(), return_stages=True)
        t2 = time.time()
    print("# %s: oracle fp32 %.1f s, fp64 %.1f s" % (name, t1 - t0, t2 - t1), file=sys.stderr)
    st32 = o32["stages"]
    hw = tuple(o32["disp"].shape[-2:])
    rf32 = lambda dq: O.refine_from(w, cfg, dq, st32["fmap4_l"], st32["fmap4_r"], hw)[0]
    sides = {"fp64": _pack(o64), "oracle32": _pack(o32)}
    if ref is not None:                       # the reference's own fp32 output (score / coarse captured only for some fixtures)
        sides["ref32"] = ref
    if gpu:
        sides["gpu-split"] = _gpu("split", md, o32, hw)
        sides["gpu-fp32"] = _gpu("fp32", md, o32, hw)
    rows = []
    with torch.no_grad():
        for a, b in (("ref32", "fp64"), ("oracle32", "fp64"), ("gpu-split", "fp64"), ("gpu-fp32", "fp64"),
                     ("oracle32", "ref32"), ("gpu-split", "ref32"), ("gpu-fp32", "ref32"), ("gpu-split", "oracle32"),
                     ("gpu-fp32", "oracle32"), ("gpu-split", "gpu-fp32")):
            if a in sides and b in sides and "score" in sides[a] and "score" in sides[b]:
                rows.append(dict(compare(sides[a], sides[b], rf32 if a.startswith("gpu") or a == "oracle32" else None),
                                 case=name, x=a, y=b, pixels=int(o32["disp"].numel())))
            elif a in sides and b in sides:   # outputs only
                s = disp_stats(sides[a]["disp"], sides[b]["disp"])
                rows.append(dict(case=name, x=a, y=b, pixels=int(o32["disp"].numel()), px_gt_0p5=int(round(s["frac_gt_0p5"] * o32["disp"].numel())),
                                 rate_gt_0p5=s["frac_gt_0p5"], epe=s["epe"], median=s["median"], max=s["max"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kitti", action="store_true", help="add one synthetic 1242x375 pair (fp64 oracle: ~1 min on 8 cores)")
    ap.add_argument("--cases", default="e2e_a,e2e_b,e2e_c,e2e_d")
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    gpu = torch.cuda.is_available()
    rows = []
    for name in [c for c in a.cases.split(",") if c]:
        g = golden(name)
        i1, i2 = golden_images(g)
        ref = dict(disp=t(g["disp"]))
        if "infer_score" in g:                 # fixtures with stage captures of the reference (forward hooks, gen_golden.py)
            b = i1.shape[0]
            n = g["proposal"].shape[-1]
            h8, w8 = (g["disp_pred"].shape[-2] // 8, g["disp_pred"].shape[-1] // 8)
            coarse, score = unshuffle_heads(t(g["infer_delta"]), t(g["infer_score"]), t(g["proposal"]).reshape(-1, n), (b, h8, w8, n))
            ref.update(score=score, coarse=coarse, disp_curr=t(g["disp_curr"]), proposal=t(g["proposal"]),
                       seeds=t(g["seeds"]).float())
        rows += run_case(name, i1, i2, int(g["max_disp"]), ref, gpu)
    if a.kitti:
        from nmrf_amd.utils.hashinit import synthetic_pair
        l, r, _ = synthetic_pair(375, 1242, seed=1000)
        rows += run_case("kitti_1242x375", l[None], r[None], 320, None, gpu)
    hdr = ("case", "X", "vs Y", "pixels", "WTA decisions differing", "largest Y-margin there", "px off > 0.5", "rate", "raw EPE",
           "median", "max", "EPE of X on its own decisions", "max")
    lines = ["| " + " | ".join(hdr) + " |", "|" + "---|" * len(hdr)]
    for r in rows:
        f = lambda k, fmt: (fmt % r[k]) if k in r else "-"
        lines.append("| %s | %s | %s | %d | %s | %s | %d | %.1e | %.2e | %.1e | %.2f | %s | %s |" % (
            r["case"], r["x"], r["y"], r["pixels"], f("wta_flips", "%d"), f("margin_max", "%.1e"), r["px_gt_0p5"], r["rate_gt_0p5"],
            r["epe"], r["median"], r["max"], f("cond_epe", "%.1e"), f("cond_max", "%.1e")))
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(text + "\n")
        with open(os.path.splitext(a.out)[0] + ".jsonl", "w") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
