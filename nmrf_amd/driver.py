"""Batched inference driver: the counterpart of the reference's demo / submission loop
(inference.py:61-126) and of the timing convention of nmrf/utils/evaluation.py:203-267, without its batch-1
synchronous structure: pairs are grouped into batches, staged in pinned host memory, copied H2D on a side stream
while the previous batch computes, and the finished disparities come back D2H asynchronously.

    python -m nmrf_amd.driver --left 'L/*.png' --right 'R/*.png' --output out/ [--ckpt kitti.pth] [--batch 8]

Disparity maps are written as KITTI 16-bit PNGs (uint16 = round(disp * 256), nmrf/utils/frame_utils.py:237-239)
or .npy.  Works with any callable `model(sample) -> {'disp': [B,H,W]}`; `model=None` builds nmrf_amd's NMRF.
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch


def encode_kitti_disp(disp):
    """float disparity [H,W] -> uint16 array, KITTI convention (0 = invalid is never produced here)."""
    return np.clip(np.round(np.asarray(disp, dtype=np.float64) * 256.0), 0, 65535).astype(np.uint16)


def load_rgb(path):
    """-> float32 [3,H,W] in 0..255 (the reference feeds raw 0..255 RGB, nmrf/data/datasets.py:54-62)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)
    return torch.from_numpy(img).permute(2, 0, 1).contiguous()


def batches(items, size):
    """Group consecutive items of equal image size into batches of at most `size` (order preserved)."""
    cur, shape = [], None
    for it in items:
        s = tuple(it[1].shape)
        if cur and (s != shape or len(cur) == size):
            yield cur
            cur = []
        cur.append(it)
        shape = s
    if cur:
        yield cur


class StereoStream:
    """Double-buffered pipeline: H2D of batch i+1 and D2H of batch i-1 overlap the compute of batch i."""

    def __init__(self, model, device="cuda", batch=8):
        self.model, self.device, self.batch = model, torch.device(device), batch
        self.on_gpu = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.on_gpu else None

    def _stage(self, group):
        left = torch.stack([g[1] for g in group])
        right = torch.stack([g[2] for g in group])
        if not self.on_gpu:
            return left, right, None
        left, right = left.pin_memory(), right.pin_memory()
        with torch.cuda.stream(self.copy_stream):
            dl = left.to(self.device, non_blocking=True)
            dr = right.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return dl, dr, ev

    def run(self, pairs):
        """pairs: iterable of (key, left [3,H,W], right [3,H,W]) -> yields (key, disparity [H,W] CPU tensor)."""
        pending = None                       # (keys, host tensor, event) of the previous batch's D2H
        staged = None
        it = batches(pairs, self.batch)
        group = next(it, None)
        if group is not None:
            staged = (group, self._stage(group))
        while staged is not None:
            group, (dl, dr, ev) = staged
            nxt = next(it, None)
            staged = (nxt, self._stage(nxt)) if nxt is not None else None      # H2D of the next batch starts now
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            with torch.no_grad():
                disp = self.model({"img1": dl, "img2": dr})["disp"]
            if self.on_gpu:
                dl.record_stream(torch.cuda.current_stream())
                dr.record_stream(torch.cuda.current_stream())
                host = torch.empty(disp.shape, dtype=disp.dtype, pin_memory=True)
                host.copy_(disp, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
            else:
                host, done = disp, None
            if pending is not None:
                yield from self._drain(pending)
            pending = ([g[0] for g in group], host, done)
        if pending is not None:
            yield from self._drain(pending)

    @staticmethod
    def _drain(pending):
        keys, host, done = pending
        if done is not None:
            done.synchronize()
        for k, d in zip(keys, host):
            yield k, d


def build_default_model(ckpt=None, opts=(), device="cuda"):
    from .config import get_cfg
    from .models import build_model
    cfg = get_cfg()
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    model = build_model(cfg)[0].eval()
    if ckpt:
        sd = torch.load(ckpt, map_location="cpu")
        model.load_state_dict(sd.get("model", sd), strict=cfg.SOLVER.STRICT_RESUME)
    return model.to(device)


def main(argv=None):
    ap = argparse.ArgumentParser(description="NMRF-Stereo batched inference on MI355X")
    ap.add_argument("--left", required=True, help="glob of left images")
    ap.add_argument("--right", required=True, help="glob of right images")
    ap.add_argument("--output", required=True)
    ap.add_argument("--ckpt", default=None, help="reference checkpoint (.pth, {'model': state_dict} or bare)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--format", choices=("kitti16", "npy"), default="kitti16")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    args = ap.parse_args(argv)
    lefts, rights = sorted(glob.glob(args.left)), sorted(glob.glob(args.right))
    assert lefts and len(lefts) == len(rights), "need as many left as right images"
    os.makedirs(args.output, exist_ok=True)
    model = build_default_model(args.ckpt, args.opts)
    pairs = ((os.path.splitext(os.path.basename(l))[0], load_rgb(l), load_rgb(r)) for l, r in zip(lefts, rights))
    t0, n = time.perf_counter(), 0
    for key, disp in StereoStream(model, batch=args.batch).run(pairs):
        if args.format == "npy":
            np.save(os.path.join(args.output, key + ".npy"), disp.numpy())
        else:
            from PIL import Image
            Image.fromarray(encode_kitti_disp(disp.numpy())).save(os.path.join(args.output, key + ".png"))
        n += 1
    dt = time.perf_counter() - t0
    print("%d pairs in %.2f s (%.1f pairs/s incl. image I/O)" % (n, dt, n / dt), file=sys.stderr)


if __name__ == "__main__":
    main()
