"""CPU ORACLE for the superpixel-guided disparity downsample.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference ships neither the source of this operator (README.md:48 announces it under ops/, the
snapshot has no such file), nor `frame_utils.downsample_disp`, nor a test, nor a producer of `super_pixel_label`.
What is restated here is reconstructed from the only evidence there is -- the call site nmrf/utils/evaluation.py:361-378:

    mini_disp_gt = frame_utils.downsample_disp(disp_gt_clone[None], superpixel_label[None])[0]     # [H/8, W/8, K]
    ... cdist(mini_disp_gt[..., None], proposal[..., None], p=1); epe[mini_disp_gt == 0, :] = 1e6; min over (K, N)

i.e. per 8x8 cell up to K representative ground-truth disparities ("modes"), 0 = empty slot, 0 = invalid input pixel --
and from the operator's name (modes = the superpixel segments intersecting the cell).  The choices the call site does
not determine are fixed here and in the HIP kernel as:
  * a mode = the MEAN of the valid (> 0) disparities of one segment inside the cell (fp32, pixels added in row-major order);
  * modes are ordered by pixel count descending, ties by smaller label id; the first K are kept, the rest dropped;
  * H and W are truncated to multiples of 8 (the call site crops the proposals to `mini_disp_gt.shape[:2]`).
Only tests/ may import this file.
"""
import numpy as np


def downsample_disp(disp, labels, k=4):
    """disp [B,H,W] float32 (0 = invalid), labels [B,H,W] integer -> [B, H//8, W//8, k] float32."""
    disp = np.asarray(disp, dtype=np.float32)
    labels = np.asarray(labels)
    b, h, w = disp.shape
    ht, wd = h // 8, w // 8
    out = np.zeros((b, ht, wd, k), dtype=np.float32)
    for bi in range(b):
        for cy in range(ht):
            for cx in range(wd):
                d = disp[bi, 8 * cy:8 * cy + 8, 8 * cx:8 * cx + 8].reshape(-1)
                l = labels[bi, 8 * cy:8 * cy + 8, 8 * cx:8 * cx + 8].reshape(-1)
                groups = {}
                for p in range(64):                      # row-major, sequential fp32 accumulation
                    if d[p] > 0:
                        s, c = groups.get(int(l[p]), (np.float32(0), 0))
                        groups[int(l[p])] = (np.float32(s + d[p]), c + 1)
                order = sorted(groups.items(), key=lambda kv: (-kv[1][1], kv[0]))
                for r, (_, (s, c)) in enumerate(order[:k]):
                    out[bi, cy, cx, r] = np.float32(s) / np.float32(c)
    return out
