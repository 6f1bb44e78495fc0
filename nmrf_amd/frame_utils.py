"""A1: input padding (nmrf/utils/frame_utils.py:259-281).  Pure index bookkeeping + a replicate pad."""
import torch.nn.functional as F


class InputPadder:
    """Pads images so that H and W are divisible by `divis_by`.  mode 'proposal' (the one NMRF.forward
    uses, NMRF.py:204) pads right/bottom only; 'sintel' pads symmetrically; anything else pads width
    symmetrically and height at the bottom."""

    def __init__(self, dims, mode="sintel", divis_by=8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (-self.ht) % divis_by
        pad_wd = (-self.wd) % divis_by
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        elif mode == "proposal":
            self._pad = [0, pad_wd, 0, pad_ht]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        assert all(x.ndim == 4 for x in inputs)
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        assert x.ndim == 4
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]:ht - self._pad[3], self._pad[0]:wd - self._pad[1]]


def downsample_disp(disp, super_pixel_label, num_modes=4):
    """Superpixel-guided disparity downsample with the call signature of the reference's evaluator
    (nmrf/utils/evaluation.py:366: `frame_utils.downsample_disp(disp_gt[None], superpixel_label[None])[0]`).
    PARITY UNPINNED -- the reference does not ship this function or the operator behind it; see include/nmrf_hip.h.
    disp [B,H,W] (0 = invalid), super_pixel_label [B,H,W] integer -> [B, H//8, W//8, num_modes]."""
    import torch
    from . import kernels as K
    return K.superpixel_downsample(disp.float().contiguous(), super_pixel_label.to(torch.int32).contiguous(), num_modes)
