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
