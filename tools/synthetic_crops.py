"""The training data of tools/gen_trained_golden.py (the reference on CPU) and tools/train_synthetic.py (this package on the MI355X):
one seeded stream of random crops, so the two runs see the same batches in the same order."""
import numpy as np
import torch

from nmrf_amd.utils.hashinit import synthetic_pair


class Crops:
    """Random 96x192 crops of synthetic pairs; the ground truth of a crop is the canvas disparity at its pixels (the right view
    of a crop at column x0 is the canvas's right view at the same columns: the disparity is unchanged by cropping both)."""

    def __init__(self, n_canvas=12, ch=160, cw=416, h=96, w=192, seed=7):
        self.pairs = [synthetic_pair(ch, cw, seed=3000 + i) for i in range(n_canvas)]
        self.g = np.random.default_rng(seed)
        self.ch, self.cw, self.h, self.w = ch, cw, h, w

    def batch(self, b):
        l, r, d = [], [], []
        for _ in range(b):
            i = int(self.g.integers(len(self.pairs)))
            y = int(self.g.integers(0, self.ch - self.h + 1))
            x = int(self.g.integers(0, self.cw - self.w + 1))
            L, R, D = self.pairs[i]
            l.append(L[:, y:y + self.h, x:x + self.w])
            r.append(R[:, y:y + self.h, x:x + self.w])
            d.append(D[y:y + self.h, x:x + self.w])
        return torch.stack(l).float(), torch.stack(r).float(), torch.stack(d).float()
