"""Training criterion of NMRF-Stereo: the loss terms of nmrf/models/NMRF.py:276-429 on the output dictionary of
`NMRF.forward`, restated in this build's own formulation.  Plain PyTorch (the losses are a few elementwise passes over the
outputs, nothing for a hand-written kernel); works on CPU and on the MI355X, differentiable through torch autograd.

In eval mode `NMRF.forward` produces the eval-mode dictionary (no `aux_outputs`) and this module gives a caller of the reference's drivers
the loss / EPE logging of an evaluation pass; in training mode the dictionary carries `aux_outputs` (NMRF.py:259-273) and, with
`model.enable_grad_slice()`, an autograd graph: `sum_k weight_dict[k] * loss_dict[k]` is then the loss `nmrf_amd.train.train_step`
back-propagates (main.py:413-420).  It is also the criterion object `build_model(cfg)` returns (nmrf/models/__init__.py:9-10).

Terms (names and reductions as the reference's training loop reads them, main.py:413-420):
  loss_prop   smooth-L1 between every valid ground-truth pixel of an 8x8 cell and the NEAREST of the cell's N proposals (x8 px),
              summed, divided by the number of valid pixels (NMRF.py:301-320)
  init        cross-entropy between the initial cost-volume distribution `prob` [B*h*w, D] and a soft target histogram per cell:
              every valid pixel spreads unit mass linearly over the two bins around gt/8 (bins clamped to D-1), the histogram is
              normalised, -sum(label * log clamp(prob, 1e-6)) / number of cells with a valid pixel (NMRF.py:322-365)
  loss_disp   L1 (or smooth-L1) of the refined disparity `disp_pred * 4` over valid pixels (NMRF.py:378-384)
  loss_coarse_disp_i / loss_disp_i   the same on intermediate predictions, score-weighted for the coarse ones (NMRF.py:367-376)
  epe_train   mean |disp - gt| over valid pixels (logging)
A pixel is valid for the proposal / init terms if 0 < gt < 320 (the network's own range, hard-coded in the reference), for the
disparity terms if 0 < gt < SOLVER.MAX_DISP.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _cells(x, c=8):
    """[B, c*h, c*w] -> [B, h*w, c*c]: the pixels of every c x c cell, row-major inside the cell."""
    b, hh, ww = x.shape
    return x.reshape(b, hh // c, c, ww // c, c).permute(0, 1, 3, 2, 4).reshape(b, (hh // c) * (ww // c), c * c)


class Criterion(nn.Module):
    def __init__(self, weight_dict, cfg):
        super().__init__()
        if cfg.SOLVER.LOSS_TYPE not in ("L1", "SMOOTH_L1"):
            raise AssertionError(f"unrecognized loss type {cfg.SOLVER.LOSS_TYPE}")
        self.weight_dict = weight_dict
        self.max_disp = cfg.SOLVER.MAX_DISP
        self.loss_fn = F.smooth_l1_loss if cfg.SOLVER.LOSS_TYPE == "SMOOTH_L1" else F.l1_loss

    # ---- proposals ---------------------------------------------------------------------------------------------------------
    def loss_prop(self, disp_prop, gt_disp):
        """disp_prop [B, h*w, N] in pixels; gt_disp [B, 8h, 8w]."""
        tgt = _cells(torch.where(gt_disp >= 320, torch.zeros_like(gt_disp), gt_disp))           # [B, hw, 64]
        nearest = (tgt.unsqueeze(-1) - disp_prop.unsqueeze(-2)).abs().argmin(-1)                # first minimum, like torch.min
        matched = disp_prop.gather(-1, nearest)
        valid = (tgt > 0) & (tgt < self.max_disp)
        # (masked sums instead of the reference's boolean indexing -- the same terms, no nonzero / index_put launches and no read-back)
        total = (F.smooth_l1_loss(matched, tgt, reduction="none") * valid).sum()
        return {"loss_prop": total / (valid.sum() + 1e-6)}

    # ---- initial distribution ------------------------------------------------------------------------------------------------
    @staticmethod
    def loss_init(prob, gt_disp):
        """prob [B*h*w, D]; gt_disp [B, 8h, 8w]."""
        nd = prob.shape[-1]
        gt = gt_disp.clamp(min=0)
        xs = torch.arange(gt.shape[-1], device=prob.device).view(1, 1, -1)
        valid = (gt > 0) & (gt < 320) & (xs - gt >= 0)                   # the match must lie inside the right view
        t = _cells(gt / 8).reshape(-1, 64)                                # target in bins
        wgt = _cells(valid.to(gt.dtype)).reshape(-1, 64)
        lo = t.floor()
        frac = t - lo
        lo = lo.long()
        label = torch.zeros_like(prob)
        label.scatter_add_(-1, lo.clamp(max=nd - 1), (1 - frac) * wgt)
        label.scatter_add_(-1, (lo + 1).clamp(max=nd - 1), frac * wgt)
        label = label / label.sum(-1, keepdim=True).clamp(min=1e-3)
        nll = -(prob.clamp(min=1e-6).log() * label).sum()                # (entries with label 0 add nothing: the reference indexes label > 0)
        cells = (wgt.sum(-1) > 0).sum()
        loss = nll / (cells + 1e-6)
        assert not torch.isnan(loss).any()
        return {"init": loss}

    # ---- disparities --------------------------------------------------------------------------------------------------------------
    def _valid(self, disp_gt, valid=None):
        """(mask, number of valid pixels as a tensor, any valid pixel at all) -- forward() computes them once per target and hands them to
        every term (the reference re-derives them, with one `valid.any()` read-back, per auxiliary output)."""
        if valid is not None:
            return valid
        m = (disp_gt > 0) & (disp_gt < self.max_disp)
        return m, m.sum(), bool(m.any())

    def loss_coarse(self, disp_pred, logits_pred, disp_gt, valid=None):
        m, cnt, has = self._valid(disp_gt, valid)
        if not has:                             # keeps the graph alive with a zero loss (NMRF.py:374-375)
            return {"loss_coarse_disp": F.smooth_l1_loss(disp_pred, disp_pred.detach()) + F.smooth_l1_loss(logits_pred, logits_pred.detach())}
        err = self.loss_fn(disp_pred, disp_gt.unsqueeze(-1).expand_as(disp_pred), reduction="none")
        return {"loss_coarse_disp": ((F.softmax(logits_pred, -1) * err).sum(-1) * m).sum() / cnt}      # the mean over the valid pixels

    def loss_disp(self, disp_pred, disp_gt, valid=None):
        m, cnt, has = self._valid(disp_gt, valid)
        if not has:
            return {"loss_disp": F.smooth_l1_loss(disp_pred, disp_pred.detach())}
        return {"loss_disp": (self.loss_fn(disp_pred, disp_gt, reduction="none") * m).sum() / cnt}

    def forward(self, outputs, targets, log=True):
        """outputs: the dictionary of NMRF.forward; targets: {'disp' [B,H,W], 'valid' bool [B,H,W]} -> dict of scalar losses."""
        disp = outputs["disp"]
        gt = targets["disp"].to(disp.device)
        gt[~targets["valid"].to(disp.device)] = 0            # (in place, as the reference does: callers see the masked target)
        losses = self.loss_prop(outputs["proposal"] * 8, gt)
        losses.update(self.loss_init(outputs["prob"], gt))
        vm = self._valid(gt)
        if "disp_pred" in outputs:
            losses.update(self.loss_disp(outputs["disp_pred"] * 4, gt, vm))
        if log:
            losses["epe_train"] = ((disp - gt).abs() * vm[0]).sum() / vm[1]
        for i, aux in enumerate(outputs.get("aux_outputs", ())):
            if "logits_pred" in aux:
                part = self.loss_coarse(aux["disp_pred"] * 8, aux["logits_pred"], gt, vm)
            else:
                part = self.loss_disp(aux["disp_pred"] * 4, gt, vm)
            losses.update({f"{k}_{i}": v for k, v in part.items()})
        return losses


def build_criterion(cfg):
    """The weight dictionary of nmrf/models/NMRF.py:432-447."""
    n_inf, n_ref = cfg.NMP.NUM_INFER_LAYERS, cfg.NMP.NUM_REFINE_LAYERS
    lw = cfg.SOLVER.LOSS_WEIGHTS
    assert len(lw) == n_inf + n_ref
    weights = {"proposal_disp": 1, "init": 1}
    if cfg.SOLVER.AUX_LOSS:
        for i in range(n_inf + n_ref - 1):
            weights[(f"loss_coarse_disp_{i}" if i < n_inf else f"loss_disp_{i}")] = lw[i]
    weights["loss_disp"] = lw[-1]
    return Criterion(weights, cfg)
