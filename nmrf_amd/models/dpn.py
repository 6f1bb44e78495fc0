"""Disparity proposal network (nmrf/models/DPN.py) on the HIP seed kernels: rows A3-A8 of SURVEY 8(a)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from .nmp import MLP, Propagation, PropagationLayer


class DPN(nn.Module):
    def __init__(self, cost_group, num_proposals, feat_dim, context_dim, num_prop_layers, prop_embed_dim, mlp_ratio,
                 split_size, prop_n_heads, normalize_before=True, **_unused):
        super().__init__()
        # 1-D filter along the disparity axis; parameters only -- evaluated by nmrf_dpn_filter_softmax_f32
        self.mlp = nn.Sequential(nn.Conv1d(cost_group, 8, 5, 1, 2), nn.ReLU(inplace=True),
                                 nn.Conv1d(8, 16, 5, 1, 2), nn.ReLU(inplace=True), nn.Conv1d(16, 1, 5, 1, 2))
        self.eps = 1e-3
        self.num_proposals = num_proposals
        self.cost_group = cost_group
        # visual context (stock convs)
        self.proj = nn.Sequential(nn.Conv2d(feat_dim, 128, 3, 1, 1, bias=False), nn.InstanceNorm2d(128),
                                  nn.ReLU(inplace=True), nn.Conv2d(128, context_dim, 1, 1, 0, bias=False))
        layers = nn.ModuleList(PropagationLayer(prop_embed_dim, mlp_ratio, context_dim, split_size, prop_n_heads,
                                                normalize_before) for _ in range(num_prop_layers))
        self.propagation = Propagation(prop_embed_dim, cost_group, layers, nn.LayerNorm(prop_embed_dim))
        self.prop_head = MLP(prop_embed_dim, prop_embed_dim, 1, 3)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if isinstance(m, nn.Conv1d) and m.bias is not None:           # DPN.py:91-94
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias) if m.bias is not None else None
        nn.init.zeros_(self.prop_head.layers[-1].weight)
        nn.init.zeros_(self.prop_head.layers[-1].bias)

    @classmethod
    def from_config(cls, cfg):
        return cls(cost_group=cfg.DPN.COST_GROUP, num_proposals=cfg.DPN.NUM_PROPOSALS,
                   feat_dim=cfg.BACKBONE.OUT_CHANNELS, context_dim=cfg.DPN.CONTEXT_DIM,
                   num_prop_layers=cfg.NMP.NUM_PROP_LAYERS, prop_embed_dim=cfg.NMP.PROP_EMBED_DIM,
                   mlp_ratio=cfg.NMP.MLP_RATIO, split_size=cfg.NMP.SPLIT_SIZE, prop_n_heads=cfg.NMP.PROP_N_HEADS,
                   normalize_before=cfg.NMP.NORMALIZE_BEFORE)

    # one wave per pixel with the row in registers (+ the seed features in the same launch) up to this many pixels: 22 us against
    # 34 + 10 + 5 us of the three launches it replaces at KITTI batch 1 (7 332 pixels); it is bound by the scalar unit (one row per
    # wave), so from ~50 000 pixels on the LDS form with 16 / 64 rows per wave catches up (109 vs 103 us at KITTI batch 8)
    SELECT_MAX_PIXELS = 1 << 15

    def seeds(self, cost_volume, features=None):
        """cost_volume [P,G,D] -> (prob [P,D], label_seeds [P,K] int64)   (DPN.py:117-125).
        features = (Fourier normalizer, row stride of the encoding): also hand back, in the third slot, what
        Propagation.forward would gather from these seeds (seeds as float, cost taps, Fourier encoding: K.seed_select)."""
        m = self.mlp
        prob = K.dpn_filter_softmax(cost_volume, m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias)
        if features is not None and prob.shape[0] <= self.SELECT_MAX_PIXELS:
            seeds, seeds_f, cost, enc = K.seed_select(prob, cost_volume, self.num_proposals, self.eps, *features)
            return prob, seeds, (seeds_f, cost, enc)
        seeds = K.nms_topk(prob, self.num_proposals, self.eps)
        return (prob, seeds) if features is None else (prob, seeds, None)

    def context(self, fmap, token_major=False):
        """proj (Conv3x3 - IN - ReLU - Conv1x1, DPN.py:45-49) of the 1/8-resolution left feature map -> [B,Cctx,H,W];
        token_major: [B,H,W,Cctx], the rows the propagation consumes (DPN.py:120), written as such by the 1x1 kernel."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.proj.parameters()):
            raise NotImplementedError("nmrf_amd implements the inference path only: call under torch.no_grad()")
        if not hasattr(self, "_c3"):
            self._c3 = {}
        raw = K.conv3x3_auto(fmap, self.proj[0].weight, self._c3).contiguous()
        w1 = self.proj[3].weight
        if w1.shape[1] in (64, 128) and w1.shape[0] % 64 == 0:
            from .nmp import _FusedCache                    # IN + ReLU folded into the 1x1 conv's operand load (csrc/conv1x1.hip)
            if not hasattr(self, "_c1"):
                self._c1 = _FusedCache()
            return K.conv1x1_in_relu(raw, 0, w1.shape[1], K.instance_stats(raw), self._c1.get((w1,), lambda: K.pack_conv1x1(w1)),
                                     token_major=token_major)
        out = self.proj[3](K.instance_norm(raw, relu=True))     # conv3x3 - IN - ReLU fused
        return out.permute(0, 2, 3, 1).contiguous() if token_major else out

    def forward(self, cost_volume, fmap1_list, context=None, context_ready=None):
        """cost_volume: [B,G,D,H,W] (reference layout) or token-major [B*H*W,G,D].  context: [B,H,W,Cctx] rows
        (self.context(fmap, token_major=True)) precomputed by the
        caller (possibly on another stream: context_ready = the event recorded behind it, waited for where it is first used).
        Returns (cost_volume [P,G,D], prob [P,D], label_seeds [P,N] float, labels [1,P,N])."""
        if cost_volume.dim() == 5:
            b, g, d, h, w = cost_volume.shape
            cost_volume = cost_volume.permute(0, 3, 4, 1, 2).reshape(b * h * w, g, d).contiguous()
        prob, seeds, feats = self.seeds(cost_volume, features=self.propagation.seed_feature_args(context))
        if context is None:
            context = self.context(fmap1_list[0], token_major=True)
        elif context_ready is not None:
            torch.cuda.current_stream().wait_event(context_ready)
        memory, seeds_f = self.propagation(cost_volume, seeds, context, feats=feats)
        # labels = relu(prop_head(memory) + seeds) (DPN.py:131-132): the add and the ReLU leave with the head's rows
        labels = self.prop_head(memory, row_add=seeds_f.reshape(-1, 1), relu_out=True).view(-1, *seeds_f.shape)
        return cost_volume, prob, seeds_f, labels
