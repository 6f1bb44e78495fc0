"""NMRF top module: same constructor arguments, forward API and state-dict names as
nmrf/models/NMRF.py:21-262, with the hot path on libnmrf_hip.so.  model.eval() is the product path: fused forward-only kernels.
model.train() runs the reference's TRAINING-mode forward (no input padding, per-layer intermediates, `aux_outputs`,
NMRF.py:203-205, 216-223, 259-273) on the same kernels, so that the `Criterion` (models/criterion.py) can be evaluated on it; with
enable_grad_slice() its outputs hang on an autograd graph whose backward is csrc/backward.hip (models/autograd_ops.py,
nmrf_amd/train.py) -- the message-passing stages and heads, or with full=True every parameter of either configuration.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from ..frame_utils import InputPadder
from .backbone import create_backbone
from .dpn import DPN
from .nmp import MLP, Inference, InferenceLayer, Refinement, RefinementLayer, _FusedCache


def _conv_head(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, 128, 3, 1, 1, bias=False), nn.InstanceNorm2d(128), nn.ReLU(inplace=True),
                         nn.Conv2d(128, cout, 1, 1, 0, bias=False))


class NMRF(nn.Module):
    def __init__(self, backbone, dpn=None, num_proposals=4, max_disp=320, num_infer_layers=5, num_refine_layers=5,
                 infer_embed_dim=128, infer_n_heads=4, mlp_ratio=4, window_size=6, refine_window_size=4,
                 with_refinement=True, attn_drop=0., proj_drop=0., drop_path=0., dropout=0.,
                 return_intermediate=False, normalize_before=True, activation="gelu", aux_loss=False,
                 divis_by=8, compat=True):
        if dpn is None and hasattr(backbone, "BACKBONE"):       # called as NMRF(cfg)
            kwargs = self.from_config(backbone)
            backbone = kwargs.pop("backbone")
            return self.__init__(backbone, **kwargs)
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("the fused kernels implement GELU (every shipped config; relu / glu: NMP.py:984-992)")
        # Dropout / stochastic depth (default.py:56-59, NMP.py:198, 343-349): identities in eval mode, so a model configured with
        # them evaluates exactly like the reference; only a TRAINING-mode forward would have to draw masks, and raises.
        self.drop_rates = {"attn_drop": float(attn_drop), "proj_drop": float(proj_drop), "drop_path": float(drop_path),
                           "dropout": float(dropout)}
        if not with_refinement:
            raise NotImplementedError("refinement is always on in the reference (NMRF.py:131-152)")
        self.num_proposals, self.max_disp, self.divis_by = num_proposals, max_disp, divis_by
        self.aux_loss = aux_loss
        feat_dim = backbone.output_dim
        self.concatconv = _conv_head(feat_dim, 64)
        self.gw = _conv_head(feat_dim, 256)
        infer_layers = nn.ModuleList(
            InferenceLayer(infer_embed_dim, mlp_ratio, window_size, 0 if i % 2 == 0 else window_size // 2,
                           infer_n_heads, normalize_before) for i in range(num_infer_layers))
        self.inference = Inference(32, infer_embed_dim, infer_layers, nn.LayerNorm(infer_embed_dim), return_intermediate)
        self.infer_head = MLP(infer_embed_dim, infer_embed_dim, 8 * 8, 3)
        self.infer_score_head = nn.Linear(infer_embed_dim, 8 * 8)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
        self.with_refinement = True
        refine_layers = nn.ModuleList(
            RefinementLayer(infer_embed_dim, mlp_ratio, refine_window_size,
                            0 if i % 2 == 0 else refine_window_size // 2, infer_n_heads, normalize_before)
            for i in range(num_refine_layers))
        self.refinement = Refinement(32, infer_embed_dim, refine_layers, nn.LayerNorm(infer_embed_dim), return_intermediate)
        self.refine_head = MLP(infer_embed_dim, infer_embed_dim, 4 * 4, 3)
        self.dpn = dpn
        self.compat = compat
        if compat:
            self.backbone = backbone
        else:
            self.image_encoder = backbone
        self.register_buffer("device_indicator_tensor", torch.empty(0))
        self._head_cache8, self._head_cache4 = _FusedCache(), _FusedCache()
        self._side_stream = None
        # fp16 range guard of the split-operand kernels (include/nmrf_hip.h): True = forward() ends with check_range() -- one
        # 4-byte read-back and a device sync, like the .cpu() every caller of the reference does next (inference.py:74) -- so that
        # an out-of-range activation raises instead of returning inf / NaN.  False: the caller checks itself where it synchronises
        # anyway (nmrf_amd.driver, bench.py); never checked inside a hipGraph capture.
        self.range_check = True

    @classmethod
    def from_config(cls, cfg):
        return dict(backbone=create_backbone(cfg), dpn=DPN.from_config(cfg), num_proposals=cfg.DPN.NUM_PROPOSALS,
                    max_disp=cfg.DPN.MAX_DISP, aux_loss=cfg.SOLVER.AUX_LOSS, num_infer_layers=cfg.NMP.NUM_INFER_LAYERS,
                    num_refine_layers=cfg.NMP.NUM_REFINE_LAYERS, infer_embed_dim=cfg.NMP.INFER_EMBED_DIM,
                    infer_n_heads=cfg.NMP.INFER_N_HEADS, mlp_ratio=cfg.NMP.MLP_RATIO, window_size=cfg.NMP.WINDOW_SIZE,
                    refine_window_size=cfg.NMP.REFINE_WINDOW_SIZE, attn_drop=cfg.NMP.ATTN_DROP,
                    proj_drop=cfg.NMP.PROJ_DROP, drop_path=cfg.NMP.DROP_PATH, dropout=cfg.NMP.DROPOUT,
                    normalize_before=cfg.NMP.NORMALIZE_BEFORE, return_intermediate=cfg.NMP.RETURN_INTERMEDIATE,
                    divis_by=cfg.DATASETS.DIVIS_BY, compat=cfg.BACKBONE.COMPAT)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    @property
    def device(self):
        return self.device_indicator_tensor.device

    def extract_feature(self, img1, img2):
        """-> (fmap1_list, fmap2_list), each [1/8-res, 1/4-res] (low to high), NMRF.py:172-187"""
        enc = self.backbone if self.compat else self.image_encoder
        feats = enc(torch.cat((img1, img2), 0))[::-1]
        b = img1.shape[0]
        self._joint_feats = feats          # the un-split [2B,C,H,W] maps: hot_path re-uses them instead of re-concatenating the views
        return [f[:b].contiguous() for f in feats], [f[b:].contiguous() for f in feats]

    @torch.no_grad()
    def forward(self, sample):
        """model(sample) of NMRF.py:189-262.  The fused HIP kernels are forward-only, so the body runs under no_grad; in training mode
        (model.train()) it is the reference's training-mode forward -- no input padding (NMRF.py:203-205: H and W must be multiples of
        divis_by), no un-padding, per-layer intermediates when return_intermediate, `aux_outputs` when aux_loss (NMRF.py:259-273).
        Without enable_grad_slice() the returned tensors carry no grad_fn (losses can be EVALUATED; a loss.backward() fails loudly); with
        it the outputs hang on an autograd graph of torch.autograd.Functions whose backward is csrc/backward.hip (models/autograd_ops.py)
        -- the message-passing stages and heads, or with full=True every parameter (nmrf_amd/train.py)."""
        enc = self.backbone if self.compat else self.image_encoder
        from .backbone import Backbone
        if self.training and any(self.drop_rates.values()):
            raise NotImplementedError("training-mode forward with non-zero dropout / drop-path rates %r: the fused kernels draw no "
                                      "masks (eval mode is exact: the rates are identities there)" % (self.drop_rates,))
        if self.training and not getattr(self, "grad_slice", False) and not getattr(self, "_warned_train", False):
            import warnings
            warnings.warn("nmrf_amd: the model is in TRAINING mode (nn.Module's default after build_model -- call model.eval() for "
                          "inference): this runs the training-mode FORWARD only -- no padding / un-padding, aux_outputs returned, forward-only "
                          "HIP kernels, no autograd graph (enable_grad_slice() builds one)")
            self._warned_train = True
        if self.device.type != "cuda":
            raise RuntimeError("the NMRF hot path runs on an MI355X through libnmrf_hip.so; there is no CPU "
                               "fallback (move the model with .to('cuda'))")
        image1 = sample["img1"].to(self.device)
        image2 = sample["img2"].to(self.device)
        h0, w0 = image1.shape[-2:]
        if image1.dtype != image2.dtype:
            image1, image2 = image1.float(), image2.float()
        if isinstance(enc, Backbone) and image1.dtype in (torch.float32, torch.uint8) and image1.shape == image2.shape:
            # (uint8 images -- decoded PNGs as the batched driver ships them over PCIe -- are converted inside the staging kernel)
            # pad (A1) + stack + normalise in one HIP pass, straight into the encoder
            b = image1.shape[0]
            hp, wp = h0 + (-h0) % self.divis_by, w0 + (-w0) % self.divis_by
            if self.training and (hp, wp) != (h0, w0):
                raise ValueError("the model is in TRAINING mode, which does not pad its input (NMRF.py:203-205): %dx%d is not a multiple of "
                                 "%d -- did you forget model.eval()?  (build_model returns the module in nn.Module's default training "
                                 "state, as the reference's does; inference.py:150 calls .eval())" % (h0, w0, self.divis_by))
            stem = enc.conv1
            if self._grad_full():
                # N4, the whole model: the encoder on stock PyTorch-ROCm autograd (its stock branch: models/backbone.py:_hip_ok), fed by
                # the same staging kernel; the hot path below attaches its own Functions to these maps
                x = K.prep_images(image1.contiguous(), image2.contiguous(), hp, wp)
                with torch.enable_grad():
                    feats = enc(x, normalized=True)[::-1]
                    fmap1_list, fmap2_list = [f[:b] for f in feats], [f[b:] for f in feats]
                self._joint_feats = feats
            elif (enc.fused and image1.shape[1] == 3 and hp % 2 == 0 and wp % 2 == 0 and stem.weight.shape[0] % 64 == 0
                    and tuple(stem.weight.shape[1:]) == (3, 7, 7) and stem.stride == (2, 2) and stem.padding == (3, 3)):
                # the stem (7x7 / stride 2) runs as a 4x4 convolution over the 2x2 space-to-depth image: staged in that layout
                feats = enc(K.prep_images_s2d(image1.contiguous(), image2.contiguous(), hp, wp), normalized="s2d")[::-1]
            else:
                feats = enc(K.prep_images(image1.contiguous(), image2.contiguous(), hp, wp), normalized=True)[::-1]
            if not self._grad_full():
                self._joint_feats = feats
                fmap1_list, fmap2_list = [f[:b] for f in feats], [f[b:] for f in feats]
        else:
            image1, image2 = image1.float(), image2.float()
            if not self.training:
                padder = InputPadder(image1.shape, mode="proposal", divis_by=self.divis_by)
                image1, image2 = padder.pad(image1, image2)
            if self._grad_full():                                   # (the Swin-T trunk + neck: stock autograd, MSDA through its Function)
                with torch.enable_grad():
                    fmap1_list, fmap2_list = self.extract_feature(image1, image2)
            else:
                fmap1_list, fmap2_list = self.extract_feature(image1, image2)
        try:
            out = self.hot_path(fmap1_list, fmap2_list, (h0, w0))
        finally:
            self._joint_feats = None
        if self.range_check and not torch.cuda.is_current_stream_capturing():
            K.check_range(self.device)
        return out

    def _grad_full(self):
        return bool(self.training and getattr(self, "grad_slice", False) and getattr(self, "grad_full", False))

    def enable_grad_slice(self, on=True, full=False):
        """N4 (round 5): in training mode, build an autograd graph over the three message-passing stages and their heads -- the WHOLE
        propagation, inference and refinement stages (seed embedding / ffn, every layer's norm1 / q | k | v / stripe, sibling or window
        attention with its LePE kernels or relative-position table / proj / norm2 / MLP, the stage-final norms) and `prop_head`,
        `infer_head`, `infer_score_head`, `refine_head`: 315 of the 340 parameter tensors (models/autograd_ops.py: every Function's forward value is the
        fused HIP launch's, backward = csrc/backward.hip).  The reference detaches `labels_curr` and `disp_curr` (NMRF.py:215,231), so
        `Criterion(model(sample)).backward()` leaves the REFERENCE's own gradients in `.grad` of the 206 tensors behind the hand-over; the
        propagation stage is reached by the proposal loss only, which carries no weight in the reference's weight_dict (add
        `weight_dict['loss_prop']` to train it).  Encoder, matching heads and DPN context convolutions stay forward-only:
        `.grad is None`, loudly.  Off by default; eval mode ignores it.
        full=True: the WHOLE model.  The convolutional modules -- encoder, `concatconv`, `gw`, `dpn.proj` -- run on stock PyTorch-ROCm
        autograd in the training-mode forward (north_star keeps them on stock ROCm; the fused forward-only conv kernels serve eval mode),
        and the three kernels that read their maps get a backward (autograd_ops.CostVolumeFn / SeedTapsFn / WarpCorrFn; the seed filter and
        the propagation's q | k return the gradient of the cost volume and of the context rows): every parameter's `.grad` is the
        reference's."""
        self.grad_slice = bool(on)
        self.grad_full = bool(on and full)
        self.inference.keep_pre_norm = self.refinement.keep_pre_norm = self.dpn.propagation.keep_pre_norm = bool(on)
        return self

    def enable_training(self, convolutions=True):
        """model.train() + the autograd graph of enable_grad_slice: every parameter (convolutions=True, the reference's training), or the
        message-passing stages, heads and seed filter with the convolutional side frozen on its fused kernels (convolutions=False)."""
        return self.train().enable_grad_slice(True, full=convolutions)

    @staticmethod
    def _stage_rows_with_grad(stage, maps=None, labels=None):
        """The residual stream of EVERY layer of an NMP stage (inference: self-edge + window sites, four labels per pixel; refinement:
        window sites, one label) on the dense grid, as an autograd graph over ALL of the stage's parameters:
            ffn -> per site [norm1 | enc -> q | k | v -> attention (siblings / windows with the relative-position table) -> proj + residual
                             (+ norm2 + MLP at window sites)].
        Every Function's forward value is the tensor the fused forward already produced (the tape of Inference._run; the sibling
        attention, evaluated inside the block kernel there, is re-run by its own kernel); backward = csrc/backward.hip.  q | k | v rows
        that the forward wrote as split fp16 pairs for the window kernel (kv16) are decoded to the fp32 values those kernels multiply.
        maps = (fmap1, fmap2, fmap1_gw, fmap2_gw) NCHW with a graph + the (constant) labels: the stage's input rows become a function of
        them (WarpCorrFn).  None when the forward did not record a tape."""
        from .autograd_ops import BlockFn, FfnFn, ProjFn, QkvFn, SelfAttnFn, WarpCorrFn, WindowAttnFn
        tape = getattr(stage, "_tape", None)
        if tape is None or len(tape["qkv"]) != len(stage._sites) or len(tape["x"]) != len(stage._sites) + 1:
            return None
        b, hp, wp, n = tape["pdims"]
        to_p = tape["to_p"]
        keep = None if to_p is None else to_p.long()
        x0, enc = tape["x"][0], tape["enc"]
        ffn = stage.ffn
        wcc = tape["wcc"]
        if maps is not None:
            wcc = WarpCorrFn.apply(*maps, labels.reshape(-1).contiguous(), n, stage.cost_group, lambda v=wcc: v)
        xd = FfnFn.apply(wcc, ffn.fc1.weight, ffn.fc1.bias, ffn.fc2.weight, ffn.fc2.bias,
                         lambda t: x0 if keep is None else x0.index_select(0, keep))
        xg = xd if keep is None else torch.zeros_like(x0).index_copy(0, keep, xd)          # (the zero-padded grid, NMP.py:745-762)
        rows = []
        for i, (kind, m) in enumerate(stage._sites):
            qkv_i, x_next = tape["qkv"][i], tape["x"][i + 1]
            if kind == "win" and stage._site_kv16[i]:
                qkv_i = K.from_kv16(qkv_i)
            wb = [t for l in ((m.q, m.k, m.v) if hasattr(m, "q") else (m.qkv,)) for t in (l.weight, l.bias)]
            qkv = QkvFn.apply(xg, enc, m.norm1.weight, m.norm1.bias, m.norm1.eps, lambda v=qkv_i: v, *wb)
            if kind == "self":
                msg = SelfAttnFn.apply(qkv, n, m.num_heads)
                xg = ProjFn.apply(xg, msg, m.proj.weight, m.proj.bias, lambda v=x_next: v)
                continue
            geom = (b, hp, wp, n, m.attn.num_heads, m.attn.window_size[0], m.attn.shift_size, n > 1)
            msg = WindowAttnFn.apply(qkv, m.attn.relative_position_enc_table, geom, lambda v=tape["msg"][i]: v)
            xg = BlockFn.apply(xg, msg, m.proj.weight, m.proj.bias, m.norm2.weight, m.norm2.bias, m.mlp.fc1.weight, m.mlp.fc1.bias,
                               m.mlp.fc2.weight, m.mlp.fc2.bias, m.norm2.eps, lambda v=x_next: v)
            rows.append(xg if keep is None else xg.index_select(0, keep))
        return rows

    @staticmethod
    def _propagation_rows_with_grad(prop, cv=None, seeds=None, context=None):
        """The propagation stage's output rows (before its final norm) as an autograd graph over ALL of its parameters: the seed embedding
        (cost_encoder on the 9 x 4 cost taps, proj on [features | Fourier]) and five layers of [norm1 | context -> q, k, v -> cross-stripe
        attention with LePE -> proj + residual + norm2 + MLP].  Forward values from the tape of Propagation.forward (kv16 rows decoded),
        backward = csrc/backward.hip.  The label seeds (integer NMS output), the cost taps and the context rows are constants: their
        producers -- seed stage, DPN context convs -- are forward-only -- unless handed in with a graph: cv [P,G,D] (+ the int64 seeds) makes
        the cost taps a function of the volume (SeedTapsFn), context [B,Cctx,H,W] the q | k operand rows one of DPN.proj.  None without a tape."""
        from .autograd_ops import BlockFn, FfnFn, LinearFn, QkvFn, SeedTapsFn, StripeAttnFn
        tape = getattr(prop, "_tape", None)
        if tape is None or "x" not in tape or len(tape["qkv"]) != len(prop.layers):
            return None
        b, h, wd, n = tape["dims"]
        ce0, ce2 = prop.cost_encoder[0], prop.cost_encoder[2]
        cost = tape["cost"][:, : ce0.in_features].contiguous()
        if cv is not None:
            cost = SeedTapsFn.apply(cv, seeds, lambda v=cost: v)

        def encoder(t):                                               # (the fused seed-embedding chain does not materialise this layer)
            _, hid = K.bias_act(K.linear_forward(t, ce0.weight), ce0.bias, 2)
            return K.bias_act(K.linear_forward(hid, ce2.weight), ce2.bias, 0)[1]
        feat = FfnFn.apply(cost, ce0.weight, ce0.bias, ce2.weight, ce2.bias, encoder)
        cat = torch.cat((feat, tape["enc"][:, : prop.proj.in_features - feat.shape[1]]), 1)
        x = LinearFn.apply(cat, prop.proj.weight, None, lambda t, v=tape["x"][0]: v)
        ctx_rows = tape["ctx"] if context is None else context.permute(0, 2, 3, 1).reshape(tape["ctx"].shape)
        ctx_tok = ctx_rows.repeat_interleave(n, 0).contiguous()       # the context row of a pixel, once per label (NMP.py:548)
        for i, layer in enumerate(prop.layers):
            m = layer.nmp
            qkv_i = K.from_kv16(tape["qkv"][i]) if tape["kv16"] else tape["qkv"][i]
            wb = [t for l in (m.q, m.k, m.v) for t in (l.weight, l.bias)]
            qkv = QkvFn.apply(x, ctx_tok, m.norm1.weight, m.norm1.bias, m.norm1.eps, lambda v=qkv_i: v, *wb)
            msg = StripeAttnFn.apply(qkv, m.attns[0].get_v.weight, m.attns[1].get_v.weight, (b, h, wd, n), lambda v=tape["msg"][i]: v)
            x = BlockFn.apply(x, msg, m.proj.weight, m.proj.bias, m.norm2.weight, m.norm2.bias, m.mlp.fc1.weight, m.mlp.fc1.bias,
                              m.mlp.fc2.weight, m.mlp.fc2.bias, m.norm2.eps, lambda v=tape["x"][i + 1]: v)
        return x

    def _tail_with_grad(self, labels_curr, dims8, heads4, tok4, out_hw, prob, label_seeds, cv_rows, graph=None):
        """The tail of hot_path in training mode with grad_slice: norms + heads of every layer under autograd.  graph (full mode): the
        encoder's 1/8 maps, the DPN context and the matching heads' maps at 1/8 and 1/4 WITH their autograd graphs (stock convolutions)."""
        from .autograd_ops import CostVolumeFn, DpnFilterFn, LayerNormFn, LinearFn, MlpHeadFn, refine_epilogue_torch
        from .nmp import _ChainLauncher
        b, h8, w8, n = dims8
        if not hasattr(self, "_score"):
            self._score = _ChainLauncher(3, (self.infer_score_head,), (128,), self.infer_score_head.out_features)
        un = lambda x: x.reshape(b, h8, w8, n, 8, 8).permute(0, 1, 4, 2, 5, 3).reshape(b, h8 * 8, w8 * 8, n)
        lab = labels_curr.reshape(-1, 1)

        def head(mlp, rows):
            l = mlp.layers
            return MlpHeadFn.apply(rows, l[0].weight, l[0].bias, l[1].weight, l[1].bias, l[2].weight, l[2].bias, lambda t: mlp(t))

        def last_rows(stage):
            """The last layer's residual stream as a function of that layer's OWN block parameters (proj, norm2, fc1, fc2): its
            operands x, msg are constants of the backward (msg comes out of a forward-only attention kernel), its output is the
            fused launch's -- the rows every earlier layer contributed stay detached."""
            from .autograd_ops import BlockFn
            x_in, msg, x_out, m, keep = stage._last_block
            y = BlockFn.apply(x_in, msg, m.proj.weight, m.proj.bias, m.norm2.weight, m.norm2.bias, m.mlp.fc1.weight, m.mlp.fc1.bias,
                              m.mlp.fc2.weight, m.mlp.fc2.bias, m.norm2.eps, lambda: x_out)
            return y if keep is None else y.index_select(0, keep)

        with torch.enable_grad():
            # the matching distribution as a function of the seed filter's parameters (the `init` loss of the Criterion, NMRF.py:300-330)
            fm = self.dpn.mlp
            cv_rows = cv_rows.contiguous()
            cv_g = None
            if graph is not None:
                cv_g = cv_rows = CostVolumeFn.apply(*graph["f8"], cv_rows.shape[2], cv_rows.shape[1], lambda v=cv_rows: v)
            prob = DpnFilterFn.apply(cv_rows, fm[0].weight, fm[0].bias, fm[2].weight, fm[2].bias, fm[4].weight, fm[4].bias,
                                     lambda v=prob: v)
            # disparity proposals (DPN.py:131-132): labels = relu(prop_head(norm(last propagation block)) + seeds) as a function of the
            # propagation stage's last block, its final norm and the head -- the loss_prop branch of the Criterion
            prop = self.dpn.propagation
            if graph is None:
                mem = self._propagation_rows_with_grad(prop)
            else:
                mem = self._propagation_rows_with_grad(prop, cv_g, label_seeds.reshape(-1, n).long().contiguous(), graph["context"])
            if mem is None:
                mem = last_rows(prop)
            if prop.norm is not None:
                mem = LayerNormFn.apply(mem, prop.norm.weight, prop.norm.bias, prop.norm.eps)
            proposal = torch.relu(head(self.dpn.prop_head, mem).view(-1, n) + label_seeds.reshape(-1, n)).reshape(b, -1, n)
            nm = self.inference.norm
            aux, delta, score = [], None, None
            pres = self._stage_rows_with_grad(self.inference, graph and graph["heads8"], labels_curr)
            if pres is None:
                pres = list(self.inference._pre_norm)
                pres[-1] = last_rows(self.inference)
            for pre in pres:
                rows = LayerNormFn.apply(pre, nm.weight, nm.bias, nm.eps)
                delta = head(self.infer_head, rows)
                score = LinearFn.apply(rows, self.infer_score_head.weight, self.infer_score_head.bias, lambda t: self._score(t, 128))
                aux.append({"disp_pred": un(torch.relu(lab + delta)), "logits_pred": un(0.25 * score)})
            disp_curr = K.wta_median(delta.detach().contiguous(), score.detach().contiguous(), labels_curr.reshape(-1).contiguous(), b, h8, w8, n)
            fmap1, fmap2, fmap1_gw, fmap2_gw = heads4
            with torch.no_grad():
                self.refinement(disp_curr, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=tok4)
            nm4 = self.refinement.norm
            preds = []
            pres = self._stage_rows_with_grad(self.refinement, graph and graph["heads4"], disp_curr)
            if pres is None:
                pres = list(self.refinement._pre_norm)
                pres[-1] = last_rows(self.refinement)
            for pre in pres:
                rows = LayerNormFn.apply(pre, nm4.weight, nm4.bias, nm4.eps)
                preds.append(refine_epilogue_torch(head(self.refine_head, rows), disp_curr))
        out = {"proposal": proposal, "prob": prob, "initial_proposal": label_seeds.reshape(b, -1, n),
               "disp": preds[-1][0], "disp_pred": preds[-1][1]}
        if self.aux_loss:
            out["aux_outputs"] = aux + [{"disp_pred": p[1]} for p in preds[:-1]]
        return out

    def check_range(self):
        """Raise NmrfHipError if an activation left the fp16 range of the split-operand kernels since the last check (syncs)."""
        return K.check_range(self.device)

    def _match_heads(self, left, right, cache):
        """concatconv / gw on both views as ONE stock 3x3 convolution: the two heads share their input, so their
        first convs are stacked along the output channels and the two views along the batch (conv3x3 ->
        InstanceNorm -> ReLU are per-(sample, channel) independent: same arithmetic as NMRF.py:211-214,233-236),
        then one 1x1 conv per head on its 128-channel slice.  Returns ((fmap1, fmap2, fmap1_gw, fmap2_gw), token_major):
        with the shipped head widths (64 / 256 channels) the maps are written token-major [B,H,W,C] -- their only consumer is the
        warp + correlation kernel, which then reads every tap as one contiguous row."""
        heads = (self.concatconv, self.gw)
        w3 = cache.get(tuple(h[0].weight for h in heads), lambda: torch.cat([h[0].weight for h in heads], 0).contiguous())
        b = left.shape[0]
        if not hasattr(cache, "wino"):
            cache.wino = {}
        both = None
        for jf in getattr(self, "_joint_feats", None) or ():       # left / right are the two halves of one encoder output
            if (jf.shape[0] == 2 * b and jf.shape[1:] == left.shape[1:] and jf.is_contiguous() and jf.data_ptr() == left.data_ptr()
                    and jf[b:].data_ptr() == right.data_ptr()):
                both = jf
        if both is None:
            both = torch.cat((left, right), 0)
        raw = K.conv3x3_auto(both, w3, cache.wino).contiguous()
        wf, wg = self.concatconv[3].weight, self.gw[3].weight
        if wf.shape[1] == 128 and wg.shape[1] == 128 and wf.shape[0] % 64 == 0 and wg.shape[0] % 64 == 0:
            # InstanceNorm + ReLU + the two 1x1 convs read the 3x3 output once: statistics pass, then one fused kernel per head
            if not hasattr(cache, "c1"):
                cache.c1 = (_FusedCache(), _FusedCache())
            stats = K.instance_stats(raw)
            tok = wf.shape[0] == 64 and wg.shape[0] == 256 and self.inference.cost_group == 32
            f = K.conv1x1_in_relu(raw, 0, 128, stats, cache.c1[0].get((wf,), lambda: K.pack_conv1x1(wf)), token_major=tok)
            g = K.conv1x1_in_relu(raw, 128, 128, stats, cache.c1[1].get((wg,), lambda: K.pack_conv1x1(wg)), token_major=tok)
        else:
            tok = False
            y = K.instance_norm(raw, relu=True)
            f = F.conv2d(y[:, 0:128], wf)
            g = F.conv2d(y[:, 128:256], wg)
        return (f[:b].contiguous(), f[b:].contiguous(), g[:b].contiguous(), g[b:].contiguous()), tok

    def hot_path(self, fmap1_list, fmap2_list, out_hw, stages=None):
        """Everything after the backbone (NMRF.py:207-262): fmap lists are [1/8-res, 1/4-res] NCHW maps of the
        left / right view; out_hw the un-padded image size.  This is the region the bench's hot-path timer brackets.
        `stages`: optional dict that receives the tensors around the one discrete decision of the path -- the winner-take-all
        of NMRF.py:228 -- (`infer_tgt`, `infer_delta` [T,64], `infer_score` [T,64] = the score head WITHOUT the 0.25 factor,
        `disp_curr`, `refine_tgt`), the counterpart of the reference's forward hooks, for the parity chain of tests/util.py."""
        n = self.num_proposals
        h0, w0 = out_hw

        # ---- disparity proposals -------------------------------------------------------------------
        # The matching heads (stock convs, needed only from the inference stage on) run on a side stream while
        # the proposal stage -- latency-bound attention kernels that leave most of the chip idle at batch 1 --
        # runs on the main one; joined before their first consumer.  Captured as parallel branches by hipGraph.
        main = torch.cuda.current_stream()
        full = self._grad_full() and stages is None                # (N4, whole model: the convolutional heads run under autograd below)
        overlap = os.environ.get("NMRF_OVERLAP", "1") != "0" and not full
        side = main
        if overlap:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=fmap1_list[0].device)
            side = self._side_stream
            side.wait_stream(main)
        context, ctx_ready = None, None
        graph = None
        if full:
            # N4, the whole model: the convolutional heads on stock PyTorch-ROCm autograd (both views as one batch: conv - InstanceNorm -
            # ReLU - conv are per-sample, NMRF.py:211-214,233-236; DPN.py:127); the HIP stages below read the detached NCHW maps
            with torch.enable_grad():
                cat8, cat4 = torch.cat((fmap1_list[0], fmap2_list[0]), 0), torch.cat((fmap1_list[1], fmap2_list[1]), 0)
                bq = fmap1_list[0].shape[0]
                ctx_g = self.dpn.proj(fmap1_list[0])
                split2 = lambda t: (t[:bq], t[bq:])
                heads8_g = split2(self.concatconv(cat8)) + split2(self.gw(cat8))
                heads4_g = split2(self.concatconv(cat4)) + split2(self.gw(cat4))
            graph = {"f8": (fmap1_list[0], fmap2_list[0]), "context": ctx_g, "heads8": heads8_g, "heads4": heads4_g}
            context = ctx_g.detach().permute(0, 2, 3, 1).contiguous()
            heads8, tok8 = tuple(t.detach().contiguous() for t in heads8_g), False
            heads4, tok4 = tuple(t.detach().contiguous() for t in heads4_g), False
            fmap1_list, fmap2_list = [f.detach() for f in fmap1_list], [f.detach() for f in fmap2_list]
        with torch.cuda.stream(side):
            if overlap:
                # the DPN context convs first: the seed stage (cost volume, conv1d + softmax, NMS: latency-bound) runs beside them
                context = self.dpn.context(fmap1_list[0], token_major=True)
                ctx_ready = torch.cuda.Event()
                ctx_ready.record(side)
                context.record_stream(main)
            if not full:
                heads8, tok8 = self._match_heads(fmap1_list[0], fmap2_list[0], self._head_cache8)
                heads4, tok4 = self._match_heads(fmap1_list[1], fmap2_list[1], self._head_cache4)
            if overlap:
                for t in heads8 + heads4:
                    t.record_stream(main)

        cost_volume = K.cost_volume(fmap1_list[0], fmap2_list[0], self.max_disp // 8, self.dpn.cost_group)
        cv_rows, prob, label_seeds, labels = self.dpn(cost_volume, fmap1_list, context=context, context_ready=ctx_ready)      # cv_rows [P,G,D]
        labels_curr = labels[-1]                                            # [P, N]
        if overlap:
            main.wait_stream(side)
        fmap1, fmap2, fmap1_gw, fmap2_gw = heads8

        # ---- neural MRF inference at 1/8 -------------------------------------------------------------
        tgt_all = self.inference(labels_curr, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=tok8)    # [1 | layers, P, N, C]
        tgt = tgt_all[-1].reshape(-1, self.inference.dim)
        b, h8, w8 = fmap1_list[0].shape[0], fmap1_list[0].shape[2], fmap1_list[0].shape[3]
        if self.training and getattr(self, "grad_slice", False) and stages is None:
            if not (self.inference.return_intermediate and self.refinement.return_intermediate):
                raise NotImplementedError("grad_slice: the training-mode forward with NMP.RETURN_INTERMEDIATE (the reference's default)")
            return self._tail_with_grad(labels_curr, (b, h8, w8, n), heads4, tok4, (h0, w0), prob, label_seeds, cv_rows, graph)
        from .nmp import _ChainLauncher, _FusedCache, _split
        hl = self.infer_head.layers
        if (stages is None and not self.training and _split() and n == 4 and len(hl) == 3
                and all(l.in_features == 128 for l in hl) and hl[0].out_features == 128 and hl[1].out_features == 128
                and hl[2].out_features == 64 and self.infer_score_head.in_features == 128 and self.infer_score_head.out_features == 64):
            # disparity head + score head + winner-take-all + medians in one launch (nmrf_heads_wta_f32): their [T,64] rows stay on
            # the CU.  (The parity chain of the tests asks for those rows through `stages` and takes the three launches below.)
            if not hasattr(self, "_heads_wta"):
                self._heads_wta = _FusedCache()
            ws = (hl[0].weight, hl[1].weight, hl[2].weight, self.infer_score_head.weight)
            bs = (hl[0].bias, hl[1].bias, hl[2].bias, self.infer_score_head.bias)
            stream, st, inv = self._heads_wta.get(ws + tuple(x for x in bs if x is not None), lambda: K.heads_wta_stream(*ws))
            disp_curr = K.heads_wta(tgt.contiguous(), stream, st, inv, bs, labels_curr.reshape(-1).contiguous(), b, h8, w8, n)
            disp_delta = score = None                                       # (only the training-mode outputs and `stages` use them)
        else:
            disp_delta = self.infer_head(tgt)                               # [T,64]
            if _split() and self.infer_score_head.in_features == 128 and self.infer_score_head.out_features <= 64:
                if not hasattr(self, "_score"):
                    self._score = _ChainLauncher(3, (self.infer_score_head,), (128,), self.infer_score_head.out_features)
                score = self._score(tgt, 128)
            else:
                score = K.linear_smalln(tgt, self.infer_score_head.weight, self.infer_score_head.bias)   # [T,64]; the 0.25 factor
            #                                                                 does not change the arg-max
            disp_curr = K.wta_median(disp_delta, score, labels_curr.reshape(-1).contiguous(), b, h8, w8, n)
        if stages is not None:
            stages.update(infer_tgt=tgt, infer_delta=disp_delta, infer_score=score, disp_curr=disp_curr)

        # ---- refinement at 1/4 ---------------------------------------------------------------------------
        fmap1, fmap2, fmap1_gw, fmap2_gw = heads4
        tgt4_all = self.refinement(disp_curr, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=tok4)    # [1 | layers, P4, C]
        tgt = tgt4_all[-1].reshape(-1, self.refinement.dim)
        rl = self.refine_head.layers
        if (not self.training and _split() and len(rl) == 3 and all(l.in_features == 128 for l in rl) and rl[0].out_features == 128
                and rl[1].out_features == 128 and rl[2].out_features == 16):
            # the head's [T,16] rows are the 4 x 4 patches: shuffled, scaled and cropped straight from the chain kernel's registers
            if not hasattr(self, "_refine_epi"):
                self._refine_epi = _FusedCache()
            ws = tuple(l.weight for l in rl)
            bs = tuple(l.bias for l in rl)
            stream, st, inv = self._refine_epi.get(ws + tuple(x for x in bs if x is not None),
                                                   lambda: K.chain_stream(list(ws), (128, 128, 128)))
            disp, disp_pred = K.refine_head_epilogue(tgt.contiguous(), stream, st, inv, bs, disp_curr, h0, w0)
        else:
            disp, disp_pred = K.refine_epilogue(self.refine_head(tgt), disp_curr, h0, w0)
        if stages is not None:
            stages["refine_tgt"] = tgt

        out = {"proposal": labels_curr.reshape(b, -1, n), "prob": prob,
               "initial_proposal": label_seeds.reshape(b, -1, n), "disp": disp, "disp_pred": disp_pred}
        if self.aux_loss and self.training:
            out["aux_outputs"] = self._aux_outputs(tgt_all, tgt4_all, labels_curr, disp_curr, (b, h8, w8, n), (h0, w0),
                                                   (disp_delta, score))
        return out

    def _aux_outputs(self, tgt_all, tgt4_all, labels, disp_curr, dims8, out_hw, last_heads):
        """_set_aux_loss of the reference (NMRF.py:216-223, 240-244, 264-273): every inference layer's normalised tokens through the
        SAME heads -> {disp_pred: relu(label + delta) [B,8H,8W,N], logits_pred: 0.25 x score}, then every refinement layer's but the
        last -> {disp_pred [B,4H4,4W4]}.  The heads are this module's HIP chains; the 8x8 un-shuffle is a view + permute."""
        b, h8, w8, n = dims8
        un = lambda x: x.reshape(b, h8, w8, n, 8, 8).permute(0, 1, 4, 2, 5, 3).reshape(b, h8 * 8, w8 * 8, n)
        lab = labels.reshape(-1, 1)
        res = []
        for i in range(tgt_all.shape[0]):
            if i + 1 == tgt_all.shape[0]:
                delta, score = last_heads                                   # the heads of the last layer ran on the hot path already
            else:
                t = tgt_all[i].reshape(-1, self.inference.dim).contiguous()
                delta = self.infer_head(t)
                score = self._score(t, 128) if hasattr(self, "_score") else K.linear_smalln(t, self.infer_score_head.weight,
                                                                                             self.infer_score_head.bias)
            res.append({"disp_pred": un(torch.relu(lab + delta)), "logits_pred": un(0.25 * score)})
        for i in range(tgt4_all.shape[0] - 1):
            t = tgt4_all[i].reshape(-1, self.refinement.dim).contiguous()
            res.append({"disp_pred": K.refine_epilogue(self.refine_head(t), disp_curr, *out_hw)[1]})
        return res


def build(cfg):
    """(model, criterion) like nmrf/models/NMRF.py:432-447: the model in nn.Module's default training state, as the reference returns it
    (callers of the inference path call .eval(), inference.py:150); the criterion (models/criterion.py, plain PyTorch) evaluates the
    reference's loss terms on its output dictionary -- differentiable once model.enable_grad_slice() is on (nmrf_amd/train.py)."""
    from .criterion import build_criterion
    kwargs = NMRF.from_config(cfg)
    return NMRF(**kwargs), build_criterion(cfg)
