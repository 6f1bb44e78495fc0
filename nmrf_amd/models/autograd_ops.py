"""N4: torch.autograd.Functions over the HIP kernels, so that a loss evaluated on the training-mode forward can be
differentiated with respect to every parameter behind the `labels_curr` hand-over.

The reference differentiates its whole forward with autograd (nmrf/models/NMRF.py:387-429, main.py:413-430) and detaches the two
discrete hand-overs between stages (`labels_curr`, NMRF.py:215; `disp_curr`, NMRF.py:231).  Everything from the inference stage's ffn to
the outputs therefore depends only on tensors the forward already holds.  The Functions here cover that part completely: FfnFn (timm Mlp),
QkvFn (LayerNorm | extra -> one or three linears), SelfAttnFn (sibling labels of a pixel), WindowAttnFn ((shifted) windows with the
relative-position q / k / v embeddings), ProjFn / BlockFn (proj + residual [+ norm2 + MLP]), LayerNormFn, MlpHeadFn / LinearFn (the heads).
What has no backward yet: the propagation's stripe attention, the seed stage (cost volume, conv1d filter, NMS), warp + correlation with
respect to the feature maps, the convolutions.

Every Function's FORWARD is the product's fused launch (or the tensor that launch already produced: the same bits the forward-only path
returns); its backward recomputes what it needs from the saved inputs and composes the pieces of csrc/backward.hip: split-operand fp16
MFMA GEMMs for dgrad / wgrad (fp32 accumulation), attention backward kernels in fp32, deterministic reductions over tokens and windows.
The only torch arithmetic in a backward is plumbing: concatenating operand rows, adding two gradients where a residual branch joins."""
import torch

from .. import kernels as K


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class LinearFn(torch.autograd.Function):
    """y = x W^T + b.  fwd(x) -> y is the product's launch for this layer."""

    @staticmethod
    def forward(ctx, x, w, b, fwd):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return fwd(x)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = K.linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        return dx, K.linear_wgrad(dy, x), (K.bias_grad(dy) if ctx.has_bias else None), None


class MlpHeadFn(torch.autograd.Function):
    """y = L3(relu(L2(relu(L1 x))))  (MLP of NMP.py:54-66: the prediction heads).  fwd(x) -> y: the fused chain launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, fwd):
        ctx.save_for_backward(x, w1, b1, w2, b2, w3)
        return fwd(x)

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, w3 = ctx.saved_tensors
        dy = _c(dy)
        p1, a1 = K.bias_act(K.linear_forward(x, w1), b1, 1)              # recomputed, not saved: [T,128] rows twice
        p2, a2 = K.bias_act(K.linear_forward(a1, w2), b2, 1)
        dw3, db3 = K.linear_wgrad(dy, a2), K.bias_grad(dy)
        d2 = K.act_backward(p2, K.linear_dgrad(dy, w3), 1)
        dw2, db2 = K.linear_wgrad(d2, a1), K.bias_grad(d2)
        d1 = K.act_backward(p1, K.linear_dgrad(d2, w2), 1)
        dw1, db1 = K.linear_wgrad(d1, x), K.bias_grad(d1)
        dx = K.linear_dgrad(d1, w1) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, dw3, db3, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (the stage-final `norm` of Inference / Refinement, NMP.py:777-798, 879-898)."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return K.layer_norm(x, g, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg, db = K.layer_norm_backward(x, g, _c(dy), ctx.eps)
        return (dx if ctx.needs_input_grad[0] else None), dg, db, None


class BlockFn(torch.autograd.Function):
    """One message-passing block without its q stage (SwinNMP / CSWinNMP.forward_pre, NMP.py:337-364, 537-574):
           x1 = x + proj(msg);   x2 = x1 + fc2(gelu(fc1(LayerNorm2(x1))))
    fwd() -> x2 is the product's fused nmp_block16 launch on the same operands."""

    @staticmethod
    def forward(ctx, x, msg, wp, bp, g2, b2n, w1, b1, w2, b2, eps, fwd):
        ctx.save_for_backward(x, msg, wp, bp, g2, b2n, w1, b1, w2)
        ctx.eps = eps
        return fwd()

    @staticmethod
    def backward(ctx, dy):
        x, msg, wp, bp, g2, b2n, w1, b1, w2 = ctx.saved_tensors
        dy = _c(dy)
        _, proj = K.bias_act(K.linear_forward(msg, wp), bp, 0)
        x1 = x + proj
        ln = K.layer_norm(x1, g2, b2n, ctx.eps)
        p, h = K.bias_act(K.linear_forward(ln, w1), b1, 2)
        dw2, db2 = K.linear_wgrad(dy, h), K.bias_grad(dy)
        dp = K.act_backward(p, K.linear_dgrad(dy, w2), 2)
        dw1, db1 = K.linear_wgrad(dp, ln), K.bias_grad(dp)
        dx1_ln, dg2, db2n = K.layer_norm_backward(x1, g2, K.linear_dgrad(dp, w1), ctx.eps)
        dx1 = dy + dx1_ln                                              # the residual branch joins
        dwp, dbp = K.linear_wgrad(dx1, msg), K.bias_grad(dx1)
        dmsg = K.linear_dgrad(dx1, wp) if ctx.needs_input_grad[1] else None
        return (dx1 if ctx.needs_input_grad[0] else None), dmsg, dwp, dbp, dg2, db2n, dw1, db1, dw2, db2, None, None


class WindowAttnFn(torch.autograd.Function):
    """(Shifted-)window attention with relative-position embeddings on fp32 q | k | v rows (WindowAttention.forward, NMP.py:185-289).
    fwd() -> the forward kernel's output on the same operands (or the tensor the fused forward already produced)."""

    @staticmethod
    def forward(ctx, qkv, table, geom, fwd):
        ctx.save_for_backward(qkv, table)
        ctx.geom = geom                                              # (b, hp, wp, n, heads, win, shift, sibling_mask)
        return fwd()

    @staticmethod
    def backward(ctx, dout):
        qkv, table = ctx.saved_tensors
        dqkv, dtab = K.window_attn_backward(qkv, table, _c(dout), *ctx.geom)
        return (dqkv if ctx.needs_input_grad[0] else None), dtab, None, None


class QkvFn(torch.autograd.Function):
    """The q | k | v projection of a message-passing block on [LayerNorm1(x) | extra]: one fused linear (SwinNMP.qkv, NMP.py:343-349) or
    three (BasicAttention / CSWinNMP q, k on [LN(x) | extra], v on LN(x) alone, NMP.py:90-96): y = cat_i([LN(x) | extra][:, :K_i] W_i^T + b_i).
    Arguments after fwd: w_1, b_1, w_2, b_2, ... (W_i [N_i, K_i], K_i <= C + E).  fwd() -> y (the fused launch's q_out, fp32 rows)."""

    @staticmethod
    def forward(ctx, x, extra, g, b, eps, fwd, *wb):
        ctx.save_for_backward(x, extra, g, b, *wb[0::2])
        ctx.eps = eps
        ctx.has_bias = [bb is not None for bb in wb[1::2]]
        return fwd()

    @staticmethod
    def backward(ctx, dy):
        x, extra, g, b, *ws = ctx.saved_tensors
        dy = _c(dy)
        c = x.shape[1]
        kmax = max(w.shape[1] for w in ws)
        cat = torch.cat((K.layer_norm(x, g, b, ctx.eps), extra[:, : kmax - c]), 1).contiguous()        # (plumbing: the operand rows)
        wf = torch.cat([w if w.shape[1] == kmax else torch.nn.functional.pad(w, (0, kmax - w.shape[1])) for w in ws], 0).contiguous()
        dwf, dbf = K.linear_wgrad(dy, cat), K.bias_grad(dy)
        dcat = K.linear_dgrad(dy, wf)
        dx, dg, db = K.layer_norm_backward(x, g, dcat[:, :c].contiguous(), ctx.eps)
        grads, r0 = [], 0
        for w, hb in zip(ws, ctx.has_bias):
            n = w.shape[0]
            grads += [dwf[r0:r0 + n, : w.shape[1]].contiguous(), dbf[r0:r0 + n].contiguous() if hb else None]
            r0 += n
        dextra = None
        if ctx.needs_input_grad[1]:                                  # (the context rows of the propagation stage: DPN.proj is trainable)
            dextra = torch.zeros_like(extra)
            dextra[:, : kmax - c] = dcat[:, c:]
        return ((dx if ctx.needs_input_grad[0] else None), dextra, dg, db, None, None, *grads)


class SelfAttnFn(torch.autograd.Function):
    """Per-pixel self-edge attention over the N sibling labels (BasicAttention.forward_pre, NMP.py:97-103) on fp32 q | k | v rows."""

    @staticmethod
    def forward(ctx, qkv, n, heads):
        ctx.save_for_backward(qkv)
        ctx.n, ctx.heads = n, heads
        return K.self_attn(qkv, n, heads)

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        return K.self_attn_backward(qkv, _c(dout), ctx.n, ctx.heads), None, None


class StripeAttnFn(torch.autograd.Function):
    """Cross-stripe attention with LePE on fp32 q | k | v rows (CSWinAttention.forward, NMP.py:429-505).  fwd() -> the message rows."""

    @staticmethod
    def forward(ctx, qkv, lepe_v, lepe_h, geom, fwd):
        ctx.save_for_backward(qkv, lepe_v, lepe_h)
        ctx.geom = geom                                              # (b, h, w, n)
        return fwd()

    @staticmethod
    def backward(ctx, dout):
        qkv, lepe_v, lepe_h = ctx.saved_tensors
        dqkv, dlv, dlh = K.stripe_attn_backward(qkv, lepe_v.contiguous(), lepe_h.contiguous(), _c(dout), *ctx.geom)
        return (dqkv if ctx.needs_input_grad[0] else None), dlv, dlh, None, None


class DpnFilterFn(torch.autograd.Function):
    """The seed filter: prob = softmax_D(conv1d(relu(conv1d(relu(conv1d(cost volume)))))) (DPN.mlp, DPN.py:32-38,117-119), differentiated
    with respect to its six parameters and, when it carries a graph (CostVolumeFn: the full training mode), the cost volume.
    fwd() -> prob of the fused kernel.  Each Conv1d(kernel 5, padding 2) over D is a Linear on 5-tap columns (K.unfold5)."""

    @staticmethod
    def forward(ctx, cv, w0, b0, w2, b2, w4, b4, fwd):
        prob = fwd()
        ctx.save_for_backward(cv, w0, b0, w2, b2, w4, prob)
        return prob

    @staticmethod
    def backward(ctx, dprob):
        cv, w0, b0, w2, b2, w4, prob = ctx.saved_tensors
        p, g, d = cv.shape
        f = lambda w: w.reshape(w.shape[0], -1).contiguous()                     # [O, C, 5] -> [O, 5C] (c-major, tap-minor: unfold5's order)
        col0 = K.unfold5(cv, p, g, d, src_pcd=True)
        y1, a1 = K.bias_act(K.linear_forward(col0, f(w0)), b0, 1)
        col1 = K.unfold5(a1, p, w0.shape[0], d)
        y2, a2 = K.bias_act(K.linear_forward(col1, f(w2)), b2, 1)
        col2 = K.unfold5(a2, p, w2.shape[0], d)
        dz = K.softmax_backward(prob, _c(dprob)).reshape(p * d, 1)
        dw4, db4 = K.linear_wgrad(dz, col2), K.bias_grad(dz)
        dy2 = K.act_backward(y2, K.fold5(K.linear_dgrad(dz, f(w4)), p, w2.shape[0], d), 1)
        dw2, db2 = K.linear_wgrad(dy2, col1), K.bias_grad(dy2)
        dy1 = K.act_backward(y1, K.fold5(K.linear_dgrad(dy2, f(w2)), p, w0.shape[0], d), 1)
        dw0, db0 = K.linear_wgrad(dy1, col0), K.bias_grad(dy1)
        dcv = None
        if ctx.needs_input_grad[0]:                                  # rows (p, d) x G -> the volume's [P, G, D]
            dcv = K.fold5(K.linear_dgrad(dy1, f(w0)), p, g, d).view(p, d, g).permute(0, 2, 1).contiguous()
        return dcv, dw0.view_as(w0), db0, dw2.view_as(w2), db2, dw4.view_as(w4), db4, None


class ProjFn(torch.autograd.Function):
    """x + proj(msg): a block without MLP (the self-edge block, NMP.py:104-108).  fwd() -> the fused launch's x_out."""

    @staticmethod
    def forward(ctx, x, msg, wp, bp, fwd):
        ctx.save_for_backward(msg, wp)
        return fwd()

    @staticmethod
    def backward(ctx, dy):
        msg, wp = ctx.saved_tensors
        dy = _c(dy)
        dmsg = K.linear_dgrad(dy, wp) if ctx.needs_input_grad[1] else None
        return (dy if ctx.needs_input_grad[0] else None), dmsg, K.linear_wgrad(dy, msg), K.bias_grad(dy), None


class FfnFn(torch.autograd.Function):
    """timm Mlp: y = fc2(gelu(fc1 x))  (Inference.ffn / Refinement.ffn, NMP.py:675, 735-741).  fwd(x) -> y: the fused chain launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, fwd):
        ctx.save_for_backward(x, w1, b1, w2)
        return fwd(x)

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2 = ctx.saved_tensors
        dy = _c(dy)
        p, h = K.bias_act(K.linear_forward(x, w1), b1, 2)
        dw2, db2 = K.linear_wgrad(dy, h), K.bias_grad(dy)
        dp = K.act_backward(p, K.linear_dgrad(dy, w2), 2)
        dx = K.linear_dgrad(dp, w1) if ctx.needs_input_grad[0] else None
        return dx, K.linear_wgrad(dp, x), K.bias_grad(dp), dw2, db2, None


def refine_epilogue_torch(delta16, disp_curr, training_hw=None):
    """relu(disp_curr + delta) pixel-shuffled 4x4 (NMRF.py:238-245) as differentiable torch views for the training-mode loss:
    delta16 [B*H4*W4, 16], disp_curr [B,H4,W4] -> (disp = 4 * disp_pred, disp_pred [B, 4*H4, 4*W4]).  (No un-padding: the training-mode
    forward does not pad, NMRF.py:203-205.)"""
    b, h4, w4 = disp_curr.shape
    pred = torch.relu(disp_curr.reshape(b, h4, w4, 1) + delta16.view(b, h4, w4, 16))
    pred = pred.view(b, h4, w4, 4, 4).permute(0, 1, 3, 2, 4).reshape(b, 4 * h4, 4 * w4)
    return pred * 4, pred


class CostVolumeFn(torch.autograd.Function):
    """Group-wise correlation volume of the 1/8 maps (build_correlation_volume, submodule.py:4-23) -> [B*H*W, G, D].  fwd() -> the volume the
    forward built (K.cost_volume).  Backward = nmrf_cost_volume_bwd_f32; from there the encoder is stock PyTorch autograd."""

    @staticmethod
    def forward(ctx, f1, f2, num_disp, groups, fwd):
        ctx.save_for_backward(f1, f2)
        ctx.dg = (num_disp, groups)
        return fwd()

    @staticmethod
    def backward(ctx, dcv):
        f1, f2 = ctx.saved_tensors
        df1, df2 = K.cost_volume_backward(_c(f1), _c(f2), _c(dcv), *ctx.dg)
        return df1, df2, None, None, None


class SeedTapsFn(torch.autograd.Function):
    """The 9 cost taps around every label seed (Propagation.sample_cost, NMP.py:619-634) as a function of the cost volume [P, G, D]; the
    seeds are integer NMS output.  fwd() -> the [P*N, 9 G] rows the seed kernel gathered."""

    @staticmethod
    def forward(ctx, cv, seeds, fwd):
        ctx.save_for_backward(seeds)
        ctx.gd = cv.shape[1:]
        return fwd()

    @staticmethod
    def backward(ctx, dcost):
        (seeds,) = ctx.saved_tensors
        return K.seed_taps_backward(_c(dcost), seeds, *ctx.gd), None, None


class WarpCorrFn(torch.autograd.Function):
    """[left features | right features warped at x - label | group correlation] rows of a message-passing stage (Inference.sample_fmap /
    corr, NMP.py:683-741) as a function of the four matching-head maps (NCHW); the labels are constants (NMP.py:694, NMRF.py:215,232).
    fwd() -> the rows the forward built (K.warp_corr_concat)."""

    @staticmethod
    def forward(ctx, f1, f2, g1, g2, labels, n, groups, fwd):
        ctx.save_for_backward(f1, f2, g1, g2, labels)
        ctx.ng = (n, groups)
        return fwd()

    @staticmethod
    def backward(ctx, drow):
        f1, f2, g1, g2, labels = ctx.saved_tensors
        grads = K.warp_corr_concat_backward(_c(labels), _c(drow), _c(f1), _c(f2), _c(g1), _c(g2), *ctx.ng)
        return (*grads, None, None, None, None)
