"""CPU ORACLE for the NMRF-Stereo inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU; fp32, or fp64 when handed double tensors --
tools/flip_floor.py) *restatement* of the reference algorithm, written from SURVEY.md §8 and the cited reference lines; it is the
checker the HIP kernels are compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product (``nmrf_amd``) never does: it fails loudly when the
HIP library is missing.

Pinning: every stage returned by :func:`forward` is checked against golden
vectors captured from the real reference (imported in the build container by
``tools/gen_golden.py``) in ``tests/test_oracle_golden.py``.  The arithmetic of
third-party ATen ops the reference calls (``topk`` tie order, ``grid_sample``,
``median``) is restated explicitly here (see ``topk_ties``, ``warp_row``) and
in ``oracle/topk_ref.c``, and pinned by the same goldens.  Pinned configurations: the
default one (hash weights: e2e_a ... e2e_d; the TRAINED reference checkpoint: e2e_t), the Swin-T encoder (e2e_swin), the training-mode
forward (e2e_train) and NMP.NORMALIZE_BEFORE False -- the forward_post blocks (e2e_post).  The functions are differentiable torch code: the
backward kernels are compared with torch autograd of these restatements in fp64 (tests/test_hip_kernels.py), the model-level gradients
with the reference's own autograd (tests/golden/e2e_train*.npz).

All citations are ``file:line`` relative to the reference repo root.
Functional style: ``w`` is a flat dict of tensors keyed by the reference's
state-dict names.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class OracleCfg:
    """Hot-path keys of nmrf/config/default.py:29-61,83."""
    max_disp: int = 320
    cost_group: int = 4
    num_proposals: int = 4
    context_dim: int = 64
    embed_dim: int = 128
    mlp_ratio: int = 4
    window_size: int = 6
    refine_window_size: int = 4
    prop_heads: int = 4
    infer_heads: int = 4
    num_prop_layers: int = 5
    num_infer_layers: int = 5
    num_refine_layers: int = 5
    divis_by: int = 8
    eps: float = 1e-3          # DPN.py:39
    backbone_prefix: str = "backbone"   # "image_encoder" when COMPAT=False (NMRF.py:108-113)
    normalize_before: bool = True       # NMP.NORMALIZE_BEFORE: False = the forward_post form of every block (NMP.py:110-135, 366-382, 576-591)


# --------------------------------------------------------------------------- #
# small helpers
# --------------------------------------------------------------------------- #
def _lin(x, w, pre, bias=True):
    return F.linear(x, w[pre + ".weight"], w[pre + ".bias"] if bias else None)


def _ln(x, w, pre):
    return F.layer_norm(x, (x.shape[-1],), w[pre + ".weight"], w[pre + ".bias"], 1e-5)


def _relu_mlp(x, w, pre, n=3):
    """NMP.py:54-66 (MLP): Linear-ReLU ... Linear."""
    for i in range(n):
        x = _lin(x, w, f"{pre}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def _gelu_mlp(x, w, pre):
    """timm Mlp as used at NMP.py:337,537,675: fc1 - GELU(erf) - fc2."""
    return _lin(F.gelu(_lin(x, w, pre + ".fc1")), w, pre + ".fc2")


# --------------------------------------------------------------------------- #
# A1  input padding (frame_utils.py:259-281, mode 'proposal')
# --------------------------------------------------------------------------- #
def pad_amounts(h, w, divis_by):
    return (-h) % divis_by, (-w) % divis_by


def pad_images(img1, img2, divis_by):
    ph, pw = pad_amounts(img1.shape[-2], img1.shape[-1], divis_by)
    f = lambda x: F.pad(x, (0, pw, 0, ph), mode="replicate")
    return f(img1), f(img2), (ph, pw)


# --------------------------------------------------------------------------- #
# stock CNN backbone + stock conv heads (out of hot-path scope, needed to feed it)
# backbone.py:16-98 ; NMRF.py:56-65 ; DPN.py:45-49
# --------------------------------------------------------------------------- #
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def _resblock(x, w, pre, stride):
    y = F.relu(_inorm(F.conv2d(x, w[pre + ".conv1.weight"], None, stride, 1)))
    y = F.relu(_inorm(F.conv2d(y, w[pre + ".conv2.weight"], None, 1, 1)))
    if (pre + ".downsample.0.weight") in w:
        x = _inorm(F.conv2d(x, w[pre + ".downsample.0.weight"], w[pre + ".downsample.0.bias"], stride))
    return F.relu(x + y)


def cnn_backbone(img, w, pre):
    x = 2 * (img / 255.0) - 1.0
    x = F.relu(_inorm(F.conv2d(x, w[pre + ".conv1.weight"], None, 2, 3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
        x = _resblock(x, w, f"{pre}.{name}.0", stride)
        x = _resblock(x, w, f"{pre}.{name}.1", 1)
    x = F.conv2d(x, w[pre + ".conv2.weight"], w[pre + ".conv2.bias"])
    return x, F.avg_pool2d(x, 2, 2)          # 1/4 res, 1/8 res


def conv_head(x, w, pre):
    """conv3x3 - InstanceNorm - ReLU - conv1x1 (concatconv / gw / dpn.proj)."""
    y = F.relu(_inorm(F.conv2d(x, w[pre + ".0.weight"], None, 1, 1)))
    return F.conv2d(y, w[pre + ".3.weight"])


# --------------------------------------------------------------------------- #
# A2  group-wise correlation volume (submodule.py:4-23)
# vol[b,g,d,y,x] = mean_c f1[b,g*cpg+c,y,x] * f2[b,g*cpg+c,y,x-d]  (0 for x<d)
# returned token-major: [B*H*W, G, D]  (the permute of DPN.py:117)
# --------------------------------------------------------------------------- #
def cost_volume(f1, f2, num_disp, groups):
    b, c, h, wd = f1.shape
    cpg = c // groups
    vol = f1.new_zeros(b, h, wd, groups, num_disp)
    for d in range(min(num_disp, wd)):
        prod = f1[:, :, :, d:] * f2[:, :, :, : wd - d]
        vol[:, :, d:, :, d] = prod.view(b, groups, cpg, h, wd - d).mean(2).permute(0, 2, 3, 1)
    return vol.reshape(b * h * wd, groups, num_disp)


# --------------------------------------------------------------------------- #
# A3  1-D conv filter along D + softmax (DPN.py:32-38,117-119)
# --------------------------------------------------------------------------- #
def dpn_filter_softmax(cv, w):
    x = F.relu(F.conv1d(cv, w["dpn.mlp.0.weight"], w["dpn.mlp.0.bias"], padding=2))
    x = F.relu(F.conv1d(x, w["dpn.mlp.2.weight"], w["dpn.mlp.2.bias"], padding=2))
    x = F.conv1d(x, w["dpn.mlp.4.weight"], w["dpn.mlp.4.bias"], padding=2)[:, 0]
    return F.softmax(x, dim=-1)


# --------------------------------------------------------------------------- #
# A4  label-seed NMS + top-k (DPN.py:120-125)
# --------------------------------------------------------------------------- #
def nms_suppress(prob, eps):
    """A bin survives if it is >= both neighbours (== max_pool1d(k=3) output);
    every other bin that is > eps is overwritten with exactly eps."""
    ninf = prob.new_full((prob.shape[0], 1), float("-inf"))
    left = torch.cat((ninf, prob[:, :-1]), 1)
    right = torch.cat((prob[:, 1:], ninf), 1)
    is_peak = (prob >= left) & (prob >= right)
    return torch.where(~is_peak & (prob > eps), prob.new_tensor(eps), prob)


def topk_ties(vals, k):
    """Pure-Python restatement of ATen's CPU top-k (TopKImpl.h, largest=True,
    sorted=True, k*64 > n branch): libstdc++ ``std::nth_element(b, b+k-1, e)``
    (introselect: median-of-3 pivot, unguarded Hoare partition, insertion sort
    below 4 elements, heap-select on depth exhaustion) followed by
    ``std::sort(b, b+k-1)``, on (value, index) pairs with comparator
    ``x>y or (isnan(x) and not isnan(y))``.  Slow; for small pinned cases.
    Same algorithm as oracle/topk_ref.c and the HIP kernel."""
    out = []
    for row in vals.tolist():
        q = [(v, i) for i, v in enumerate(row)]
        _nth_element(q, k - 1)
        head = q[: k - 1]
        _insertion_sort(head, 0, len(head))          # std::sort on <=16 elems == insertion sort
        q[: k - 1] = head
        out.append([q[j][1] for j in range(k)])
    return torch.tensor(out, dtype=torch.int64)


def _gt(x, y):
    xn, yn = x[0] != x[0], y[0] != y[0]
    return (xn and not yn) or (x[0] > y[0])


def _insertion_sort(q, first, last):
    if first == last:
        return
    for i in range(first + 1, last):
        val = q[i]
        if _gt(val, q[first]):
            q[first + 1:i + 1] = q[first:i]
            q[first] = val
        else:
            j = i
            while _gt(val, q[j - 1]):
                q[j] = q[j - 1]
                j -= 1
            q[j] = val


def _nth_element(q, nth):
    first, last = 0, len(q)
    if first == last or nth == last:
        return
    depth = 2 * int(math.floor(math.log2(last - first)))
    while last - first > 3:
        if depth == 0:
            _heap_select(q, first, nth + 1, last)
            q[first], q[nth] = q[nth], q[first]
            return
        depth -= 1
        mid = first + (last - first) // 2
        a, b, c = first + 1, mid, last - 1
        # __move_median_to_first(first, a, b, c)
        if _gt(q[a], q[b]):
            if _gt(q[b], q[c]):
                m = b
            elif _gt(q[a], q[c]):
                m = c
            else:
                m = a
        elif _gt(q[a], q[c]):
            m = a
        elif _gt(q[b], q[c]):
            m = c
        else:
            m = b
        q[first], q[m] = q[m], q[first]
        # __unguarded_partition(first+1, last, pivot=first)
        lo, hi = first + 1, last
        while True:
            while _gt(q[lo], q[first]):
                lo += 1
            hi -= 1
            while _gt(q[first], q[hi]):
                hi -= 1
            if not lo < hi:
                break
            q[lo], q[hi] = q[hi], q[lo]
            lo += 1
        cut = lo
        if cut <= nth:
            first = cut
        else:
            last = cut
    _insertion_sort(q, first, last)


def _heap_select(q, first, middle, last):   # libstdc++ __heap_select, comparator _gt
    def sift(start, length, hole, val):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if _gt(q[start + child], q[start + child - 1]):
                child -= 1
            q[start + hole] = q[start + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            q[start + hole] = q[start + child - 1]
            hole = child - 1
        parent = (hole - 1) // 2
        while hole > top and _gt(q[start + parent], val):
            q[start + hole] = q[start + parent]
            hole = parent
            parent = (hole - 1) // 2
        q[start + hole] = val

    n = middle - first
    if n >= 2:
        parent = (n - 2) // 2
        while True:
            sift(first, n, parent, q[first + parent])
            if parent == 0:
                break
            parent -= 1
    for i in range(middle, last):
        if _gt(q[i], q[first]):
            val = q[i]
            q[i] = q[first]
            sift(first, n, 0, val)


def nms_topk(prob, k, eps):
    """DPN.py:120-125.  torch.topk on CPU *is* the reference's arithmetic here."""
    return torch.topk(nms_suppress(prob, eps), k, dim=-1).indices


# --------------------------------------------------------------------------- #
# Fourier feature of a scalar disparity (NMP.py:35-51); op order matters (H2):
# c = coord*normalizer ; f_i = c * 2^i ; [sin f (15) | cos f (15) | c]
# --------------------------------------------------------------------------- #
def fourier_embed(coord, normalizer, n_freq=15):
    c = coord.unsqueeze(-1) * normalizer
    f = c * (2.0 ** torch.arange(n_freq, dtype=coord.dtype))
    return torch.cat((f.sin(), f.cos(), c), -1)


# --------------------------------------------------------------------------- #
# A6  seed embedding (NMP.py:619-649)
# --------------------------------------------------------------------------- #
def sample_cost(cv, seeds):
    """cv [P,G,D], seeds [P,N] int64 -> [P,N,G*9]; taps seed-4..seed+4 clamped,
    output order group-major, tap-minor (NMP.py:628-633)."""
    p, g, d = cv.shape
    n = seeds.shape[1]
    idx = (seeds[:, :, None] + torch.arange(-4, 5)).clamp(0, d - 1)        # [P,N,9]
    gathered = torch.gather(cv[:, None].expand(p, n, g, d), 3, idx[:, :, None, :].expand(p, n, g, 9))
    return gathered.reshape(p, n, g * 9)


def seed_embed(cv, seeds, w):
    pre = "dpn.propagation"
    cost = sample_cost(cv, seeds)
    feat = F.linear(F.gelu(_lin(cost, w, pre + ".cost_encoder.0")), w[pre + ".cost_encoder.2.weight"],
                    w[pre + ".cost_encoder.2.bias"])
    enc = fourier_embed(seeds.to(cv.dtype), 3.14 / 64)
    return F.linear(torch.cat((feat, enc), -1), w[pre + ".proj.weight"])


# --------------------------------------------------------------------------- #
# A7  cross-stripe (CSWin) message passing, SPLIT_SIZE=1 (NMP.py:401-600)
# --------------------------------------------------------------------------- #
def stripe_attention(q, k, v, lepe_w, axis, scale):
    """q,k,v: [B,H,W,N,heads,hd] (one channel half).  axis=0: vertical stripes
    (one per column, tokens (y,n)); axis=1: horizontal (one per row, tokens (x,n)).
    A label attends to itself and to every label of the *other* pixels of the
    stripe, never to its siblings (NMP.py:196-208,495).
    LePE (NMP.py:433-449, H3) for width-1 stripes, per channel c:
      rpe_j(p) = w_c v_j(p) + sum_k [ w_- v_k(p-1) + w_+ v_k(p+1) ]
    with (w_-,w_c,w_+) the centre column (axis 0) / centre row (axis 1) of the
    depthwise 3x3 kernel ``lepe_w`` [heads*hd,1,3,3], zero outside the stripe."""
    b, h, wd, n, heads, hd = q.shape
    if axis == 0:
        perm = lambda t: t.permute(0, 2, 4, 1, 3, 5)      # B,W,heads,H,N,hd
        taps = lepe_w[:, 0, :, 1]
    else:
        perm = lambda t: t.permute(0, 1, 4, 2, 3, 5)      # B,H,heads,W,N,hd
        taps = lepe_w[:, 0, 1, :]
    qs, ks, vs = perm(q), perm(k), perm(v)
    L = qs.shape[3]
    qf = (qs * scale).reshape(*qs.shape[:3], L * n, hd)
    kf = ks.reshape(*ks.shape[:3], L * n, hd)
    vf = vs.reshape(*vs.shape[:3], L * n, hd)
    logits = qf @ kf.transpose(-1, -2)
    pix = torch.arange(L * n) // n
    sib = (pix[:, None] == pix[None, :]) & ~torch.eye(L * n, dtype=torch.bool)
    logits = logits.masked_fill(sib, float("-inf"))
    out = torch.softmax(logits, -1) @ vf
    taps = taps.reshape(heads, hd, 3)                      # [heads,hd,(-,c,+)]
    vsum = vs.sum(4, keepdim=True)                         # sum over labels  [..,L,1,hd]
    zero = torch.zeros_like(vsum[:, :, :, :1])
    prev = torch.cat((zero, vsum[:, :, :, :-1]), 3)
    nxt = torch.cat((vsum[:, :, :, 1:], zero), 3)
    tm, tc, tp = (taps[None, None, :, None, None, :, i] for i in range(3))
    rpe = tc * vs + tm * prev + tp * nxt                   # [B,S,heads,L,N,hd]
    out = out.reshape(*vs.shape) + rpe
    if axis == 0:
        return out.permute(0, 3, 1, 4, 2, 5)               # back to B,H,W,N,heads,hd
    return out.permute(0, 1, 3, 4, 2, 5)


def cswin_layer(x, ctx, w, pre, dims, post=False):
    """One PropagationLayer = CSWinNMP.forward_pre (NMP.py:561-574); post: forward_post (NMP.py:576-591).
    x [T,128]; ctx [B,H,W,64]; dims=(B,H,W,N)."""
    b, h, wd, n = dims
    c = x.shape[-1]
    xn = x if post else _ln(x, w, pre + ".norm1")
    qk_in = torch.cat((xn.view(b, h, wd, n, c), ctx[:, :, :, None, :].expand(b, h, wd, n, ctx.shape[-1])), -1)
    q = _lin(qk_in, w, pre + ".q")
    k = _lin(qk_in, w, pre + ".k")
    v = _lin(xn.view(b, h, wd, n, c), w, pre + ".v")
    half, heads = c // 2, 2
    hd = half // heads
    scale = hd ** -0.5
    outs = []
    for axis in (0, 1):
        sl = slice(axis * half, (axis + 1) * half)
        sp = lambda t: t[..., sl].reshape(b, h, wd, n, heads, hd)
        o = stripe_attention(sp(q), sp(k), sp(v), w[f"{pre}.attns.{axis}.get_v.weight"], axis, scale)
        outs.append(o.reshape(b, h, wd, n, half))
    msg = torch.cat(outs, -1).reshape(-1, c)
    x = x + _lin(msg, w, pre + ".proj")
    if post:
        x = _ln(x, w, pre + ".norm1")
        return _ln(x + _gelu_mlp(x, w, pre + ".mlp"), w, pre + ".norm2")
    return x + _gelu_mlp(_ln(x, w, pre + ".norm2"), w, pre + ".mlp")


def propagation(cv, seeds, ctx, w, cfg, dims, stages=None):
    x = seed_embed(cv, seeds, w).reshape(-1, cfg.embed_dim)
    if stages is not None:
        stages["seed_embed"] = x
    for i in range(cfg.num_prop_layers):
        x = cswin_layer(x, ctx, w, f"dpn.propagation.layers.{i}.nmp", dims, post=not cfg.normalize_before)
        if stages is not None:
            stages[f"prop_layer{i}"] = x
    return _ln(x, w, "dpn.propagation.norm")


# --------------------------------------------------------------------------- #
# A9  warp right features at x - label, correlate, concat (NMP.py:683-741)
# --------------------------------------------------------------------------- #
def warp_row(fmap, disp):
    """fmap [B,C,H,W]; disp [B,H,W,N] -> [B,H,W,N,C].
    Restates F.grid_sample(bilinear, zeros, align_corners=True) on the grid
    built at NMP.py:694-705, *including its float round trip*:
      gx = 2*(x-d)/(W-1) - 1 ; ix = (gx+1)*((W-1)/2)   (same for y with d=0;
      the unnormalise form is ATen's vectorised CPU kernel, GridSamplerKernel.cpp)
    so iy is not exactly y and a ~1e-7 share of the adjacent row leaks in (H6).
    Weights follow ATen's vectorised CPU kernel: w=ix-floor(ix), e=1-w, ..."""
    b, c, h, wd = fmap.shape
    n = disp.shape[-1]
    xs = torch.arange(wd, dtype=fmap.dtype).view(1, 1, wd, 1)
    ys = torch.arange(h, dtype=fmap.dtype).view(1, h, 1, 1).expand(1, h, wd, 1)
    gx = 2 * (xs + (-disp)) / (wd - 1) - 1
    gy = (2 * (ys + torch.zeros_like(disp)) / (h - 1) - 1)
    ix = (gx + 1) * ((wd - 1) / 2)
    iy = (gy + 1) * ((h - 1) / 2)
    x0, y0 = ix.floor(), iy.floor()
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = 1 - wx1, 1 - wy1
    src = fmap.permute(0, 2, 3, 1).reshape(b, h * wd, c)
    out = fmap.new_zeros(b, h, wd, n, c)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy <= h - 1) & (xx >= 0) & (xx <= wd - 1)
            lin = (yy.clamp(0, h - 1) * wd + xx.clamp(0, wd - 1)).long().reshape(b, -1)
            tap = torch.gather(src, 1, lin[:, :, None].expand(-1, -1, c)).view(b, h, wd, n, c)
            out = out + tap * (wy * wx * ok).unsqueeze(-1)
    return out


def warp_corr_concat(labels, f1, f2, g1, g2, groups=32):
    """-> [B*H*W*N, 64+64+32] = [left | warped right | group corr] (NMP.py:735-741).
    corr group g = mean over channels 8g..8g+7 of g1*warp(g2) (NMP.py:716-719)."""
    b, c, h, wd = f1.shape
    n = labels.shape[-1]
    disp = labels.view(b, h, wd, n)
    wg = warp_row(g2, disp)                                            # [B,H,W,N,256]
    left_g = g1.permute(0, 2, 3, 1)[:, :, :, None, :]
    corr = (left_g * wg).view(b, h, wd, n, groups, -1).mean(-1)
    wf = warp_row(f2, disp)
    left = f1.permute(0, 2, 3, 1)[:, :, :, None, :].expand(b, h, wd, n, c)
    return torch.cat((left, wf, corr), -1).reshape(-1, 2 * c + groups)


# --------------------------------------------------------------------------- #
# A10  per-pixel self-edge attention (NMP.py:90-108)
# --------------------------------------------------------------------------- #
def self_attention_layer(x, enc, w, pre, n, heads, post=False):
    """BasicAttention.forward_pre (NMP.py:90-108); post: forward_post (NMP.py:110-128)."""
    t, c = x.shape
    xn = x if post else _ln(x, w, pre + ".norm1")
    qk_in = torch.cat((xn, enc), -1)
    hd = c // heads
    q = _lin(qk_in, w, pre + ".q").view(t // n, n, heads, hd).transpose(1, 2)
    k = _lin(qk_in, w, pre + ".k").view(t // n, n, heads, hd).transpose(1, 2)
    v = _lin(xn, w, pre + ".v").view(t // n, n, heads, hd).transpose(1, 2)
    attn = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), -1)
    out = (attn @ v).transpose(1, 2).reshape(t, c)
    x = x + _lin(out, w, pre + ".proj")
    return _ln(x, w, pre + ".norm1") if post else x


# --------------------------------------------------------------------------- #
# A10/A13  (shifted-)window message passing (NMP.py:142-394)
# --------------------------------------------------------------------------- #
def window_attention(qkv, table, dims, win, shift, heads, sibling_mask):
    """qkv [B,Hp,Wp,N,3C] (q|k|v chunks of C).  Scalar definition (H5):
    tokens of one window are (a,b,n) on the *rolled* grid (roll by -shift);
    for query i=(a,b,n), key j=(a',b',n'):
      rel   = (a-a'+win-1)*(2win-1) + (b-b'+win-1)
      eq,ek,ev = table[rel].view(heads,3*hd)[head].split(hd)
      logit = s*q_i.k_j + s*q_i.ek + s*k_j.eq          s = hd^-0.5
      masked (-inf) if sibling_mask and same pixel and n != n'
      masked (-inf) if shift>0 and region(i) != region(j)   (Swin regions on the rolled grid)
      out_i = sum_j softmax_j(logit) * (v_j + ev)
    """
    b, hp, wp, n, c3 = qkv.shape
    c = c3 // 3
    hd = c // heads
    s = hd ** -0.5
    if shift:
        qkv = torch.roll(qkv, (-shift, -shift), (1, 2))
    nh, nw = hp // win, wp // win
    t = win * win * n
    x = qkv.view(b, nh, win, nw, win, n, 3, heads, hd).permute(6, 0, 1, 3, 7, 2, 4, 5, 8)
    x = x.reshape(3, b, nh, nw, heads, t, hd)
    q, k, v = x[0] * s, x[1], x[2]
    pa = torch.arange(win).view(win, 1, 1).expand(win, win, n).reshape(-1)
    pb = torch.arange(win).view(1, win, 1).expand(win, win, n).reshape(-1)
    rel = (pa[:, None] - pa[None, :] + win - 1) * (2 * win - 1) + (pb[:, None] - pb[None, :] + win - 1)
    emb = table[rel.reshape(-1)].view(t, t, heads, 3, hd).permute(2, 3, 0, 1, 4)   # [heads,3,i,j,hd]
    eq, ek, ev = emb[:, 0] * s, emb[:, 1], emb[:, 2]
    logits = q @ k.transpose(-1, -2)
    logits = logits + torch.einsum("...hic,hijc->...hij", q, ek) + torch.einsum("...hjc,hijc->...hij", k, eq)
    pix = torch.arange(t) // n
    if sibling_mask:
        sib = (pix[:, None] == pix[None, :]) & ~torch.eye(t, dtype=torch.bool)
        logits = logits.masked_fill(sib, float("-inf"))
    if shift:
        def region(length):
            r = torch.zeros(length, dtype=torch.long)
            r[length - win:length - shift] = 1
            r[length - shift:] = 2
            return r
        reg = region(hp)[:, None] * 3 + region(wp)[None, :]                    # [Hp,Wp] on rolled grid
        reg = reg.view(nh, win, nw, win).permute(0, 2, 1, 3).reshape(nh, nw, win * win)
        reg = reg[:, :, pix]                                                    # [nh,nw,t]
        diff = reg[:, :, :, None] != reg[:, :, None, :]
        logits = logits.masked_fill(diff[None, :, :, None], float("-inf"))
    p = torch.softmax(logits, -1)
    out = p @ v + torch.einsum("...hij,hijc->...hic", p, ev)
    out = out.view(b, nh, nw, heads, win, win, n, hd).permute(0, 1, 4, 2, 5, 6, 3, 7).reshape(b, hp, wp, n, c)
    if shift:
        out = torch.roll(out, (shift, shift), (1, 2))
    return out


def swin_layer(x, enc, w, pre, dims, win, shift, heads, sibling_mask, post=False):
    """SwinNMP.forward_pre (NMP.py:350-364); post: forward_post (NMP.py:366-382)."""
    b, hp, wp, n = dims
    c = x.shape[-1]
    qkv = _lin(torch.cat((x if post else _ln(x, w, pre + ".norm1"), enc), -1), w, pre + ".qkv")
    msg = window_attention(qkv.view(b, hp, wp, n, 3 * c), w[pre + ".attn.relative_position_enc_table"],
                           dims, win, shift, heads, sibling_mask)
    x = x + _lin(msg.reshape(-1, c), w, pre + ".proj")
    if post:
        x = _ln(x, w, pre + ".norm1")
        return _ln(x + _gelu_mlp(x, w, pre + ".mlp"), w, pre + ".norm2")
    return x + _gelu_mlp(_ln(x, w, pre + ".norm2"), w, pre + ".mlp")


def _pad_tokens(x, dims, win):
    """Zero-pad the (H,W) token grid to a multiple of win, top=pad//2 (NMP.py:745-762)."""
    b, h, wd, n = dims
    ph, pw = (-h) % win, (-wd) % win
    top, left = ph // 2, pw // 2
    xg = x.view(b, h, wd, n, -1)
    xg = F.pad(xg, (0, 0, 0, 0, left, pw - left, top, ph - top))
    return xg.reshape(-1, x.shape[-1]), (b, h + ph, wd + pw, n), (top, left)


def _crop_tokens(x, pdims, dims, off):
    b, hp, wp, n = pdims
    _, h, wd, _ = dims
    return x.view(b, hp, wp, n, -1)[:, off[0]:off[0] + h, off[1]:off[1] + wd].reshape(-1, x.shape[-1])


def inference(labels, f1, f2, g1, g2, w, cfg, stages=None, intermediate=None):
    """Inference.forward (NMP.py:722-798). labels [B*H*W,N] in 1/8-px units.
    intermediate: a list that receives norm(crop(x)) after EVERY layer (return_intermediate in training mode, NMP.py:777-796;
    the last entry is the returned tensor itself)."""
    b, _, h, wd = f1.shape
    n = labels.shape[-1]
    dims = (b, h, wd, n)
    x = _gelu_mlp(warp_corr_concat(labels, f1, f2, g1, g2), w, "inference.ffn")
    enc = fourier_embed(labels.reshape(-1), 3.14 / 64)
    if stages is not None:
        stages["infer_ffn"] = x
    x, pdims, off = _pad_tokens(x, dims, cfg.window_size)
    enc, _, _ = _pad_tokens(enc, dims, cfg.window_size)
    for i in range(cfg.num_infer_layers):
        pre = f"inference.layers.{i}"
        x = self_attention_layer(x, enc, w, pre + ".self_nmp", n, cfg.infer_heads, post=not cfg.normalize_before)
        shift = 0 if i % 2 == 0 else cfg.window_size // 2
        x = swin_layer(x, enc, w, pre + ".nmp", pdims, cfg.window_size, shift, cfg.infer_heads, True, post=not cfg.normalize_before)
        if stages is not None:
            stages[f"infer_layer{i}"] = x
        if intermediate is not None:
            intermediate.append(_ln(_crop_tokens(x, pdims, dims, off), w, "inference.norm"))
    return _ln(_crop_tokens(x, pdims, dims, off), w, "inference.norm")


def refinement(disp_q, f1, f2, g1, g2, w, cfg, stages=None, intermediate=None):
    """Refinement.forward (NMP.py:828-900). disp_q [B,H4,W4] in 1/4-px units, N=1.  intermediate: as in inference (NMP.py:879-898)."""
    b, _, h, wd = f1.shape
    dims = (b, h, wd, 1)
    labels = disp_q.reshape(-1, 1)
    x = _gelu_mlp(warp_corr_concat(labels, f1, f2, g1, g2), w, "refinement.ffn")
    enc = fourier_embed(labels.reshape(-1), 3.14 / 128)
    if stages is not None:
        stages["refine_ffn"] = x
    win = cfg.refine_window_size
    x, pdims, off = _pad_tokens(x, dims, win)
    enc, _, _ = _pad_tokens(enc, dims, win)
    for i in range(cfg.num_refine_layers):
        shift = 0 if i % 2 == 0 else win // 2
        x = swin_layer(x, enc, w, f"refinement.layers.{i}.nmp", pdims, win, shift, cfg.infer_heads, False, post=not cfg.normalize_before)
        if stages is not None:
            stages[f"refine_layer{i}"] = x
        if intermediate is not None:
            intermediate.append(_ln(_crop_tokens(x, pdims, dims, off), w, "refinement.norm"))
    return _ln(_crop_tokens(x, pdims, dims, off), w, "refinement.norm")


# --------------------------------------------------------------------------- #
# A11/A12  heads, WTA + 4x4 lower median (NMRF.py:218-232)
# --------------------------------------------------------------------------- #
def coarse_heads(tgt, labels, w, dims):
    b, h, wd, n = dims
    delta = _relu_mlp(tgt, w, "infer_head")                               # [T,64]
    coarse = F.relu(labels.reshape(-1, 1) + delta)
    score = 0.25 * _lin(tgt, w, "infer_score_head")
    unshuffle = lambda t: t.view(b, h, wd, n, 8, 8).permute(0, 1, 4, 2, 5, 3).reshape(b, h * 8, wd * 8, n)
    return unshuffle(coarse), unshuffle(score)


def wta_median(coarse, score):
    """first-max over N (torch.max on CPU), x2, 4x4 blocks -> lower median (8th of 16)."""
    idx = score.max(-1, keepdim=True).indices
    d = torch.gather(coarse, -1, idx)[..., 0] * 2
    b, h, wd = d.shape
    blocks = d.view(b, h // 4, 4, wd // 4, 4).permute(0, 1, 3, 2, 4).reshape(b, h // 4, wd // 4, 16)
    return blocks.sort(-1).values[..., 7]


# --------------------------------------------------------------------------- #
# A14  refine head + pixel shuffle + scale + unpad (NMRF.py:238-251)
# --------------------------------------------------------------------------- #
def refine_epilogue(tgt, disp_q, w, pad_hw, out_hw):
    b, h, wd = disp_q.shape
    delta = _relu_mlp(tgt, w, "refine_head").view(b, h, wd, 4, 4)
    pred = F.relu(disp_q[..., None, None] + delta).permute(0, 1, 3, 2, 4).reshape(b, h * 4, wd * 4)
    disp = (pred * 4)[:, : out_hw[0], : out_hw[1]]
    return disp, pred


# --------------------------------------------------------------------------- #
# A15  multi-scale deformable attention core (ops/functions/ms_deform_attn_func.py:49-71,
#      ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299)
# --------------------------------------------------------------------------- #
def msda_core(value, shapes, loc, wgt):
    """value [B,S,M,D]; shapes list[(H,W)]; loc [B,Lq,M,L,P,2] in [0,1] (x,y);
    wgt [B,Lq,M,L,P] -> [B,Lq,M*D].  Sample position h_im = loc_y*H - 0.5
    (align_corners=False), 4-tap bilinear, zero outside."""
    bsz, s, m, dch = value.shape
    _, lq, _, nl, npnt, _ = loc.shape
    out = value.new_zeros(bsz, lq, m, dch)
    start = 0
    for li, (hh, ww) in enumerate(shapes):
        val = value[:, start:start + hh * ww]                              # [B,HW,M,D]
        start += hh * ww
        x = loc[:, :, :, li, :, 0] * ww - 0.5
        y = loc[:, :, :, li, :, 1] * hh - 0.5
        x0, y0 = x.floor(), y.floor()
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                lw, lh = x - x0, y - y0                      # cuh:48-50: hh=1-lh, hw=1-lw
                wt = (lh if dy else 1 - lh) * (lw if dx else 1 - lw)
                ok = (xx >= 0) & (xx <= ww - 1) & (yy >= 0) & (yy <= hh - 1)
                lin = (yy.clamp(0, hh - 1) * ww + xx.clamp(0, ww - 1)).long()  # [B,Lq,M,P]
                g = torch.gather(val.permute(0, 2, 1, 3), 2,
                                 lin.permute(0, 2, 1, 3).reshape(bsz, m, lq * npnt, 1).expand(-1, -1, -1, dch))
                g = g.view(bsz, m, lq, npnt, dch).permute(0, 2, 1, 3, 4)
                out = out + (g * (wt * ok * wgt[:, :, :, li])[..., None]).sum(3)
    return out.reshape(bsz, lq, m * dch)


# --------------------------------------------------------------------------- #
# full forward (NMRF.py:189-262), CNN backbone
# --------------------------------------------------------------------------- #
def hot_path(w, cfg, feats8, feats4, pad_hw, out_hw, stages=None, seeds=None, aux=False):
    """Everything after the backbone (NMRF.py:207-262).  feats8 / feats4: [2B,C,H,W] maps, left views first.
    aux: also return `aux_outputs` as the reference does in training mode with SOLVER.AUX_LOSS and NMP.RETURN_INTERMEDIATE
    (NMRF.py:216-223, 240-244, 259-273): one {disp_pred: coarse [B,8H,8W,N], logits_pred} per inference layer from that layer's
    normalised tokens through the SAME heads, then one {disp_pred [B,4H4,4W4]} per refinement layer but the last.
    seeds: [P,N] int64 label seeds to continue from instead of the oracle's own NMS + top-k (test infrastructure: at a pixel
    whose candidates tie within the fp32 noise of `prob`, the checker continues from the candidate's choice -- tests/util.py
    seeds_explained_by_prob_noise -- so that everything downstream is compared on identical seeds)."""
    b = feats8.shape[0] // 2
    l8, r8 = feats8[:b], feats8[b:]
    l4, r4 = feats4[:b], feats4[b:]
    num_disp = cfg.max_disp // 8
    _, _, h8, w8 = l8.shape
    n = cfg.num_proposals
    dims8 = (b, h8, w8, n)

    cv = cost_volume(l8, r8, num_disp, cfg.cost_group)
    prob = dpn_filter_softmax(cv, w)
    seeds = nms_topk(prob, n, cfg.eps) if seeds is None else seeds.long().reshape(-1, n)
    ctx = conv_head(l8, w, "dpn.proj").permute(0, 2, 3, 1)
    mem = propagation(cv, seeds, ctx, w, cfg, dims8, stages)
    labels = F.relu(_relu_mlp(mem, w, "dpn.prop_head").view(-1, n) + seeds.to(mem.dtype))

    f1, f2 = conv_head(l8, w, "concatconv"), conv_head(r8, w, "concatconv")
    g1, g2 = conv_head(l8, w, "gw"), conv_head(r8, w, "gw")
    inter8 = [] if aux else None
    proposal = labels                       # the returned proposals keep their graph (the proposal loss); the stages below see constants:
    labels = labels.detach()                # `labels_curr = labels[-1].detach()` (NMRF.py:215)
    tgt = inference(labels, f1, f2, g1, g2, w, cfg, stages, inter8)
    coarse, score = coarse_heads(tgt, labels, w, dims8)
    disp_q = wta_median(coarse, score).detach()                        # `disp_curr = disp_curr.detach()` (NMRF.py:232)

    f1, f2 = conv_head(l4, w, "concatconv"), conv_head(r4, w, "concatconv")
    g1, g2 = conv_head(l4, w, "gw"), conv_head(r4, w, "gw")
    inter4 = [] if aux else None
    tgt4 = refinement(disp_q, f1, f2, g1, g2, w, cfg, stages, inter4)
    disp, pred = refine_epilogue(tgt4, disp_q, w, pad_hw, out_hw)
    aux_outputs = None
    if aux:
        aux_outputs = []
        for t in inter8:                                                   # every inference layer, the last included
            c_i, s_i = coarse_heads(t, labels, w, dims8)
            aux_outputs.append({"disp_pred": c_i, "logits_pred": s_i})
        for t in inter4[:-1]:                                              # refinement layers but the last (NMRF.py:271-272)
            aux_outputs.append({"disp_pred": refine_epilogue(t, disp_q, w, pad_hw, out_hw)[1]})

    out = {
        "proposal": proposal.view(b, -1, n),
        "prob": prob,
        "initial_proposal": seeds.to(mem.dtype).view(b, -1, n),
        "disp": disp,
        "disp_pred": pred,
    }
    if aux_outputs is not None:
        out["aux_outputs"] = aux_outputs
    if stages is not None:
        stages.update(cost_volume=cv, seeds=seeds, context=ctx, prop_memory=mem, infer_tgt=tgt,
                      coarse=coarse, score=score, disp_curr=disp_q, refine_tgt=tgt4,
                      fmap8_l=l8, fmap8_r=r8, fmap4_l=l4, fmap4_r=r4)
        out["stages"] = stages
    return out


def refine_from(w, cfg, disp_q, l4, r4, out_hw):
    """The path from the winner-take-all result on (NMRF.py:233-251): refinement + epilogue from a GIVEN `disp_q`
    [B,H4,W4] (1/4-px units) and the 1/4-res encoder maps.  Used by the parity chain of tests/util.py to compare the GPU
    refinement with the oracle's on identical discrete decisions."""
    f1, f2 = conv_head(l4, w, "concatconv"), conv_head(r4, w, "concatconv")
    g1, g2 = conv_head(l4, w, "gw"), conv_head(r4, w, "gw")
    tgt4 = refinement(disp_q, f1, f2, g1, g2, w, cfg)
    return refine_epilogue(tgt4, disp_q, w, None, out_hw)


def forward(w, cfg, img1, img2, return_stages=False, training=False):
    """training: the reference's model.train() forward (NMRF.py:203-205, 250-251, 259-260): no input padding (the crop size is
    assumed adequate: H, W multiples of divis_by), no un-padding, aux_outputs returned.  No dropout / batch statistics exist in
    the CNN configuration, so everything else is the eval arithmetic."""
    stages = {} if return_stages else None
    h0, w0 = img1.shape[-2:]
    if training:
        assert h0 % cfg.divis_by == 0 and w0 % cfg.divis_by == 0, "training mode does not pad (NMRF.py:203-205)"
        img1, img2, pad_hw = img1.float(), img2.float(), (0, 0)
    else:
        img1, img2, pad_hw = pad_images(img1, img2, cfg.divis_by)
    feats4, feats8 = cnn_backbone(torch.cat((img1, img2), 0), w, cfg.backbone_prefix)
    return hot_path(w, cfg, feats8, feats4, pad_hw, (h0, w0), stages, aux=training)
