"""Neural message passing blocks of NMRF on the HIP kernels.

The module tree and parameter names mirror nmrf/models/NMP.py so that reference checkpoints load with strict=True.  The arithmetic
does not: on token-major fp32 buffers [T, C] (T = B*H*W*N) every block is
    [HIP] attention kernel  ->  [HIP] nmp_block16: proj + residual + LayerNorm + fc1 + GELU + fc2 + residual + the NEXT block's
                                 LayerNorm | side columns -> q|k|v                                     (ONE launch, split-fp16 MFMA)
with the weights of a launch packed into one MFMA-fragment stream per parameter version.
NMRF_LINEAR=fp32 (A/B parity runs, tests) switches the per-token linears to the round-1 chain of fp32-MFMA token_linear kernels
(+ hipBLASLt fc2) of the tools / test library libnmrf_hip_debug.so; the product library does not contain them.
There is no CPU path: the kernels raise on non-CUDA tensors.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K


def _split():
    """Default: the per-token linears of a block run in ONE kernel on split-operand fp16 MFMA (csrc/nmp_block16.hip, fp32-grade
    products).  NMRF_LINEAR=fp32: the reference chain of fp32-MFMA token_linear kernels + hipBLASLt fc2 (debug library)."""
    return os.environ.get("NMRF_LINEAR", "split") != "fp32"

FOURIER_DIM = 31        # 15 sin + 15 cos + the scaled coordinate (NMP.py:35-51)


class MLP(nn.Module):
    """Linear-ReLU stack (NMP.py:54-66); params `layers.{i}`."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.num_layers = num_layers
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x, row_add=None, relu_out=False):
        """row_add [T, n_out] / relu_out (HIP chain only): [relu](MLP(x) + row_add) in the chain kernel's store pass."""
        if (_split() and x.is_cuda and self.num_layers == 3 and self.layers[0].in_features == 128
                and self.layers[0].out_features == 128 and self.layers[1].out_features == 128 and self.layers[2].out_features <= 64):
            if not hasattr(self, "_chain"):
                self._chain = _ChainLauncher(2, self.layers, (128, 128, 128), self.layers[2].out_features)
            shp = x.shape
            return self._chain(x.reshape(-1, 128).contiguous(), 128, row_add=row_add, relu_out=relu_out).view(
                *shp[:-1], self.layers[2].out_features)
        if row_add is not None or relu_out:
            y = self.forward(x)
            y = y if row_add is None else y + row_add.view_as(y)
            return F.relu(y) if relu_out else y
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            if (x.is_cuda and layer.out_features % 32 == 0 and layer.out_features > 64
                    and layer.in_features in (32, 64, 128, 160)):
                if not hasattr(self, "_lin"):
                    self._lin = {}
                lin = self._lin.setdefault(i, _Lin(layer))
                shp = x.shape
                y = lin(x.reshape(-1, shp[-1]).contiguous(), act=0 if last else "relu")
                x = y.view(*shp[:-1], layer.out_features)
            elif x.is_cuda and layer.out_features <= 64 and layer.in_features % 4 == 0 and layer.in_features <= 128:
                shp = x.shape
                y = K.linear_smalln(x.reshape(-1, shp[-1]).contiguous(), layer.weight, layer.bias, relu=not last)
                x = y.view(*shp[:-1], layer.out_features)
            else:
                # no silent stock-torch tail: a shape no HIP kernel covers is an unsupported configuration, like every other one here
                raise NotImplementedError("MLP layer %d (%d -> %d) on %s: no HIP kernel covers this shape (token_linear: in 32/64/128/160, "
                                          "out a multiple of 32 above 64; linear_smalln: out <= 64, in <= 128 and a multiple of 4)"
                                          % (i, layer.in_features, layer.out_features, x.device))
        return x


class Mlp(nn.Module):
    """fc1 - GELU(erf) - fc2 (timm's Mlp as used at NMP.py:337,537,675); params fc1, fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and self.fc1.in_features in (64, 128, 160) and self.fc1.out_features % 32 == 0:
            if not hasattr(self, "_lin1"):
                self._lin1 = _Lin(self.fc1)
            return self.fc2(self._lin1(x.contiguous(), act="gelu"))
        raise NotImplementedError("Mlp (%d -> %d -> %d) on %s: the unfused form exists for the NMRF_LINEAR=fp32 A/B chain only (2-D CUDA rows, "
                                  "fc1 in 64/128/160, hidden a multiple of 32); the product runs it inside nmp_block16 / mlp_chain"
                                  % (self.fc1.in_features, self.fc1.out_features, self.fc2.out_features, x.device))

    def forward_ln(self, x, y, norm):
        """fc2(GELU(fc1(LayerNorm(x + y)))) with add, norm, fc1 and GELU in one kernel -> (x + y, out)."""
        if not hasattr(self, "_lin1"):
            self._lin1 = _Lin(self.fc1)
        pw, b, k, n = _packed(self._lin1.cache, (self.fc1.weight,), (self.fc1.bias,))
        r = K.token_linear(x, pw, n, k, b, ln=(norm.weight, norm.bias, norm.eps), y=y, act="gelu")
        x, h = r if y is not None else (x, r)
        return x, self.fc2(h)


class _FusedCache:
    """Caches tensors derived from parameters, keyed on (data_ptr, version) of the sources."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, params, builder):
        key = tuple((p.data_ptr(), p._version, p.device) for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = builder()
            self._key = key
        return self._val


def _pad_cols(w, k):
    return w if w.shape[1] == k else F.pad(w, (0, k - w.shape[1]))


def _packed(cache, weights, biases):
    """(packed q|k|.. weight in MFMA fragment order, concatenated bias or None, K, N) for K.token_linear."""
    def build():
        k = max(w.shape[1] for w in weights)
        w = torch.cat([_pad_cols(w, k) for w in weights], 0).contiguous()
        b = None if biases[0] is None else torch.cat(list(biases)).contiguous()
        return K.pack_linear_weight(w), b, k, w.shape[0]
    return cache.get(tuple(weights) + tuple(b for b in biases if b is not None), build)


class _Lin:
    """token_linear front end of one nn.Linear (packed weight cached per parameter version)."""

    def __init__(self, linear):
        self.linear = linear
        self.cache = _FusedCache()

    def __call__(self, x, act=0, **kw):
        pw, b, k, n = _packed(self.cache, (self.linear.weight,), (self.linear.bias,))
        return K.token_linear(x, pw, n, k, b, act=act, **kw)


class _BlockLauncher:
    """One nmp_block launch site of a stage: x1 = x + proj(msg); x2 = x1 + mlp(norm2(x1)); then the NEXT block's q|k|v on
    [norm(x2) | side] or the stage's final norm.  The weight stream and the fused bias are rebuilt when a parameter changes."""

    def __init__(self, proj=None, mlp=None, nxt_norm=None, nxt_linears=(), kq=0, ln_out=False, kv16=False):
        self.proj, self.mlp, self.nxt_norm, self.nxt_linears, self.kq, self.ln_out = proj, mlp, nxt_norm, tuple(nxt_linears), kq, ln_out
        self.kv16 = kv16        # the q|k|v it produces goes to an attention kernel that takes k | v as split fp16 pairs (include/nmrf_hip.h)
        self.cache = _FusedCache()

    def _params(self):
        ps = []
        if self.proj is not None:
            ps += [self.proj.weight, self.proj.bias]
        if self.mlp is not None:
            ps += [self.mlp[1].fc1.weight, self.mlp[1].fc2.weight]
        for l in self.nxt_linears:
            ps += [l.weight, l.bias]
        return tuple(ps)

    def _build(self):
        wq = bq = wq_amax = None
        if self.nxt_linears:
            k = max(l.in_features for l in self.nxt_linears)
            wq = torch.cat([_pad_cols(l.weight, k) for l in self.nxt_linears], 0).contiguous()
            bq = torch.cat([l.bias for l in self.nxt_linears]).contiguous()
            am = [K.cached_amax(l.weight) for l in self.nxt_linears]           # (training: prefetched by train_step; zero padding adds nothing)
            wq_amax = None if any(a is None for a in am) else max(am)
        stream, stages, inv = K.block_stream16(None if self.proj is None else self.proj.weight.contiguous(),
                                               None if self.mlp is None else self.mlp[1].fc1.weight.contiguous(),
                                               None if self.mlp is None else self.mlp[1].fc2.weight.contiguous(), wq, self.kq, wq_amax)
        return stream, stages, inv, bq, (0 if wq is None else wq.shape[0])

    def __call__(self, x, msg=None, extra=None, extra_div=1, want_x=True, ln_out=None, ln_out_map=None, attn_qkv=None):
        stream, stages, inv, bq, nq = self.cache.get(self._params(), self._build)
        mlp = None
        if self.mlp is not None:
            n2, m = self.mlp
            mlp = (n2.weight, n2.bias, n2.eps, m.fc1.bias, m.fc2.bias)
        q = None
        if self.nxt_norm is not None:
            q = dict(g=self.nxt_norm.weight, b=self.nxt_norm.bias, eps=self.nxt_norm.eps, extra=extra if self.kq > 128 else None,
                     extra_div=extra_div, bias=bq, kq=self.kq, nq=nq, ln_out=self.ln_out, kv16=self.kv16 and nq == 384)
        return K.nmp_block(x, stream, stages, inv, msg, None if self.proj is None else self.proj.bias, mlp, q, want_x=want_x,
                           ln_out=ln_out, ln_out_map=ln_out_map, attn_qkv=attn_qkv)


class _ChainLauncher:
    """One mlp_chain launch site (csrc/mlp_chain.hip): weights packed per parameter version."""

    def __init__(self, kind, linears, kps, n_out):
        self.kind, self.linears, self.kps, self.n_out = kind, tuple(linears), tuple(kps), n_out
        self.cache = _FusedCache()

    def __call__(self, x, k1, extra=None, out=None, out_map=None, row_add=None, relu_out=False):
        ps = tuple(l.weight for l in self.linears) + tuple(l.bias for l in self.linears if l.bias is not None)
        stream, stages, inv = self.cache.get(ps, lambda: K.chain_stream([l.weight for l in self.linears], self.kps))
        return K.mlp_chain(self.kind, x, k1, stream, stages, inv, [l.bias for l in self.linears], self.n_out, extra, out, out_map,
                           row_add=row_add, relu_out=relu_out)


class _PairLauncher:
    """A full block's launch site and the self-edge block's behind it as ONE launch (nmrf_nmp_block16_pair_f32): the two weight
    streams back to back, rebuilt when a parameter of either changes."""

    def __init__(self, first, second):
        self.a, self.b = first, second
        self.cache = _FusedCache()

    def _build(self):
        import ctypes
        sa, na, ia, bqa, nqa = self.a._build()
        sb, nb, ib, bqb, nqb = self.b._build()
        inv = (ctypes.c_float * 6)(ia[0], ia[1], ia[2], ia[3], ib[0], ib[3])
        return torch.cat((sa, sb)).contiguous(), na + nb, inv, bqa, bqb, nqb

    @staticmethod
    def ok(first, second):
        # (first.proj is None and first.mlp is None: the launch that opens the stage -- a q stage alone)
        return ((first.proj is None) == (first.mlp is None) and first.kq == 160 and len(first.nxt_linears) == 3
                and all(l.out_features == 128 for l in first.nxt_linears) and second.proj is not None and second.mlp is None
                and second.kq == 160 and second.nxt_norm is not None and not second.ln_out)

    def __call__(self, x, msg, extra, extra_div=1, want_x=True):
        a, b = self.a, self.b
        stream, stages, inv, bqa, bqb, nqb = self.cache.get(a._params() + b._params(), self._build)
        mlp = None
        if a.mlp is not None:
            n2, m = a.mlp
            mlp = (n2.weight, n2.bias, n2.eps, m.fc1.bias, m.fc2.bias)
        q = dict(g=a.nxt_norm.weight, b=a.nxt_norm.bias, eps=a.nxt_norm.eps, extra=extra, extra_div=extra_div, bias=bqa)
        q2 = dict(g=b.nxt_norm.weight, b=b.nxt_norm.bias, eps=b.nxt_norm.eps, extra=extra, extra_div=extra_div, bias=bqb, nq=nqb,
                  kv16=b.kv16 and nqb == 384)
        return K.nmp_block_pair(x, msg, stream, stages, inv, None if a.proj is None else a.proj.bias, mlp, q, b.proj.bias, q2,
                                want_x=want_x)


def _pad_maps(dims, win, device, cache):
    """int32 row maps between the dense token grid (b, h, w, n) and the grid zero-padded to a multiple of `win` (top = pad // 2,
    NMP.py:745-762): (pdims, to_padded [T], to_dense [Tp] with -1 at pad tokens), or (dims, None, None) when nothing is padded."""
    b, h, wd, n = dims
    ph, pw = (-h) % win, (-wd) % win
    if ph == 0 and pw == 0:
        return dims, None, None
    key = (dims, win, str(device))
    if key not in cache:
        hp, wp = h + ph, wd + pw
        top, left = ph // 2, pw // 2
        bi, yi, xi, ni = torch.meshgrid(torch.arange(b), torch.arange(h), torch.arange(wd), torch.arange(n), indexing="ij")
        to_p = (((bi * hp + yi + top) * wp + xi + left) * n + ni).reshape(-1).to(torch.int32)
        to_d = torch.full((b * hp * wp * n,), -1, dtype=torch.int32)
        to_d[to_p.long()] = torch.arange(to_p.numel(), dtype=torch.int32)
        cache[key] = ((b, hp, wp, n), to_p.to(device), to_d.to(device))
    return cache[key]


def _block_ok(*mods):
    """The fused kernel is built for the dimensions of every shipped config: 128-wide tokens, 512-wide hidden layer."""
    for m in mods:
        if isinstance(m, Mlp) and (m.fc1.in_features != 128 or m.fc1.out_features != 512 or m.fc2.out_features != 128):
            return False
        if isinstance(m, nn.Linear) and m.out_features != 128:
            return False
    return True


def _ln(x, norm):
    return K.ln_concat(x, norm.weight, norm.bias, None, 1, x.shape[1], norm.eps)


def _add_ln(x, y, norm, extra=None, extra_div=1, ld=None):
    """The residual stream travels as a pair (x, y) meaning x + y; the pending add is folded into the next
    LayerNorm kernel.  Returns (materialised x + y, [LN(x + y) | extra | 0])."""
    ld = ld or x.shape[1]
    if y is None:
        return x, K.ln_concat(x, norm.weight, norm.bias, extra, extra_div, ld, norm.eps)
    return K.add_ln_concat(x, y, norm.weight, norm.bias, extra, extra_div, ld, norm.eps)


# --------------------------------------------------------------------------------------------------
# NMP.NORMALIZE_BEFORE False: the forward_post form of the three block types (NMP.py:110-135, 366-382, 576-591).  No shipped config
# sets it, so it is not fused: every linear is one launch of the split-fp16 GEMM (nmrf_gemm_split_f32) + its bias / GELU pass, every
# LayerNorm one nmrf_layernorm_f32, around the same attention kernels on fp32 q | k | v rows.
# --------------------------------------------------------------------------------------------------
def _rows_linear(x, lin, act=0):
    """act(x W^T + b) of an nn.Linear on token rows [T,K]; act 0 identity, 2 GELU(erf)."""
    y = K.linear_forward(x if x.is_contiguous() else x.contiguous(), lin.weight)
    if lin.bias is None and not act:
        return y
    return K.bias_act(y, lin.bias, act, want_pre=not act)[1]


def _post_norm_tail(m, x, msg):
    """x = norm1(x + proj(msg)) and, for a block with an MLP, x = norm2(x + mlp(x))  (NMP.py:124-126, 374-380, 586-590)."""
    x = K.layer_norm(x + _rows_linear(msg, m.proj), m.norm1.weight, m.norm1.bias, m.norm1.eps)
    if hasattr(m, "mlp"):
        h = _rows_linear(x, m.mlp.fc1, act=2)
        x = K.layer_norm(x + _rows_linear(h, m.mlp.fc2), m.norm2.weight, m.norm2.bias, m.norm2.eps)
    return x


# --------------------------------------------------------------------------------------------------
# self edges (BasicAttention, NMP.py:70-139)
# --------------------------------------------------------------------------------------------------
class BasicAttention(nn.Module):
    def __init__(self, dim, qk_dim, num_heads=8, normalize_before=True):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.num_heads = num_heads
        self.norm1 = nn.LayerNorm(dim)
        self.q, self.k, self.v = nn.Linear(qk_dim, dim), nn.Linear(qk_dim, dim), nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)
        self._packed_qkv = _FusedCache()
        self._proj = _Lin(self.proj)

    def forward_pair(self, x, y, abs_encoding, n):
        """(x, y) = residual stream x + y; returns the next pair.  (The fp32-MFMA reference chain; the default path runs the
        whole stage through Inference._run_blocks.)"""
        pw, b, k, nn_ = _packed(self._packed_qkv, (self.q.weight, self.k.weight, self.v.weight),
                                (self.q.bias, self.k.bias, self.v.bias))
        r = K.token_linear(x, pw, nn_, k, b, ln=(self.norm1.weight, self.norm1.bias, self.norm1.eps), y=y,
                           extra=abs_encoding)
        x, qkv = r if y is not None else (x, r)
        return x, self._proj(K.self_attn(qkv, n, self.num_heads))

    def forward_post(self, x, abs_encoding, n):
        """BasicAttention.forward_post (NMP.py:110-128): q, k on [x | enc], v on x -- no norm in front --, x = norm1(x + proj(attn))."""
        cat = torch.cat((x, abs_encoding[:, : self.q.in_features - x.shape[1]]), 1)
        qkv = torch.cat((_rows_linear(cat, self.q), _rows_linear(cat, self.k), _rows_linear(x, self.v)), 1)
        return _post_norm_tail(self, x, K.self_attn(qkv, n, self.num_heads))

    def forward(self, label_rep, abs_encoding, n):
        """label_rep [T,C], abs_encoding [T,31] -> [T,C]; n = labels per pixel."""
        if not self.normalize_before:
            return self.forward_post(label_rep, abs_encoding, n)
        x, y = self.forward_pair(label_rep, None, abs_encoding, n)
        return x + y


# --------------------------------------------------------------------------------------------------
# neighbour edges in (shifted) windows (WindowAttention / SwinNMP, NMP.py:142-398)
# --------------------------------------------------------------------------------------------------
class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, shift_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.shift_size, self.num_heads = dim, window_size, shift_size, num_heads
        wh, ww = window_size
        self.relative_position_enc_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), dim * 3))
        ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
        coords = torch.stack((ys.reshape(-1), xs.reshape(-1)))
        rel = coords[:, :, None] - coords[:, None, :]
        idx = (rel[0] + wh - 1) * (2 * ww - 1) + (rel[1] + ww - 1)
        self.register_buffer("relative_position_index", idx)      # state-dict parity; the kernel derives it itself

    def kv16_ok(self, n):
        """this window geometry has a kernel that takes k | v as split fp16 pairs (include/nmrf_hip.h): 6 x 6 windows of 4 labels"""
        return tuple(self.window_size) == (6, 6) and n == 4 and self.num_heads == 4

    def forward(self, qkv, dims, sibling_mask, checked=False, kv16=False):
        """qkv [T,3C] token-major on the padded grid dims=(B,Hp,Wp,N) -> [T,C].  checked: qkv comes from nmp_block16, which has
        range-checked it (include/nmrf_hip.h, "fp16 range"); otherwise the entry point scans it first.  kv16: ... and wrote its
        k | v thirds as split fp16 operand pairs."""
        b, hp, wp, n = dims
        return K.window_attn(qkv, self.relative_position_enc_table, b, hp, wp, n, self.num_heads,
                             self.window_size[0], self.shift_size, sibling_mask, checked=checked, kv16=kv16)


class SwinNMP(nn.Module):
    def __init__(self, dim, qkv_dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., normalize_before=True):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        assert 0 <= shift_size < window_size
        self.dim, self.window_size, self.shift_size = dim, window_size, shift_size
        self.qkv = nn.Linear(qkv_dim, 3 * dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, (window_size, window_size), shift_size, num_heads)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self._qkv = _Lin(self.qkv)
        self._proj = _Lin(self.proj)

    def forward_pair(self, x, y, abs_encoding, dims, sibling_mask):
        r = self._qkv(x, ln=(self.norm1.weight, self.norm1.bias, self.norm1.eps), y=y, extra=abs_encoding)
        x, qkv = r if y is not None else (x, r)
        return self.mlp.forward_ln(x, self._proj(self.attn(qkv, dims, sibling_mask)), self.norm2)

    def forward_post(self, x, abs_encoding, dims, sibling_mask):
        """SwinNMP.forward_post (NMP.py:366-382) on the padded token grid dims = (B, Hp, Wp, N)."""
        qkv = _rows_linear(torch.cat((x, abs_encoding[:, : self.qkv.in_features - self.dim]), 1), self.qkv)
        return _post_norm_tail(self, x, self.attn(qkv, dims, sibling_mask))

    def forward(self, label_rep, abs_encoding, dims, sibling_mask):
        if not self.normalize_before:
            return self.forward_post(label_rep, abs_encoding, dims, sibling_mask)
        x, y = self.forward_pair(label_rep, None, abs_encoding, dims, sibling_mask)
        return x + y


# --------------------------------------------------------------------------------------------------
# cross-shaped stripes (CSWinAttention / CSWinNMP, NMP.py:401-600)
# --------------------------------------------------------------------------------------------------
class CSWinAttention(nn.Module):
    """Holds the LePE depthwise 3x3 kernel `get_v`; the attention itself is nmrf_stripe_attn_f32."""

    def __init__(self, dim, idx, split_size=1, num_heads=2):
        super().__init__()
        if split_size != 1:
            raise NotImplementedError("the HIP stripe kernel implements SPLIT_SIZE=1 (every shipped config)")
        self.dim, self.idx, self.num_heads = dim, idx, num_heads
        self.get_v = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim, bias=False)


class CSWinNMP(nn.Module):
    def __init__(self, dim, qk_dim, v_dim, num_heads, split_size=1, mlp_ratio=4., normalize_before=True):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        if v_dim != dim:
            raise NotImplementedError("v_dim > dim (fourier_grid_embed branch, NMP.py:552-555) is dead in every config")
        self.dim = dim
        self.q, self.k, self.v = nn.Linear(qk_dim, dim), nn.Linear(qk_dim, dim), nn.Linear(v_dim, dim)
        self.norm1 = nn.LayerNorm(dim)
        self.proj = nn.Linear(dim, dim)
        self.attns = nn.ModuleList(CSWinAttention(dim // 2, i, split_size, num_heads // 2) for i in range(2))
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim)
        self.norm2 = nn.LayerNorm(dim)
        self._packed_qkv = _FusedCache()
        self._proj = _Lin(self.proj)

    def forward_post(self, x, context_tok, dims):
        """CSWinNMP.forward_post (NMP.py:576-591); context_tok [T, Cctx]: the context row of a token's pixel."""
        b, h, wd, n = dims
        cat = torch.cat((x, context_tok), 1)
        qkv = torch.cat((_rows_linear(cat, self.q), _rows_linear(cat, self.k), _rows_linear(x, self.v)), 1)
        return _post_norm_tail(self, x, K.stripe_attn(qkv, self.attns[0].get_v.weight, self.attns[1].get_v.weight, b, h, wd, n))

    def forward(self, seed_rep, context, dims):
        """seed_rep [T,C]; context [B*H*W, Cctx] (per pixel, shared by its N labels); dims=(B,H,W,N)."""
        if not self.normalize_before:
            return self.forward_post(seed_rep, context.repeat_interleave(dims[3], 0), dims)
        x, y = self.forward_pair(seed_rep, None, context, dims)
        return x + y

    def forward_pair(self, x, y, context, dims):
        b, h, wd, n = dims
        pw, bias, k, nn_ = _packed(self._packed_qkv, (self.q.weight, self.k.weight, self.v.weight),
                                   (self.q.bias, self.k.bias, self.v.bias))
        r = K.token_linear(x, pw, nn_, k, bias, ln=(self.norm1.weight, self.norm1.bias, self.norm1.eps), y=y,
                           extra=context, extra_div=n)
        x, qkv = r if y is not None else (x, r)
        msg = K.stripe_attn(qkv, self.attns[0].get_v.weight, self.attns[1].get_v.weight, b, h, wd, n)
        return self.mlp.forward_ln(x, self._proj(msg), self.norm2)


class PropagationLayer(nn.Module):
    def __init__(self, embed_dim, mlp_ratio, context_dim, split_size, n_heads, normalize_before=True, **_unused):
        super().__init__()
        self.nmp = CSWinNMP(embed_dim, embed_dim + context_dim, embed_dim, n_heads, split_size, mlp_ratio,
                            normalize_before)

    def forward(self, tgt, context, dims):
        return self.nmp(tgt, context, dims)

    def forward_pair(self, x, y, context, dims):
        return self.nmp.forward_pair(x, y, context, dims)


class InferenceLayer(nn.Module):
    def __init__(self, embed_dim, mlp_ratio, window_size, shift_size, n_heads, normalize_before=True, **_unused):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        self.self_nmp = BasicAttention(embed_dim, embed_dim + FOURIER_DIM, n_heads, normalize_before)
        self.nmp = SwinNMP(embed_dim, embed_dim + FOURIER_DIM, n_heads, window_size, shift_size, mlp_ratio,
                           normalize_before)

    def forward(self, tgt, abs_encoding, dims):
        tgt = self.self_nmp(tgt, abs_encoding, dims[3])
        return self.nmp(tgt, abs_encoding, dims, True)

    def forward_pair(self, x, y, abs_encoding, dims):
        x, y = self.self_nmp.forward_pair(x, y, abs_encoding, dims[3])
        return self.nmp.forward_pair(x, y, abs_encoding, dims, True)


class RefinementLayer(nn.Module):
    def __init__(self, dim, mlp_ratio, window_size, shift_size, n_heads, normalize_before=True, **_unused):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        self.nmp = SwinNMP(dim, dim + FOURIER_DIM, n_heads, window_size, shift_size, mlp_ratio, normalize_before)

    def forward(self, tgt, abs_encoding, dims):
        return self.nmp(tgt, abs_encoding, dims, False)

    def forward_pair(self, x, y, abs_encoding, dims):
        return self.nmp.forward_pair(x, y, abs_encoding, dims, False)


# --------------------------------------------------------------------------------------------------
# stage drivers
# --------------------------------------------------------------------------------------------------
class Propagation(nn.Module):
    """Label-seed propagation (NMP.py:603-667)."""

    def _forward_blocks(self, x, ctx, dims):
        """5 x [stripe attention -> one fused block kernel]; the block of layer i also produces layer i+1's q|k|v."""
        b, h, wd, n = dims
        L = [l.nmp for l in self.layers]
        kv16 = n == 4            # the stripe kernels take k | v pre-split at four labels per pixel (every shipped config)
        if not hasattr(self, "_launch") or self._launch_kv16 != kv16:
            self._launch_kv16 = kv16
            self._launch = [_BlockLauncher(nxt_norm=L[0].norm1, nxt_linears=_qkv_of(L[0]), kq=192, kv16=kv16)]
            for i, m in enumerate(L):
                if i + 1 < len(L):
                    self._launch.append(_BlockLauncher(m.proj, (m.norm2, m.mlp), L[i + 1].norm1, _qkv_of(L[i + 1]), 192, kv16=kv16))
                elif self.norm is not None:
                    self._launch.append(_BlockLauncher(m.proj, (m.norm2, m.mlp), self.norm, (), 128, ln_out=True))
                else:
                    self._launch.append(_BlockLauncher(m.proj, (m.norm2, m.mlp)))
        _, qkv, _ = self._launch[0](x, None, ctx, n, want_x=False)
        keep_pre = self.training and getattr(self, "keep_pre_norm", False)      # NMRF.enable_grad_slice: see Inference._run_blocks
        tape = getattr(self, "_tape", None) if keep_pre else None
        if tape is not None:
            tape.update(x=[x], qkv=[qkv], msg=[], dims=dims, ctx=ctx, kv16=kv16)
        for i, m in enumerate(L):
            msg = K.stripe_attn(qkv, m.attns[0].get_v.weight, m.attns[1].get_v.weight, b, h, wd, n, kv16=kv16)
            last = i + 1 == len(L)
            x_in = x
            x, qkv, ln = self._launch[i + 1](x, msg, ctx, n, want_x=not last or self.norm is None or keep_pre)
            if last and keep_pre:
                self._last_block = (x_in, msg, x, m, None)
            if tape is not None:
                tape["msg"].append(msg)
                tape["x"].append(x)
                if qkv is not None:
                    tape["qkv"].append(qkv)
        return ln if self.norm is not None else x

    def __init__(self, embed_dim, cost_group, layers, norm=None):
        super().__init__()
        self.cost_encoder = nn.Sequential(nn.Linear(cost_group * 9, embed_dim), nn.GELU(), nn.Linear(embed_dim, embed_dim))
        self.proj = nn.Linear(embed_dim + FOURIER_DIM, embed_dim, bias=False)
        self.embed_dim, self.layers, self.norm = embed_dim, layers, norm

    def _split_ok(self, cc):
        return (_split() and cc == 64 and self.embed_dim == 128 and self.cost_encoder[0].in_features <= 48
                and self.cost_encoder[0].in_features % 4 == 0 and all(_block_ok(l.nmp.proj, l.nmp.mlp) for l in self.layers))

    def seed_feature_args(self, context):
        """(Fourier normalizer, encoding row stride) of the seed features this module consumes (NMP.py:646-647), for the producer of
        the seeds to gather them in its own launch (DPN.seeds / K.seed_select)."""
        cc = context.shape[-1] if context is not None else self.layers[0].nmp.q.in_features - self.embed_dim   # rows [B,H,W,Cctx]
        return (3.14 / 64, 32 if self._split_ok(cc) else 31)

    def forward(self, cost_volume, label_seed, context, feats=None):
        """cost_volume [P,G,D]; label_seed [P,N] int64; context [B,H,W,Cctx] -> ([1,P*N,C], seeds.float()).
        feats: (seeds as float, cost taps, Fourier encoding) already gathered by the seed kernel, or None."""
        b, h, wd, cc = context.shape
        n = label_seed.shape[-1]
        dims = (b, h, wd, n)
        if not self.layers[0].nmp.normalize_before:
            return self._forward_post(cost_volume, label_seed, context.reshape(b * h * wd, cc), dims)
        split = self._split_ok(cc)
        ctx = context.reshape(b * h * wd, cc)
        seeds_f = None
        if feats is not None:
            seeds_f, cost, enc = feats
            if enc.shape[1] != (32 if split else 31):
                feats = None
        if split:
            if feats is None:
                cost, enc = K.seed_features(cost_volume, label_seed, 3.14 / 64, 32)
            if not hasattr(self, "_embed"):
                self._embed = _ChainLauncher(1, (self.cost_encoder[0], self.cost_encoder[2], self.proj), (48, 128, 160), 128)
            x = self._embed(cost, cost.shape[1], extra=enc)
            if self.training and getattr(self, "keep_pre_norm", False):
                self._tape = {"cost": cost, "enc": enc}          # (NMRF.enable_grad_slice: the seed embedding's operands; _forward_blocks adds the rest)
            return self._forward_blocks(x, ctx, dims).unsqueeze(0), (label_seed.float() if seeds_f is None else seeds_f)
        if feats is None:
            cost, enc = K.seed_features(cost_volume, label_seed, 3.14 / 64)
        seeds_f = label_seed.float() if seeds_f is None else seeds_f
        x = self.proj(torch.cat((self.cost_encoder(cost), enc), -1))
        if _split() and cc == 64 and self.embed_dim == 128 and all(_block_ok(l.nmp.proj, l.nmp.mlp) for l in self.layers):
            return self._forward_blocks(x.contiguous(), ctx, dims).unsqueeze(0), seeds_f
        y = None
        for layer in self.layers:
            x, y = layer.forward_pair(x, y, ctx, dims)
        if self.norm is not None:
            x = _add_ln(x, y, self.norm)[1]
        elif y is not None:
            x = x + y
        return x.unsqueeze(0), seeds_f


def _propagation_forward_post(self, cost_volume, label_seed, ctx, dims):
    """Propagation.forward with NMP.NORMALIZE_BEFORE False (NMP.py:636-667 around CSWinNMP.forward_post): seed embedding = cost_encoder on
    the 9 x G cost taps, proj on [features | Fourier]; five post-norm layers; the stage's final norm."""
    cost, enc = K.seed_features(cost_volume, label_seed, 3.14 / 64)
    ce = self.cost_encoder
    feat = _rows_linear(_rows_linear(cost, ce[0], act=2), ce[2])
    x = _rows_linear(torch.cat((feat, enc[:, : self.proj.in_features - feat.shape[1]]), 1), self.proj)
    ctx_tok = ctx.repeat_interleave(dims[3], 0).contiguous()
    for layer in self.layers:
        x = layer.nmp.forward_post(x, ctx_tok, dims)
    if self.norm is not None:
        x = K.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
    return x.unsqueeze(0), label_seed.float()


Propagation._forward_post = _propagation_forward_post


def _qkv_of(nmp):
    return (nmp.q, nmp.k, nmp.v) if hasattr(nmp, "q") else (nmp.qkv,)


def _pad_grid(x, dims, win):
    """zero-pad the (H,W) token grid to a multiple of `win`, top = pad//2 (NMP.py:745-762)."""
    b, h, wd, n = dims
    ph, pw = (-h) % win, (-wd) % win
    if ph == 0 and pw == 0:
        return x, dims, (0, 0)
    top, left = ph // 2, pw // 2
    xg = F.pad(x.view(b, h, wd, n, -1), (0, 0, 0, 0, left, pw - left, top, ph - top))
    return xg.reshape(-1, x.shape[-1]), (b, h + ph, wd + pw, n), (top, left)


def _crop_grid(x, pdims, dims, off):
    if pdims == dims:
        return x
    b, hp, wp, n = pdims
    _, h, wd, _ = dims
    return x.view(b, hp, wp, n, -1)[:, off[0]:off[0] + h, off[1]:off[1] + wd].reshape(-1, x.shape[-1])


class Inference(nn.Module):
    """Neural MRF inference (NMP.py:670-798)."""

    normalizer = 3.14 / 64

    def __init__(self, cost_group, dim, layers, norm, return_intermediate=False):
        super().__init__()
        self.ffn = Mlp(dim + cost_group, dim, dim)
        self.dim, self.layers, self.norm, self.cost_group = dim, layers, norm, cost_group
        self.return_intermediate = return_intermediate     # NMP.py:777,879: honoured in training mode only

    def _run(self, labels_flat, n, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=False, collect=None):
        """collect: a list that receives norm(crop(x)) after every layer (return_intermediate, NMP.py:777-796 / 879-898); its last
        entry is the returned tensor."""
        if token_major:                                   # maps [B,H,W,C] (written so by the heads' 1x1 kernels)
            b, h, wd, _ = fmap1.shape
        else:
            b, _, h, wd = fmap1.shape
        dims = (b, h, wd, n)
        split = _split() and self.dim == 128 and all(
            _block_ok(l.nmp.proj, l.nmp.mlp, *((l.self_nmp.proj,) if hasattr(l, "self_nmp") else ())) for l in self.layers)
        win = self.layers[0].window_size
        fused = (split and self.ffn.fc1.in_features == 160 and self.ffn.fc1.out_features == 128 and self.ffn.fc2.out_features == 128)
        if fused:
            # ffn and Fourier rows are written straight into the zero-padded token grid; the final norm is cropped on the way out
            if not hasattr(self, "_maps"):
                self._maps = {}
                self._ffn = _ChainLauncher(0, (self.ffn.fc1, self.ffn.fc2), (160, 128), 128)
            dev = fmap1.device
            pdims, to_p, to_d = _pad_maps(dims, win, dev, self._maps)
            tp = pdims[0] * pdims[1] * pdims[2] * pdims[3]
            ebuf = None
            if to_p is not None:
                # the padded grids are persistent: their padding rows are zeroed once and never written (the row maps only address
                # real tokens), the interior is overwritten by every forward -- no fill kernels on the hot path
                # keyed like the row maps (geometry, not size): two inputs with the same padded size but different padding
                # (KITTI 1242x375 vs 1224x370 -> both 48x156 cells) must not share a buffer, or the rows that were real tokens of
                # the first become non-zero "padding" of the second
                gkey = (dims, win, str(dev))
                if not hasattr(self, "_grids"):
                    self._grids = {}
                if gkey not in self._grids:
                    self._grids[gkey] = (torch.zeros(tp, self.dim, device=dev), torch.zeros(tp, 32, device=dev))
                xbuf, ebuf = self._grids[gkey]
            if token_major:
                # token-major maps: the Fourier rows of the labels leave with the warp kernel's tokens (one launch less per stage)
                wcc, enc = K.warp_corr_concat(labels_flat, fmap1, fmap2, fmap1_gw, fmap2_gw, n, self.cost_group, token_major=True,
                                              fourier=(self.normalizer, ebuf, to_p))
            else:
                wcc = K.warp_corr_concat(labels_flat, fmap1, fmap2, fmap1_gw, fmap2_gw, n, self.cost_group)
                enc = K.fourier_embed(labels_flat, self.normalizer, 32, out=ebuf, out_map=to_p)
            x = self._ffn(wcc, 160) if to_p is None else self._ffn(wcc, 160, out=xbuf, out_map=to_p)
            t_dense = dims[0] * dims[1] * dims[2] * dims[3]
            if not self.layers[0].nmp.normalize_before:
                if collect is not None:
                    raise NotImplementedError("return_intermediate (training mode) with NMP.NORMALIZE_BEFORE False")
                return self._run_blocks_post(x, enc, pdims, to_d)
            if collect is not None and getattr(self, "keep_pre_norm", False):
                # the tape of the training-mode forward (NMRF.enable_grad_slice): every tensor a backward through the WHOLE stage needs --
                # the ffn operand, the Fourier rows and, appended by _run_blocks, each layer's residual stream, q | k | v and message
                # (clones of the persistent padded grids: the next forward overwrites them)
                self._tape = {"wcc": wcc, "enc": enc.clone(), "x": [x.clone()], "qkv": [], "msg": [], "pdims": pdims, "to_p": to_p}
            return self._run_blocks(x, enc, pdims, to_d, t_dense, collect)
        if collect is not None:
            raise NotImplementedError("return_intermediate is implemented on the fused block path (128-wide tokens, shipped head shapes)")
        wcc = K.warp_corr_concat(labels_flat, fmap1, fmap2, fmap1_gw, fmap2_gw, n, self.cost_group, token_major=token_major)
        x = self.ffn(wcc)
        enc = K.fourier_embed(labels_flat, self.normalizer, 32 if split else 31)
        x, pdims, off = _pad_grid(x, dims, win)
        enc, _, _ = _pad_grid(enc, dims, win)
        x, enc = x.contiguous(), enc.contiguous()
        if split:
            out = self._run_blocks(x, enc, pdims)
            return _crop_grid(out, pdims, dims, off).contiguous()
        y = None
        for layer in self.layers:
            x, y = layer.forward_pair(x, y, enc, pdims)
        x = _crop_grid(x, pdims, dims, off).contiguous()
        if y is not None:
            y = _crop_grid(y, pdims, dims, off).contiguous()
        if self.norm is not None:
            return _add_ln(x, y, self.norm)[1]
        return x if y is None else x + y

    def _run_blocks(self, x, enc, pdims, to_dense=None, t_dense=None, collect=None):
        """Per layer: [self-edge attention -> fused block (proj + residual -> window q|k|v)] (inference only), then
        window attention -> fused block (proj + residual + MLP -> the next layer's first q|k|v, or the final norm).
        collect (training-mode forward): after every layer but the last, the stage's final LayerNorm of the layer's residual stream on
        the dense token grid (one ln kernel per layer; the last layer's entry is the fused kernel's own ln output)."""
        if collect is not None and self.norm is None:
            raise NotImplementedError("return_intermediate without a final norm (never configured by the reference, NMRF.py:79,102)")
        keep = None
        if collect is not None and to_dense is not None:
            keep = (to_dense >= 0).nonzero().squeeze(1)
        # keep_pre_norm (set by NMRF.enable_grad_slice): the training-mode forward also keeps every layer's residual stream BEFORE the
        # stage-final LayerNorm on the dense grid (`_pre_norm`), the saved input of that norm's backward (models/autograd_ops.py)
        keep_pre = collect is not None and getattr(self, "keep_pre_norm", False)
        if keep_pre:
            self._pre_norm = []
        n = pdims[3]
        if not hasattr(self, "_launch"):
            sites = []                                                   # (kind, module) in execution order
            for l in self.layers:
                if hasattr(l, "self_nmp"):
                    sites.append(("self", l.self_nmp))
                sites.append(("win", l.nmp))
            self._sites = sites
            first = sites[0][1]
            # a launch site that feeds a window attention with a pre-split form writes k | v as split fp16 pairs (kv16)
            kv = lambda kind_m: kind_m[0] == "win" and kind_m[1].attn.kv16_ok(n)
            self._site_kv16 = [kv(sm) for sm in sites]
            self._launch = [_BlockLauncher(nxt_norm=first.norm1, nxt_linears=_qkv_of(first), kq=160, kv16=self._site_kv16[0])]
            for i, (kind, m) in enumerate(sites):
                mlp = (m.norm2, m.mlp) if kind == "win" else None
                if i + 1 < len(sites):
                    nx = sites[i + 1][1]
                    self._launch.append(_BlockLauncher(m.proj, mlp, nx.norm1, _qkv_of(nx), 160, kv16=self._site_kv16[i + 1]))
                elif self.norm is not None:
                    self._launch.append(_BlockLauncher(m.proj, mlp, self.norm, (), 128, ln_out=True))
                else:
                    self._launch.append(_BlockLauncher(m.proj, mlp))
        ln = None
        # a window block and the self-edge block behind it run as one launch (the self-edge q | k | v stay on the CU) where the pair has
        # the shipped shape; the training-mode intermediates need the window block's own output, so they take the two launches
        if not hasattr(self, "_pairs"):
            self._pairs = {}
            for i, (kind, m) in enumerate(self._sites):
                if (kind == "win" and i + 2 < len(self._sites) and self._sites[i + 1][0] == "self" and n == 4
                        and self._sites[i + 1][1].num_heads == 4 and _PairLauncher.ok(self._launch[i + 1], self._launch[i + 2])):
                    self._pairs[i] = _PairLauncher(self._launch[i + 1], self._launch[i + 2])
            # ... and the stage's opening launch (a q stage alone) with the first self-edge block
            if (len(self._sites) > 1 and self._sites[0][0] == "self" and n == 4 and self._sites[0][1].num_heads == 4
                    and _PairLauncher.ok(self._launch[0], self._launch[1])):
                self._pairs[-1] = _PairLauncher(self._launch[0], self._launch[1])
        skip = False
        if collect is None and -1 in self._pairs and _split():
            x, qkv, ln = self._pairs[-1](x, None, enc, 1)
            skip = True
        else:
            _, qkv, _ = self._launch[0](x, None, enc, 1, want_x=False)
        tape = getattr(self, "_tape", None) if keep_pre else None
        if tape is not None:
            tape["qkv"].append(qkv)
        for i, (kind, m) in enumerate(self._sites):
            if skip:                                     # this self-edge site ran inside the previous launch
                skip = False
                continue
            if kind == "win" and collect is None and i in self._pairs and _split():
                msg = m.attn(qkv, pdims, n > 1, checked=True, kv16=self._site_kv16[i])
                x, qkv, ln = self._pairs[i](x, msg, enc, 1)
                skip = True
                continue
            attn_qkv = None
            if kind == "self" and n == 4 and m.num_heads == 4 and qkv.shape[1] == 384 and i + 1 < len(self._sites):
                msg, attn_qkv = None, qkv               # the 4 x 4 self-edge attention is evaluated inside the block kernel
            elif kind == "self":
                msg = K.self_attn(qkv, n, m.num_heads)
            else:
                msg = m.attn(qkv, pdims, n > 1, checked=True, kv16=self._site_kv16[i])   # sibling mask for N > 1 (inference), none for refinement
            last = i + 1 == len(self._sites)
            if last and self.norm is not None and to_dense is not None:       # final norm, cropped to the dense grid on the way out
                ln = torch.empty(t_dense, self.dim, device=x.device)
                xo, _, _ = self._launch[i + 1](x, msg, enc, 1, want_x=keep_pre, ln_out=ln, ln_out_map=to_dense)
                if collect is not None:
                    collect.append(ln)
                if keep_pre:
                    self._pre_norm.append(xo.index_select(0, keep).contiguous())
                    self._last_block = (x, msg, xo, m, keep)        # the last block's operands on the padded grid (autograd_ops.BlockFn)
                    if tape is not None:
                        tape["msg"].append(msg)
                        tape["x"].append(xo)
                return ln
            x_in = x
            x, qkv, ln = self._launch[i + 1](x, msg, enc, 1, want_x=not last or self.norm is None or keep_pre, attn_qkv=attn_qkv)
            if keep_pre and last and kind == "win":
                self._last_block = (x_in, msg, x, m, keep)
            if tape is not None:
                tape["msg"].append(msg)
                tape["x"].append(x)
                if qkv is not None:
                    tape["qkv"].append(qkv)
            if collect is not None and kind == "win":
                xd = None
                if not last or keep_pre:
                    xd = (x if keep is None else x.index_select(0, keep)).contiguous()
                if keep_pre:
                    self._pre_norm.append(xd)
                if last:
                    collect.append(ln)
                else:
                    collect.append(K.ln_concat(xd, self.norm.weight, self.norm.bias, ld=self.dim, eps=self.norm.eps))
        if to_dense is not None:
            keep = (to_dense >= 0).nonzero().squeeze(1)
            return (ln if self.norm is not None else x).index_select(0, keep)
        return ln if self.norm is not None else x

    def _run_blocks_post(self, x, enc, pdims, to_dense):
        """The layers of the stage in their forward_post form (NMP.NORMALIZE_BEFORE False) on the padded token grid, then the crop to the
        dense grid and the stage's final norm (NMP.py:764-790, 866-892)."""
        n = pdims[3]
        for l in self.layers:
            if hasattr(l, "self_nmp"):
                x = l.self_nmp.forward_post(x, enc, n)
            x = l.nmp.forward_post(x, enc, pdims, hasattr(l, "self_nmp"))
        if to_dense is not None:
            x = x.index_select(0, (to_dense >= 0).nonzero().squeeze(1))
        if self.norm is not None:
            x = K.layer_norm(x.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def forward(self, labels, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=False):
        """labels [B*H*W, N] -> [1, B*H*W, N, C]   (token_major: the four maps are [B,H,W,C] instead of [B,C,H,W]);
        [num_layers, B*H*W, N, C] in training mode with return_intermediate (NMP.py:777-798)."""
        n = labels.shape[-1]
        collect = [] if (self.return_intermediate and self.training) else None
        out = self._run(labels.reshape(-1).contiguous(), n, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major, collect)
        if collect is not None:
            return torch.stack(collect).view(len(collect), -1, n, self.dim)
        return out.view(1, -1, n, self.dim)


class Refinement(Inference):
    """Refinement at 1/4 resolution, one label per pixel (NMP.py:801-900)."""

    normalizer = 3.14 / 128

    def forward(self, labels, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major=False):
        """labels [B,H,W] -> [1, B*H*W, C]; [num_layers, B*H*W, C] in training mode with return_intermediate (NMP.py:879-900)"""
        collect = [] if (self.return_intermediate and self.training) else None
        out = self._run(labels.reshape(-1).contiguous(), 1, fmap1, fmap2, fmap1_gw, fmap2_gw, token_major, collect)
        if collect is not None:
            return torch.stack(collect).view(len(collect), -1, self.dim)
        return out.view(1, -1, self.dim)
