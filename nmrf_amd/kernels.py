"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch, pass raw device
pointers and the CURRENT torch stream.  PyTorch is plumbing here (memory + streams); all arithmetic
of these ops happens in libnmrf_hip.so.  Every wrapper refuses non-CUDA tensors: there is no
fallback path."""
import functools
import os

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors, dtype=torch.float32):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.is_cuda:
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise _lib.NmrfHipError("tensors of one kernel call live on different devices: %s and %s" % (dev, t.device))
        if not t.is_cuda:
            raise _lib.NmrfHipError("NMRF hot-path kernels run on the MI355X only: got a %s tensor "
                                    "(no CPU fallback exists by design)" % t.device)
        if not t.is_contiguous():
            raise _lib.NmrfHipError("tensor must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise _lib.NmrfHipError("expected %s, got %s" % (dtype, t.dtype))


def _p(t):
    return None if t is None else t.data_ptr()


# ---- fp16 range guard of the split-operand kernels (include/nmrf_hip.h, "fp16 range"): one sticky int32 per device --------------
_range_flags = {}


def range_flag(device=None):
    """The device int32 the guarded kernels OR their out-of-range bit into (allocated on first use, one per device)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    f = _range_flags.get(dev.index)
    if f is None:
        f = _range_flags[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
    return f


def _rf(t):
    return range_flag(t.device).data_ptr()


class BlockKernelClock:
    """Shader clock of the CUs under the block kernels (nmrf_nmp_block16_clock_records): `with BlockKernelClock() as c: <eager launches of
    K.nmp_block / K.nmp_block_pair>`; afterwards c.ghz = sum of the blocks' shader-clock cycles / sum of their 100 MHz ticks, of the LAST
    launch inside the block (each launch overwrites the records), c.ghz_min / c.ghz_max over its blocks.  A measurement helper of
    bench.py; off in every product call."""

    def __init__(self, device=None, capacity=1024):
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.buf = torch.zeros(capacity, 4, dtype=torch.int64, device=self.dev)
        self.capacity = capacity
        self.ghz = self.ghz_min = self.ghz_max = None

    def __enter__(self):
        _lib.check(_lib.load().nmrf_nmp_block16_clock_records(self.buf.data_ptr(), self.capacity), "clock_records")
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize(self.dev)
        _lib.check(_lib.load().nmrf_nmp_block16_clock_records(None, 0), "clock_records")
        b = self.buf.cpu()
        b = b[b[:, 3] > b[:, 2]]
        self.blocks = int(b.shape[0])
        if self.blocks:
            cyc, ticks = (b[:, 1] - b[:, 0]).double(), (b[:, 3] - b[:, 2]).double()
            per = cyc / ticks * 0.1
            self.ghz, self.ghz_min, self.ghz_max = float(cyc.sum() / ticks.sum() * 0.1), float(per.min()), float(per.max())
            self.block_us = float(ticks.mean()) * 1e-2
        return False


def check_range(device=None, reset=True):
    """Synchronise with `device` and raise NmrfHipError if any guarded kernel since the last check converted an activation that does
    not fit the fp16 range of the split-operand arithmetic (|x| >= 65520, or NaN).  The results of those launches are invalid (inf
    / NaN or, downstream of a ReLU or a masked softmax, silently wrong).  The reference computes in plain fp32 and has no such
    limit; weights have none here either (rescaled by a power of two at pack time)."""
    f = range_flag(device)
    v = int(f.item())
    if v and reset:
        f.zero_()
    if v:
        raise _lib.NmrfHipError("an activation left the fp16 range of the split-operand MFMA kernels (|x| >= 65520 or NaN): the "
                                "outputs since the last check are invalid.  The reference's fp32 arithmetic has no such limit; "
                                "rescale the inputs / checkpoint (csrc/split_mfma.h)")
    return True


def _on_device(fn):
    """Launch on the device the tensors live on (like the reference's device-guarded ATen ops): the stream handed to the C ABI
    is that device's current torch stream and the HIP current device is switched for the duration of the call."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
        return fn(*args, **kw)
    return wrapper


@_on_device
def mfma_selftest(a, bm):
    _chk(a, bm)
    k = a.shape[1]
    out = torch.empty(32, 32, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_selftest_mfma_f32(_p(a), _p(bm), k, _p(out), _stream()), "mfma selftest")
    return out


@_on_device
def mfma16x16_selftest(a, bm):
    _chk(a, bm)
    out = torch.empty(16, 16, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_selftest_mfma16x16_f16split(_p(a), _p(bm), a.shape[1], _p(out), _stream()), "mfma 16x16 selftest")
    return out


@_on_device
def mfma_f16split_selftest(a, bm, mode=0):
    _chk(a, bm)
    out = torch.empty(32, 32, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_selftest_mfma_f16split(_p(a), _p(bm), a.shape[1], mode, _p(out), _stream()), "mfma f16 split selftest")
    return out


@_on_device
def lds_dma_selftest(src):
    _chk(src)
    dst = torch.empty_like(src)
    _lib.check(_lib.load().nmrf_selftest_lds_dma(_p(src), _p(dst), src.numel() // 4, _stream()), "lds dma selftest")
    return dst


@_on_device
def cost_volume(f1, f2, num_disp, groups):
    """[B,C,H,W] x2 -> [B*H*W, G, D]"""
    _chk(f1, f2)
    b, c, h, w = f1.shape
    vol = torch.empty(b * h * w, groups, num_disp, device=f1.device, dtype=torch.float32)
    _hb("cost_volume", row="A2", bound="hbm", bytes=4.0 * (2 * f1.numel() + vol.numel()), flops=2.0 * b * h * w * num_disp * c,
        label="cost_volume_kernel (group-wise correlation volume, A2)", pmc=["cost_volume_kernel<"])
    _lib.check(_lib.load().nmrf_cost_volume_f32(_p(f1), _p(f2), b, c, h, w, num_disp, groups, _p(vol), _stream()),
               "cost_volume")
    _he("cost_volume")
    return vol


@_on_device
def dpn_filter_softmax(vol, w0, b0, w1, b1, w2, b2):
    _chk(vol, w0, b0, w1, b1, w2, b2)
    p, g, d = vol.shape
    prob = torch.empty(p, d, device=vol.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_dpn_filter_softmax_f32(_p(vol), _p(w0), _p(b0), _p(w1), _p(b1), _p(w2), _p(b2), p, g, d,
                                                       _p(prob), _stream()), "dpn_filter_softmax")
    return prob


@_on_device
def nms_topk(prob, k, eps, do_nms=True):
    _chk(prob)
    p, d = prob.shape
    seeds = torch.empty(p, k, device=prob.device, dtype=torch.int64)
    _lib.check(_lib.load().nmrf_nms_topk_f32(_p(prob), p, d, k, float(eps), int(do_nms), _p(seeds), _stream()),
               "nms_topk")
    return seeds


@_on_device
def seed_select(prob, vol, k, eps, normalizer, enc_ld=31, do_nms=True):
    """NMS + top-k and the seed features in one launch (csrc/seed.hip seed_select_kernel): -> (seeds [P,k] int64, seeds as
    float [P,k], cost [P*k, G*9], enc [P*k, enc_ld]).  Same seeds, bit for bit, as nms_topk; same features as seed_features."""
    _chk(prob, vol)
    p, d = prob.shape
    g = vol.shape[1]
    seeds = torch.empty(p, k, device=prob.device, dtype=torch.int64)
    seeds_f = torch.empty(p, k, device=prob.device, dtype=torch.float32)
    cost = torch.empty(p * k, g * 9, device=prob.device, dtype=torch.float32)
    enc = torch.empty(p * k, enc_ld, device=prob.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_seed_select_f32(_p(prob), _p(vol), p, g, d, k, float(eps), int(do_nms), float(normalizer), _p(seeds),
                                                _p(seeds_f), _p(cost), _p(enc), enc_ld, _stream()), "seed_select")
    return seeds, seeds_f, cost, enc


@_on_device
def seed_features(vol, seeds, normalizer, enc_ld=31):
    _chk(vol)
    _chk(seeds, dtype=torch.int64)
    p, g, d = vol.shape
    n = seeds.shape[1]
    cost = torch.empty(p * n, g * 9, device=vol.device, dtype=torch.float32)
    enc = torch.empty(p * n, enc_ld, device=vol.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_seed_features_f32(_p(vol), _p(seeds), p, g, d, n, float(normalizer), _p(cost), _p(enc), enc_ld,
                                                  _stream()), "seed_features")
    return cost, enc


@_on_device
def fourier_embed(coord, normalizer, ld=31, out=None, out_map=None):
    """-> [T, ld]: 31 Fourier columns, the rest (ld = 32: 16-byte rows for the fused block kernel) zero.
    out / out_map: write token t to row out_map[t] of the given (pre-zeroed, padded-grid) tensor instead."""
    _chk(coord, out)
    _chk(out_map, dtype=torch.int32)
    t = coord.numel()
    enc = out if out is not None else torch.empty(t, ld, device=coord.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_fourier_embed_f32(_p(coord), t, float(normalizer), _p(enc), enc.shape[-1], _p(out_map), _stream()),
               "fourier_embed")
    return enc


@_on_device
def ln_concat(x, gamma, beta, extra=None, extra_div=1, ld=None, eps=1e-5):
    """-> [T, ld] = [LN(x) | extra[t // extra_div] | 0-pad]"""
    _chk(x, gamma, beta, extra)
    t, c = x.shape
    e = 0 if extra is None else extra.shape[-1]
    if ld is None:
        ld = (c + e + 3) // 4 * 4
    out = torch.empty(t, ld, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_ln_concat_f32(_p(x), _p(gamma), _p(beta), float(eps), _p(extra), e, extra_div, t, c,
                                              _p(out), ld, _stream()), "ln_concat")
    return out


@_on_device
def add_ln_concat(x, y, gamma, beta, extra=None, extra_div=1, ld=None, eps=1e-5):
    """x_new = x + y (returned, fresh tensor) and [LN(x_new) | extra | 0-pad] in one pass."""
    _chk(x, y, gamma, beta, extra)
    t, c = x.shape
    e = 0 if extra is None else extra.shape[-1]
    if ld is None:
        ld = (c + e + 3) // 4 * 4
    x_new = torch.empty_like(x)
    out = torch.empty(t, ld, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_add_ln_concat_f32(_p(x), _p(y), _p(x_new), _p(gamma), _p(beta), float(eps), _p(extra), e,
                                                  extra_div, t, c, _p(out), ld, _stream()), "add_ln_concat")
    return x_new, out


# optional observer used by bench.py to bracket kernel launches with HIP events: called as hook("begin", name, meta) /
# hook("end", name, None) around one launch, on the launching stream.  meta: row, bound ("mfma"|"hbm"),
# flops / bytes (ALGORITHMIC work of the launch, SURVEY 8(d) formulas), label, pmc (kernel name(s) in the rocprofv3 tables)
kernel_hook = None


def _hb(name, **meta):
    if kernel_hook is not None:
        kernel_hook("begin", name, meta)


def _he(name):
    if kernel_hook is not None:
        kernel_hook("end", name, None)


@_on_device
def stripe_attn(qkv, lepe_v, lepe_h, b, h, w, n, kv16=False, two_launches=False):
    """kv16: the k | v thirds of qkv are split fp16 operand pairs (nmp_block(q=dict(kv16=True)) / to_kv16); with four labels both axes
    then run in ONE launch (two_launches=True: the vertical and the horizontal kernel one after the other -- tests, A/B)."""
    _chk(qkv, lepe_v, lepe_h)
    t, c3 = qkv.shape
    c = c3 // 3
    assert t == b * h * w * n
    out = torch.empty(t, c, device=qkv.device, dtype=torch.float32)
    fn = _lib.load().nmrf_stripe_attn_f32
    rf = None if kv16 else _rf(qkv)                      # (pre-split operands were range-checked by their producer)
    # per (row, head): QK^T and PV, 2*T^2*32 FLOPs each, T = W*N, 2 heads of 32 channels (columns: T = H*N)
    fl_h, fl_v = b * h * 2 * 4.0 * 32 * (w * n) ** 2, b * w * 2 * 4.0 * 32 * (h * n) ** 2
    if kv16 and n == 4 and not two_launches:
        # pre-split rows, four labels: both axes in one launch (stripe_attn_both_kernel: the horizontal items first, the vertical ones
        # fill in behind them)
        _hb("stripe_attn_both", row="A7", bound="mfma", split=True, flops=fl_h + fl_v, bytes=4.0 * (qkv.numel() + t * c),
            label="stripe_attn_both_kernel (vertical + horizontal stripes in one launch, A7)", pmc=["stripe_attn_both_kernel<2, true>"])
        _lib.check(fn(_p(qkv), _p(lepe_v), _p(lepe_h), b, h, w, n, c, 3, 1, _p(out), rf, _stream()), "stripe_attn(both axes)")
        _he("stripe_attn_both")
        return out
    _lib.check(fn(_p(qkv), _p(lepe_v), _p(lepe_h), b, h, w, n, c, 1, int(kv16), _p(out), rf, _stream()), "stripe_attn(vertical)")
    _hb("stripe_attn_horizontal", row="A7", bound="mfma", split=True, flops=fl_h, bytes=4.0 * (qkv.numel() / 2 + t * c / 2),
        label="stripe_attn_kernel<1> (horizontal stripes, A7)",
        pmc=["stripe_attn_kernel<1, 2, 1, false, %s>" % ("true" if kv16 else "false"), "stripe_attn_kernel<1, 2, 2, false,"])
    _lib.check(fn(_p(qkv), _p(lepe_v), _p(lepe_h), b, h, w, n, c, 2, int(kv16), _p(out), rf, _stream()), "stripe_attn(horizontal)")
    _he("stripe_attn_horizontal")
    return out


@_on_device
def warp_corr_concat(labels, f1, f2, g1, g2, n, groups=32, ld=None, token_major=False, fourier=None):
    """f1, f2 [B,Cf,H,W], g1, g2 [B,Cg,H,W] (token_major: [B,H,W,C], Cf 64 / Cg 256 / 32 groups) -> [T, ld].
    fourier = (normalizer, enc [rows, enc_ld] or None, out_map int32 [T] or None): the Fourier embedding of the labels is written by
    the same launch (token_major only; what fourier_embed(labels, normalizer, enc_ld, out=enc, out_map=out_map) writes) and the
    result is (rows, enc)."""
    _chk(labels, f1, f2, g1, g2)
    if token_major:
        b, h, w, cf = f1.shape
        cg = g1.shape[3]
    else:
        b, cf, h, w = f1.shape
        cg = g1.shape[1]
    if ld is None:
        ld = 2 * cf + groups
    t = b * h * w * n
    assert labels.numel() == t
    out = torch.empty(t, ld, device=f1.device, dtype=torch.float32)
    enc = None
    if fourier is not None:
        if not token_major:
            raise NmrfHipError("warp_corr_concat: the Fourier rows ride with the token-major kernel only")
        normalizer, enc, out_map = fourier
        if enc is None:
            enc = torch.empty(t, 32, device=f1.device, dtype=torch.float32)
        _chk(enc)
        if out_map is not None:
            _chk(out_map, dtype=torch.int32)
            assert out_map.numel() == t
    _hb("warp_corr_concat_n%d" % n, row="A9" if n > 1 else "A13", bound="hbm",
        bytes=4.0 * (2 * f1.numel() + 2 * g1.numel() + t + out.numel()), flops=2.0 * t * (cg + 2 * cf),
        label="warp_corr_concat_kernel (warp + 32-group correlation + concat, %s)" % ("A9, 1/8" if n > 1 else "A13, 1/4"),
        pmc=["warp_corr_concat_tok_kernel" if token_major else "warp_corr_concat_kernel"])
    if fourier is None:
        _lib.check(_lib.load().nmrf_warp_corr_concat_f32(_p(labels), _p(f1), _p(f2), _p(g1), _p(g2), b, h, w, n, cf, cg,
                                                         groups, _p(out), ld, int(token_major), _stream()), "warp_corr_concat")
    else:
        _lib.check(_lib.load().nmrf_warp_corr_concat_fourier_f32(
            _p(labels), _p(f1), _p(f2), _p(g1), _p(g2), b, h, w, n, cf, cg, groups, _p(out), ld, int(token_major), float(normalizer),
            _p(enc), enc.shape[-1], _p(out_map), _stream()), "warp_corr_concat_fourier")
    _he("warp_corr_concat_n%d" % n)
    return out if fourier is None else (out, enc)


def to_kv16(qkv):
    """[T, 384] fp32 q | k | v -> the same tensor with k and v as split fp16 operand pairs (the kv16 format of include/nmrf_hip.h;
    torch restatement of what nmrf_nmp_block16_f32 writes with kv16 != 0: tests and tools)."""
    t = qkv.shape[0]
    out = qkv.clone()
    words = out.view(torch.int32)

    def split(x):
        hi = x.half()
        lo = (x - hi.float()).half()
        return hi.view(torch.int16).int() & 0xffff, lo.view(torch.int16).int() & 0xffff
    kh, kl = split(qkv[:, 128:256].reshape(t, 4, 32))
    pack2 = lambda a: a[..., 0::2] | (a[..., 1::2] << 16)                         # two fp16 per 32-bit word, little endian
    words[:, 128:256] = torch.cat((pack2(kh), pack2(kl)), -1).reshape(t, 128)     # per head: 16 words of hi, 16 words of lo
    vh, vl = split(qkv[:, 256:384])
    words[:, 256:384] = vh | (vl << 16)
    return out


@_on_device
def self_attn(qkv, n, heads):
    _chk(qkv)
    t, c3 = qkv.shape
    c = c3 // 3
    out = torch.empty(t, c, device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_self_attn_f32(_p(qkv), t, n, c, heads, _p(out), _stream()), "self_attn")
    return out


_TABLE_OK = {}


def _table_fits_fp16(table):
    """|table| < 32 and finite, read back once per (tensor object, version).  Keyed on the object's id WITH a weak reference to it: a
    (data_ptr, version) key alone is not an identity -- a new tensor allocated at a freed table's address would inherit its verdict."""
    import weakref
    ent = _TABLE_OK.get(id(table))
    if ent is not None and ent[0]() is table and ent[1] == (table._version, table.data_ptr()):
        return ent[2]
    if len(_TABLE_OK) > 256:
        for k in [k for k, e in _TABLE_OK.items() if e[0]() is None]:
            del _TABLE_OK[k]
    with torch.no_grad():
        ok = bool(torch.isfinite(table).all() and float(table.abs().max()) < 32.0)
    _TABLE_OK[id(table)] = (weakref.ref(table), (table._version, table.data_ptr()), ok)
    return ok


_TABLE_PACK = {}
W6_HEAD_BYTES = 49152          # csrc/window_attn6.hip


def window_table_packed(table, heads):
    """The relative-position table of a window-attention layer in the layout the persistent 6 x 6 x 4 kernel copies into LDS
    (nmrf_window_table_pack_f32), made once per (tensor object, version) like the range verdict above."""
    import weakref
    ent = _TABLE_PACK.get(id(table))
    key = (table._version, table.data_ptr())
    if ent is not None and ent[0]() is table and ent[1] == key:
        return ent[2]
    if len(_TABLE_PACK) > 256:
        for k in [k for k, e in _TABLE_PACK.items() if e[0]() is None]:
            del _TABLE_PACK[k]
    packed = torch.zeros(heads * W6_HEAD_BYTES, dtype=torch.uint8, device=table.device)
    _lib.check(_lib.load().nmrf_window_table_pack_f32(_p(table), heads * 32, heads, _p(packed), _stream()), "window_table_pack")
    _TABLE_PACK[id(table)] = (weakref.ref(table), key, packed)
    return packed


WINDOW6 = True      # tests / tools set False for the A/B against the two-windows-per-block kernel of rounds 2-5 (never read from the environment)


@_on_device
def window_attn(qkv, table, b, hp, wp, n, heads, win, shift, sibling_mask, checked=False, kv16=False):
    """checked: the producer of qkv range-checked it (no scan pass).  kv16: the k | v thirds of qkv are split fp16 operand pairs
    (nmp_block(q=dict(kv16=True)) / to_kv16; 6 x 6 windows of four labels; implies checked).  The kv16 kernel contracts q / k with
    the table on the fp16 matrix pipe (table staged x 2^10 as split fp16, window_attn.hip P0M): that form needs |table| < 32, checked
    HERE once per parameter version (one read-back) for every caller; a table beyond it (or non-finite) takes the same kernel with
    the relative-position dot products on the VALU in fp32 (kv16 = 2 at the C ABI) instead of returning inf / NaN."""
    _chk(qkv, table)
    if kv16:
        kv16 = 1 if _table_fits_fp16(table) else 2
    t, c3 = qkv.shape
    c = c3 // 3
    assert t == b * hp * wp * n
    out = torch.empty(t, c, device=qkv.device, dtype=torch.float32)
    tw = win * win * n                                 # reference form: 5 contractions of tw^2 x 32 MACs per (window, head)
    if kv16 == 1 and WINDOW6 and c == 128 and heads == 4 and qkv.numel() < (1 << 30):
        # the persistent kernel: one block per (CU, head) keeps the packed table in LDS, one wave per query tile
        packed = window_table_packed(table, heads)
        _hb("window_attn_w6_n4", row="A10", bound="mfma", split=True,
            flops=b * (hp // win) * (wp // win) * heads * 5 * 2.0 * tw * tw * 32, bytes=4.0 * (qkv.numel() + t * c),
            label="window_attn6_kernel (6 x 6 x 4 inference windows, persistent, A10)", pmc=["window_attn6_kernel<"])
        _lib.check(_lib.load().nmrf_window_attn6_f32(_p(qkv), _p(packed), b, hp, wp, c, heads, shift, int(bool(sibling_mask)), _p(out),
                                                     _stream()), "window_attn6")
        _he("window_attn_w6_n4")
        return out
    fast = {(6, 4): ("window_attn_fast_kernel<5, 6, 4, 2, 3, false, 1, true, true, false>" if kv16 == 1 else
                     "window_attn_fast_kernel<5, 6, 4, 2, 3, false, 1, true, false, false>" if kv16 else
                     "window_attn_fast_kernel<5, 6, 4, 2, 3, false, 1, false, false, false>"),
            (4, 1): "window_attn_fast_kernel<1, 4, 1, 4, 3,"}
    _hb("window_attn_w%d_n%d" % (win, n), row="A10" if n > 1 else "A13", bound="mfma", split=True,
        flops=b * (hp // win) * (wp // win) * heads * 5 * 2.0 * tw * tw * 32, bytes=4.0 * (qkv.numel() + t * c),
        label="%s (%s windows, %s)" % (fast.get((win, n), "window_attn_kernel<%d>" % ((tw + 31) // 32)).split("<")[0] +
                                       "<win %d, N %d>" % (win, n), "inference" if n > 1 else "refinement",
                                       "A10" if n > 1 else "A13"),
        pmc=[fast.get((win, n), "window_attn_kernel<%d>" % ((tw + 31) // 32))] + (["window_attn_fast_kernel<1, 4, 1, 8, 2, false, 2>"]
                                                                                    if (win, n) == (4, 1) else []))
    _lib.check(_lib.load().nmrf_window_attn_f32(_p(qkv), _p(table), b, hp, wp, n, c, heads, win, shift,
                                                int(bool(sibling_mask)), int(kv16), _p(out), None if (checked or kv16) else _rf(qkv),
                                                _stream()),
               "window_attn")
    _he("window_attn_w%d_n%d" % (win, n))
    return out


@_on_device
def linear_smalln(x, weight, bias=None, relu=False):
    """[T,K] x [N,K]^T (+bias, optional ReLU) for the narrow prediction heads (N <= 64)."""
    _chk(x, weight, bias)
    t, k = x.shape
    n = weight.shape[0]
    out = torch.empty(t, n, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_linear_smalln_f32(_p(x), _p(weight), _p(bias), t, k, n, int(relu), _p(out), _stream()),
               "linear_smalln")
    return out


@_on_device
def pack_linear_weight(weight):
    """[N,K] nn.Linear weight -> MFMA fragment order for token_linear (N % 32 == 0).  Debug library (include/nmrf_hip_debug.h)."""
    _chk(weight)
    n, k = weight.shape
    packed = torch.empty(n * ((k + 31) // 32 * 32) + n // 32, device=weight.device, dtype=torch.float32)
    _lib.check(_lib.load_debug().nmrf_pack_linear_weight_f32(_p(weight.contiguous()), n, k, _p(packed), _stream()), "pack_linear_weight")
    return packed


@_on_device
def token_linear(x, packed_w, n, k, bias=None, ln=None, y=None, extra=None, extra_div=1, act=0, residual=None):
    """fp32-MFMA reference linear of the tools / test build (libnmrf_hip_debug.so; NMRF_LINEAR=fp32 A/B runs).
    out[T,n] = act(P(x) W^T + bias) + residual on the fused MFMA kernel.
    ln = (gamma, beta, eps): P(x) = [LayerNorm(x + y) | extra[t // extra_div]]; returns (x + y, out) when y is given.
    ln = None: P(x) = x (k columns).  act: 0 / 'relu' / 'gelu'."""
    _chk(x, packed_w, bias, y, extra, residual)
    t, cx = x.shape
    act = {0: 0, 1: 1, 2: 2, None: 0, "relu": 1, "gelu": 2}[act]
    e = 0 if extra is None else extra.shape[-1]
    g = bt = None
    eps = 0.0
    if ln is not None:
        g, bt, eps = ln
        _chk(g, bt)
    out = torch.empty(t, n, device=x.device, dtype=torch.float32)
    x_out = torch.empty_like(x) if y is not None else None
    hook_name = None
    if kernel_hook is not None:
        hook_name = "token_linear_ln%d_k%d_n%d_act%d" % (int(ln is not None), k, n, act)
        kc, tf = (k + 31) // 32, lambda v: "true" if v else "false"
        pmc = (["token_linear_pipe_kernel<%d, %s, 2>" % (kc, tf(act == 2)), "token_linear_pipe_kernel<%d, %s, 1>" % (kc, tf(act == 2))]
               if ln is not None and residual is None and n % 128 == 0 else [])
        pmc.append("token_linear_kernel<%d, %s, %s>" % (kc, tf(ln is not None), tf(act == 2)))
        _hb(hook_name, row="A7/A10/A13 (N3)", bound="mfma", flops=2.0 * t * k * n, bytes=4.0 * t * (cx * (3 if y is not None else 1) + e + n),
            label="token_linear%s_kernel<%d,%s,%s> (%s%d->%d%s)" % ("_pipe" if len(pmc) > 1 else "", kc, "LN" if ln is not None else "plain",
                                                                  "GELU" if act == 2 else "-", "LayerNorm+" if ln is not None else "", k, n,
                                                                  "+GELU" if act == 2 else ""), pmc=pmc)
    _lib.check(_lib.load_debug().nmrf_token_linear_f32(_p(x), _p(y), _p(x_out), _p(g), _p(bt), float(eps), _p(extra), e, extra_div,
                                                 _p(packed_w), _p(bias), _p(residual), act, t, cx, k, n, _p(out), _stream()),
               "token_linear")
    if hook_name is not None:
        _he(hook_name)
    return (x_out, out) if y is not None else out


# max |w| of the weights about to be packed, fetched for MANY tensors with one read-back (nmrf_amd.train.train_step: the optimizer bumps
# every parameter's version each step, so every launch site re-packs -- one `float(w.abs().max())` each was 89 device read-backs, 89
# reductions and 107 abs kernels per training step).  Keyed on (data_ptr, version) of live parameters; the same value bit for bit.
_AMAX_CACHE = {}


def prefetch_amax(tensors):
    ts = [t for t in tensors if t.is_cuda and t.dtype == torch.float32 and t.numel() > 0]
    _AMAX_CACHE.clear()
    if not ts:
        return 0
    with torch.no_grad():
        try:
            vals = torch._foreach_norm(ts, float("inf"))
        except (RuntimeError, TypeError):
            vals = [t.abs().max() for t in ts]
        host = torch.stack([v.reshape(()) for v in vals]).tolist()             # ONE synchronising read
    for t, v in zip(ts, host):
        _AMAX_CACHE[(t.data_ptr(), t._version)] = float(v)
    return len(ts)


def cached_amax(t):
    """The prefetched max |t| of a live parameter (this version), or None."""
    return _AMAX_CACHE.get((t.data_ptr(), t._version))


def _amax(weight, amax=None):
    if amax is None:
        amax = cached_amax(weight)
    return float(weight.abs().max()) if amax is None else float(amax)


@_on_device
def pack_split_weight(weight, kp=None):
    """[N,K] nn.Linear weight -> (split-fp16 MFMA fragment pairs [N/32, Kp/16, 512] (int32 view of 2 KB pairs), 1/scale) for
    nmp_block.  scale = the power of two that brings max|w| into [2^13, 2^14) (host-side, once per parameter version)."""
    import math
    _chk(weight)
    n, k = weight.shape
    kp = kp or (k + 15) // 16 * 16
    amax = _amax(weight)
    scale = 1.0 if not (amax > 0 and math.isfinite(amax)) else 2.0 ** min(40, max(-40, math.floor(math.log2(16383.0 / amax))))
    out = torch.empty(n // 32, kp // 16, 512, device=weight.device, dtype=torch.int32)
    _lib.check(_lib.load().nmrf_pack_split_weight_f32(_p(weight), n, k, kp, scale, _p(out), _stream()), "pack_split_weight")
    return out, 1.0 / scale


def _pack_scaled(weight, kp, fn_name, rows, kchunk, amax=None):
    import math
    _chk(weight)
    n, k = weight.shape
    amax = _amax(weight, amax)
    scale = 1.0 if not (amax > 0 and math.isfinite(amax)) else 2.0 ** min(40, max(-40, math.floor(math.log2(16383.0 / amax))))
    out = torch.empty(n // rows, kp // kchunk, 512, device=weight.device, dtype=torch.int32)
    _lib.check(getattr(_lib.load(), fn_name)(_p(weight), n, k, kp, scale, _p(out), _stream()), fn_name)
    return out, 1.0 / scale


@_on_device
def pack_split_weight16(weight, kp, amax=None):
    """[N,K] -> (pairs [N/16, Kp/32, 512] for nmp_block16 (16-row strips x 32-deep chunks), 1/scale).  amax: max |weight| if the caller
    knows it (a concatenation of parameters whose maxima were prefetched)."""
    return _pack_scaled(weight, kp, "nmrf_pack_split_weight16_f32", 16, 32, amax)


def block_stream16(wp=None, w1=None, w2=None, wq=None, kq=0, wq_amax=None):
    """Weight stream of one nmp_block16 launch (include/nmrf_hip.h): proj | W1 strip pairs interleaved with W2 k chunks | q;
    within proj / W1 / q two adjacent 16-row strips are interleaved chunk by chunk (the kernel feeds them to two accumulators)."""
    import ctypes
    parts, inv = [], [1.0, 1.0, 1.0, 1.0]
    def ilv(pk):                                                          # [strips][chunks] -> strip pairs interleaved chunk by chunk
        n, c = pk.shape[0], pk.shape[1]
        return pk.view(n // 2, 2, c, 512).permute(0, 2, 1, 3).contiguous()
    if wp is not None:
        pk, inv[0] = pack_split_weight16(wp, 128)
        parts.append(ilv(pk).view(-1, 512))
    if w1 is not None:
        p1, inv[1] = pack_split_weight16(w1, 128)                          # [32 strips][4 chunks] = [16 groups][8 pairs]
        p2, inv[2] = pack_split_weight16(w2, 512)                          # [8 strips][16 chunks]
        p1 = ilv(p1).view(16, 8, 512)
        p2 = p2.permute(1, 0, 2).contiguous()                             # [16 groups][8 strips]
        seq = [p1[0]]
        for h in range(15):
            seq += [p1[h + 1], p2[h]]
        seq.append(p2[15])
        parts.append(torch.stack(seq).view(-1, 512))
    if wq is not None:
        pk, inv[3] = pack_split_weight16(wq, kq, wq_amax)
        parts.append(ilv(pk).view(-1, 512))
    stream = torch.cat(parts).contiguous()
    assert stream.shape[0] % 8 == 0
    return stream, stream.shape[0] // 8, (ctypes.c_float * 4)(*inv)


def block_stream(wp=None, w1=None, w2=None, wq=None, kq=0):
    """Weight stream of one nmp_block launch, in the kernel's consumption order (include/nmrf_hip.h, nmrf_nmp_block_f32):
    proj pairs | W1 strips interleaved with W2 k-slices | q-stage pairs.
    Returns (int32 tensor [stages*8, 512], stages, (1/scale of proj, fc1, fc2, q))."""
    import ctypes
    parts = []
    inv = [1.0, 1.0, 1.0, 1.0]
    if wp is not None:
        pk, inv[0] = pack_split_weight(wp, 128)
        parts.append(pk.view(-1, 512))
    if w1 is not None:
        p1, inv[1] = pack_split_weight(w1, 128)                                          # [16 hidden strips][8 chunks]
        p2, inv[2] = pack_split_weight(w2, 512)
        p2 = p2.view(4, 16, 2, 512).permute(1, 0, 2, 3).reshape(16, 8, 512)             # [hidden strip][n, c]
        seq = [p1[0]]
        for h in range(15):
            seq += [p1[h + 1], p2[h]]
        seq.append(p2[15])
        parts.append(torch.stack(seq).view(-1, 512))
    if wq is not None:
        pk, inv[3] = pack_split_weight(wq, kq)
        parts.append(pk.view(-1, 512))
    stream = torch.cat(parts).contiguous()
    assert stream.shape[0] % 8 == 0
    return stream, stream.shape[0] // 8, (ctypes.c_float * 4)(*inv)


def chain_stream(weights, kps):
    """Weight stream of one mlp_chain launch: the layers' pairs in order, rows zero-padded to a multiple of 32, zero pairs up to a
    whole stage.  weights: list of [N,K] tensors; kps: padded K per layer.  -> (int32 [stages*8, 512], stages, 1/scales x3)"""
    import ctypes
    parts, inv = [], [1.0, 1.0, 1.0]
    for i, (w, kp) in enumerate(zip(weights, kps)):
        n = w.shape[0]
        if n % 32:
            w = torch.nn.functional.pad(w, (0, 0, 0, 32 - n % 32))
        pk, inv[i] = pack_split_weight(w.contiguous(), kp)
        parts.append(pk.view(-1, 512))
    stream = torch.cat(parts)
    if stream.shape[0] % 8:
        stream = torch.cat((stream, stream.new_zeros(8 - stream.shape[0] % 8, 512)))
    return stream.contiguous(), stream.shape[0] // 8, (ctypes.c_float * 3)(*inv)


@_on_device
def mlp_chain(kind, x, k1, stream, stages, inv_scales, biases, n_out, extra=None, out=None, out_map=None, row_add=None, relu_out=False):
    """nmrf_mlp_chain_f32: kind 0 ffn | 1 seed embed | 2 three-layer ReLU head | 3 single Linear.  x [T, ld] (k1 live columns).
    row_add [T, >= n_out] / relu_out: out = [relu](chain(x) + row_add) in the kernel's store pass."""
    _chk(x, extra, out, row_add, *[b for b in biases if b is not None])
    _chk(stream, out_map, dtype=torch.int32)
    t, ld = x.shape
    if out is None:
        out = torch.empty(t, n_out, device=x.device, dtype=torch.float32)
    b = list(biases) + [None] * (3 - len(biases))
    name = "mlp_chain_kind%d" % kind
    if kernel_hook is not None:
        widths = {0: k1 * 128 + 128 * 128, 1: k1 * 128 + 128 * 128 + 160 * 128, 2: 2 * 128 * 128 + 128 * n_out, 3: 128 * n_out}[kind]
        _hb(name + "_n%d" % n_out, row="A6/A9/A11/A13/A14 (N3)", bound="mfma", flops=2.0 * t * widths, bytes=4.0 * t * (k1 + n_out), split=True,
            label="mlp_chain_kernel kind %d (%s, split-fp16 MFMA)" % (kind, ("ffn 160->128->128", "seed embed 36->128->128|32->128",
                                                                            "head 128->128->128->%d" % n_out, "Linear 128->%d" % n_out)[kind]),
            pmc=["mlp_chain_kernel<%s" % {0: "10, 4, 2, true, 0, 0, 0,", 1: "3, 4, 2, true, 0, 10, 4,", 2: "8, 4, 1, true, 1, 8, %d," % (2 if n_out > 32 else 1),
                                          3: "8, %d, 0, false, 0, 0, 0," % (2 if n_out > 32 else 1)}[kind]])
    _lib.check(_lib.load().nmrf_mlp_chain_f32(kind, _p(x), ld, k1, _p(stream), stages, _p(b[0]), _p(b[1]), _p(b[2]), _p(extra),
                                              0 if extra is None else extra.shape[-1], inv_scales, t, _p(out), out.shape[-1], n_out,
                                              _p(out_map), _p(row_add), 0 if row_add is None else row_add.shape[-1], int(bool(relu_out)),
                                              _rf(x), _stream()), "mlp_chain")
    if kernel_hook is not None:
        _he(name + "_n%d" % n_out)
    return out


@_on_device
def refine_head_epilogue(tgt, stream, stages, inv_scales, biases, disp_curr, out_h, out_w):
    """nmrf_refine_head_epilogue_f32: refine_head (chain_stream of its three layers, n_out 16) + refine_epilogue in one launch.
    tgt [B*H4*W4, 128], disp_curr [B,H4,W4] -> (disp [B,out_h,out_w], disp_pred [B,4H4,4W4])."""
    _chk(tgt, disp_curr, *[x for x in biases if x is not None])
    _chk(stream, dtype=torch.int32)
    b, h4, w4 = disp_curr.shape
    assert tgt.shape == (b * h4 * w4, 128)
    pred = torch.empty(b, 4 * h4, 4 * w4, device=tgt.device, dtype=torch.float32)
    disp = torch.empty(b, out_h, out_w, device=tgt.device, dtype=torch.float32)
    b1, b2, b3 = biases
    _hb("refine_head_epilogue", row="A14 (N3)", bound="mfma", flops=2.0 * tgt.shape[0] * (2 * 128 * 128 + 128 * 16),
        bytes=4.0 * tgt.shape[0] * 129 + 4.0 * (pred.numel() + disp.numel()), split=True,
        label="mlp_chain_kernel EPI form (refine head 128->128->128->16 + pixel shuffle + crop, split-fp16 MFMA)",
        pmc=["mlp_chain_kernel<8, 4, 1, true, 1, 8, 1, false, true>"])
    _lib.check(_lib.load().nmrf_refine_head_epilogue_f32(_p(tgt), b, h4, w4, _p(stream), stages, _p(b1), _p(b2), _p(b3), inv_scales,
                                                         _p(disp_curr), out_h, out_w, _p(pred), _p(disp), _rf(tgt), _stream()),
               "refine_head_epilogue")
    _he("refine_head_epilogue")
    return disp, pred


@_on_device
def nmp_block_pair(x, msg, stream, stages, inv_scales, bp, mlp, q, bp2, q2, want_x=True, ln_out=None, ln_out_map=None):
    """nmrf_nmp_block16_pair_f32: a full block (proj + MLP -> the self-edge q | k | v) and the self-edge block behind it in one launch.
    stream / inv_scales: the two launches' weight streams back to back, 6 floats.  mlp = (ln2_gamma, ln2_beta, eps, b1, b2);
    q / q2 = dict(g, b, eps, extra, extra_div, bias[, nq, kv16, ln_out]) of the first / second block.  Returns what the second launch
    would: (x_out | None, q_out | None, ln_out | None)."""
    _chk(x, msg, bp, bp2)
    _chk(stream, dtype=torch.int32)
    t = x.shape[0]
    ln2_g, ln2_b, eps2, b1, b2 = mlp if mlp is not None else (None, None, 0.0, None, None)     # (msg is None: a q stage alone in front)
    _chk(ln2_g, ln2_b, b1, b2, q["g"], q["b"], q["extra"], q.get("bias"), q2["g"], q2["b"], q2["extra"], q2.get("bias"))
    nq2, want_ln = q2.get("nq", 0), q2.get("ln_out", False)
    x_out = torch.empty_like(x) if want_x else None
    q_out = torch.empty(t, nq2, device=x.device, dtype=torch.float32) if nq2 else None
    if ln_out is None and want_ln:
        ln_out = torch.empty_like(x)
    flops = 2.0 * t * ((128 * 128 + 2 * 128 * 512 if msg is not None else 0) + 160 * 384 + 128 * 128 + 160 * nq2)
    _hb("nmp_block_pair" if msg is not None else "nmp_block_pair_entry", row="A10 (N3)", bound="mfma", split=True, flops=flops, bytes=4.0 * t * (2 * 128 + 32 + 128 + nq2),
        label="nmp_block16_kernel<true,5,FUSE> (full block + the self-edge block behind it in one launch)",
        pmc=["nmp_block16_kernel<true, 5, 0, true>" if msg is not None else "nmp_block16_kernel<false, 5, 0, true>"])
    _lib.check(_lib.load().nmrf_nmp_block16_pair_f32(
        _p(x), _p(msg), _p(stream), stages, _p(bp), _p(ln2_g), _p(ln2_b), float(eps2), _p(b1), _p(b2),
        _p(q["g"]), _p(q["b"]), float(q["eps"]), _p(q["extra"]), q["extra"].shape[-1], q.get("extra_div", 1), _p(q.get("bias")),
        _p(bp2), _p(q2["g"]), _p(q2["b"]), float(q2["eps"]), _p(q2["extra"]), q2["extra"].shape[-1], q2.get("extra_div", 1),
        _p(q2.get("bias")), nq2, t, inv_scales, _p(x_out), _p(q_out), _p(ln_out), _p(ln_out_map), int(bool(q2.get("kv16", False))),
        _rf(x), _stream()), "nmp_block_pair")
    _he("nmp_block_pair" if msg is not None else "nmp_block_pair_entry")
    return x_out, q_out, ln_out


def heads_wta_stream(w1, w2, w3, ws):
    """Weight stream of heads_wta: the pairs of W1, Ws, W2, W3 (the score layer runs on layer 1's operand, right behind it).
    -> (int32 [96, 512], 12, 1/scales of W1, W2, W3, Ws)"""
    import ctypes
    parts, inv = [], {}
    for name, w in (("1", w1), ("s", ws), ("2", w2), ("3", w3)):
        pk, inv[name] = pack_split_weight(w.contiguous(), 128)
        parts.append(pk.view(-1, 512))
    stream = torch.cat(parts).contiguous()
    assert stream.shape[0] == 96
    return stream, 12, (ctypes.c_float * 4)(inv["1"], inv["2"], inv["3"], inv["s"])


@_on_device
def heads_wta(tgt, stream, stages, inv_scales, biases, labels, b, h, w, n=4):
    """nmrf_heads_wta_f32: infer_head + infer_score_head + winner-take-all + x2 + 4x4 lower medians in one launch.
    tgt [T,128], labels [T], biases = (b1, b2, b3, bs) -> disp_curr [B, 2H, 2W]."""
    _chk(tgt, labels, *[x for x in biases if x is not None])
    _chk(stream, dtype=torch.int32)
    t = tgt.shape[0]
    assert t == b * h * w * n and tgt.shape[1] == 128 and labels.numel() == t
    out = torch.empty(b, 2 * h, 2 * w, device=tgt.device, dtype=torch.float32)
    _hb("heads_wta", row="A11/A12 (N3)", bound="mfma", flops=2.0 * t * (2 * 128 * 128 + 2 * 128 * 64), bytes=4.0 * t * 129 + 4.0 * out.numel(), split=True,
        label="mlp_chain_kernel WTA form (head 128->128->128->64 + score 128->64 + winner-take-all + medians, split-fp16 MFMA)",
        pmc=["mlp_chain_kernel<8, 4, 1, true, 1, 8, 2, true, false>"])
    b1, b2, b3, bs = biases
    _lib.check(_lib.load().nmrf_heads_wta_f32(_p(tgt), b, h, w, n, _p(stream), stages, _p(b1), _p(b2), _p(b3), _p(bs), inv_scales,
                                              _p(labels), _p(out), _rf(tgt), _stream()), "heads_wta")
    _he("heads_wta")
    return out


@_on_device
def nmp_block(x, stream, stages, inv_scales, msg=None, bp=None, mlp=None, q=None, want_x=True, ln_out=None, ln_out_map=None,
              tokens_per_wave=16, attn_qkv=None):
    """One fused message-passing block (nmrf_nmp_block16_f32; tokens_per_wave=32: the debug library's 32-token form).
    mlp = (ln2_gamma, ln2_beta, eps, b1, b2) or None;  q = dict(g, b, eps, extra=None, extra_div=1, bias=None, kq=0|128|160|192,
    nq=0 -> no q_out, ln_out=False, kv16=False: k | v of q_out as split fp16 pairs, include/nmrf_hip.h) or None.  attn_qkv [T,384] (instead of msg, proj-only blocks): q | k | v of the self-edge
    attention among the 4 sibling labels of a pixel, evaluated on the way in.  Returns (x_out | None, q_out | None, ln_out | None)."""
    _chk(x, msg, bp, attn_qkv)
    _chk(stream, dtype=torch.int32)
    t = x.shape[0]
    ln2_g = ln2_b = b1 = b2 = None
    eps2 = 0.0
    if mlp is not None:
        ln2_g, ln2_b, eps2, b1, b2 = mlp
        _chk(ln2_g, ln2_b, b1, b2)
    lq_g = lq_b = extra = bq = None
    epsq, kq, nq, ld, div, want_ln = 0.0, 0, 0, 0, 1, False
    if q is not None:
        lq_g, lq_b, epsq = q["g"], q["b"], q["eps"]
        extra, div, bq, kq, nq = q.get("extra"), q.get("extra_div", 1), q.get("bias"), q["kq"], q.get("nq", 0)
        want_ln = q.get("ln_out", False)
        _chk(lq_g, lq_b, extra, bq)
        if extra is not None:
            ld = extra.shape[-1]
    x_out = torch.empty_like(x) if want_x else None
    q_out = torch.empty(t, nq, device=x.device, dtype=torch.float32) if nq else None
    if want_ln and ln_out is None:
        ln_out = torch.empty_like(x)
    _chk(ln_out)
    _chk(ln_out_map, dtype=torch.int32)
    if kernel_hook is not None:
        has_msg = msg is not None or attn_qkv is not None
        flops = 2.0 * t * ((128 * 128 if has_msg else 0) + (2 * 128 * 512 if mlp is not None else 0) + kq * nq)
        nbytes = 4.0 * t * (128 * (1 + (msg is not None) + 3 * (attn_qkv is not None) + bool(want_x) + bool(want_ln)) + nq) + (
            4.0 * extra.numel() if extra is not None else 0)
        name = "nmp_block_p%d_m%d_q%dx%d" % (1 if msg is not None else (2 if attn_qkv is not None else 0), mlp is not None, kq, nq)
        _hb(name, row="A7/A10/A13 (N3)", bound="mfma", flops=flops, bytes=nbytes, split=True,
            label="%s<%s,%d> (%s%s%s fused, split-fp16 MFMA, %d tokens per wave)" % (
                "nmp_block_kernel" if tokens_per_wave == 32 else "nmp_block16_kernel",
                "true" if mlp is not None else "false", kq // 16 if tokens_per_wave == 32 else (kq + 31) // 32,
                "proj+residual " if msg is not None else ("self-edge attention+proj+residual " if attn_qkv is not None else ""),
                "LN+fc1+GELU+fc2 " if mlp is not None else "", ("LN+%d->%d" % (kq, nq)) if nq else ("final LN" if want_ln else ""),
                tokens_per_wave),
            pmc=["nmp_block_kernel<%s, %d, 1, 4, false, 0>" % ("true" if mlp is not None else "false", kq // 16) if tokens_per_wave == 32
                 else "nmp_block16_kernel<%s, %d, 0, false>" % ("true" if mlp is not None else "false", (kq + 31) // 32)])
    head = (_p(stream), stages, _p(bp), _p(ln2_g), _p(ln2_b), float(eps2), _p(b1), _p(b2), _p(lq_g), _p(lq_b), float(epsq), _p(extra), ld,
            div, _p(bq), int(mlp is not None), kq, nq, t, inv_scales, _p(x_out), _p(q_out), _p(ln_out), _p(ln_out_map))
    kv16 = int(bool(q is not None and q.get("kv16", False)))
    if tokens_per_wave == 32:
        if attn_qkv is not None or kv16:
            raise ValueError("the 32-token debug form takes the message as a tensor and writes fp32 q | k | v")
        _lib.check(_lib.load_debug().nmrf_nmp_block_f32(_p(x), _p(msg), *head, _rf(x), _stream()), "nmp_block")
    else:
        _lib.check(_lib.load().nmrf_nmp_block16_f32(_p(x), _p(msg), _p(attn_qkv), 4 if attn_qkv is not None else 0, *head, kv16, _rf(x),
                                                    _stream()), "nmp_block")
    if kernel_hook is not None:
        _he(name)
    return x_out, q_out, ln_out


@_on_device
def wino_pack_filter(weight):
    """[Co,Ci,3,3] conv weight -> Winograd-domain filter in MFMA fragment order for conv3x3_wino."""
    _chk(weight)
    co, ci, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    packed = torch.empty(16 * co * ci, device=weight.device, dtype=torch.float32)
    _lib.check(_lib.load_debug().nmrf_wino_pack_filter_f32(_p(weight.contiguous()), co, ci, _p(packed), _stream()), "wino_pack_filter")
    return packed


@_on_device
def conv3x3_wino(x, packed_u, co):
    """3x3 / stride 1 / pad 1 / no-bias convolution (NCHW fp32) as fused Winograd F(2x2,3x3) on fp32 MFMA: the round-1 kernel, kept
    in the debug library as a reference (include/nmrf_hip_debug.h); the product runs conv3x3_split."""
    _chk(x, packed_u)
    b, ci, h, w = x.shape
    y = torch.empty(b, co, h, w, device=x.device, dtype=torch.float32)
    # N2 (stock-conv band, not a north_star hot-path row).  SURVEY 8(d) counts the direct convolution's FLOPs; the kernel
    # executes the Winograd F(2x2,3x3) form = 1/2.25 of those multiplies, which is what its MFMA roofline is priced on
    _hb("conv3x3_wino", row="N2", bound="mfma", flops=2.0 * 9 * b * ci * co * h * w / 2.25, direct_flops=2.0 * 9 * b * ci * co * h * w,
        bytes=4.0 * (x.numel() + y.numel()), label="conv3x3_wino_kernel (3x3 stride-1 convs of the backbone / conv heads, N2; "
        "mean over layers; Winograd-form FLOPs)", pmc=["conv3x3_wino_kernel"])
    _lib.check(_lib.load_debug().nmrf_conv3x3_wino_f32(_p(x), _p(packed_u), b, ci, h, w, co, _p(y), _stream()), "conv3x3_wino")
    _he("conv3x3_wino")
    return y


def _conv3_plan(co, tiles):
    """(strips, groups) of conv3x3_split for `co` output channels: 32-channel strips per block x channel groups (blockIdx.y).
    Two strips per block (three resident blocks per CU, most blocks) unless the launch still fills the chip with four (the
    halo is then staged once for 128 channels instead of twice; measured at KITTI B=1: 128->256 at 1/4 res, 480 blocks of four
    strips 93 us vs 960 of two 103 us; 128->128 at 1/4 res, 240 vs 480 blocks: 56 vs 53 us; 128->256 at 1/8 res: 47 vs 34 us)."""
    k = co // 32
    if k % 4 == 0 and tiles * (k // 4) >= 400:
        return 4, k // 4
    if k % 2 == 0:
        return 2, k // 2
    if k % 3 == 0:
        return 3, k // 3
    return None


@_on_device
def pack_conv3x3(weight, strips, groups):
    """[Co,Ci,KT,KT] conv weight (KT = 3 or 4) -> (stream [groups, KT*KT*Ci/16, strips, 512] int32, 1/scale) for conv3x3_split /
    conv_split: the matrix Wm[co][((ci/16 * KT + dy) * KT + dx) * 16 + ci%16] as split-fp16 MFMA fragment pairs, chunk-major
    within a channel group."""
    co, ci, kt = weight.shape[0], weight.shape[1], weight.shape[2]
    wm = weight.reshape(co, ci // 16, 16, kt, kt).permute(0, 1, 3, 4, 2).reshape(co, kt * kt * ci).contiguous()
    pk, inv = pack_split_weight(wm, kt * kt * ci)                             # [co/32, kt*kt*ci/16, 512]
    return pk.view(groups, strips, kt * kt * ci // 16, 512).permute(0, 2, 1, 3).contiguous(), inv


@_on_device
def conv_split(x, packed, co, kt=3, stride=1, pad=1, stats=None, eps=1e-5):
    """KTxKT convolution (NCHW fp32, no bias, padding `pad` before and KT-1-pad after) as a direct implicit GEMM on the
    split-operand fp16 MFMA (csrc/conv3x3.hip); stats (instance_stats(x)): conv(relu(InstanceNorm(x))) with the normalisation
    folded into the operand load.  packed = (stream, strips, groups, 1/scale) of pack_conv3x3."""
    stream, strips, groups, inv = packed
    _chk(x, stats)
    _chk(stream, dtype=torch.int32)
    b, ci, h, w = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty(b, co, ho, wo, device=x.device, dtype=torch.float32)
    name = "conv_split_k%d_s%d_%d" % (kt, stride, strips)
    if kt == 3 and stride == 1:
        label = ("conv3x3_split_kernel<%d,3,1> (3x3 stride-1 convs of the backbone / conv heads, N2; mean over layers; "
                 "direct-form FLOPs)" % strips)
    else:
        label = "conv3x3_split_kernel<%d,%d,%d> (%s, N2)" % (strips, kt, stride, "7x7/2 stem as 4x4 over space-to-depth" if kt == 4
                                                             else "3x3 stride-2 conv of layer2")
    _hb(name, row="N2", bound="mfma", flops=2.0 * kt * kt * b * ci * co * ho * wo, bytes=4.0 * (x.numel() + y.numel()), split=True,
        label=label, pmc=["conv3x3_split_kernel<%d, %d, %d," % (strips, kt, stride)])
    _lib.check(_lib.load().nmrf_conv_split_f32(_p(x), b, ci, h, w, _p(stats), 0 if stats is None else stats.shape[1], float(eps),
                                               _p(stream), kt, stride, pad, strips, groups, float(inv), co, _p(y), _rf(x), _stream()),
               "conv_split")
    _he(name)
    return y


def conv3x3_split(x, packed, co, stats=None, eps=1e-5):
    """3x3 / stride 1 / pad 1 / no-bias convolution: conv_split with its defaults."""
    return conv_split(x, packed, co, 3, 1, 1, stats, eps)


def _cached_pack(cache, key, make):
    if cache.get("key") != key:
        with torch.no_grad():
            cache["packed"] = make()
        cache["key"] = key
    return cache["packed"]


def conv3x3_s2_auto(x, weight, cache):
    """3x3 / stride 2 / pad 1 / no-bias convolution (layer2.0.conv1): split-fp16 MFMA kernel; channel counts it is not built for
    go to the stock torch convolution (MIOpen)."""
    co, ci = weight.shape[0], weight.shape[1]
    k = co // 32
    strips = 3 if k % 3 == 0 else (2 if k % 2 == 0 else 0)
    if not x.is_cuda or x.dtype != torch.float32 or ci % 16 or co % 32 or not strips:
        return torch.nn.functional.conv2d(x, weight, None, 2, 1)
    groups = k // strips
    stream, inv = _cached_pack(cache, (weight.data_ptr(), weight._version, "s2"), lambda: pack_conv3x3(weight, strips, groups))
    return conv_split(x.contiguous(), (stream, strips, groups, inv), co, 3, 2, 1)


def stem_s2d_weight(weight):
    """[Co,3,7,7] stride-2 / pad-3 stem filter -> [Co,16,4,4] filter of the equivalent stride-1 convolution (pad 2 before, 1 after)
    over the 2x2 space-to-depth image (channel c*4 + p*2 + q = pixel (2Y+p, 2X+q) of colour c):
    W'[co, c*4+p*2+q, ty, tx] = W[co, c, 2*ty+p-1, 2*tx+q-1] (zero where that index leaves 0..6)."""
    co = weight.shape[0]
    wp = torch.zeros(co, 3, 8, 8, device=weight.device, dtype=weight.dtype)
    wp[:, :, 1:, 1:] = weight                                                # index k+1 = 2*t + p
    w2 = wp.view(co, 3, 4, 2, 4, 2).permute(0, 1, 3, 5, 2, 4).reshape(co, 12, 4, 4)       # [co, (c,p,q), ty, tx]
    out = torch.zeros(co, 16, 4, 4, device=weight.device, dtype=weight.dtype)
    out[:, :12] = w2
    return out


@_on_device
def prep_images_s2d(img1, img2, hp, wp):
    """prep_images written as the space-to-depth image of the stem: [2B,16,hp/2,wp/2].  Images float32 or uint8 (0..255)."""
    _chk(img1, img2, dtype=img1.dtype if img1.dtype == torch.uint8 else torch.float32)
    b, c, h, w = img1.shape
    assert c == 3 and hp % 2 == 0 and wp % 2 == 0
    out = torch.empty(2 * b, 16, hp // 2, wp // 2, device=img1.device, dtype=torch.float32)
    fn = _lib.load().nmrf_prep_images_s2d_u8 if img1.dtype == torch.uint8 else _lib.load().nmrf_prep_images_s2d_f32
    _lib.check(fn(_p(img1), _p(img2), b, h, w, hp, wp, _p(out), _stream()), "prep_images_s2d")
    return out


def stem_conv_s2d(x_s2d, weight, cache):
    """The 7x7 / stride-2 / pad-3 stem convolution on the space-to-depth image (prep_images_s2d): [2B,16,H/2,W/2] -> [2B,Co,H/2,W/2]."""
    co = weight.shape[0]
    stream, inv = _cached_pack(cache, (weight.data_ptr(), weight._version, "stem"),
                               lambda: pack_conv3x3(stem_s2d_weight(weight), 2, co // 64))
    return conv_split(x_s2d, (stream, 2, co // 64, inv), co, 4, 1, 2)


def conv3x3_auto(x, weight, cache, stats=None):
    """3x3 / stride 1 / pad 1 / no-bias convolution [of relu(InstanceNorm(x)) when `stats` = instance_stats(x) is given] on the
    direct split-fp16 MFMA kernel (conv3x3.hip).  Channel counts the kernel is not built for (Ci % 16, Co % 32, or a normalised
    input wider than the 256-channel affine table) go to the stock torch convolution (MIOpen).
    `cache`: a dict owned by the caller, holds the packed filter per weight version."""
    co, ci = weight.shape[0], weight.shape[1]
    b, _, h, w = x.shape
    key = (weight.data_ptr(), weight._version)
    if x.is_cuda and x.dtype == torch.float32 and ci % 16 == 0 and co % 32 == 0 and (stats is None or ci <= 256):
        plan = _conv3_plan(co, b * ((h + 7) // 8) * ((w + 31) // 32))
        if plan is not None:
            if cache.get("key") != key + plan:
                with torch.no_grad():
                    cache["packed"] = pack_conv3x3(weight, *plan)
                cache["key"] = key + plan
            stream, inv = cache["packed"]
            return conv3x3_split(x.contiguous(), (stream, plan[0], plan[1], inv), co, stats)
    if stats is not None:                                  # the stock path takes the normalised activation
        x = instance_norm(x.contiguous(), relu=True)
    return torch.nn.functional.conv2d(x, weight, None, 1, 1)


@_on_device
def superpixel_downsample(disp, labels, k=4):
    """A16 (parity unpinned, see include/nmrf_hip.h): disp [B,H,W] f32 (0 invalid), labels [B,H,W] int32 -> [B,H//8,W//8,k]."""
    _chk(disp)
    _chk(labels, dtype=torch.int32)
    b, h, w = disp.shape
    out = torch.empty(b, h // 8, w // 8, k, device=disp.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_superpixel_downsample_f32(_p(disp), _p(labels), b, h, w, k, _p(out), _stream()),
               "superpixel_downsample")
    return out


@_on_device
def wta_median(delta, score, labels, b, h, w, n):
    _chk(delta, score, labels)
    out = torch.empty(b, 2 * h, 2 * w, device=delta.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_wta_median_f32(_p(delta), _p(score), _p(labels), b, h, w, n, _p(out), _stream()),
               "wta_median")
    return out


@_on_device
def refine_epilogue(delta, disp_curr, out_h, out_w):
    _chk(delta, disp_curr)
    b, h4, w4 = disp_curr.shape
    pred = torch.empty(b, 4 * h4, 4 * w4, device=delta.device, dtype=torch.float32)
    disp = torch.empty(b, out_h, out_w, device=delta.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_refine_epilogue_f32(_p(delta), _p(disp_curr), b, h4, w4, out_h, out_w, _p(pred), _p(disp),
                                                    _stream()), "refine_epilogue")
    return disp, pred


@_on_device
def instance_norm(x, relu=False, residual=None, relu_out=False, eps=1e-5):
    """InstanceNorm2d (no affine) of an NCHW tensor fused with ReLU / residual add / ReLU:
    y = [relu] IN(x); y = [relu_out](y + residual)."""
    _chk(x, residual)
    b, c, h, w = x.shape
    hw = h * w
    ws = torch.empty(2 * b * c * ((hw + 8191) // 8192), device=x.device, dtype=torch.float32)
    y = torch.empty_like(x)
    _lib.check(_lib.load().nmrf_instance_norm_f32(_p(x), _p(residual), b * c, hw, float(eps), int(relu), int(relu_out),
                                                  _p(ws), _p(y), _stream()), "instance_norm")
    return y


@_on_device
def instance_apply(x, stats, relu=False, residual=None, relu_out=False, residual_stats=None, residual_relu=False, eps=1e-5):
    """The apply pass of instance_norm on given statistics (instance_stats(x)); with residual_stats the residual operand is a raw
    convolution output that is normalised (+ ReLU) on the way in: y = [relu_out]([relu] IN(x) + [relu] IN(residual))."""
    _chk(x, residual, stats, residual_stats)
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().nmrf_instance_apply_f32(_p(x), _p(stats), _p(residual), _p(residual_stats), int(residual_relu), b * c, h * w,
                                                   float(eps), int(relu), int(relu_out), _p(y), _stream()), "instance_apply")
    return y


@_on_device
def instance_stats(x):
    """Statistics pass of InstanceNorm2d: x [B,C,H,W] -> per-chunk (mean, M2) workspace for conv1x1_in_relu."""
    _chk(x)
    b, c, h, w = x.shape
    ws = torch.empty(b * c, (h * w + 8191) // 8192, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_instance_stats_f32(_p(x), b * c, h * w, _p(ws), _stream()), "instance_stats")
    return ws


@_on_device
def conv1x1_in_relu(x, c0, k, stats, packed, bias=None, eps=1e-5, token_major=False):
    """out = Conv1x1(relu(InstanceNorm(x[:, c0:c0+k])))  (stats None: Conv1x1(x[:, c0:c0+k])).  packed = (stream, stages, 1/scale, N).
    token_major: the result is [B,H,W,N] (a pixel's channels contiguous) instead of [B,N,H,W]."""
    _chk(x, stats, bias)
    stream, stages, inv, n = packed
    _chk(stream, dtype=torch.int32)
    b, cx, h, w = x.shape
    out = torch.empty((b, h, w, n) if token_major else (b, n, h, w), device=x.device, dtype=torch.float32)
    _hb("conv1x1_k%d_n%d" % (k, n), row="N2", bound="hbm", bytes=4.0 * b * h * w * (k + n), flops=2.0 * b * h * w * k * n, split=True,
        label="conv1x1_kernel (InstanceNorm + ReLU + 1x1 conv %d->%d of the conv heads, N2)" % (k, n), pmc=["conv1x1_kernel<%d>" % (k // 16)])
    _lib.check(_lib.load().nmrf_conv1x1_in_relu_f32(_p(x), b, cx, h * w, c0, k, _p(stats), 0 if stats is None else stats.shape[1],
                                                    float(eps), _p(stream), stages, float(inv), _p(bias), n, _p(out), int(token_major), _rf(x),
                                                    _stream()), "conv1x1_in_relu")
    _he("conv1x1_k%d_n%d" % (k, n))
    return out


def pack_conv1x1(weight):
    """[N,K,1,1] conv weight -> (stream, stages, 1/scale, N) for conv1x1_in_relu / conv1x1: rows zero-padded to a multiple of 64,
    columns to 64 (K <= 64) or 128."""
    n, k = weight.shape[0], weight.shape[1]
    kp, npad = (64 if k <= 64 else 128), (n + 63) // 64 * 64
    w = weight.reshape(n, k)
    if npad != n:
        w = torch.nn.functional.pad(w, (0, 0, 0, npad - n))
    pk, inv = pack_split_weight(w.contiguous(), kp)
    return pk.view(-1, 512).contiguous(), pk.shape[0] * pk.shape[1] // 8, inv, n


@_on_device
def conv1x1(x, packed, k, stride=1, bias=None):
    """Plain (strided) 1x1 convolution [B,Cx,H,W] -> [B,N,Ho,Wo] on the conv1x1 kernel (the encoder's down-sampling shortcuts)."""
    _chk(x, bias)
    stream, stages, inv, n = packed
    _chk(stream, dtype=torch.int32)
    b, cx, h, w = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty(b, n, ho, wo, device=x.device, dtype=torch.float32)
    _hb("conv1x1_s%d_k%d_n%d" % (stride, k, n), row="N2", bound="hbm", bytes=4.0 * b * ho * wo * (k + n), flops=2.0 * b * ho * wo * k * n,
        split=True, label="conv1x1_kernel (1x1 stride-%d shortcut %d->%d of the encoder, N2)" % (stride, k, n),
        pmc=["conv1x1_kernel<%d>" % (4 if k <= 64 else 8)])
    _lib.check(_lib.load().nmrf_conv1x1_f32(_p(x), b, cx, h, w, stride, 0, k, None, 0, 1e-5, _p(stream), stages, float(inv), _p(bias), n,
                                            _p(out), 0, _rf(x), _stream()), "conv1x1")
    _he("conv1x1_s%d_k%d_n%d" % (stride, k, n))
    return out


@_on_device
def prep_images(img1, img2, hp, wp):
    """[B,3,H,W] x2 (0..255, float32 or uint8) -> [2B,3,hp,wp]: replicate-padded right/bottom, stacked, normalised to [-1,1]."""
    _chk(img1, img2, dtype=img1.dtype if img1.dtype == torch.uint8 else torch.float32)
    b, c, h, w = img1.shape
    out = torch.empty(2 * b, c, hp, wp, device=img1.device, dtype=torch.float32)
    fn = _lib.load().nmrf_prep_images_u8 if img1.dtype == torch.uint8 else _lib.load().nmrf_prep_images_f32
    _lib.check(fn(_p(img1), _p(img2), b, c, h, w, hp, wp, _p(out), _stream()), "prep_images")
    return out


def host_copy_nt(dst, src):
    """dst (pinned CPU tensor or a contiguous view of one) <- src (CPU tensor, same dtype and element count), written with
    non-temporal stores (nmrf_host_copy_nt): what the driver stages its input images with."""
    if dst.is_cuda or src.is_cuda or dst.dtype != src.dtype or dst.numel() != src.numel() or not dst.is_contiguous():
        raise _lib.NmrfHipError("host_copy_nt: contiguous CPU tensors of one dtype and size")
    src = src.contiguous()
    _lib.check(_lib.load().nmrf_host_copy_nt(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()), "host_copy_nt")
    return dst


def host_read_evict(src):
    """A pageable copy of `src` (a contiguous view of a pinned CPU tensor); the lines of `src` are flushed from the CPU cache
    afterwards (nmrf_host_read_evict): what the driver hands results out with."""
    if src.is_cuda or not src.is_contiguous():
        raise _lib.NmrfHipError("host_read_evict: a contiguous CPU tensor")
    dst = torch.empty(src.shape, dtype=src.dtype)
    _lib.check(_lib.load().nmrf_host_read_evict(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size()), "host_read_evict")
    return dst


@_on_device
def bias_avgpool2(y, bias):
    """y [B,C,H,W] (bias-free conv output) -> (y + bias[c], its 2x2 average) in one pass."""
    _chk(y, bias)
    b, c, h, w = y.shape
    x = torch.empty_like(y)
    pooled = torch.empty(b, c, h // 2, w // 2, device=y.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_bias_avgpool2_f32(_p(y), _p(bias), b * c, c, h, w, _p(x), _p(pooled), _stream()), "bias_avgpool2")
    return x, pooled


@_on_device
def avgpool2(x):
    """x [B,C,H,W] (H, W even) -> its 2x2 average, summed row by row like ATen's avg_pool2d."""
    _chk(x)
    b, c, h, w = x.shape
    pooled = torch.empty(b, c, h // 2, w // 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_bias_avgpool2_f32(_p(x), None, b * c, c, h, w, None, _p(pooled), _stream()), "avgpool2")
    return pooled


@_on_device
def msda_forward(value, shapes, lvl_start, loc, w):
    dt = value.dtype
    if dt not in (torch.float32, torch.float64):
        raise _lib.NmrfHipError("msda supports fp32/fp64")
    _chk(value, loc, w, dtype=dt)
    _chk(shapes, lvl_start, dtype=torch.int64)
    b, s, m, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    out = torch.empty(b, lq, m * d, device=value.device, dtype=dt)
    fn = _lib.load().nmrf_msda_forward_f32 if dt == torch.float32 else _lib.load().nmrf_msda_forward_f64
    _hb("msda_forward", row="A15", bound="hbm", bytes=float(value.element_size()) * (value.numel() + loc.numel() + w.numel() + out.numel()),
        flops=2.0 * b * lq * m * d * l * p * 5, label="msda_fwd_d8_tiled_kernel / msda_fwd_d8_kernel / msda_fwd_kernel (multi-scale deformable "
        "attention forward, A15; the tiled form when the queries are a grid over one level of 8 x 8-channel heads)",
        pmc=["msda_fwd_d8_tiled_kernel<", "msda_fwd_d8_kernel<", "msda_fwd_kernel"])
    _lib.check(fn(_p(value), _p(shapes), _p(lvl_start), _p(loc), _p(w), b, s, m, d, l, lq, p, _p(out), _stream()),
               "msda_forward")
    _he("msda_forward")
    return out


@_on_device
def msda_backward(value, shapes, lvl_start, loc, w, grad_out):
    dt = value.dtype
    _chk(value, loc, w, grad_out, dtype=dt)
    _chk(shapes, lvl_start, dtype=torch.int64)
    b, s, m, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    gv = torch.zeros_like(value)
    gl = torch.empty_like(loc)
    gw = torch.empty_like(w)
    fn = _lib.load().nmrf_msda_backward_f32 if dt == torch.float32 else _lib.load().nmrf_msda_backward_f64
    _lib.check(fn(_p(value), _p(shapes), _p(lvl_start), _p(loc), _p(w), _p(grad_out), b, s, m, d, l, lq, p, _p(gv), _p(gl),
                  _p(gw), _stream()), "msda_backward")
    return gv, gl, gw


# ---- N4: the pieces of a backward pass through the token-linear chains (csrc/backward.hip, include/nmrf_hip.h) -------------
def _sum_parts(parts, n, out=None):
    """parts [S, n] (contiguous) -> [n]: rounds of 32 parts per group (nmrf_sum_partials_grouped_f32), a fixed reduction tree."""
    s_ = parts.shape[0]
    cur = parts
    if s_ > 32 and n < 8192:
        # many parts of a narrow row (bias / LayerNorm / table sums): the whole tree in one launch (nmrf_sum_partials_tree_f32)
        if out is None:
            out = torch.empty(n, device=parts.device, dtype=torch.float32)
        _lib.check(_lib.load().nmrf_sum_partials_tree_f32(_p(cur), s_, n, n, _p(out), _stream()), "sum_partials_tree")
        return out
    # (wide rows -- a weight gradient's K-split parts -- have a thread per element to keep the chip busy: one pass over all parts; narrow
    #  rows -- bias and LayerNorm sums, 64 ... 512 elements -- would be one long serial chain per thread: rounds of 32)
    while s_ > 32 and not (n >= 8192 and s_ <= 256):
        g = (s_ + 31) // 32
        nxt = torch.empty(g, n, device=parts.device, dtype=torch.float32)
        _lib.check(_lib.load().nmrf_sum_partials_grouped_f32(_p(cur), s_, n, n, 32, _p(nxt), _stream()), "sum_partials")
        cur, s_ = nxt, g
    if out is None:
        out = torch.empty(n, device=parts.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_sum_partials_grouped_f32(_p(cur), s_, n, n, s_, _p(out), _stream()), "sum_partials")
    return out


_AMAX = {}


def grad_amax(dy):
    """max |dy| as a device scalar (no read-back), shared by the dgrad and the wgrad of one gradient tensor: nmrf_gemm_split_f32 rescales
    its A operand from it by a power of two before the fp16 split (gradients of a mean loss are ~1 / (B H W), far below the range in
    which the unscaled split keeps 22 bits).  One nmrf_absmax_f32 launch per (tensor object, version)."""
    import weakref
    ent = _AMAX.get(id(dy))
    key = (dy._version, dy.data_ptr())
    if ent is not None and ent[0]() is dy and ent[1] == key:
        return ent[2]
    if len(_AMAX) > 64:
        _AMAX.clear()
    with torch.no_grad():
        d = dy.detach()
        if d.is_contiguous() and d.dtype == torch.float32 and d.data_ptr() % 16 == 0 and d.numel() > 0:
            m = _zeroed_scalar(d.device)
            _lib.check(_lib.load().nmrf_absmax_f32(_p(d), d.numel(), _p(m), _stream()), "absmax")
        else:
            m = torch.linalg.vector_norm(d.reshape(-1), ord=float("inf")).reshape(1)
    _AMAX[id(dy)] = (weakref.ref(dy), key, m)
    return m


_ZPOOL = {}


def _zeroed_scalar(device):
    """A one-float device tensor holding 0: slices of a pool that is refilled (one torch.zeros) every 1024 requests -- a scalar per
    gradient tensor without a fill kernel each."""
    ent = _ZPOOL.get(device)
    if ent is None or ent[1] >= ent[0].numel():
        ent = [torch.zeros(1024, device=device, dtype=torch.float32), 0]
        _ZPOOL[device] = ent
    i = ent[1]
    ent[1] = i + 4                                                 # 16-byte spacing
    return ent[0][i:i + 1]


def _gemm(a, sa_i, sa_k, b, sb_k, sb_j, m, n, k, splits=1, a_amax=None):
    """C[m,n] = op(A) . op(B) on the split-fp16 MFMA with explicit element strides; splits > 1: K split + fixed-order sum.
    a_amax: device scalar max |A| (grad_amax) when A is a gradient."""
    _chk(a, b)
    out = torch.empty(m, n, device=a.device, dtype=torch.float32)
    if splits <= 1:
        _lib.check(_lib.load().nmrf_gemm_split_f32(_p(a), sa_i, sa_k, _p(b), sb_k, sb_j, m, n, k, _p(out), n, 1, 0, _p(a_amax), _rf(a),
                                                   _stream()), "gemm_split")
        return out
    parts = torch.empty(splits, m, n, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_gemm_split_f32(_p(a), sa_i, sa_k, _p(b), sb_k, sb_j, m, n, k, _p(parts), n, splits, m * n, _p(a_amax),
                                               _rf(a), _stream()), "gemm_split")
    _sum_parts(parts.view(splits, m * n), m * n, out=out.view(-1))
    return out


@_on_device
def linear_forward(x, w):
    """x [T,K] . w [N,K]^T -> [T,N] (no bias): the recomputation of a saved-for-backward pre-activation."""
    t, k = x.shape
    return _gemm(x, k, 1, w, 1, k, t, w.shape[0], k)


@_on_device
def linear_dgrad(dy, w):
    """dy [T,N], w [N,K] -> dx [T,K] = dy . w."""
    t, n = dy.shape
    return _gemm(dy, n, 1, w, w.shape[1], 1, t, w.shape[1], n, a_amax=grad_amax(dy))


@_on_device
def linear_wgrad(dy, x):
    """dy [T,N], x [T,K] -> dW [N,K] = dy^T . x (reduction over the T tokens, split into <= 256 deterministic parts)."""
    t, n = dy.shape
    k = x.shape[1]
    tiles = ((n + 31) // 32) * ((k + 31) // 32)
    splits = max(1, min(256, (t + 511) // 512, max(1, 2048 // tiles)))
    return _gemm(dy, 1, n, x, k, 1, n, k, t, splits=splits, a_amax=grad_amax(dy))


@_on_device
def bias_grad(dy):
    """dy [T,N] -> [N] column sums (two deterministic passes)."""
    _chk(dy)
    t, n = dy.shape
    rpb = max(64, (t + 1023) // 1024)
    nb = (t + rpb - 1) // rpb
    parts = torch.empty(nb, n, device=dy.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_colsum_partials_f32(_p(dy), t, n, rpb, _p(parts), _stream()), "colsum_partials")
    return _sum_parts(parts, n)


@_on_device
def bias_act(pre, bias, act, want_pre=True):
    """(pre + bias, act(pre + bias)); act 0 identity, 1 ReLU, 2 GELU(erf)."""
    _chk(pre, bias)
    t, n = pre.shape
    pre_out = torch.empty_like(pre) if want_pre else None
    act_out = torch.empty_like(pre) if act else None
    _lib.check(_lib.load().nmrf_bias_act_f32(_p(pre), _p(bias), t, n, act, _p(pre_out), _p(act_out), _stream()), "bias_act")
    return pre_out, (act_out if act else pre_out)


@_on_device
def act_backward(pre, dy, act):
    _chk(pre, dy)
    dx = torch.empty_like(dy)
    _lib.check(_lib.load().nmrf_act_bwd_f32(_p(pre), _p(dy), dy.numel(), act, _p(dx), _stream()), "act_bwd")
    return dx


@_on_device
def layer_norm(x, g, b, eps):
    _chk(x, g, b)
    y = torch.empty_like(x)
    _lib.check(_lib.load().nmrf_layernorm_f32(_p(x), _p(g), _p(b), x.shape[0], x.shape[1], float(eps), _p(y), _stream()), "layernorm")
    return y


@_on_device
def layer_norm_backward(x, g, dy, eps):
    """-> (dx [T,C], dg [C], db [C])"""
    _chk(x, g, dy)
    t, c = x.shape
    blocks = max(1, min(512, (t + 31) // 32))
    dx = torch.empty_like(x)
    pg = torch.empty(4 * blocks, c, device=x.device, dtype=torch.float32)
    pb = torch.empty_like(pg)
    _lib.check(_lib.load().nmrf_layernorm_bwd_f32(_p(x), _p(g), _p(dy), t, c, float(eps), blocks, _p(dx), _p(pg), _p(pb), _stream()), "layernorm_bwd")
    return dx, _sum_parts(pg, c), _sum_parts(pb, c)


@_on_device
def window_attn_backward(qkv, table, dout, b, hp, wp, n, heads, win, shift, sibling_mask):
    """Backward of window_attn on fp32 rows: -> (dqkv [T,3C], dtable [(2 win - 1)^2, 3C])."""
    _chk(qkv, table, dout)
    t, c3 = qkv.shape
    c = c3 // 3
    assert t == b * hp * wp * n and dout.shape == (t, c)
    nwin = (hp // win) * (wp // win)
    tw, r = win * win * n, (2 * win - 1) ** 2
    dqkv = torch.empty_like(qkv)
    parts = torch.empty(b * nwin, r, c3, device=qkv.device, dtype=torch.float32)
    scratch = torch.empty(2 * b * heads * nwin * tw * tw, device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_window_attn_bwd_f32(_p(qkv), _p(table), _p(dout), b, hp, wp, n, c, heads, win, shift, int(bool(sibling_mask)),
                                                    _p(dqkv), _p(parts), _p(scratch), _stream()), "window_attn_bwd")
    return dqkv, _sum_parts(parts.view(b * nwin, r * c3), r * c3).view(r, c3)


@_on_device
def self_attn_backward(qkv, dout, n, heads):
    """Backward of self_attn: qkv [T,3C] fp32, dout [T,C] -> dqkv [T,3C]."""
    _chk(qkv, dout)
    t, c3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.load().nmrf_self_attn_bwd_f32(_p(qkv), _p(dout), t, n, c3 // 3, heads, _p(dqkv), _stream()), "self_attn_bwd")
    return dqkv


def from_kv16(qkv):
    """The inverse of to_kv16 (torch views and bit operations -- plumbing): rows whose k | v thirds are split fp16 pairs -> fp32 rows with
    k = hi + lo, v = hi + lo (the value the attention kernels multiply: up to 2^-22 relative of what the producer split)."""
    t = qkv.shape[0]
    if qkv.is_cuda and qkv.is_contiguous() and qkv.dtype == torch.float32 and qkv.shape[1] == 384 and qkv.data_ptr() % 16 == 0 and t > 0:
        out = torch.empty_like(qkv)                                              # one launch (nmrf_from_kv16_f32): the training tape asks per layer
        _lib.check(_lib.load().nmrf_from_kv16_f32(_p(qkv), t, _p(out), _stream()), "from_kv16")
        return out
    return _from_kv16_torch(qkv)


def _from_kv16_torch(qkv):
    """from_kv16 as torch views and bit operations (rows the kernel does not take; the restatement its test compares it with)."""
    t = qkv.shape[0]
    out = qkv.clone()
    words = qkv.view(torch.int32)
    half = lambda w16: (w16 & 0xffff).to(torch.int16).view(torch.float16).float()
    kw = words[:, 128:256].reshape(t, 4, 32)                                   # per head: 16 words of hi halves, 16 words of lo halves
    unpack = lambda w: torch.stack((half(w), half(w >> 16)), -1).reshape(t, 4, 32)
    out[:, 128:256] = (unpack(kw[..., :16]) + unpack(kw[..., 16:])).reshape(t, 128)
    vw = words[:, 256:384]
    out[:, 256:384] = half(vw) + half(vw >> 16)
    return out


@_on_device
def stripe_attn_backward(qkv, lepe_v, lepe_h, dout, b, h, w, n):
    """Backward of stripe_attn on fp32 rows: -> (dqkv [T,384], dlepe_v [64,1,3,3], dlepe_h [64,1,3,3])."""
    _chk(qkv, lepe_v, lepe_h, dout)
    t = qkv.shape[0]
    assert t == b * h * w * n and qkv.shape[1] == 384 and dout.shape == (t, 128)
    dqkv = torch.empty_like(qkv)
    pv = torch.empty(b * w, 64 * 3, device=qkv.device, dtype=torch.float32)
    ph = torch.empty(b * h, 64 * 3, device=qkv.device, dtype=torch.float32)
    scratch = torch.empty(2 * b * 2 * max(w * (h * n) ** 2, h * (w * n) ** 2), device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_stripe_attn_bwd_f32(_p(qkv), _p(lepe_v), _p(lepe_h), _p(dout), b, h, w, n, _p(dqkv), _p(pv), _p(ph), _p(scratch),
                                                    _stream()), "stripe_attn_bwd")
    gv, gh = _sum_parts(pv, 192).view(64, 3), _sum_parts(ph, 192).view(64, 3)
    dlv, dlh = torch.zeros_like(lepe_v), torch.zeros_like(lepe_h)
    dlv[:, 0, :, 1] = gv                                     # centre column: taps over dy
    dlh[:, 0, 1, :] = gh                                     # centre row: taps over dx
    return dqkv, dlv, dlh


@_on_device
def unfold5(src, p, c, d, src_pcd=False):
    """5-tap columns of a Conv1d(kernel 5, padding 2) over the disparity axis: src rows (p, d) x c (or [p][c][d]) -> [p*d, 5c]."""
    _chk(src)
    col = torch.empty(p * d, 5 * c, device=src.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_unfold5_f32(_p(src), p, c, d, int(bool(src_pcd)), _p(col), _stream()), "unfold5")
    return col


@_on_device
def fold5(dcol, p, c, d):
    _chk(dcol)
    out = torch.empty(p * d, c, device=dcol.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_fold5_f32(_p(dcol), p, c, d, _p(out), _stream()), "fold5")
    return out


@_on_device
def softmax_backward(prob, dprob):
    _chk(prob, dprob)
    dz = torch.empty_like(prob)
    _lib.check(_lib.load().nmrf_softmax_bwd_f32(_p(prob), _p(dprob), prob.shape[0], prob.shape[1], _p(dz), _stream()), "softmax_bwd")
    return dz


@_on_device
def cost_volume_backward(f1, f2, dcv, num_disp, groups):
    """Backward of cost_volume: f1, f2 [B,C,H,W], dcv [B*H*W, G, D] -> (df1, df2)."""
    _chk(f1, f2, dcv)
    b, c, h, w = f1.shape
    assert dcv.shape == (b * h * w, groups, num_disp)
    df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
    _lib.check(_lib.load().nmrf_cost_volume_bwd_f32(_p(f1), _p(f2), _p(dcv), b, c, h, w, num_disp, groups, _p(df1), _p(df2), _stream()),
               "cost_volume_bwd")
    return df1, df2


@_on_device
def seed_taps_backward(dcost, seeds, g, d):
    """Backward of seed_features' cost taps: dcost [P*N, ldc >= 9 g], seeds [P, N] int64 -> dcv [P, g, d]."""
    _chk(dcost)
    _chk(seeds, dtype=torch.int64)
    p, n = seeds.shape
    assert dcost.shape[0] == p * n and dcost.shape[1] >= 9 * g
    dcv = torch.empty(p, g, d, device=dcost.device, dtype=torch.float32)
    _lib.check(_lib.load().nmrf_seed_taps_bwd_f32(_p(dcost), _p(seeds), p, n, g, d, dcost.shape[1], _p(dcv), _stream()), "seed_taps_bwd")
    return dcv


@_on_device
def warp_corr_concat_backward(labels, drow, f1, f2, g1, g2, n, groups=32):
    """Backward of warp_corr_concat (NCHW maps): drow [T, 2 Cf + groups] -> (df1, df2, dg1, dg2)."""
    _chk(labels, drow, f1, f2, g1, g2)
    b, cf, h, w = f1.shape
    cg = g1.shape[1]
    assert labels.numel() == b * h * w * n and drow.shape == (b * h * w * n, 2 * cf + groups)
    out = [torch.empty_like(t) for t in (f1, f2, g1, g2)]
    _lib.check(_lib.load().nmrf_warp_corr_concat_bwd_f32(_p(labels), _p(drow), _p(f1), _p(f2), _p(g1), _p(g2), b, h, w, n, cf, cg, groups,
                                                         *[_p(t) for t in out], _stream()), "warp_corr_concat_bwd")
    return tuple(out)
