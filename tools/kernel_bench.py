#!/usr/bin/env python
"""Micro-bench of the hand-written attention kernels at BASELINE sizes (for rocprofv3 --pmc passes).
    python tools/kernel_bench.py [--iters 10] [--batch 1] [--which window,stripe,refine,warp]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrf_amd import kernels as K  # noqa: E402
from nmrf_amd.utils.hashinit import unit_noise  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--which", default="window,stripe,refine,warp")
ap.add_argument("--xscale", type=float, default=1.0, help="conv section: scale of the input (fp16 subnormal probe)")
ap.add_argument("--lib", default=None, help="A/B runs: load this build of the library instead of nmrf_amd/lib/libnmrf_hip.so")
args = ap.parse_args()
if args.lib:
    import nmrf_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(args.lib)
dev = "cuda"
b, h, w, n = args.batch, 47, 156, 4


def mk(key, *shape):
    import numpy as np
    return torch.from_numpy(unit_noise(key, int(np.prod(shape))).reshape(shape)).to(dev)


def timeit(name, fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-64s %9.2f us / call" % (name, e0.elapsed_time(e1) * 1e3 / args.iters), flush=True)


import ctypes
from nmrf_amd import _lib
# the probes (nmrf_debug_*) live in the tools-only debug build of the same sources: build it if needed and make the
# kernels module of this process use it (a superset of the product library)
from nmrf_amd.build import build_library
if not args.lib:                 # (--lib: an A/B build of the tools library, tools/build_ab_lib.sh)
    _lib.LIB_PATH = build_library(debug=True, verbose=False)
_l = _lib.load()
for _n, _at in (("nmrf_debug_window_occupancy", None), ("nmrf_debug_window_timing", None), ("nmrf_debug_stripe_census", None),
                ("nmrf_debug_mfma_peak", None), ("nmrf_debug_attn_core_peak", None), ("nmrf_debug_token_linear_timing", None),
                ("nmrf_debug_wino_timing", None), ("nmrf_debug_nmp_block_variant", None)):
    getattr(_l, _n).restype = ctypes.c_int
a, b2 = ctypes.c_int(-1), ctypes.c_int(-1)
_l.nmrf_debug_window_occupancy(ctypes.byref(a), ctypes.byref(b2))
print("runtime occupancy (blocks/CU): infer-window", a.value, " refine-window", b2.value, flush=True)
which = args.which.split(",")
if "mfma16" in which:
    out = torch.empty(256 * 256, device=dev)
    _l.nmrf_debug_mfma16_peak.restype = ctypes.c_int
    for chains in (1, 2, 3):
        iters = 2000
        fn = lambda: _l.nmrf_debug_mfma16_peak(chains, iters, 256, ctypes.c_void_p(out.data_ptr()), None)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n_mfma = 256 * 4 * iters * (24 // chains) * chains
        print("mfma 32x32x16 f16, %d chain(s) per wave, 1 wave/SIMD: %.1f TFLOP/s, %.1f cycles/MFMA at 2.4 GHz" % (
            chains, n_mfma * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * (24 // chains) * chains)), flush=True)
if "mfma16x16" in which:
    out = torch.empty(256 * 512, device=dev)
    _l.nmrf_debug_mfma16x16_peak.restype = ctypes.c_int
    for threads in (256, 512):
        for chains in (1, 2, 4, 8):
            iters = 2000
            fn = lambda: _l.nmrf_debug_mfma16x16_peak(chains, threads, iters, 256, ctypes.c_void_p(out.data_ptr()), None)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            per_simd = iters * 24 * (threads // 256)
            print("mfma 16x16x32 f16, %d chain(s) per wave, %d wave(s)/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA slot of a SIMD at 2.4 GHz" % (
                chains, threads // 256, 256 * 4 * per_simd * 16384 / ms / 1e9, ms * 1e-3 * 2.4e9 / per_simd), flush=True)
if "block" in which:
    # the fused block kernel against the four launches it replaces (KITTI padded inference grid, batch b)
    import torch.nn.functional as F
    T = b * 48 * 156 * 4
    x, msg, enc = mk("bx", T, 128), mk("bm", T, 128), mk("be", T, 32)
    wp, w1, w2, wq = mk("wp", 128, 128) * 0.1, mk("w1", 512, 128) * 0.1, mk("w2", 128, 512) * 0.05, mk("wq", 384, 159) * 0.1
    bp, b1, b2, bq = mk("bp", 128), mk("b1", 512), mk("b2", 128), mk("bq", 384)
    g, be = mk("g", 128) * 0.1 + 1, mk("bb", 128) * 0.1
    stream, stages, inv = K.block_stream(wp, w1, w2, wq, 160)
    qd = dict(g=g, b=be, eps=1e-5, extra=enc, extra_div=1, bias=bq, kq=160, nq=384)
    for var, tag in ((0, "FD1 touch (product)"), (1, "FD1 no touch"), (2, "FD2 touch"), (3, "FD2 no touch"), (4, "FD1 PF4 touch"),
                     (10, "DBG no barrier"), (11, "DBG no commit/fetch"), (12, "DBG no LDS fragment reads"), (13, "DBG no MFMA"),
                     (14, "DBG no barrier, no commit/fetch"), (15, "DBG no barrier/commit/fetch/fragment reads")):
        _l.nmrf_debug_nmp_block_variant(var)
        timeit("nmp_block proj+mlp+qkv " + tag, lambda: K.nmp_block(x, stream, stages, inv, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=32))
    _l.nmrf_debug_nmp_block_variant(0)
    import numpy as np
    stamps = torch.zeros(64 * 4 * 16, dtype=torch.int64, device=dev)
    _l.nmrf_debug_nmp_block_timing.restype = ctypes.c_int
    _l.nmrf_debug_nmp_block_timing(ctypes.c_void_p(stamps.data_ptr()))
    K.nmp_block(x, stream, stages, inv, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=32)
    torch.cuda.synchronize()
    _l.nmrf_debug_nmp_block_timing(None)
    st = stamps.cpu().numpy().reshape(64, 4, 16).astype(np.int64)
    names = {1: "prologue (2 stages to LDS, barrier)", 2: "x/msg loads + split + proj stage (4 stages, 96 MFMAs)",
             3: "LN2 + split + park x1", 4: "MLP (32 stages, 768 MFMAs)", 5: "x_out staging + row stores",
             6: "LNq + extra loads + split", 7: "q group 0 (5 stages, 120 MFMAs)", 8: "q group 1", 9: "q group 2", 15: "tail"}
    prev = 0
    print("nmp_block phases, s_memtime ticks (100 MHz = 10 ns), median over 64 blocks x 4 waves:")
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15):
        d = (st[:, :, k] - st[:, :, prev]).reshape(-1)
        print("   %-52s median %7.0f  min %7.0f  max %7.0f ticks" % (names[k], np.median(d), d.min(), d.max()))
        prev = k
    for k, nm in ((10, "sum over stages: barrier wait"), (11, "sum over stages: consume (LDS reads + MFMAs + VALU)"), (12, "sum over stages: commit + fetch")):
        d = st[:, :, k].reshape(-1)
        print("   %-52s median %7.0f  min %7.0f  max %7.0f ticks" % (nm, np.median(d), d.min(), d.max()))
    tot = (st[:, :, 15] - st[:, :, 0]).reshape(-1)
    print("   total per wave: median %.0f ticks; first-start to last-end over blocks: %.0f ticks" % (np.median(tot), st[:, :, 15].max() - st[:, :, 0].min()))
    s16, st16, i16 = K.block_stream16(wp, w1, w2, wq, 160)
    timeit("nmp_block16 proj+mlp+qkv (16 tokens / wave)", lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
    _l.nmrf_debug_nmp_block16_variant.restype = ctypes.c_int
    for var, tag in ((1, "no barrier"), (2, "no commit/fetch"), (4, "no LDS fragment reads"), (8, "no MFMA"), (7, "no barrier/commit/fetch/reads"), (16, "no GELU"), (24, "no GELU, no MFMA")):
        _l.nmrf_debug_nmp_block16_variant(var)
        timeit("nmp_block16 DBG " + tag, lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
    _l.nmrf_debug_nmp_block16_variant(0)
    _l.nmrf_debug_nmp_block16_timing.restype = ctypes.c_int
    def stamp16(tag, call):
        st16s = torch.zeros(64 * 8 * 16 + 4 * 16384, dtype=torch.int64, device=dev)
        call(); call()
        _l.nmrf_debug_nmp_block16_timing(ctypes.c_void_p(st16s.data_ptr()))
        call()
        torch.cuda.synchronize()
        _l.nmrf_debug_nmp_block16_timing(None)
        s_ = st16s.cpu().numpy()[:64 * 8 * 16].reshape(64, 8, 16).astype(np.int64)
        nm = {1: "prologue: params + 2 stages to LDS + barrier", 2: "x / msg row loads + split", 3: "proj (4 stages, 96 MFMAs) + residual",
              4: "LN2 + split + park x1", 5: "MLP (32 stages, 768 MFMAs, GELU)", 6: "x_out staging + row stores", 7: "LNq + extra + split",
              8: "q group 0", 9: "q group 1", 10: "q group 2", 11: "drain (vmcnt 0)"}
        print("nmp_block16 %s: per-wave phases in s_memtime ticks (100 MHz: 1 tick = 10 ns), median / min / max over 64 blocks x 8 waves" % tag)
        prev = 0
        for k in range(1, 12):
            if (s_[:, :, k] == 0).all():
                continue
            d = (s_[:, :, k] - s_[:, :, prev]).reshape(-1)
            print("   %-48s %7.0f %7.0f %7.0f" % (nm[k], np.median(d), d.min(), d.max()))
            prev = k
        for k, nm_ in ((12, "sum over stages: barrier wait"), (13, "sum over stages: consume (LDS reads + MFMAs + VALU between)"), (14, "sum over stages: commit + fetch issue")):
            d = s_[:, :, k].reshape(-1)
            print("   %-48s %7.0f %7.0f %7.0f" % (nm_, np.median(d), d.min(), d.max()))
        print("   total per wave median %.0f ticks; first start to last end %.0f ticks" % (
            np.median(s_[:, :, 11] - s_[:, :, 0]), s_[:, :, 11].max() - s_[:, :, 0].min()))
    stamp16("proj+mlp+qkv", lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
    # A/B inside this process: the first tile's rows requested before the prologue (product) vs behind it (variant 256)
    ref_out = [t.clone() for t in K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16) if t is not None]
    for rep in range(2):
        _l.nmrf_debug_nmp_block16_variant(256)
        alt_out = [t.clone() for t in K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16) if t is not None]
        timeit("nmp_block16 proj+mlp+qkv, rows loaded behind the prologue (variant 256)", lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
        _l.nmrf_debug_nmp_block16_variant(0)
        timeit("nmp_block16 proj+mlp+qkv, product (rows requested before the prologue)", lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
    print("variant 256 vs product: max |diff|", [float((p_ - q_).abs().max()) for p_, q_ in zip(ref_out, alt_out)])
    _l.nmrf_debug_nmp_block16_variant(256)
    stamp16("proj+mlp+qkv, rows behind the prologue", lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
    _l.nmrf_debug_nmp_block16_variant(0)
    s16b, st16b, i16b = K.block_stream16(wp, None, None, wq, 160)
    stamp16("proj+qkv (self block)", lambda: K.nmp_block(x, s16b, st16b, i16b, msg, bp, None, qd, tokens_per_wave=16))
    timeit("nmp_block16 proj+qkv (self block)", lambda: K.nmp_block(x, s16b, st16b, i16b, msg, bp, None, qd, tokens_per_wave=16))
    s16c, st16c, i16c = K.block_stream16(wp, w1, w2, None, 0)
    timeit("nmp_block16 proj+mlp", lambda: K.nmp_block(x, s16c, st16c, i16c, msg, bp, (g, be, 1e-5, b1, b2), None, tokens_per_wave=16))
    s2, st2, i2 = K.block_stream(wp, None, None, wq, 160)
    timeit("nmp_block proj+qkv (self block)", lambda: K.nmp_block(x, s2, st2, i2, msg, bp, None, qd, tokens_per_wave=32))
    s3, st3, i3 = K.block_stream(None, None, None, wq, 160)
    timeit("nmp_block qkv only", lambda: K.nmp_block(x, s3, st3, i3, None, None, None, qd, want_x=False, tokens_per_wave=32))
    s4, st4, i4 = K.block_stream(wp, w1, w2, None, 0)
    timeit("nmp_block proj+mlp", lambda: K.nmp_block(x, s4, st4, i4, msg, bp, (g, be, 1e-5, b1, b2), None, tokens_per_wave=32))
    pwp, pw1, pwq = (K.pack_linear_weight(v.contiguous()) for v in (wp, w1, wq))
    enc31 = enc[:, :31].contiguous()

    def old():
        y = K.token_linear(msg, pwp, 128, 128, bp)
        x1, h = K.token_linear(x, pw1, 512, 128, b1, ln=(g, be, 1e-5), y=y, act="gelu")
        y2 = F.linear(h, w2, b2)
        return K.token_linear(x1, pwq, 384, 159, bq, ln=(g, be, 1e-5), y=y2, extra=enc31)
    timeit("round-1 sequence (4 launches, fp32)", old)
if "block_timeline" in which:
    # when and where the workgroups of ONE block-kernel launch run (debug build, s_memrealtime = the chip-wide 100 MHz counter):
    # marks on the stream before and after, entry / exit of every block, its CU
    import numpy as np
    for bmul in (1, 8):
        T = bmul * 48 * 156 * 4
        x, msg, enc = mk("btx%d" % bmul, T, 128), mk("btm%d" % bmul, T, 128), mk("bte%d" % bmul, T, 32)
        wp, w1, w2, wq = mk("wp", 128, 128) * 0.1, mk("w1", 512, 128) * 0.1, mk("w2", 128, 512) * 0.05, mk("wq", 384, 159) * 0.1
        bp, b1, b2, bq = mk("bp", 128), mk("b1", 512), mk("b2", 128), mk("bq", 384)
        g, be = mk("g", 128) * 0.1 + 1, mk("bb", 128) * 0.1
        s16, st16, i16 = K.block_stream16(wp, w1, w2, wq, 160)
        qd = dict(g=g, b=be, eps=1e-5, extra=enc, extra_div=1, bias=bq, kq=160, nq=384)
        call = lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16)
        timeit("nmp_block16 proj+mlp+qkv, %d tokens (product build)" % T, call)
        buf = torch.zeros(64 * 8 * 16 + 4 * 16384, dtype=torch.int64, device=dev)
        marks = torch.zeros(4, dtype=torch.int64, device=dev)
        _l.nmrf_debug_nmp_block16_timing.restype = ctypes.c_int
        _l.nmrf_debug_realtime_mark.restype = ctypes.c_int
        cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        call(); call()
        _l.nmrf_debug_nmp_block16_timing(ctypes.c_void_p(buf.data_ptr()))
        call(); torch.cuda.synchronize()
        buf.zero_()
        _l.nmrf_debug_realtime_mark(ctypes.c_void_p(marks.data_ptr()), cs)
        call()
        _l.nmrf_debug_realtime_mark(ctypes.c_void_p(marks.data_ptr() + 8), cs)
        torch.cuda.synchronize()
        _l.nmrf_debug_nmp_block16_timing(None)
        rec = buf.cpu().numpy()[64 * 8 * 16:].reshape(-1, 4)
        rec = rec[rec[:, 0] > 0]
        mk_ = marks.cpu().numpy()
        t0 = mk_[0]
        s_, e_ = rec[:, 0] - t0, rec[:, 1] - t0
        cu = ((rec[:, 2] >> 32) << 16) | (rec[:, 2] & 0xff00)
        print("  %d blocks on %d CUs; times in 10 ns ticks after the mark kernel in front: first entry %d, last entry %d, first exit %d, "
              "last exit %d, mark behind %d" % (len(rec), len(np.unique(cu)), s_.min(), s_.max(), e_.min(), e_.max(), mk_[1] - t0))
        life = e_ - s_
        print("  block life: min %d  median %.0f  max %d ticks; entry-time deciles %s" % (
            life.min(), np.median(life), life.max(), " ".join("%d" % v for v in np.percentile(s_, [10, 30, 50, 70, 90]))))
        print("  exit-time deciles %s" % " ".join("%d" % v for v in np.percentile(e_, [10, 30, 50, 70, 90, 99])))
        per = {}
        for c in cu:
            per[int(c)] = per.get(int(c), 0) + 1
        cnt = np.array(list(per.values()))
        print("  blocks per CU: min %d max %d; CUs with 2+: %d" % (cnt.min(), cnt.max(), int((cnt > 1).sum())), flush=True)
        # s_memtime ticks per 10 ns realtime tick over the life of the first 64 blocks: what s_memtime counts
        w = buf.cpu().numpy()[:64 * 8 * 16].reshape(64, 8, 16)
        allrec = buf.cpu().numpy()[64 * 8 * 16:].reshape(-1, 4)[:64]
        mt = (w[:, 0, 11] - w[:, 0, 0]).astype(np.float64)
        rt = (allrec[:, 1] - allrec[:, 0]).astype(np.float64)
        okk = (rt > 0) & (mt > 0)
        print("  s_memtime ticks per microsecond of s_memrealtime over a block's life: median %.1f (min %.1f max %.1f)" % (
            np.median(mt[okk] / rt[okk] * 100), (mt[okk] / rt[okk] * 100).min(), (mt[okk] / rt[okk] * 100).max()), flush=True)

if "msda" in which:
    # A15 at the config-5 shapes of the Swin-T neck: [2, 96 256 queries, 8 heads, 8 channels], one level, 4 points (4 calls per forward,
    # levels 256x376 ... 32x47).  A/B against the round-3 library if a copy is present (nmrf_amd/lib/ab_main: the generic kernel).
    import numpy as np
    lq = 256 * 376
    # sampling locations as the neck produces them: the query's own reference point (pixel centres of the 256 x 376 query grid in
    # normalised coordinates) plus an offset of a few pixels of the sampled level (adaptor_modules.py:78-90) -- NOT uniform noise over
    # the map, which turns every tap into an L2 / HBM miss (3.4x slower, and not what the model does)
    ys, xs = torch.meshgrid((torch.arange(256, device=dev) + 0.5) / 256, (torch.arange(376, device=dev) + 0.5) / 376, indexing="ij")
    ref = torch.stack((xs, ys), -1).reshape(1, lq, 1, 1, 1, 2)
    noise = mk("mloc", 2, lq, 8, 1, 4, 2)
    wgt = torch.softmax(mk("mw", 2, lq, 8, 1, 4) * 2, -1).contiguous()
    old = None
    abp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmrf_amd", "lib", "ab_main", "libnmrf_hip.so")
    if os.path.exists(abp):
        old = ctypes.CDLL(abp)
        old.nmrf_msda_forward_f32.argtypes = _lib.PROTOTYPES["nmrf_msda_forward_f32"]
    for (hh, ww) in ((256, 376), (128, 188), (64, 94), (32, 47)):
        value = mk("mv%d" % hh, 2, hh * ww, 8, 8)
        loc = (ref + noise * 4.0 / torch.tensor([ww, hh], device=dev, dtype=torch.float32)).contiguous()      # +- ~4 px (noise is unit-range)
        shapes = torch.tensor([[hh, ww]], device=dev)
        start = torch.tensor([0], device=dev)
        out_new = K.msda_forward(value, shapes, start, loc, wgt)
        for var, tag in ((0, "product: tiled, taps from an LDS-staged bounding box (round 6)"),
                         (4, "untiled: persistent, operands prefetched, 2 points per load batch (rounds 4-5)"), (3, "one item per thread, no prefetch"),
                         (5, "tiled, 8 x 4 tiles, two 256-thread blocks per CU"), (1, "generic kernel"), (0, "product again"), (4, "untiled again"), (5, "8 x 4 tiles again")):
            _l.nmrf_debug_msda_variant(var)
            timeit("msda_forward level %dx%d (%s)" % (hh, ww, tag), lambda: K.msda_forward(value, shapes, start, loc, wgt))
        # offsets too large for the staged box (+- 14 px): the tiled kernel's blocks take their taps from global memory
        loc_far = (ref + noise * 14.0 / torch.tensor([ww, hh], device=dev, dtype=torch.float32)).contiguous()
        for var, tag in ((0, "tiled kernel, +-14 px offsets: boxes do not fit, taps from global memory"), (4, "untiled, +-14 px offsets")) * 2:
            _l.nmrf_debug_msda_variant(var)
            timeit("msda_forward level %dx%d (%s)" % (hh, ww, tag), lambda: K.msda_forward(value, shapes, start, loc_far, wgt))
        _l.nmrf_debug_msda_variant(4)
        out_untiled = K.msda_forward(value, shapes, start, loc, wgt)
        _l.nmrf_debug_msda_variant(0)
        print("   tiled vs untiled: max |diff| %.2e; algorithmic bytes %.1f MB" % (float((out_new - out_untiled).abs().max()),
              (loc.numel() + wgt.numel() + out_new.numel() + value.numel()) * 4 / 1e6), flush=True)
        if old is not None:
            out_old = torch.empty_like(out_new)
            call_old = lambda: old.nmrf_msda_forward_f32(value.data_ptr(), shapes.data_ptr(), start.data_ptr(), loc.data_ptr(), wgt.data_ptr(),
                                                        2, hh * ww, 8, 8, 1, lq, 4, out_old.data_ptr(), None)
            call_old(); torch.cuda.synchronize()
            timeit("msda_forward level %dx%d (round-3 library)" % (hh, ww), call_old)
            print("   max |new - old| %.3e ; algorithmic bytes %.1f MB" % (float((out_new - out_old).abs().max()),
                  (loc.numel() + wgt.numel() + out_new.numel() + value.numel()) * 4 / 1e6))
if "block16" in which:
    # the product block kernel only (PMC passes at KITTI batch-1 and batch-32 token counts)
    T = b * 48 * 156 * 4
    x, msg, enc = mk("bx", T, 128), mk("bm", T, 128), mk("be", T, 32)
    wp, w1, w2, wq = mk("wp", 128, 128) * 0.1, mk("w1", 512, 128) * 0.1, mk("w2", 128, 512) * 0.05, mk("wq", 384, 159) * 0.1
    bp, b1, b2, bq = mk("bp", 128), mk("b1", 512), mk("b2", 128), mk("bq", 384)
    g, be = mk("g", 128) * 0.1 + 1, mk("bb", 128) * 0.1
    qd = dict(g=g, b=be, eps=1e-5, extra=enc, extra_div=1, bias=bq, kq=160, nq=384)
    s16, st16, i16 = K.block_stream16(wp, w1, w2, wq, 160)
    timeit("nmp_block16 proj+mlp+qkv T=%d" % T, lambda: K.nmp_block(x, s16, st16, i16, msg, bp, (g, be, 1e-5, b1, b2), qd, tokens_per_wave=16))
if "window" in which:
    _l.nmrf_debug_window_pack1.restype = ctypes.c_int
    hp, wp = 48, 156
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", 121, 384)
    q16 = K.to_kv16(qkv)
    for rep in range(2):
        for shift in (0, 3):
            timeit("window_attn 6x6x4 shift=%d" % shift, lambda: K.window_attn(qkv, table, b, hp, wp, n, 4, 6, shift, True, checked=True))
            o_new = K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True)
            for rep in range(2):
                timeit("window_attn 6x6x4 shift=%d, kv16, MFMA phase 0, two windows per 10-wave block (product)" % shift, lambda: K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True))
                _l.nmrf_debug_window_pack1(11)
                o_two = K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True)
                timeit("window_attn 6x6x4 shift=%d, kv16, MFMA phase 0, ONE window per 5-wave block" % shift, lambda: K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True))
                _l.nmrf_debug_window_pack1(10)
                o_old = K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True)
                timeit("window_attn 6x6x4 shift=%d, kv16, phase 0 on the VALU (round 3)" % shift, lambda: K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True))
                _l.nmrf_debug_window_pack1(0)
                if not rep:
                    print("   one vs two windows per block: max |diff| %.3e" % float((o_new - o_two).abs().max()))
            print("   MFMA vs VALU phase 0: max |diff| %.3e (output scale %.3e)" % (float((o_new - o_old).abs().max()), float(o_old.abs().max())))
if "window6" in which:
    # round 6: the persistent 6 x 6 x 4 kernel (csrc/window_attn6.hip) against the two-windows-per-block kernel, interleaved, min of 3 rounds
    hp, wp = 48, 156
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", 121, 384)
    q16 = K.to_kv16(qkv)
    def t_us6(fn, nrep):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / nrep
    for shift in (0, 3):
        best = {}
        outs = {}
        for rnd_ in range(3):
            for tag, flag in (("two windows per block (rounds 2-5)", False), ("persistent, one wave per query tile (round 6)", True)):
                K.WINDOW6 = flag
                outs[tag] = K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True)
                best[tag] = min(best.get(tag, 1e9), t_us6(lambda: K.window_attn(q16, table, b, hp, wp, n, 4, 6, shift, True, kv16=True), args.iters))
        K.WINDOW6 = True
        a_, b_ = list(outs.values())
        print("window_attn 6x6x4 kv16 batch %d shift %d: %s; max |diff| %.2e" % (b, shift, ", ".join("%s %.1f us" % kv for kv in best.items()),
                                                                                  float((a_ - b_).abs().max())), flush=True)
if "window6_census" in which:
    # timing ablations / variants of the persistent window kernel (debug build: nmrf_debug_window6_variant), interleaved, min of 3
    _l.nmrf_debug_window6_variant.restype = ctypes.c_int
    hp, wp = 48, 156
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", 121, 384)
    q16 = K.to_kv16(qkv)
    names = {0: "product (8 waves per block)", 1: "12 waves per block", 3: "4 waves per block", 101: "- value-embedding term",
             102: "- 4x4x4 passes (1 chunk of 8)", 104: "- P V MFMAs", 108: "- K Q^T MFMAs", 116: "- exponentials", 132: "- per-tile K / V loads",
             131: "- all of the arithmetic above", 163: "- everything (loads + arithmetic)"}
    best = {}
    for rnd_ in range(3):
        for v in names:
            _l.nmrf_debug_window6_variant(v)
            f = lambda: K.window_attn(q16, table, b, hp, wp, n, 4, 6, 3, True, kv16=True)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1) * 1e3 / args.iters)
    _l.nmrf_debug_window6_variant(0)
    for v, nm in names.items():
        print("window_attn6 batch %d  %-44s %8.1f us" % (b, nm, best[v]), flush=True)
if "window6_stamps" in which:
    # per-phase s_memtime timeline of a wave's first item in the persistent window kernel (timed instantiations of the debug build)
    _l.nmrf_debug_window6_variant.restype = ctypes.c_int
    _l.nmrf_debug_window6_stamps.restype = ctypes.c_int
    hp, wp = 48, 156
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", 121, 384)
    q16 = K.to_kv16(qkv)
    for v, nw, nm in ((10, 8, "8 waves per block"), (11, 12, "12 waves per block"), (12, 4, "4 waves per block")):
        st = torch.zeros(256 * nw * 64, dtype=torch.int64, device=dev)
        _l.nmrf_debug_window6_stamps(ctypes.c_void_p(st.data_ptr()))
        _l.nmrf_debug_window6_variant(v)
        for _ in range(3):
            K.window_attn(q16, table, b, hp, wp, n, 4, 6, 3, True, kv16=True)
        torch.cuda.synchronize()
        _l.nmrf_debug_window6_variant(0)
        t_ = st.view(256 * nw, 64).cpu().double()
        t_ = t_[t_[:, 33] > 0]
        d = lambda a, c: float((t_[:, a] - t_[:, c]).mean())
        tiles = []
        for kt in range(5):
            prev = 1 if kt == 0 else 7 + 6 * (kt - 1)
            tiles.append("[4x4x4 %5.0f | KQ %5.0f | softmax %5.0f | PV %5.0f | ev %5.0f]" % (d(2 + 6 * kt, prev), d(3 + 6 * kt, 2 + 6 * kt), d(4 + 6 * kt, 3 + 6 * kt),
                                                                                           d(5 + 6 * kt, 4 + 6 * kt), d(7 + 6 * kt, 5 + 6 * kt)))
        print("window_attn6 batch %d, %s: first item of %d waves: prologue %.0f cycles, item %.0f, store %.0f; tiles:" % (b, nm, len(t_), d(1, 0), d(33, 0), d(33, 32)))
        for x in tiles:
            print("    " + x)
if "stripe" in which:
    qkv = mk("q2", b * h * w * n, 384)
    lv, lh = mk("lv", 64, 1, 3, 3), mk("lh", 64, 1, 3, 3)
    timeit("stripe_attn (both axes)", lambda: K.stripe_attn(qkv, lv, lh, b, h, w, n))
    so = torch.empty(b * h * w * n, 128, device=dev)
    q16 = K.to_kv16(qkv)
    for rep in range(2):
        for ax, nm in ((1, "vertical"), (2, "horizontal")):
            for fmt, src in ((0, qkv), (1, q16)):
                timeit("stripe_attn %s only, %s" % (nm, "k | v pre-split (kv16)" if fmt else "fp32 rows"), lambda: _l.nmrf_stripe_attn_f32(
                    ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(lv.data_ptr()), ctypes.c_void_p(lh.data_ptr()), b, h, w, n, 128, ax,
                    fmt, ctypes.c_void_p(so.data_ptr()), None, None))
if "stripe_both" in which:
    # both axes of a propagation layer in one launch against the two single-axis launches (kv16 rows), interleaved, min of 3 rounds
    qkv = mk("q2", b * h * w * n, 384)
    lv, lh = mk("lv", 64, 1, 3, 3), mk("lh", 64, 1, 3, 3)
    q16 = K.to_kv16(qkv)
    one = K.stripe_attn(q16, lv, lh, b, h, w, n, kv16=True)
    two = K.stripe_attn(q16, lv, lh, b, h, w, n, kv16=True, two_launches=True)
    print("stripe_attn kv16, batch %d: one launch == two launches bit for bit: %s" % (b, bool(torch.equal(one, two))))
    def t_us(fn, nrep):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / nrep
    best = {}
    for rnd_ in range(3):
        for tag, kw in (("two launches", dict(two_launches=True)), ("one launch", dict())):
            best[tag] = min(best.get(tag, 1e9), t_us(lambda: K.stripe_attn(q16, lv, lh, b, h, w, n, kv16=True, **kw), args.iters))
    print("stripe_attn kv16, batch %d: two launches %.1f us, one launch %.1f us" % (b, best["two launches"], best["one launch"]), flush=True)
if "refine" in which:
    hp, wp = 96, 312
    qkv, table = mk("q3", b * hp * wp, 384), mk("t3", 49, 384)
    for shift in (0, 2):
        timeit("window_attn 4x4x1 shift=%d" % shift, lambda: K.window_attn(qkv, table, b, hp, wp, 1, 4, 4, shift, False))
    _l.nmrf_debug_window_pack1.restype = ctypes.c_int
    for var, tag in ((1, "ONE window per tile"), (2, "8 tiles per block, launch bound 2"), (3, "2 tiles per block"),
                     (4, "8 tiles per block, launch bound 1")):
        _l.nmrf_debug_window_pack1(var)
        timeit("window_attn 4x4x1 shift=0, " + tag, lambda: K.window_attn(qkv, table, b, hp, wp, 1, 4, 4, 0, False))
    _l.nmrf_debug_window_pack1(0)
if "warp" in which:
    f1, f2 = mk("f1", b, 64, h, w), mk("f2", b, 64, h, w)
    g1, g2 = mk("g1", b, 256, h, w), mk("g2", b, 256, h, w)
    lab = mk("lab", b * h * w * n).abs() * 40
    timeit("warp_corr_concat 1/8", lambda: K.warp_corr_concat(lab, f1, f2, g1, g2, n))

if "timing" in which:
    import numpy as np
    hp, wp = 48, 156
    qkv, table = mk("q", b * hp * wp * n, 384), mk("t", 121, 384)
    out = torch.empty(b * hp * wp * n, 128, device=dev)
    nblk = (hp // 6) * (wp // 6) * 4 * b
    nblk = ((hp // 6) * (wp // 6) + 1) // 2 * 4 * b          # two windows per block
    q16 = K.to_kv16(qkv)
    for p0_shift, p0_tag in ((0, "phase 0 on the 4x4x4 MFMA (product)"), (100, "phase 0 on the VALU (round 3)"), (200, "ONE window per five-wave block")):
        print("---- window 6x6x4, kv16 rows,", p0_tag)
        wpb = 1 if p0_shift >= 200 else 2
        nblk = ((hp // 6) * (wp // 6) + wpb - 1) // wpb * 4 * b
        stamps = torch.zeros(64 * 10 * 16 + nblk * 3, dtype=torch.int64, device=dev)
        for _ in range(3):
            _l.nmrf_debug_window_timing(ctypes.c_void_p(q16.data_ptr()), ctypes.c_void_p(table.data_ptr()), b, hp, wp, p0_shift,
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stamps.data_ptr()), None)
        torch.cuda.synchronize()
        allst = stamps.cpu().numpy().astype(np.int64)
        st = allst[:64 * 5 * wpb * 16].reshape(64, 5 * wpb, 16)[:, :5]
        cen = allst[64 * 5 * wpb * 16:64 * 5 * wpb * 16 + nblk * 3].reshape(nblk, 3)
        names = ["vec+tables issued", "barrier1 wait", "ev issue+phase0", "q/k/v issue+barrier2", "ev store+barrier3",
                 "tile0", "tile1", "tile2", "tile3", "tile4", "normalise+exchange", "stores"]
        d = np.diff(st[:, :, :13], axis=2)
        print("window 6x6x4 per-wave phase durations in shader cycles (mean over 64 blocks), waves 0..4:")
        for i, nm in enumerate(names):
            print("  %-20s" % nm, " ".join("%7.0f" % v for v in d[:, :, i].mean(0)))
        print("  %-20s" % "total", " ".join("%7.0f" % v for v in (st[:, :, 12] - st[:, :, 0]).mean(0)))
        # census: per-CU concurrency from realtime (100 MHz) block start/end
        t0 = cen[:, 1].min()
        dur = (cen[:, 2] - cen[:, 1]) / 100.0
        print("  blocks %d, distinct smid %d, block duration us: mean %.1f min %.1f max %.1f; kernel span %.1f us"
              % (nblk, len(set(cen[:, 0].tolist())), dur.mean(), dur.min(), dur.max(), (cen[:, 2].max() - t0) / 100.0))
        conc = []
        for sm in set(cen[:, 0].tolist()):
            rows = cen[cen[:, 0] == sm]
            ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
            cur = mx = 0
            for _, dlt in ev:
                cur += dlt
                mx = max(mx, cur)
            conc.append((len(rows), mx))
        import collections
        print("  blocks per smid histogram:", dict(collections.Counter(c[0] for c in conc)))
        print("  max concurrent blocks per smid histogram:", dict(collections.Counter(c[1] for c in conc)))

if "stripe_census" in which:
    import numpy as np, collections
    qkv = mk("q2", b * h * w * n, 384)
    lh = mk("lh", 64, 1, 3, 3)
    out = torch.empty(b * h * w * n, 128, device=dev)
    grid = (ctypes.c_int * 3)()
    cap = 20 * h * 2 * b * 3 + 64 * 4 * 16 + 64
    census = torch.zeros(cap, dtype=torch.int64, device=dev)
    for _ in range(3):
        _l.nmrf_debug_stripe_census(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(lh.data_ptr()), b, h, w,
                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(census.data_ptr()), grid, None)
    torch.cuda.synchronize()
    nblk = grid[0] * grid[1] * grid[2]
    cen = census.cpu().numpy()[: nblk * 3].reshape(nblk, 3)
    st = census.cpu().numpy()[nblk * 3: nblk * 3 + grid[0] * 4 * 16].reshape(grid[0], 4, 16).astype(np.int64)
    nm = ["setup+q/k/v issue", "tile0", "tile1", "tile2", "tile3", "tile4", "loop exit", "lepe", "o->lds", "barrier", "merge+store(w0)"]
    d = np.diff(st[:, :, :12], axis=2)
    print("stripe<1,2,1> per-wave phases in shader cycles (mean over the blocks of stripe 0/head 0), waves 0..3 (first 5 of 20 tiles):")
    for i, n_ in enumerate(nm):
        print("  %-20s" % n_, " ".join("%7.0f" % v for v in d[:, :, i].mean(0)[: (1 if i == 10 else 4)]))
    print("  %-20s" % "start->barrier", " ".join("%7.0f" % v for v in (st[:, :, 9] - st[:, :, 0]).mean(0)))
    t0 = cen[:, 1].min()
    dur = (cen[:, 2] - cen[:, 1]) / 100.0
    print("stripe<1,2,1>: blocks %d, smids %d, block us mean %.1f min %.1f max %.1f, span %.1f us" %
          (nblk, len(set(cen[:, 0].tolist())), dur.mean(), dur.min(), dur.max(), (cen[:, 2].max() - t0) / 100.0))
    conc = []
    for sm in set(cen[:, 0].tolist()):
        rows = cen[cen[:, 0] == sm]
        ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
        cur = mx = 0
        for _, dlt in ev:
            cur += dlt
            mx = max(mx, cur)
        conc.append((len(rows), mx))
    print("  blocks per smid:", dict(collections.Counter(c[0] for c in conc)))
    print("  max concurrent blocks per smid:", dict(collections.Counter(c[1] for c in conc)))

if "mfma_peak" in which:
    out = torch.empty(4096 * 256, device=dev)
    for chains in (1, 2, 4):
        for blocks in (256, 512, 1024, 2048):
            iters = 2000
            fn = lambda: _l.nmrf_debug_mfma_peak(chains, iters, blocks, ctypes.c_void_p(out.data_ptr()), None)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            fl = blocks * 4 * iters * 16 * chains * 4096.0
            print("mfma 32x32x2 f32: chains/wave %d, blocks %4d (waves/SIMD %.1f): %.1f TFLOP/s (%.3f ms)"
                  % (chains, blocks, blocks * 4 / 1024.0, fl / ms / 1e9, ms), flush=True)

if "attn_core" in which:
    out = torch.empty(4096 * 256, device=dev)
    seed = mk("seed", 256)
    for variant in (0, 1):
        for blocks in (256, 512, 768, 1024):
            iters = 2000
            fn = lambda: _l.nmrf_debug_attn_core_peak(variant, iters, blocks, ctypes.c_void_p(seed.data_ptr()),
                                                      ctypes.c_void_p(out.data_ptr()), None)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            fl = blocks * 4 * iters * 32 * 4096.0
            print("attention tile loop, registers only, variant %d, waves/SIMD %.0f: %.1f TFLOP/s, %.0f cycles/tile/wave @2.4GHz"
                  % (variant, blocks * 4 / 1024.0, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / iters), flush=True)

if "linear" in which:
    import torch.nn.functional as F
    T = b * 48 * 156 * 4
    x, y = mk("lx", T, 128), mk("ly", T, 128)
    gam, bet = mk("lg", 128) * 0.1 + 1, mk("lb", 128) * 0.1
    for (nm, e, n, act) in (("q|k|v  [LN(x+y)|F31] -> 384", 31, 384, 0), ("prop q|k|v [LN|ctx64] -> 384", 64, 384, 0),
                            ("fc1 + GELU  LN(x+y) -> 512", 0, 512, 2)):
        k = 128 + e
        extra = mk("le", T, e) if e else None
        w, bias = mk("lw%d" % n, n, k) * 0.1, mk("lbb", n)
        kp = (k + 3) // 4 * 4
        wpad = F.pad(w, (0, kp - k)).contiguous()
        pw = K.pack_linear_weight(w)

        def base():
            xn, a = K.add_ln_concat(x, y, gam, bet, extra, 1, kp)
            o = F.linear(a, wpad, bias)
            return F.gelu(o) if act == 2 else o
        timeit("hipBLASLt path : " + nm, base)
        timeit("token_linear   : " + nm, lambda: K.token_linear(x, pw, n, k, bias, (gam, bet, 1e-5), y, extra, 1, act))
    for (nm, k, n, res) in (("proj 128 -> 128", 128, 128, False), ("fc2 512 -> 128", 512, 128, False), ("ffn 160 -> 128 (+GELU)", 160, 128, False)):
        xx = mk("px%d" % k, T, k)
        w, bias = mk("pw%d" % k, n, k) * 0.1, mk("pb", n)
        pw = K.pack_linear_weight(w)
        act = 2 if k == 160 else 0
        timeit("hipBLASLt path : " + nm, lambda: F.gelu(F.linear(xx, w, bias)) if act else F.linear(xx, w, bias))
        timeit("token_linear   : " + nm, lambda: K.token_linear(xx, pw, n, k, bias, act=act))

if "linear_timing" in which:
    import numpy as np
    import torch.nn.functional as F
    T = b * 48 * 156 * 4
    x, y = mk("lx", T, 128), mk("ly", T, 128)
    gam, bet = mk("lg", 128) * 0.1 + 1, mk("lb", 128) * 0.1
    extra = mk("le", T, 31)
    stamps = torch.zeros(64 * 4 * 16, dtype=torch.int64, device=dev)
    for (nm, fn) in (("q|k|v LN KC=5 N=384", lambda: K.token_linear(x, K.pack_linear_weight(mk("lw", 384, 159)), 384, 159, mk("lb3", 384), (gam, bet, 1e-5), y, extra, 1, 0)),
                     ("proj KC=4 N=128", lambda: K.token_linear(x, K.pack_linear_weight(mk("pw", 128, 128)), 128, 128, mk("pb", 128))),
                     ("fc2 KC=16 N=128", lambda: K.token_linear(mk("hx", T, 512), K.pack_linear_weight(mk("hw", 128, 512)), 128, 512, mk("pb", 128)))):
        fn(); fn()
        stamps.zero_()
        _l.nmrf_debug_token_linear_timing(ctypes.c_void_p(stamps.data_ptr()))
        fn()
        torch.cuda.synchronize()
        _l.nmrf_debug_token_linear_timing(None)
        st = stamps.cpu().numpy().reshape(64, 4, 16).astype(np.int64)
        marks = [(0, "start"), (1, "prologue (loads, LN, LDS)"), (2, "barrier"), (3, "group 0: MFMAs + LDS staging"),
                 (4, "barrier"), (5, "group 0: row stores"), (11, "remaining groups")]
        print(nm, "-- per-wave phases in shader cycles, mean over 64 blocks, waves 0..3")
        prev = 0
        for idx, n_ in marks[1:]:
            if (st[:, :, idx] == 0).all():
                continue
            print("  %-26s" % n_, " ".join("%7.0f" % v for v in (st[:, :, idx] - st[:, :, prev]).mean(0)))
            prev = idx
        print("  %-26s" % "total", " ".join("%7.0f" % v for v in (st[:, :, 11] - st[:, :, 0]).mean(0)))

if "seed" in which:
    for pp in (7332, 8160, 58656, 261120):
        prob = torch.softmax(mk("sp%d" % pp, pp, 40) * 4.0, -1)
        vol = mk("sv%d" % pp, pp, 4, 40)
        fw = [mk("fw%d" % i, *sh) * 0.3 for i, sh in enumerate(((8, 4, 5), (8,), (16, 8, 5), (16,), (1, 16, 5), (1,)))]
        timeit("dpn_filter_softmax P=%d" % pp, lambda: K.dpn_filter_softmax(vol, *fw))
        if os.environ.get("KB_PROB"):                     # A/B of two builds (--lib): bit-compare the probabilities across processes
            pth = "gpurun_out/kb_prob_%d.pt" % pp
            got = K.dpn_filter_softmax(vol, *fw).cpu()
            if os.environ["KB_PROB"] == "save":
                os.makedirs("gpurun_out", exist_ok=True)
                torch.save(got, pth)
            else:
                print("  prob bit-identical to the saved build:", bool(torch.equal(got, torch.load(pth))))
        timeit("nms_topk (LDS rows) P=%d" % pp, lambda: K.nms_topk(prob, 4, 1e-3))
        sd = K.nms_topk(prob, 4, 1e-3)
        timeit("  + seed_features + seeds.float() P=%d" % pp, lambda: (K.seed_features(vol, sd, 3.14 / 64, 32), sd.float()))
        timeit("seed_select (one wave per row, registers; features fused) P=%d" % pp, lambda: K.seed_select(prob, vol, 4, 1e-3, 3.14 / 64, 32))
        assert torch.equal(K.seed_select(prob, vol, 4, 1e-3, 3.14 / 64, 32)[0], sd)
if "conv" in which:
    import torch.nn.functional as F
    for (bb, ci, co, hh, ww) in ((2, 64, 64, 192, 624), (2, 96, 96, 96, 312), (2, 96, 128, 96, 312), (2, 128, 128, 96, 312),
                                 (2, 128, 256, 96, 312), (2, 128, 256, 48, 156), (1, 128, 128, 48, 156)):
        xx = mk("cx%d" % ci, bb * args.batch, ci, hh, ww) * args.xscale
        wt = mk("cw%d%d" % (ci, co), co, ci, 3, 3) * 0.05
        pu = K.wino_pack_filter(wt)
        st = K.instance_stats(xx)
        fl = 2.0 * bb * args.batch * ci * co * 9 * hh * ww
        cands = [("MIOpen", lambda: F.conv2d(xx, wt, None, 1, 1)), ("wino", lambda: K.conv3x3_wino(xx, pu, co))]
        for strips in (2, 3, 4):
            if (co // 32) % strips == 0:
                pk = K.pack_conv3x3(wt, strips, co // 32 // strips)
                packed = (pk[0], strips, co // 32 // strips, pk[1])
                cands.append(("split%d" % strips, lambda packed=packed: K.conv3x3_split(xx, packed, co)))
                cands.append(("split%d+IN" % strips, lambda packed=packed: K.conv3x3_split(xx, packed, co, st)))
                if strips == 2:
                    def rows1(packed=packed):
                        _l.nmrf_debug_conv3_variant(100)
                        r = K.conv3x3_split(xx, packed, co, st)
                        _l.nmrf_debug_conv3_variant(0)
                        return r
                    cands.append(("split2+IN 4-row tiles", rows1))
        line = "conv3x3 %3d->%3d @%dx%dx%d :" % (ci, co, bb * args.batch, hh, ww)
        for nm, fn in cands:
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            line += "  %s %.1f us (%.0f TF/s)" % (nm, us, fl / us / 1e6)
        print(line, flush=True)

if "conv_census" in which:
    # timing experiments of the direct split conv (debug build): which resource the main loop waits for
    _l.nmrf_debug_conv3_variant.restype = ctypes.c_int
    for (bb, ci, co, hh, ww, strips) in ((2, 64, 64, 188, 624, 2), (2, 256, 256, 94, 312, 4), (2, 128, 128, 94, 312, 2)):
        xx = mk("cx%d" % ci, bb, ci, hh, ww)
        wt = mk("cw%d%d" % (ci, co), co, ci, 3, 3) * 0.05
        pk = K.pack_conv3x3(wt, strips, co // 32 // strips)
        packed = (pk[0], strips, co // 32 // strips, pk[1])
        for var, tag in ((0, "product (first call)"), (0, "product"), (1, "no weight DMA"), (2, "no halo restaging"), (4, "no barriers"), (8, "no MFMA"),
                         (16, "no output stores"), (3, "no DMA, no restaging"), (7, "no DMA / restaging / barriers"),
                         (23, "no DMA / restaging / barriers / stores"), (32, "no LDS fragment reads"),
                         (55, "MFMAs only (no DMA / restaging / barriers / stores / fragment reads)"), (63, "nothing (launch + prologue)")):
            _l.nmrf_debug_conv3_variant(var)
            timeit("conv3x3 %d->%d @%dx%dx%d strips %d: %s" % (ci, co, bb, hh, ww, strips, tag), lambda: K.conv3x3_split(xx, packed, co))
        _l.nmrf_debug_conv3_variant(0)

if "conv_stamps" in which:
    import numpy as np
    occ = (ctypes.c_int * 3)(-1, -1, -1)
    _l.nmrf_debug_conv3_occupancy.restype = ctypes.c_int
    _l.nmrf_debug_conv3_occupancy(occ)
    print("conv3x3_split runtime occupancy (blocks/CU): strips 2: %d  3: %d  4: %d" % tuple(occ), flush=True)
    _l.nmrf_debug_conv3_timing.restype = ctypes.c_int
    for (bb, ci, co, hh, ww, strips) in ((2, 64, 64, 188, 624, 2), (2, 256, 256, 94, 312, 4)):
        xx = mk("cx%d" % ci, bb, ci, hh, ww)
        wt = mk("cw%d%d" % (ci, co), co, ci, 3, 3) * 0.05
        pk = K.pack_conv3x3(wt, strips, co // 32 // strips)
        packed = (pk[0], strips, co // 32 // strips, pk[1])
        K.conv3x3_split(xx, packed, co); torch.cuda.synchronize()
        stamps = torch.zeros(128 * 4 * 16, dtype=torch.int64, device=dev)
        _l.nmrf_debug_conv3_timing(ctypes.c_void_p(stamps.data_ptr()))
        K.conv3x3_split(xx, packed, co)
        torch.cuda.synchronize()
        _l.nmrf_debug_conv3_timing(None)
        st = stamps.cpu().numpy().reshape(128, 4, 16).astype(np.int64)
        st = st[st[:, 0, 0] > 0]
        t0 = st[:, :, 0].min()
        print("conv3x3 %d->%d strips %d: %d stamped tiles of XCD 0; s_memtime ticks of 10 ns" % (ci, co, strips, st.shape[0]))
        gen = np.zeros(st.shape[0], dtype=bool)
        for nm, sel in (("stamped tiles", ~gen),):
            if not sel.any():
                continue
            s2 = st[sel]
            def med(a):
                return float(np.median(a))
            print("  %s (%d): (start %.0f) | first loads landed +%.0f | barrier +%.0f | halo written +%.0f | stage 0 done +%.0f | slab 0 "
                  "done +%.0f | loop done +%.0f | stores issued +%.0f | stores drained +%.0f" % (
                      nm, s2.shape[0], med(s2[:, :, 0] - t0), med(s2[:, :, 1] - s2[:, :, 0]), med(s2[:, :, 2] - s2[:, :, 1]),
                      med(s2[:, :, 3] - s2[:, :, 2]), med(s2[:, :, 4] - s2[:, :, 3]), med(s2[:, :, 5] - s2[:, :, 4]),
                      med(s2[:, :, 6] - s2[:, :, 5]), med(s2[:, :, 7] - s2[:, :, 6]), med(s2[:, :, 11] - s2[:, :, 7])))
            print("     per wave, summed over stages: waits (vmcnt + barrier) %.0f | compute (LDS reads + MFMAs) %.0f | halo restaging %.0f | "
                  "whole wave %.0f" % (med(s2[:, :, 8]), med(s2[:, :, 9]), med(s2[:, :, 10]), med(s2[:, :, 11] - s2[:, :, 0])))
            print("     of the dy == 0 waits: vmcnt %.0f ; of the restaging: until the halo is written %.0f, until the second barrier %.0f, "
                  "DMA + prefetch issue %.0f" % (med(s2[:, :, 12]), med(s2[:, :, 13]), med(s2[:, :, 14]), med(s2[:, :, 10] - s2[:, :, 14])))
        # (s_memtime counters of different CUs are not synchronised: only differences inside one wave are meaningful)

if "in_stats" in which:
    # the InstanceNorm statistics pass alone, on maps that fit / do not fit the 256 MB memory-side cache, repeated on ONE map (its lines
    # can stay cached between calls) and rotating over several maps (they cannot)
    for (bb, ch, hh, ww) in ((2, 64, 192, 624), (2, 128, 96, 312), (16, 64, 192, 624)):
        maps = [mk("is%d_%d_%d" % (bb, ch, k), bb, ch, hh, ww) for k in range(6 if bb <= 2 else 2)]
        mb = maps[0].numel() * 4 / 1e6
        def t_us(fn, nrep):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nrep):
                fn(i)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / nrep
        same = t_us(lambda i=0: K.instance_stats(maps[0]), 60)
        rot = t_us(lambda i=0: K.instance_stats(maps[i % len(maps)]), 60)
        print("in_stats %dx%dx%dx%d (%.0f MB): same map %.1f us (%.2f TB/s), rotating over %d maps %.1f us (%.2f TB/s)" % (
            bb, ch, hh, ww, mb, same, mb / same, len(maps), rot, mb / rot), flush=True)

if "conv_one" in which:
    # the 64 -> 64 convolution of layer 1 alone (for counter passes): --batch images of 192 x 624
    xx = mk("c1x%d" % args.batch, args.batch, 64, 192, 624)
    wt = mk("c1w", 64, 64, 3, 3) * 0.05
    pk = K.pack_conv3x3(wt, 2, 1)
    packed = (pk[0], 2, 1, pk[1])
    st = K.instance_stats(xx)
    timeit("conv3x3 64->64 @%dx192x624 +IN" % args.batch, lambda: K.conv3x3_split(xx, packed, 64, st))

if "conv_lds" in which:
    # A/B of the LDS request of the 3x3 kernels (debug build): variant 300 = the fixed 256-channel affine table of rounds 2-4 (one
    # 1 280-byte granule too many for a third resident block of the two-strip form), 0 = table sized by Ci.  Interleaved, min of 3.
    _l.nmrf_debug_conv3_variant.restype = ctypes.c_int
    def t_us(fn, n):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    for (bb, ci, co, hh, ww, strips) in ((2, 64, 64, 192, 624, 2), (2, 32, 64, 192, 624, 2), (8, 64, 64, 192, 624, 2), (16, 64, 64, 192, 624, 2),
                                         (64, 64, 64, 272, 480, 2), (2, 128, 128, 96, 312, 2), (16, 128, 128, 96, 312, 2), (2, 96, 96, 96, 312, 3),
                                         (16, 96, 96, 96, 312, 3), (2, 256, 256, 96, 312, 4), (16, 256, 256, 96, 312, 4), (2, 128, 128, 48, 156, 2)):
        xx = mk("clx%d_%d_%d" % (ci, bb, hh), bb, ci, hh, ww)
        wt = mk("clw%d%d" % (ci, co), co, ci, 3, 3) * 0.05
        pk = K.pack_conv3x3(wt, strips, co // 32 // strips)
        packed = (pk[0], strips, co // 32 // strips, pk[1])
        st = K.instance_stats(xx)
        _l.nmrf_debug_conv3_variant(300)
        ref = K.conv3x3_split(xx, packed, co, st).clone()
        best, same = {}, {}
        for rnd in range(3):
            for var in (300, 0):
                _l.nmrf_debug_conv3_variant(var)
                same[var] = bool(torch.equal(K.conv3x3_split(xx, packed, co, st), ref))
                best[var] = min(best.get(var, 1e9), t_us(lambda: K.conv3x3_split(xx, packed, co, st), args.iters))
        _l.nmrf_debug_conv3_variant(0)
        print("conv3x3 %3d->%3d @%dx%dx%d strips %d +IN: fixed table %.1f us, sized by Ci %.1f us (%+.1f %%)%s" % (
            ci, co, bb, hh, ww, strips, best[300], best[0], 100.0 * (best[0] / best[300] - 1.0), "" if same[0] else "  MISMATCH"), flush=True)

if "conv_timeline" in which:
    # where the workgroups of one launch run and when: start / end of EVERY tile with the CU it ran on (debug build), per XCD clock
    import numpy as np
    _l.nmrf_debug_conv3_timing_n.restype = ctypes.c_int
    for (bb, ci, co, hh, ww, strips) in ((2, 64, 64, 192, 624, 2), (8, 64, 64, 192, 624, 2), (2, 128, 128, 96, 312, 2)):
        xx = mk("ctx%d_%d" % (ci, bb), bb, ci, hh, ww)
        wt = mk("cw%d%d" % (ci, co), co, ci, 3, 3) * 0.05
        pk = K.pack_conv3x3(wt, strips, co // 32 // strips)
        packed = (pk[0], strips, co // 32 // strips, pk[1])
        st_in = K.instance_stats(xx)
        for _ in range(3):
            K.conv3x3_split(xx, packed, co, st_in)
        torch.cuda.synchronize()
        nt = bb * ((hh + 7) // 8) * ((ww + 31) // 32)
        per_xcd = (nt + 7) // 8
        stamps = torch.zeros(nt * 4 * 16, dtype=torch.int64, device=dev)
        _l.nmrf_debug_conv3_timing_n(ctypes.c_void_p(stamps.data_ptr()), nt)
        K.conv3x3_split(xx, packed, co, st_in)
        torch.cuda.synchronize()
        _l.nmrf_debug_conv3_timing_n(None, 0)
        st = stamps.cpu().numpy().reshape(nt, 4, 16)
        ok = st[:, 0, 0] > 0
        start = st[:, :, 0].min(axis=1); end = st[:, :, 11].max(axis=1)
        xcd = np.arange(nt) // per_xcd
        hw = st[:, 0, 15] & 0xffffffff
        cu = (xcd << 16) | (hw & 0xff00)                      # se_id | sh_id | cu_id
        print("conv3x3 %d->%d @%dx%dx%d: %d tiles (%d stamped), %d distinct CUs" % (ci, co, bb, hh, ww, nt, int(ok.sum()), len(np.unique(cu[ok]))))
        span = []
        for x in range(8):
            m = ok & (xcd == x)
            if not m.any():
                continue
            t0 = start[m].min()
            span.append((end[m].max() - t0))
            if x == 0:
                rel_s = (start[m] - t0); rel_e = (end[m] - t0)
                order = np.argsort(rel_s)
                edges = np.arange(0, rel_e.max() + 10000, 10000)
                print("  XCD 0: span %d cycles; per 10k-cycle bucket: blocks started / blocks running at the bucket's start / median life of those started"
                      % span[-1])
                for e in edges:
                    sel = (rel_s >= e) & (rel_s < e + 10000)
                    running = int(((rel_s <= e) & (rel_e > e)).sum())
                    life = float(np.median((rel_e - rel_s)[sel])) if sel.any() else 0.0
                    print("    %7d: %4d started  %4d running  life %7.0f" % (e, int(sel.sum()), running, life))
                cus = cu[m]
                per = {}
                for c, a_, b_ in zip(cus, rel_s, rel_e):
                    per.setdefault(int(c), []).append((int(a_), int(b_)))
                cnt = np.array([len(v) for v in per.values()])
                print("  XCD 0: %d CUs, blocks per CU min / median / max %d / %.1f / %d" % (len(per), cnt.min(), np.median(cnt), cnt.max()))
                for c in list(per)[:4]:
                    print("    CU %06x: " % c + "  ".join("%d-%d" % ab for ab in sorted(per[c])))
        mt = (st[:, 0, 11] - st[:, 0, 0]).astype(np.float64); rt = (st[:, 0, 13] - st[:, 0, 12]).astype(np.float64)
        okk = ok & (rt > 0)
        rate = mt[okk] / rt[okk] * 100
        rs, re = st[okk, 0, 12], st[okk, 0, 13]
        print("  shader clock over a block's life (s_memtime per us of s_memrealtime): median %.0f MHz (min %.0f, max %.0f); launch on the chip-wide "
              "counter: first entry to last exit %.1f us, entries within %.1f us" % (np.median(rate), rate.min(), rate.max(),
                                                                                   (re.max() - rs.min()) / 100.0, (rs.max() - rs.min()) / 100.0), flush=True)

if "conv_timing" in which:
    import numpy as np
    for (bb, ci, co, hh, ww) in ((2, 64, 64, 188, 624), (2, 128, 128, 94, 312)):
        xx = mk("cx%d" % ci, bb, ci, hh, ww)
        pu = K.wino_pack_filter(mk("cw%d%d" % (ci, co), co, ci, 3, 3) * 0.05)
        K.conv3x3_wino(xx, pu, co); K.conv3x3_wino(xx, pu, co)
        stamps = torch.zeros(64 * 8 * 32, dtype=torch.int64, device=dev)
        _l.nmrf_debug_wino_timing(ctypes.c_void_p(stamps.data_ptr()))
        K.conv3x3_wino(xx, pu, co)
        torch.cuda.synchronize()
        _l.nmrf_debug_wino_timing(None)
        st = stamps.cpu().numpy().reshape(64, 8, 32).astype(np.int64)
        nch = ci // 8
        print("conv3x3_wino %d->%d @%dx%dx%d: per-wave phases in shader cycles, mean over 64 blocks x 8 waves" % (ci, co, bb, hh, ww))
        print("  launch -> first chunk ready   %7.0f" % (st[:, :, 2] - st[:, :, 0]).mean())
        cm = np.mean([(st[:, :, 1 + 3 * (k + 1)] - st[:, :, 3 + 3 * k]).mean() for k in range(0, min(nch, 8) - 1)])
        cp = np.mean([(st[:, :, 3 + 3 * k] - st[:, :, 2 + 3 * k]).mean() for k in range(0, min(nch, 8))])
        print("  per 8-channel chunk: commit + fetch issue + barrier %7.0f   transforms + MFMAs %7.0f" % (cm, cp))
        print("  inverse transform + output     %7.0f   total %7.0f" % ((st[:, :, 29] - st[:, :, 28]).mean(), (st[:, :, 29] - st[:, :, 0]).mean()))
