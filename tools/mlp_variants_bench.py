"""the block kernel's MLP phase (LayerNorm -> fc1 -> GELU -> fc2 -> residual: 63 % of a block's FLOPs) as
stand-alone micro-kernels in OTHER decompositions, against the product's own MLP-only launch, in one gpurun call.

    python tools/mlp_variants_bench.py [--iters 30] [--tokens 29952,239616,958464]

  product   nmrf_nmp_block16_f32(msg = NULL, has_mlp = 1, KQ = 0): 8 waves x 16 tokens, two waves per SIMD        (csrc/nmp_block16.hip)
  r2-32tok  the round-2 32-token form of the same launch (v_mfma_f32_32x32x16_f16, 4 waves, no in-wave pipelining) (csrc/nmp_block.hip)
  b32/p0    tools/ab/mlp_b32.hip: 4 waves x 32 tokens on 16x16x32, TWO token tiles per weight fragment, product stage order
  b32/p1    ... GELU halves placed inside the neighbouring MFMA stages of the same wave
  b32/p2    ... and interleaved with them by sched_group_barrier
Every b32 form must return the SAME BITS as the product launch (same stream, same per-accumulator MFMA order); all are held to
test_nmp_block_fused's tolerance against fp64 on a sample of rows.  Prints us per launch, the split-fp16 MFMA fraction
(3 x 2 x T x 131 072 MAC / t / 2.5 PFLOP/s) and the ratio to the product."""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nmrf_amd import kernels as K  # noqa: E402
from nmrf_amd.utils.hashinit import unit_noise  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--tokens", default="29952,239616,958464")
ap.add_argument("--lib", default=None, help="another build of libnmrf_hip.so for the `product` row (tools/build_ab_nopk.sh)")
ap.add_argument("--so", default="tools/_ab/mlp_b32.so", help="build of tools/ab/mlp_b32.hip to load")
args = ap.parse_args()
if args.lib:
    import nmrf_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda")


def mk(key, *shape, scale=1.0):
    import numpy as np
    return (torch.from_numpy(unit_noise(key, int(np.prod(shape))).reshape(shape)) * scale).to(dev)


def build_so():
    src, so = os.path.join(ROOT, "tools/ab/mlp_b32.hip"), os.path.join(ROOT, args.so)
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.mlp_b32_f32.restype = ctypes.c_int
    return lib


def time_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    lib = build_so()
    w1, b1 = mk("w1", 512, 128, scale=0.15), mk("b1", 512, scale=0.1)
    w2, b2 = mk("w2", 128, 512, scale=0.08), mk("b2", 128, scale=0.1)
    g2, bn = 1.0 + 0.1 * mk("g2", 128), 0.1 * mk("bn", 128)
    stream16, stages16, inv16 = K.block_stream16(None, w1, w2, None, 0)
    stream32, stages32, inv32 = K.block_stream(None, w1, w2, None, 0)
    assert stages16 == 32
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    print("# product library: %s   micro-kernel build: %s" % (args.lib or "nmrf_amd/lib/libnmrf_hip.so", args.so))
    print("%-10s %10s | %9s %7s %7s | %s" % ("form", "tokens", "us/launch", "frac", "x prod", "check"))
    for T in [int(v) for v in args.tokens.split(",")]:
        x = mk("x%d" % T, T, 128, scale=2.0)
        # fp64 reference on a sample of rows (first / last tiles and a stride through the middle)
        idx = torch.cat([torch.arange(0, min(T, 256)), torch.arange(max(0, T - 160), T), torch.arange(0, T, max(1, T // 512))]).unique().to(dev)
        xs = x[idx].double()
        ln = torch.nn.functional.layer_norm(xs, (128,), g2.double(), bn.double(), 1e-5)
        ref = xs + torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(ln, w1.double(), b1.double())),
                                              w2.double(), b2.double())
        mlp = (g2, bn, 1e-5, b1, b2)
        prod = lambda: K.nmp_block(x, stream16, stages16, inv16, None, None, mlp, None, want_x=True)[0]
        r2 = lambda: K.nmp_block(x, stream32, stages32, inv32, None, None, mlp, None, want_x=True, tokens_per_wave=32)[0]
        outs = {}

        def b32(pipe, copies=1, pad=0):
            out = torch.empty_like(x)
            nbytes = stream16.numel() * 4
            stride = nbytes + pad
            if copies > 1:                                   # `copies` replicas of the stream, `stride` bytes apart (16-byte aligned)
                buf = torch.zeros(copies * stride // 4 + 4, dtype=torch.int32, device=dev)
                for c in range(copies):
                    buf[c * stride // 4: c * stride // 4 + stream16.numel()] = stream16.reshape(-1)
            else:
                buf = stream16
            def run():
                rc = lib.mlp_b32_f32(pipe, p(x), p(buf), 32, p(g2), p(bn), ctypes.c_float(1e-5), p(b1), p(b2),
                                     ctypes.c_float(inv16[1]), ctypes.c_float(inv16[2]), ctypes.c_int64(T), p(out), p(flag), st(),
                                     ctypes.c_int(copies), ctypes.c_int64(stride))
                assert rc == 0, rc
                return out
            return run
        forms = [("product", prod), ("r2-32tok", r2), ("b32/p0", b32(0)), ("b32/p1", b32(1)), ("b32/p2", b32(2)),
                 ("p0 x8", b32(0, 8)), ("p0 x32", b32(0, 32)), ("p0 x256", b32(0, 256)), ("p0 x32+4K", b32(0, 32, 4096 + 256)),
                 ("p2 x32", b32(2, 32)), ("p2 x256", b32(2, 256)),
                 ("p0 dist2", b32(10)), ("p0 dist3", b32(20)), ("p0 dist4", b32(30)), ("p2 dist3", b32(22))]
        base_us, base_out = None, None
        for name, fn in forms:
            try:
                o = fn()
                torch.cuda.synchronize()
                err = (o[idx].double() - ref).abs()
                tol_ok = bool((err <= 2e-5 + 1e-5 * ref.abs()).all())
                us = time_us(fn, args.iters)
                if name == "product":
                    base_us, base_out = us, o.clone()
                same = "" if name in ("product", "r2-32tok") else (" bit-equal to product: %s" % bool(torch.equal(o, base_out)))
                frac = 3 * 2.0 * T * 131072 / (us * 1e-6) / 2.5e15
                print("%-10s %10d | %9.2f %7.3f %7.2f | max err vs fp64 %.2e (%s)%s" % (
                    name, T, us, frac, base_us / us, float(err.max()), "within 2e-5 + 1e-5|ref|" if tol_ok else "OUT OF TOLERANCE", same), flush=True)
            except Exception as e:                                    # one broken form must not hide the others
                print("%-10s %10d | FAILED: %r" % (name, T, e), flush=True)
        assert int(flag.item()) == 0
        del x


if __name__ == "__main__":
    main()
