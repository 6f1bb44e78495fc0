"""GPU check of gelu_fast as compiled into a library build: elementwise (nmrf_bias_act_f32) over a dense range + the kind-0 / kind-1
chains of tests/test_hip_kernels.py::test_mlp_chain_fused with an error map.   python tools/dbg/gelu_check.py [lib.so]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import nmrf_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(sys.argv[1])
from nmrf_amd import kernels as K
dev = "cuda"
print("lib", sys.argv[1:] or "tree")
v = torch.cat((torch.linspace(-12, 12, 1 << 20), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 7.0, -7.0, 7.5, -7.5, 100.0, -100.0, 6e4, -6e4]))).float()
pad = (-v.numel()) % 128
v = torch.cat((v, torch.zeros(pad))).view(-1, 128).to(dev)
_, g = K.bias_act(v, None, 2, want_pre=False) if True else (None, None)
ref = F.gelu(v.double().cpu())
err = (g.double().cpu() - ref).abs()
print("elementwise gelu: max err %.3e at v=%.5f; nan %d" % (float(err.max()), float(v.cpu().view(-1)[int(err.argmax())]), int(torch.isnan(g).sum())))


def rnd(*shape, seed=0, scale=1.0):
    gg = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=gg) * 2 - 1) * scale


d = lambda t: None if t is None else t.to(dev)
for kind, t_, n_out in ((0, 300, 128), (1, 1000, 128), (1, 64, 128)):
    k1 = {0: 160, 1: 36}[kind]
    x = rnd(t_, k1, seed=1, scale=1.5)
    if kind == 0:
        ws = [rnd(128, 160, seed=2, scale=0.1), rnd(128, 128, seed=3, scale=0.1)]
        bs = [rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3)]
        kps, extra = (160, 128), None
        ref = F.gelu(x.double() @ ws[0].double().t() + bs[0].double()) @ ws[1].double().t() + bs[1].double()
    else:
        ws = [rnd(128, 36, seed=2, scale=0.2), rnd(128, 128, seed=3, scale=0.1), rnd(128, 159, seed=6, scale=0.1)]
        bs = [rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3), None]
        kps = (48, 128, 160)
        extra = rnd(t_, 32, seed=7)
        extra[:, 31] = 0
        hdn = F.gelu(x.double() @ ws[0].double().t() + bs[0].double()) @ ws[1].double().t() + bs[1].double()
        ref = torch.cat((hdn, extra[:, :31].double()), 1) @ ws[2].double().t()
    stream, stages, inv = K.chain_stream([d(w) for w in ws], kps)
    got = K.mlp_chain(kind, d(x), k1, stream, stages, inv, [d(b) for b in bs], n_out, d(extra)).double().cpu()
    e = (got - ref).abs()
    bad = e > 1e-4
    print("kind %d T %d: max err %.3e, %d bad of %d; bad rows %s ... bad cols %s" % (
        kind, t_, float(e.max()), int(bad.sum()), e.numel(), bad.any(1).nonzero().view(-1)[:12].tolist(), bad.any(0).nonzero().view(-1)[:12].tolist()))
    if bad.any():
        # which hidden unit would explain it?  replace gelu of unit u by the A-S value ... just print first bad entries
        idx = bad.nonzero()[:6]
        for r, c in idx.tolist():
            print("   [%d,%d] got %.6f want %.6f" % (r, c, float(got[r, c]), float(ref[r, c])))
