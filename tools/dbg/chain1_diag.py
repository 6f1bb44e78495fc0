"""Which layer of the kind-1 chain (csrc/mlp_chain.hip, <3,4,2,true,0,10,4>) goes wrong?   python tools/dbg/chain1_diag.py [lib.so]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import nmrf_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(sys.argv[1])
from nmrf_amd import kernels as K
dev = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    gg = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=gg) * 2 - 1) * scale


d = lambda t: None if t is None else t.to(dev)
t_ = 256
x = rnd(t_, 36, seed=1, scale=1.5)
extra = rnd(t_, 32, seed=7)
extra[:, 31] = 0
W1, W2, W3 = rnd(128, 36, seed=2, scale=0.2), rnd(128, 128, seed=3, scale=0.1), rnd(128, 159, seed=6, scale=0.1)
B1, B2 = rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3)
I = torch.eye(128)
I3 = torch.cat((I, torch.zeros(128, 31)), 1)
cases = {
    "full": (W1, B1, W2, B2, W3),
    "extra only (w3 hidden part 0)": (W1, B1, W2, B2, torch.cat((torch.zeros(128, 128), W3[:, 128:]), 1)),
    "no extra (w3 side part 0)": (W1, B1, W2, B2, torch.cat((W3[:, :128], torch.zeros(128, 31)), 1)),
    "gelu(layer 1) passed through (w2 = I, w3 = [I|0])": (W1, B1, I, torch.zeros(128), I3),
    "layer 1 pre-activation ~ via tiny weights": (W1 * 1e-3, B1 * 0, I, torch.zeros(128), I3),
    "layer 2 only (w1 makes gelu ~ linear? no: w3 = [I|0])": (W1, B1, W2, B2, I3),
}
for rep in range(2):
    for name, (w1, b1, w2, b2, w3) in cases.items():
        hdn = F.gelu(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
        ref = torch.cat((hdn, extra[:, :31].double()), 1) @ w3.double().t()
        stream, stages, inv = K.chain_stream([d(w1), d(w2), d(w3)], (48, 128, 160))
        got = K.mlp_chain(1, d(x), 36, stream, stages, inv, [d(b1), d(b2), None], 128, d(extra)).double().cpu()
        e = (got - ref).abs()
        bad = e > 1e-4 * max(1.0, float(ref.abs().max()))
        rows = bad.any(1).nonzero().view(-1).tolist()
        cols = bad.any(0).nonzero().view(-1).tolist()
        print("%-60s max err %.3e bad %6d rows %s cols %s" % (name, float(e.max()), int(bad.sum()),
              (rows[:3] + ["..."] + rows[-2:]) if len(rows) > 5 else rows, (cols[:6] + ["..."] + cols[-2:]) if len(cols) > 8 else cols))
        if "passed through" in name and bad.any():
            r = rows[0]
            pre = (x.double() @ w1.double().t() + b1.double())[r]
            for c in cols[:8]:
                print("      row %d col %d: pre %.5f  got %.6f  want %.6f   gelu_AS-like? %.6f" % (r, c, float(pre[c]), float(got[r, c]), float(ref[r, c]), 0))
