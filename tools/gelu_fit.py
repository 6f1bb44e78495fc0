import numpy as np, mpmath as mp
from scipy.optimize import linprog
from scipy.special import erfc, erf
try:
    import mpmath
except Exception: pass
def target(t):
    # -log2(erfc(t/sqrt2))/t in high precision
    out = np.empty_like(t)
    for i, x in enumerate(t):
        out[i] = float(-mp.log(mp.erfc(mp.mpf(x) / mp.sqrt(2)), 2) / mp.mpf(x))
    return out
T = 7.0
t = np.linspace(1e-4, T, 4000)
tg = target(t)
c = erfc(t / np.sqrt(2))
w = 0.5 * t * t * c * np.log(2) + 1e-12
for deg in (5, 6, 7, 8):
    V = np.vander(t, deg + 1, increasing=True)
    # vars: coeffs (deg+1), E
    n = deg + 1
    A = np.vstack([np.hstack([w[:, None] * V, -np.ones((len(t), 1))]), np.hstack([-w[:, None] * V, -np.ones((len(t), 1))])])
    b = np.concatenate([w * tg, -w * tg])
    cost = np.zeros(n + 1); cost[-1] = 1
    r = linprog(cost, A_ub=A, b_ub=b, bounds=[(None, None)] * n + [(0, None)], method="highs")
    co = r.x[:n]
    print(deg, "minimax gelu err (exact arith)", r.x[-1], "lead", co[-1])
    np.save("/tmp/gelu/co%d.npy" % deg, co)
