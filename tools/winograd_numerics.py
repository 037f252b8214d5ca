"""Construction and fp32 error study of the 1-D Winograd transforms used by msk_conv_wino.hip / msk_wgrad_wino.hip
(Cook-Toom with exact fractions): F(2,5) with points 0, 1, -1, 2, -2, inf and F(4,5) with 0, +-1, +-2, +-1/2, inf.

    python tools/winograd_numerics.py

Prints AT (2x6), G (6x5), BT (6x6) and, for K = 800 accumulated products per output (25 taps x 32 channels, as in
a 32-channel LUConv layer), the fp32 error of the direct sum and of the Winograd evaluation relative to max|y|
(float64 reference): direct ~6e-7, Winograd ~1.3e-6 -- inside the 2e-5 * sqrt(K/1000 + 1) conv tolerance of
tests/test_gpu_ops.py.  Other point sets ({0, +-1, +-1/2, inf}, {0, +-1, 2, -1/2, inf}) are about 2x worse."""
from fractions import Fraction as F

import numpy as np


def cook_toom(m, r, pts):
    n = m + r - 1
    a = [F(p) for p in pts[:n - 1]]

    def vand(cols):
        rows = [[p ** j for j in range(cols)] for p in a]
        rows.append([F(0)] * (cols - 1) + [F(1)])
        return rows

    def polymul(p, q):
        out = [F(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                out[i + j] += x * y
        return out

    at = list(map(list, zip(*vand(m))))
    g = vand(r)
    for i, p in enumerate(a):
        den = F(1)
        for j, q in enumerate(a):
            if i != j:
                den *= p - q
        g[i] = [v / den for v in g[i]]
    full = [F(1)]
    for p in a:
        full = polymul(full, [-p, F(1)])
    bt = []
    for i in range(n - 1):
        li = [F(1)]
        for j, q in enumerate(a):
            if j != i:
                li = polymul(li, [-q, F(1)])
        bt.append(li + [F(0)] * (n - len(li)))
    bt.append(full)
    return at, g, bt


def study(pts, K=800, T=2000, seed=0, m=2):
    r = 5
    at, g, bt = (np.array(x, dtype=np.float64) for x in cook_toom(m, r, pts))
    rng = np.random.default_rng(seed)
    gs = (rng.standard_normal((K, r)) / np.sqrt(K * r)).astype(np.float32)
    ds = rng.standard_normal((T, K, m + r - 1)).astype(np.float32)
    ref = np.einsum('tki,ki->t', ds[:, :, :r].astype(np.float64), gs.astype(np.float64))
    direct = np.zeros(T, np.float32)
    for i in range(r):
        direct += np.einsum('tk,k->t', ds[:, :, i], gs[:, i]).astype(np.float32)
    u = (gs @ g.T.astype(np.float32)).astype(np.float32)
    v = (ds @ bt.T.astype(np.float32)).astype(np.float32)
    y = (np.einsum('tkn,kn->tn', v, u).astype(np.float32) @ at.T.astype(np.float32))[:, 0]
    sc = np.abs(ref).max()
    return np.abs(direct - ref).max() / sc, np.abs(y - ref).max() / sc


if __name__ == "__main__":
    at, g, bt = cook_toom(2, 5, [0, 1, -1, 2, -2])
    for name, mat in (("AT", at), ("G", g), ("BT", bt)):
        print(name)
        for row in mat:
            print("  ", [str(v) for v in row])
    for pts in ([0, 1, -1, 2, -2], [0, 1, -1, F(1, 2), -F(1, 2)], [0, 1, -1, 2, F(-1, 2)]):
        d, w = study(pts)
        print("F(2,5) points %-28s direct fp32 err %.2e   winograd fp32 err %.2e" % ([str(p) for p in pts], d, w))
    # F(4,5), the transform of conv_halo_wino4_k / wgrad_wino4_k (8 multiplications per 4 outputs)
    for pts in ([0, 1, -1, 2, -2, F(1, 2), -F(1, 2)], [0, 1, -1, 2, -2, 3, -3]):
        d, w = study(pts, m=4)
        print("F(4,5) points %-36s direct fp32 err %.2e   winograd fp32 err %.2e" % ([str(p) for p in pts], d, w))
