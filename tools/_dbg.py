import sys, os, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import *
from test_gpu_ops import _desc
d = dev()
rng = np.random.default_rng(0)
N, D, H, W = 1, 12, 16, 16
x = rng.standard_normal((N, 32, D, H, W)).astype(np.float32)
xt = t_from_ncdhw(x)
k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
for (kd, kh, kw, ci, co) in [(2, 2, 2, 0, 0), (2, 2, 2, 5, 1), (2, 2, 3, 0, 0), (2, 3, 2, 0, 0), (3, 2, 2, 0, 0), (0, 2, 2, 0, 0), (4, 2, 2, 0, 2), (1, 0, 4, 17, 1)]:
    w = np.zeros((3, 32, 5, 5, 5), np.float32)
    w[co, ci, kd, kh, kw] = 1.0
    yt = t_empty(N, 3, D, H, W, fill=7.0)
    d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(vec(w.ravel())), None, yt.msk())
    y = t_to_ncdhw(yt)
    # expected: y[co][d,h,w] = x[ci][d+kd-2, h+kh-2, w+kw-2]
    xp = np.pad(x[0], ((0, 0), (4, 4), (4, 4), (4, 4)))
    best = None
    for c2 in range(3):
        for cc in ([ci] if True else range(32)):
            for dz, dy, dx in itertools.product(range(-4, 5), repeat=3):
                ref = xp[cc, 4 + dz:4 + dz + D, 4 + dy:4 + dy + H, 4 + dx:4 + dx + W]
                e = np.abs(y[0, c2, 2:-2, 4:-4, 4:-4] - ref[2:-2, 4:-4, 4:-4]).max()
                if best is None or e < best[0]:
                    best = (e, c2, dz, dy, dx)
    other = [float(np.abs(y[0, c]).max()) for c in range(3)]
    print("tap", (kd, kh, kw, ci, co), "expect shift", (kd - 2, kh - 2, kw - 2), "co", co, "-> best", best, "absmax per co", other)
