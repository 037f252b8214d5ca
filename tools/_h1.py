import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
from test_gpu_ops import _desc
from oracle import vnet_numpy as O
d = dev()
rng = np.random.default_rng(0)
for (c, shp) in ((32, (2, 16, 32, 16)), (64, (1, 16, 16, 16)), (128, (1, 8, 16, 8)), (256, (2, 8, 8, 8))):
    x = rng.standard_normal((shp[0], c) + shp[1:]).astype(np.float32)
    x = np.where(x > 0, x, 0.25 * x).astype(np.float32)      # PReLU-like activations
    w = (rng.standard_normal((c, c, 5, 5, 5)) * np.sqrt(2.0 / (125 * c))).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), (1, 1, 1), (2, 2, 2))
    for split in (3, 2):
        d.set_option("conv_split", split)
        yt = t_empty(shp[0], c, *shp[1:], fill=7.0)
        d.call("msk_conv3d_fwd", _desc((5, 5, 5), (1, 1, 1), (2, 2, 2)), t_from_ncdhw(x).msk(), vp(vec(w.ravel())), vp(vec(b)), yt.msk())
        got = t_to_ncdhw(yt)
        e = np.abs(got - ref)
        print("c=%d split %d: max rel %.2e  rms rel %.2e" % (c, split, e.max() / np.abs(ref).max(), np.sqrt((e ** 2).mean() / (ref ** 2).mean())))
d.set_option("conv_split", 3)
