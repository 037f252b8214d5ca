"""Race check of the pipelined one-kernel matrix stage: the same 'same' 5^3 convolution (forward with statistics-free epilogue and
accumulating data gradient) run N times on unchanged inputs must give bitwise identical outputs.   python tools/probes/repro_fused.py"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from medicalseg_amd._lib import MskConvDesc  # noqa: E402
from medicalseg_amd.device import Tensor, get_device  # noqa: E402

dev = get_device()
rng = np.random.default_rng(0)
bad = 0
for c, s in ((32, 128), (64, 64), (32, 64)):
    n = 2
    vox = n * s ** 3
    mk = lambda: Tensor(dev, dev.malloc(vox * c * 4), n, s, s, s, c, c, None)
    x, y, dx = mk(), mk(), mk()
    dev.h2d(x.ptr, rng.standard_normal(vox * c, dtype=np.float32))
    w = dev.malloc(c * c * 125 * 4)
    dev.h2d(w, (rng.standard_normal(c * c * 125) * 0.01).astype(np.float32))
    b = dev.small(c)
    cd = MskConvDesc(5, 5, 5, 1, 1, 1, 2, 2, 2)
    hashes = set()
    for it in range(12):
        dev.call("msk_conv3d_fwd", cd, x.msk(), C.c_void_p(w), C.c_void_p(b), y.msk())
        dev.memset(dx.ptr, 0, vox * c * 4)
        dev.call("msk_conv3d_dgrad", cd, y.msk(), C.c_void_p(w), dx.msk(), 1)
        h = hashlib.sha256(dev.d2h(y.ptr, (vox * c,), np.float32).tobytes() + dev.d2h(dx.ptr, (vox * c,), np.float32).tobytes()).hexdigest()
        hashes.add(h)
    print("c=%d %d^3: %d distinct result(s) over 12 runs" % (c, s, len(hashes)))
    bad += len(hashes) != 1
    for t in (x, y, dx):
        dev.free(t.ptr)
sys.exit(1 if bad else 0)
