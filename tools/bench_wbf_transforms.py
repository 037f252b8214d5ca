"""Micro-benchmark of the transform kernels of the 16-bit-pieces pipeline on one 5x5x5 'same' layer through the C ABI (one stream):
forward (wbf_tin_k<0> on x), data gradient (wbf_tin_k<0> on dy) and weight gradient (wbf_tin_k<0> on x + wbf_ty_k = <1> on dy), per-kernel
HIP-event times with the shapes in the tags.

    python tools/bench_wbf_transforms.py [--shape 512,512,12] [--cin 32] [--cout 32] [--iters 5] [--opt key=int ...]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512,512,12")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")
    a = ap.parse_args()
    from medicalseg_amd._lib import MskConvDesc
    from medicalseg_amd.device import Tensor, get_device
    dev = get_device()
    dev.set_option("wgrad_async", 0)
    for kv in a.opt:
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    rng = np.random.default_rng(0)
    vp = C.c_void_p
    d, h, w = (int(v) for v in a.shape.split(","))
    vox = a.batch * d * h * w

    def mk(ch):
        t = Tensor(dev, dev.malloc(vox * ch * 4), a.batch, d, h, w, ch, ch, None)
        dev.h2d(t.ptr, rng.standard_normal(vox * ch, dtype=np.float32))
        return t

    x, dy, y, dx = mk(a.cin), mk(a.cout), mk(a.cout), mk(a.cin)
    nw = a.cin * a.cout * 125
    wt, dw, db = dev.malloc(nw * 4), dev.malloc(nw * 4), dev.malloc(a.cout * 4)
    dev.h2d(wt, (rng.standard_normal(nw) * 0.02).astype(np.float32))
    cd = MskConvDesc(5, 5, 5, 1, 1, 1, 2, 2, 2)
    calls = {"fwd": lambda: dev.call("msk_conv3d_fwd", cd, x.msk(), vp(wt), vp(db), y.msk()),
             "dgrad": lambda: dev.call("msk_conv3d_dgrad", cd, dy.msk(), vp(wt), dx.msk(), 0),
             "wgrad": lambda: dev.call("msk_conv3d_wgrad", cd, x.msk(), dy.msk(), vp(dw), vp(db), 0)}
    dev.set_option("prof_shapes", 1)
    dev.set_option("prof_only_halo", 0)
    for name, fn in calls.items():
        fn()
        dev.sync()
        dev.prof_reset()
        dev.prof_enable(True)
        for _ in range(a.iters):
            fn()
        dev.sync()
        dev.prof_enable(False)
        rep = dev.prof_report()
        print("%-6s %s" % (name, " | ".join("%s %.3f" % (t, v[1] / max(v[0], 1)) for t, v in sorted(rep.items(), key=lambda kv: -kv[1][1]))))


if __name__ == "__main__":
    main()
