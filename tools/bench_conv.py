"""Micro-benchmark of one 'same' 5^3 convolution (forward, data gradient, weight gradient) through the C ABI,
with the tile/chunk tuning knobs:  python tools/bench_conv.py --c 32 --size 128 --halo-tile 1 --wgrad-chunk 1"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c", type=int, default=32)
    ap.add_argument("--cn", type=int, default=0, help="output channels (default: --c)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dhw", default=None, help="D,H,W of a non-cubic volume (overrides --size)")
    ap.add_argument("--acc", type=int, default=0, help="accumulate flag of dgrad / wgrad")
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--halo-tile", type=int, default=-1)
    ap.add_argument("--wgrad-chunk", type=int, default=-1)
    ap.add_argument("--impl", type=int, default=0, help="conv_impl knob (10 = force Winograd halo kernel, 11 = direct only, 12 = Winograd wgrad)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--zero", action="store_true", help="all-zero tensors and weights: DVFS check (a power-limited kernel "
                    "runs faster on zeros, MI355X_MICROARCH.md 'DVFS give-back')")
    ap.add_argument("--profile", action="store_true", help="also print the per-kernel HIP-event breakdown of each case")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="extra msk_set_option knobs")
    a = ap.parse_args()
    from medicalseg_amd._lib import MskConvDesc
    from medicalseg_amd.device import Tensor, get_device
    dev = get_device()
    dev.set_option("halo_tile", a.halo_tile)
    dev.set_option("wgrad_chunk", a.wgrad_chunk)
    dev.set_option("wgrad_async", 0)
    dev.set_option("conv_impl", a.impl)
    for kv in a.opt:
        key, val = kv.split("=")
        dev.set_option(key, int(val))
    n, s, c, k = a.n, a.size, a.c, a.k
    cn = a.cn or c
    D, H, W = (int(v) for v in a.dhw.split(",")) if a.dhw else (s, s, s)
    vox = n * D * H * W
    mk = lambda ch: Tensor(dev, dev.malloc(vox * ch * 4), n, D, H, W, ch, ch, None)
    x, y, dy, dx = mk(c), mk(cn), mk(cn), mk(c)
    rng = np.random.default_rng(0)
    for t in (x, dy):
        dev.h2d(t.ptr, np.zeros(vox * t.c, np.float32) if a.zero else rng.standard_normal(vox * t.c, dtype=np.float32))
    w = dev.malloc(c * cn * k ** 3 * 4)
    dev.h2d(w, (rng.standard_normal(c * cn * k ** 3) * (0.0 if a.zero else 0.01)).astype(np.float32))
    dw, b, db = dev.malloc(c * cn * k ** 3 * 4), dev.small(cn), dev.small(cn)
    cd = MskConvDesc(k, k, k, 1, 1, 1, k // 2, k // 2, k // 2)
    vp = C.c_void_p
    gf = 2.0 * k ** 3 * c * cn * vox / 1e9
    cases = {
        "fwd": lambda: dev.call("msk_conv3d_fwd", cd, x.msk(), vp(w), vp(b), y.msk()),
        "dgrad": lambda: dev.call("msk_conv3d_dgrad", cd, dy.msk(), vp(w), dx.msk(), a.acc),
        "wgrad": lambda: dev.call("msk_conv3d_wgrad", cd, x.msk(), dy.msk(), vp(dw), vp(db), a.acc),
    }
    for name, fn in cases.items():
        fn()
        dev.sync()
        dev.timer_start()
        for _ in range(a.iters):
            fn()
        ms = dev.timer_stop() / a.iters
        if a.profile:
            dev.set_option("prof_only_halo", 0)
            dev.prof_reset()
            dev.prof_enable(True)
            for _ in range(a.iters):
                fn()
            dev.sync()
            dev.prof_enable(False)
            for tag, (cnt, tms) in sorted(dev.prof_report().items(), key=lambda kv: -kv[1][1]):
                print(f"    {tag:32s} x{cnt // a.iters}  {tms / a.iters:7.3f} ms")
        print(f"c={c}->{cn} {n}x{D}x{H}x{W} k={k} tile={a.halo_tile} chunk={a.wgrad_chunk} {name:6s} {ms:8.3f} ms  {gf / ms:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
