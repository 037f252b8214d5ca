"""Micro-benchmark of the HBM-bound per-voxel kernels at the VNet full-resolution shapes:
achieved GB/s (algorithmic bytes / HIP-event time) per kernel.  python tools/bench_elementwise.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    from medicalseg_amd._lib import NULL_TENSOR
    from medicalseg_amd.device import Tensor, get_device
    dev = get_device()
    for kv in sys.argv[1:]:                      # KEY=INT msk_set_option knobs (ew_cap, reduce_cap)
        k, v = kv.split("=")
        dev.set_option(k, int(v))
    vp = lambda p: C.c_void_p(p) if p else None
    for (n, s, c) in [(2, 128, 32), (2, 128, 16), (2, 64, 64)]:
        vox = n * s ** 3
        mk = lambda: Tensor(dev, dev.malloc(vox * c * 4), n, s, s, s, c, c, None)
        x, dout, dx, res, dres = mk(), mk(), mk(), mk(), mk()
        for t in (x, dout, res):
            dev.h2d(t.ptr, np.random.default_rng(0).standard_normal(vox * c, dtype=np.float32))
        vec = lambda: dev.small(4 * c)
        scale, shift, alpha, mean, invstd, sums, stats = vec(), vec(), vec(), vec(), vec(), vec(), vec()
        for p_ in (scale, invstd):
            dev.h2d(p_, np.ones(4 * c, np.float32))
        dev.h2d(alpha, np.full(4 * c, 0.25, np.float32))
        tb = vox * c * 4 / 1e9  # GB per tensor pass
        cases = {
            "bn_stats (1 read)": (1, lambda: dev.call("msk_bn_stats", x.msk(), vp(stats))),
            "affine_act_fwd (1r+1w)": (2, lambda: dev.call("msk_affine_act_fwd", x.msk(), vp(scale), vp(shift), NULL_TENSOR,
                                                          vp(alpha), dx.msk())),
            "affine_act_fwd + amax": (2, lambda: dev.call("msk_affine_act_fwd_amax", x.msk(), vp(scale), vp(shift), NULL_TENSOR,
                                                         vp(alpha), dx.msk(), vp(stats))),
            "bwd_reduce (2r)": (2, lambda: dev.call("msk_affine_act_bwd_reduce", x.msk(), vp(scale), vp(shift), NULL_TENSOR,
                                                   vp(alpha), vp(mean), vp(invstd), dout.msk(), vp(sums))),
            "bwd_apply (2r+1w)": (3, lambda: dev.call("msk_affine_act_bwd_apply", x.msk(), vp(scale), vp(shift), NULL_TENSOR,
                                                     vp(alpha), vp(mean), vp(invstd), vp(scale), dout.msk(), vp(sums),
                                                     C.c_double(float(vox)), 1, dx.msk(), NULL_TENSOR, 0)),
            "add_act_bwd (3r+2w)": (5, lambda: dev.call("msk_add_act_bwd", x.msk(), res.msk(), vp(alpha), dout.msk(),
                                                       dx.msk(), dres.msk(), 0, vp(sums))),
            "add_act_bwd acc (4r+2w)": (6, lambda: dev.call("msk_add_act_bwd", x.msk(), res.msk(), vp(alpha), dout.msk(),
                                                           dx.msk(), dres.msk(), 1, vp(sums))),
            "copy_scale (1r+1w)": (2, lambda: dev.call("msk_copy_scale", x.msk(), None, dx.msk(), 0)),
        }
        for name, (passes, fn) in cases.items():
            for _ in range(2):
                fn()
            dev.sync()
            dev.timer_start()
            for _ in range(10):
                fn()
            ms = dev.timer_stop() / 10
            print(f"[{n}x{s}^3x{c}] {name:26s} {ms:7.3f} ms  {passes * tb / ms * 1e3:7.0f} GB/s")
        for t in (x, dout, dx, res, dres):
            dev.free(t.ptr)


if __name__ == "__main__":
    main()
