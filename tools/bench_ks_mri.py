"""Micro-benchmark of the anisotropic down / up convolutions of the MRI-spine VNet (vnet_mri_spine_seg_512_512_12_15k.yml:9-10:
kernel (2, 2, 4) / stride (2, 2, 1) at 512 x 512 x 12 <-> 256 x 256 x 9, kernel (2, 2, 2) / stride (2, 2, 1) at 256 x 256 x 9 <->
128 x 128 x 8) through the C ABI, product dispatch (streaming kernels, round 4) next to the general kernels (conv_impl 6 / 16).

    python tools/bench_ks_mri.py [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--slice", action="store_true", help="the up-convolution's fine tensors are channel slices of a buffer twice as wide "
                    "(the zero-copy concat buffers of the real step)")
    ap.add_argument("--sets", type=int, default=1, help="rotate over this many copies of every tensor (cold caches)")
    a = ap.parse_args()
    from medicalseg_amd._lib import MskConvDesc
    from medicalseg_amd.device import Tensor, get_device
    dev = get_device()
    dev.set_option("wgrad_async", 0)
    rng = np.random.default_rng(0)
    vp = C.c_void_p

    class Rot:
        def __init__(self, ts):
            self.ts, self.i, self.ptr = ts, 0, ts[0].ptr

        def msk(self):
            self.i = (self.i + 1) % len(self.ts)
            return self.ts[self.i].msk()

    def mk(shape, ch, ld=None):
        d, h, w = shape
        ld = ld or ch
        ts = []
        for _ in range(a.sets):
            t = Tensor(dev, dev.malloc(d * h * w * ld * 4), 1, d, h, w, ch, ld, None)
            dev.h2d(t.ptr, rng.standard_normal(d * h * w * ld, dtype=np.float32))
            ts.append(t)
        return Rot(ts)

    def wbuf(nelem):
        p = dev.malloc(nelem * 4)
        dev.h2d(p, (rng.standard_normal(nelem) * 0.05).astype(np.float32))
        return p

    def run(name, fn, nbytes):
        out = []
        for impl in (0, 6, 16):
            dev.set_option("conv_impl", impl)
            fn()
            dev.sync()
            dev.timer_start()
            for _ in range(a.iters):
                fn()
            out.append(dev.timer_stop() / a.iters)
        dev.set_option("conv_impl", 0)
        dev.set_option("prof_only_halo", 0)
        dev.prof_reset()
        dev.prof_enable(True)
        fn()
        dev.sync()
        dev.prof_enable(False)
        tags = " + ".join(f"{t} {v[1]:.3f}" for t, v in sorted(dev.prof_report().items(), key=lambda kv: -kv[1][1]))
        print(f"{name:40s} {out[0]:7.3f} ms ({nbytes / out[0] / 1e9:5.2f} TB/s)   general kernels {min(out[1], out[2]):7.3f} ms   [{tags}]")

    levels = [((512, 512, 12), (256, 256, 9), (2, 2, 4), 16, 32, 64, 16), ((256, 256, 9), (128, 128, 8), (2, 2, 2), 32, 64, 128, 32)]
    for fine, coarse, k, dci, dco, uci, uco in levels:
        cd = MskConvDesc(*k, 2, 2, 1, 0, 0, 0)
        vf, vc = int(np.prod(fine)), int(np.prod(coarse))
        taps = int(np.prod(k))
        xf, yc, dyc, dxf = mk(fine, dci), mk(coarse, dco), mk(coarse, dco), mk(fine, dci)
        w, dw, db = wbuf(dco * dci * taps), wbuf(dco * dci * taps), wbuf(dco)
        tag = "%dx%dx%d" % fine
        run(f"down {dci}->{dco} @ {tag} fwd", lambda: dev.call("msk_conv3d_fwd", cd, xf.msk(), vp(w), vp(db), yc.msk()), 4 * (vf * dci + vc * dco))
        run(f"down {dci}->{dco} @ {tag} dgrad", lambda: dev.call("msk_conv3d_dgrad", cd, dyc.msk(), vp(w), dxf.msk(), 0), 4 * (vf * dci + vc * dco))
        run(f"down {dci}->{dco} @ {tag} wgrad", lambda: dev.call("msk_conv3d_wgrad", cd, xf.msk(), dyc.msk(), vp(dw), vp(db), 0), 4 * (vf * dci + vc * dco))
        fl = 2 * uco if a.slice else uco
        xc, yf, dyf, dxc = mk(coarse, uci), mk(fine, uco, fl), mk(fine, uco, fl), mk(coarse, uci)
        wt, dwt, dbt = wbuf(uci * uco * taps), wbuf(uci * uco * taps), wbuf(uco)
        run(f"up   {uci}->{uco} @ {tag} fwd", lambda: dev.call("msk_convT3d_fwd", cd, xc.msk(), vp(wt), vp(dbt), yf.msk()), 4 * (vc * uci + vf * uco))
        run(f"up   {uci}->{uco} @ {tag} dgrad", lambda: dev.call("msk_convT3d_dgrad", cd, dyf.msk(), vp(wt), dxc.msk(), 0), 4 * (vc * uci + vf * uco))
        run(f"up   {uci}->{uco} @ {tag} wgrad", lambda: dev.call("msk_convT3d_wgrad", cd, xc.msk(), dyf.msk(), vp(dwt), vp(dbt), 0), 4 * (vc * uci + vf * uco))
        for r in (xf, yc, dyc, dxf, xc, yf, dyf, dxc):
            for t in r.ts:
                dev.free(t.ptr)


if __name__ == "__main__":
    main()
